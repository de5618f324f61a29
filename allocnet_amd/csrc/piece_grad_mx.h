// k_piece_grad_mx: the penalty / energy partial gradients of large batches with the two contractions against the CONSTANT
// normalised-time basis table on the FP64 matrix instructions.
//
// What the table contractions are.  With c~_k = c_k T^k every state of a piece at sample j is  sum_col tab[j][d][col] c~[col]
// (d = 0, 1, 2: position, T velocity, T^2 acceleration) and the gradient w.r.t. c~ is  sum_j sum_d tab[j][d][col] w_d[j]  with the
// per-sample weights w_d the penalty terms produce.  For res = 20 samples that is a [60 x D] constant matrix applied to the D
// coefficients of every (piece, axis) and its transpose applied to 60 weights: in k_piece_grad 198 of the ~440 FP64 instructions
// of a sample.
//
// Shape.  A wave owns 64 (trajectory, piece) pairs as NCS = 4 column sets of 16; lane (r = lane >> 4, col = lane & 15) works on
// pair 16 cs + col and on the samples j = r + 4 i', i' = 0 .. 4 -- four lanes per pair, five samples each, the assignment both
// instructions' register layouts ask for:
//   * forward, v_mfma_f64_16x16x4_f64: D[row][col] = sum_k A[row][k] B[k][col], A[row = lane & 15][k = lane >> 4] = table rows
//     (constants of the lane, read from LDS at their use), B[k = lane >> 4][col = lane & 15] = c~ of pair `col` -- a lane loads just the two
//     coefficients k = r, 4 + r of its pair --, and result register i of lane (r, col) is row 4 i + r.  The 60 (state, sample)
//     rows are numbered row = 4 u + r, u = 5 d + i', so lane (r, col) receives exactly ITS samples' states: no transposes;
//   * gradient, v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks; layout probed on the device, tools/micro/mfma_f64_layout.hip:
//     A[i = lane & 3][k = lane >> 4], B[k = lane >> 4][j = lane & 3] and D[i = lane >> 4][j = lane & 3] of block (lane >> 2) & 3):
//     block = four pairs, k = the four lanes' samples of step u, B = the lane's OWN weight of step u, A = table entries
//     tab[.][col = 4 ct + i] -- the result lands as coefficient 4 ct + r of pair `col`.  (The 16x16x4 form would waste half its
//     rows here: D coefficients are 8 of 16.)
// What the matrix instructions buy on gfx950 is NOT a second pipe: FP64 MFMA and FP64 VALU instructions of a SIMD add up
// (profiles/r06_mfma_f64_mix.txt: 64 / 16 busy cycles per instruction, mixed streams take the sum) -- it is the denser issue, 16
// resp. 4 FMA-instruction equivalents per instruction at 4.0-4.3 cycles each against 4.6-5.2 for v_fma_f64 streams and ~5.9 in
// k_piece_grad, no table loads, a third of the registers (c~ and the gradient are 2 of 8 columns per lane).
// Only res = 20 (planner.yaml:21, every BASELINE config) and orders 3 / 4 take this kernel; everything else k_piece_grad.
#pragma once
#include "minco_kernels.h"

namespace anet {

#ifndef ANET_PGMX_MINB
#define ANET_PGMX_MINB 2
#endif

constexpr int kMxRes = 20;  // samples per piece this kernel is built for: 4 lanes x 5 samples
constexpr int kMxNSL = 5;   // samples per lane
constexpr int kMxNU = 15;   // (state, sample-of-the-lane) slots per lane: u = 5 d + i'
// Column sets of 16 pairs a wave of k_piece_grad_mx walks (template parameter NCS): 4, or 8 where that still leaves four
// waves per SIMD slot pair -- a wave's first loads are cold, twice the sets per wave halve their share: 131 072 x 8 pieces 292 -> 283 us
// (16: 288; at 20 000 x 8, where 8 leave SIMDs empty, 61 -> 76: hence the threshold, launch_piece_grad).

typedef double mx_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mx_sel4(int r, double a0, double a1, double a2, double a3) {
  return r == 0 ? a0 : (r == 1 ? a1 : (r == 2 ? a2 : a3));
}

// One column set of 16 pairs on a wave: the states of every lane's five samples from the B operands cb (c~ of the lane's pair, columns r
// and 4 + r), the limit and corridor penalties, the gradient w.r.t. c~ (gN: columns r and 4 + r) -- the part k_piece_grad_mx and the
// one-launch evaluation (minco_fused_kernel.h, MX) share.  rows: the pair's parked row block in LDS ([row][4], b / mu); before(rb) /
// after(rb): called around the walk of row block rb (the callers' prefetch of the next block and its parking).  lag / laf: the
// constant A operands in LDS, read through lane_o (an index the compiler cannot hoist the reads through).
// Corridor rows: one wave-uniform test per row over the lane's five samples ahead of the per-sample tests (a compare-to-branch is
// a bubble -- one wave per SIMD in the one-launch kernel: 23.8 -> 22.6 us; two here: 311 -> 295 us -- and the zero rows a corridor is
// padded with never pass the first test).
template <int S, int RB = 16, class Before, class After>
__device__ __forceinline__ void mx_column_set(const Penalty &pp, const double inv_mu, const double inv_res, const int lane_o,
                                              const double *lag, const double *laf, const double *rows, const int M, const int nrb,
                                              const double Ti, const double (&cb)[3][2], Before &&before, After &&after,
                                              double (&gN)[3][2], double &csum, double &Rs1, double &Rs2, double &rT, double &step) {
  constexpr int NSL = kMxNSL;  // (RB: rows of a parked block, 16 or 8)
  const double wcm = pp.wc * pp.mu, wvm = pp.wv * pp.mu, wam = pp.wa * pp.mu;
  const double cv = pp.vmax * inv_mu, ca = pp.amax * inv_mu;
  rT = fast_rcp(Ti);
  step = Ti * inv_res;
  const double rT2 = rT * rT;
  const double kv = rT * inv_mu, ka = rT2 * inv_mu;
  const double thr1 = pp.vmax * Ti, thr2 = pp.amax * (Ti * Ti);
  const double K0 = step * pp.wc, K1 = step * rT * pp.wv, K2 = step * rT2 * pp.wa;
  // ---- forward, tiles 1 .. 3: velocity and acceleration of this lane's samples (and the position of its fifth) ----
  // Stage order (registers): limits -> their gradient steps u = 5 .. 14 -> position tile 0 -> corridor rows -> steps u = 0 .. 4;
  // a weight lives from its sample's penalty to its matrix instruction and no longer.
  mx_d4 V[3][4];
  auto forward_tile = [&](const int t) {
    const double af0 = laf[(t * 64 + lane_o) * 2], af1 = laf[(t * 64 + lane_o) * 2 + 1];
    const mx_d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) V[ax][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0, cb[ax][0], z, 0, 0, 0);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) V[ax][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1, cb[ax][1], V[ax][t], 0, 0, 0);
  };
  forward_tile(1);
  forward_tile(2);
  forward_tile(3);
  auto val = [&](int ax, int u) { return V[ax][u >> 2][u & 3]; };  // (u compile-time after unrolling)
  const double p4[3] = {val(0, 4), val(1, 4), val(2, 4)};          // (the position of the fifth sample sits in tile 1)
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) gN[ax][0] = gN[ax][1] = 0.0;
  auto grad_step = [&](const int u, const double (&w)[3]) {  // gN += tab[slot u]' w: two 4x4x4 instructions per axis
    const double ag0 = lag[(u * 64 + lane_o) * 2], ag1 = lag[(u * 64 + lane_o) * 2 + 1];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      gN[ax][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ag0, w[ax], gN[ax][0], 0, 0, 0);
      gN[ax][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ag1, w[ax], gN[ax][1], 0, 0, 0);
    }
  };
  csum = Rs1 = Rs2 = 0.0;
  // ---- velocity / acceleration limits (the formulas of piece_penalty_part) ----
#pragma unroll
  for (int ii = 0; ii < NSL; ++ii) {
    double a1[3], a2[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      a1[ax] = val(ax, NSL + ii);
      a2[ax] = val(ax, 2 * NSL + ii);
    }
    // (|a1| kv - cv > 0 <=> |a1| > vmax T: two maxima and two compares instead of six FMAs and five maxima)
    const double m1 = fmax(fmax(fabs(a1[0]), fabs(a1[1])), fabs(a1[2])), m2 = fmax(fmax(fabs(a2[0]), fabs(a2[1])), fabs(a2[2]));
    if (__any(m1 > thr1 || m2 > thr2)) {  // only one of +v, -v (+a, -a) can be violated: the slope has the sign of a1 (a2)
      double cost = 0.0, s1[3], s2[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double f, df;
        smoothed_l1_unit(__builtin_fma(fabs(a1[ax]), kv, -cv), f, df);
        cost = __builtin_fma(wvm, f, cost);
        s1[ax] = K1 * copysign(df, a1[ax]);
        Rs1 = __builtin_fma(s1[ax], a1[ax], Rs1);
        smoothed_l1_unit(__builtin_fma(fabs(a2[ax]), ka, -ca), f, df);
        cost = __builtin_fma(wam, f, cost);
        s2[ax] = K2 * copysign(df, a2[ax]);
        Rs2 = __builtin_fma(s2[ax], a2[ax], Rs2);
      }
      csum += cost;
      grad_step(NSL + ii, s1);
      grad_step(2 * NSL + ii, s2);
    }
  }
  // ---- corridor rows: positions and offsets in units of mu, normals as given ----
  forward_tile(0);
  double ps[3][NSL];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) ps[ax][ii] = V[ax][0][ii];
    ps[ax][4] = p4[ax];
  }
  double Fs[NSL], G[3][NSL];
#pragma unroll
  for (int ii = 0; ii < NSL; ++ii) {
    Fs[ii] = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) G[ax][ii] = 0.0;
  }
#pragma unroll 1
  for (int rb = 0; rb < nrb; ++rb) {
    // before(rb): the next block on its way while this one is walked; after(rb): it is parked (the same LDS rows: this block is
    // done with them then)
    before(rb);
    const double *src = rows;
    const int nq = M - rb * RB < RB ? M - rb * RB : RB;
#pragma unroll 1
    for (int q0 = 0; q0 < nq; q0 += 4) {  // (rows beyond M are zero rows: inside every corridor)
      double h[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) h[q][e] = src[(q0 + q) * 4 + e];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double uq[NSL];
#pragma unroll
        for (int ii = 0; ii < NSL; ++ii)
          uq[ii] = __builtin_fma(h[q][0], ps[0][ii], __builtin_fma(h[q][1], ps[1][ii], __builtin_fma(h[q][2], ps[2][ii], -h[q][3])));
        {
          double um = uq[0];
#pragma unroll
          for (int ii = 1; ii < NSL; ++ii) um = fmax(um, uq[ii]);
          if (!__any(um > 0.0)) continue;
        }
#pragma unroll
        for (int ii = 0; ii < NSL; ++ii) {
          const double u = uq[ii];
          if (__any(u > 0.0)) {  // wave-uniform: inside the corridor nothing else is computed
            const double w = fmax(u, 0.0), uc = fmin(w, 1.0), sq = uc * uc;
            Fs[ii] += w - uc;
            Fs[ii] = __builtin_fma(sq * uc, __builtin_fma(-0.5, uc, 1.0), Fs[ii]);
            const double df = sq * __builtin_fma(-2.0, uc, 3.0);
            G[0][ii] = __builtin_fma(df, h[q][0], G[0][ii]);
            G[1][ii] = __builtin_fma(df, h[q][1], G[1][ii]);
            G[2][ii] = __builtin_fma(df, h[q][2], G[2][ii]);
          }
        }
      }
    }
    after(rb);
  }
#pragma unroll
  for (int ii = 0; ii < NSL; ++ii) {
    csum = __builtin_fma(wcm, Fs[ii], csum);
    if (__any(Fs[ii] > 0.0)) {  // (no violated row at this sample in the whole wave: its weights are zeros)
      const double w0[3] = {K0 * G[0][ii], K0 * G[1][ii], K0 * G[2][ii]};
      grad_step(ii, w0);
    }
  }
}

template <int S, int NCS = 4>
__global__ void __launch_bounds__(256, ANET_PGMX_MINB) k_piece_grad_mx(PieceGradArgs a, const double *__restrict__ tab) {
  static_assert(S == 3 || S == 4, "orders 3 and 4");
  constexpr int D = 2 * S, NSL = kMxNSL, NU = kMxNU, RB = 16;
  constexpr int TST = RB * 4 + 2;              // doubles per trajectory of a row block in LDS (+2: the 16 trajectories' 16-byte reads fall on 16 bank groups)
  __shared__ double lag[NU * 64 * 2];          // gradient A operands: [u][lane][ct]
  __shared__ double laf[4 * 64 * 2];           // forward A operands: [tile][lane][ks]
  __shared__ double lrow[4][16 * TST];         // per wave: the corridor rows of the column set at work, [trajectory][row][4], b / mu
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane >> 4, col = lane & 15;
  const int i = blockIdx.y;
  const int64_t ld = a.ld;
  const Penalty pp = a.pp;
  const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
  // ---- constants of the lane ----
  // forward A operands: tile t holds the rows 16 t .. 16 t + 15; row = 4 u + r' <-> state d = u / 5 of sample r' + 4 (u % 5);
  // the position rows divided by mu (the corridor rows are evaluated in units of mu)
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = 16 * t + col, u = row >> 2, rr = row & 3, d = u / NSL, j = rr + 4 * (u % NSL), ck = 4 * ks + r;
        const bool in = u < NU && ck < D;
        const double v = tab[(size_t)(j * 4 + (in ? d : 0)) * D + (in ? ck : 0)];
        laf[(t * 64 + lane) * 2 + ks] = in ? (d == 0 ? v * inv_mu : v) : 0.0;
      }
  }
  if (wave == 0) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int ic = lane & 3, k = lane >> 4, j = k + 4 * (u % NSL), d = u / NSL, cc = 4 * ct + ic;
        lag[(u * 64 + lane) * 2 + ct] = tab[(size_t)(j * 4 + d) * D + (cc < D ? cc : 0)] * (cc < D ? 1.0 : 0.0);
      }
  }
  // energy part as two more small products per axis: with y_k = c_k' T^(S - 1 - k) (k < S: the S highest powers, column k; a
  // lane's own value) the gradient w.r.t. column i is  T^(S - i) sum_k E[i][k] y_k,  E[i][k] = 2 f_i f_k / (2 S - 1 - i - k),
  // f_k = (D - 1 - k)! / (S - 1 - k)!, and d/dT = (sum_k f_k y_k)^2 per axis: A operands E[i = lane & 3][k = lane >> 4] and f_k
  double AE, AP;
  {
    const int ie = lane & 3, ke = lane >> 4;
    double fi = 1.0, fk = 1.0;
    for (int e = 0; e < S; ++e) {
      fi *= (double)(D - 1 - ie - e);
      fk *= (double)(D - 1 - ke - e);
    }
    const bool in = ie < S && ke < S;
    AE = in ? 2.0 * fi * fk / (double)(2 * S - 1 - ie - ke) : 0.0;
    AP = ke < S ? fk : 0.0;
  }
  __syncthreads();
  const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NCS);
  if (b0 >= a.B) return;
  const int M = a.hpolys ? pp.M : 0;
  const int nrb = (M + RB - 1) / RB;                       // row blocks of 16 per pair
  double *const lr = lrow[wave];
  const bool has1 = 4 + r < D;                             // (order 3: the columns 4, 5 only)

  // Row blocks travel global -> registers (a block ahead, issued before the arithmetic of the block at work) -> LDS -> registers:
  // lane (r, col) fetches the rows r, r + 4, r + 8, r + 12 of its pair's block -- every row is fetched by ONE lane, not by the
  // four that use it -- and all four read them back.  (LDS instructions of a wave execute in order: no barrier.)
  // Addresses: a wave-uniform base (scalar registers) plus ONE 64-bit lane offset per column set -- r selects among four
  // multiples of ld, never a per-lane 64-bit multiplication.
  const int64_t rld = r == 0 ? 0 : (r == 1 ? ld : (r == 2 ? 2 * ld : 3 * ld));
  auto lane_b = [&](const int cs) {
    const int64_t bq = b0 + 16 * cs + col;
    return bq < a.B ? bq : a.B - 1;
  };
  double hn[4][4];
  int hok = 0;  // bit m: row r + 4 m of the fetched block exists
  auto fetch_rows = [&](const int cs, const int rb) {
    const int64_t lofs = 4 * rld + lane_b(cs);              // rows r + 4 m: (4 r) ld + b
    const double *const pb = a.hpolys + (int64_t)(i * pp.M) * 4 * ld;  // the pair's row 0 (uniform part)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int ru = rb * RB + 4 * m;                       // uniform part of the row index
      const bool ok = ru + r < M;
      if (m == 0) hok = 0;
      if (ru >= M) {                                        // (wave-uniform: none of the four rows exists)
#pragma unroll
        for (int e = 0; e < 4; ++e) hn[m][e] = 0.0;
        continue;
      }
      const double *const pm = pb + (int64_t)ru * 4 * ld;
      // (a lane whose row does not exist reads row `ru`, which does, and keeps zeros)
      const int64_t lo = ok ? lofs : lofs - 4 * rld;
#pragma unroll
      for (int e = 0; e < 4; ++e) hn[m][e] = (pm + (int64_t)e * ld)[lo];
      // (raw: the zeros go in where the block is parked -- a select here is a wait for the load right behind its issue)
      hok |= ok ? (1 << m) : 0;
    }
  };
  auto park_rows = [&]() {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double *dst = lr + col * TST + (r + 4 * m) * 4;
      const bool ok = (hok >> m) & 1;
      dst[0] = ok ? hn[m][0] : 0.0;
      dst[1] = ok ? hn[m][1] : 0.0;
      dst[2] = ok ? hn[m][2] : 0.0;
      dst[3] = ok ? hn[m][3] * inv_mu : 0.0;
    }
  };
  double cn[3][2], Tn;  // the next column set's coefficients and duration
  auto fetch_coeffs = [&](const int cs) {
    const int64_t b = lane_b(cs), lofs = rld + b;
    Tn = (a.T + (int64_t)i * ld)[b];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const double *const pc0 = a.coeffs + (int64_t)((i * 3 + ax) * D) * ld;
      cn[ax][0] = pc0[lofs];
      cn[ax][1] = (has1 ? pc0 + 4 * ld : pc0)[lofs];
    }
  };
  // Prefetched values must not be carried across the loop's back edge as loads in flight: the compiler then sinks the loads to
  // the end of the body and waits for everything (vmcnt counts in order) at the top.  `landed` ties a value to an empty asm
  // statement -- the wait stands where it is written, the value crosses the back edge as a plain register.
  auto landed = [](double &v) { asm volatile("" : "+v"(v)); };
  fetch_coeffs(0);
  if (nrb > 0) {
    fetch_rows(0, 0);
    park_rows();
  }

#pragma unroll 1
  for (int cs = 0; cs < NCS; ++cs) {
    const int64_t bq = b0 + 16 * cs + col;
    const bool live = bq < a.B;
    const int64_t b = live ? bq : a.B - 1;
    const double Ti = Tn;
    // (the A operands are constants of the lane: read where they are used, through an index the compiler cannot see through --
    //  hoisted out of this loop they are 46 registers the loop does not have)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    // ---- B operands: c~[ax][4 ks + r] = c T^(D - 1 - col) ----
    double tp[D];
    tp[0] = 1.0;
#pragma unroll
    for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Ti;
    double tsel[2];
    tsel[0] = mx_sel4(r, tp[D - 1], tp[D - 2], tp[D - 3], tp[D - 4]);
    if constexpr (D == 8) tsel[1] = mx_sel4(r, tp[3], tp[2], tp[1], tp[0]);
    else tsel[1] = mx_sel4(r, tp[1], tp[0], 0.0, 0.0);
    double cb[3][2];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      cb[ax][0] = cn[ax][0] * tsel[0];
      cb[ax][1] = has1 ? cn[ax][1] * tsel[1] : 0.0;
    }
    if (cs + 1 < NCS) fetch_coeffs(cs + 1);
    double gN[3][2], csum, Rs1, Rs2, rT, step;
    mx_column_set<S>(pp, inv_mu, inv_res, lane_o, lag, laf, lr + col * TST, M, nrb, Ti, cb,
                     [&](const int rb) {  // the next block on its way: this pair's next, or the next column set's first
                       const int rbn = rb + 1 < nrb ? rb + 1 : 0, csn = rb + 1 < nrb ? cs : cs + 1;
                       if (csn < NCS) fetch_rows(csn, rbn);
                     },
                     [&](const int rb) {
                       if ((rb + 1 < nrb ? cs : cs + 1) < NCS) {
#pragma unroll
                         for (int m = 0; m < 4; ++m)
#pragma unroll
                           for (int e = 0; e < 4; ++e) landed(hn[m][e]);
                         park_rows();
                       }
                     },
                     gN, csum, Rs1, Rs2, rT, step);
    // ---- d/dT at fixed c (quadrature weight and sample times, as in piece_penalty_part); the sums over the pair's four lanes as
    //      products with a matrix of ones (every lane receives the sum) ----
    double acc = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      acc = __builtin_fma(cb[ax][0] * (double)(D - 1 - r), gN[ax][0], acc);
      acc = __builtin_fma(cb[ax][1] * (double)(D - 5 - r), gN[ax][1], acc);  // (cb = 0 where the column does not exist)
    }
    double gT = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, csum * inv_res + rT * (acc - __builtin_fma(2.0, Rs2, Rs1)), 0.0, 0, 0, 0);
    const double pc = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, step * csum, 0.0, 0, 0, 0);
    // ---- d/dc = T^k d/dc~, the energy part, the stores ----
    double g0[3], g1[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      g0[ax] = gN[ax][0] * tsel[0];
      g1[ax] = gN[ax][1] * tsel[1];
    }
    if (a.with_energy) {
      // y_k = c_k T^(S - 1 - k) = c~_k T^-S for the lane's column k = r < S (zero operands A elsewhere); T^(S - r) = tsel[0] T^-(S - 1)
      const double rTS = S == 4 ? (rT * rT) * (rT * rT) : rT * (rT * rT);
      const double TA = tsel[0] * (S == 4 ? rT * (rT * rT) : rT * rT);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const double ye = cb[ax][0] * rTS;
        const double e = __builtin_amdgcn_mfma_f64_4x4x4f64(AE, ye, 0.0, 0, 0, 0);
        const double ps = __builtin_amdgcn_mfma_f64_4x4x4f64(AP, ye, 0.0, 0, 0, 0);
        g0[ax] = __builtin_fma(e, TA, g0[ax]);  // (e = 0 in the lanes r >= S)
        gT = __builtin_fma(ps, ps, gT);
      }
    }
    if (live) {
      const int64_t lofs = rld + b;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double *const pg0 = a.gdC + (int64_t)((i * 3 + ax) * D) * ld;
        pg0[lofs] = g0[ax];
        if (has1) (pg0 + 4 * ld)[lofs] = g1[ax];
      }
      if (r == 0) {
        a.gdT[(int64_t)i * ld + b] = gT;
        if (a.pcost) a.pcost[(int64_t)i * ld + b] = pc;
      }
    }
    if (cs + 1 < NCS) {
      landed(Tn);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        landed(cn[ax][0]);
        landed(cn[ax][1]);
      }
    }
  }
}

}  // namespace anet
