// Interior-point solve of the reference's inequality QP (planner/qp_solver.hpp:119-360), an alternative to
// the OSQP-faithful ADMM kernel (qp_admm.h) for callers who want the optimum rather than OSQP's iterates:
// same problem, same solution, 10-20 Newton steps instead of ~10^3 ADMM iterations.
//
// MI355X-first formulation (one 256-thread workgroup per trajectory, everything but the row state in LDS):
//  * normalised time per piece (as in qp_admm.h) AND Hermite coordinates: the unknowns are the node states
//    y_k = (p, p', .., p^(s-1)) T^d at the N+1 knots.  The reference's equality block -- C^(s-1) continuity and
//    the start/end PVA rows -- then holds by construction (fixed components are pinned), so the QP has
//    inequality rows only and its Newton matrix is SPD block tridiagonal with one 3s x 3s block per knot,
//    the shape the MINCO solve already uses (minco_core.h).
//  * piece i sees u_i = [y_i ; (T_i/T_i+1)^d y_i+1]; every row of the QP is a Hermite basis function (or a
//    derivative) at tau_j = j/res contracted with a 3-vector, so A'WA is assembled from per-sample 3x3 /
//    3-vector weights and one table of 3*res*2s numbers; Q, A, G are never formed.
//  * Mehrotra predictor-corrector; the two solves of a step share one block Cholesky.  What is sequential in a step
//    is kept short and what is idle is given work (round 3):
//      - TWISTED factorisation: the chain of knots is eliminated from both ends towards the middle by two waves at
//        once (half the sequential depth of the factor and of both substitutions, no fill-in);
//      - the dual residual, its scales and the affine right-hand side are formed by the other two waves meanwhile;
//      - the rows of the QP are visited three or four times per step, not five: the corrector's right-hand side is
//        t0 + mu_target / s with both per-sample sums taken in the affine pass, and (FUSE) the updating pass forms the
//        next step's sums from the rows it holds;
//      - slacks and multipliers (global memory, L2) are loaded a group of rows at a time ahead of the arithmetic; the
//        twelve box rows of a sample multiply no zeros;
//      - every phase starts from an opaque thread index and the scalars of the iteration live in SGPRs: what is the
//        same in every Newton step is not hoisted out of the loop into registers the phases need.
//  * Round 4: the diagonal factors are inverted in the chain and the substitutions are block-vector products carried in registers;
//    the multipliers start at the scale of the first iterate's cost gradient; every sum of a step is added in a fixed order
//    (results are the same bit for bit from launch to launch and for any launch order); the tables of (order, res, m34) are
//    built once per context (k_qp_ipm_tables); a large batch runs in TWO launches -- four steps of every problem, then the
//    unfinished ones longest-expected first (IpmArgs::it_stop / resume / cont) -- because what a batch of thousands costs beyond
//    its Newton steps is the problem that takes 40 of them and happens to start last.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "minco_core.h"  // fast_rcp
#include "minco_kernels.h"  // pair_sum
#include "wave_ops.h"  // wave_sum, wave_min_f64
#include "qp_admm.h"    // qblk1, fallf

namespace anet {

struct IpmArgs {
  const double *state;   // [B][2][3][3]
  const double *T;       // [B][N]
  const double *hpolys;  // [B][N][M][4]
  double *sl, *lam;      // [B][m] slacks / multipliers of the inequality rows, m = N*R*(M+12), sample index minor
  double *coeffs;        // [B][n]
  double *obj;           // [B]
  int *status, *iters;   // [B]
  double *res;           // [B][2] primal / dual residual at exit (relative)
  double *gradT;         // optional [B][N]
  const double *grad_z;  // optional [B][n]: d loss / d coefficients (the layout of `coeffs`) ...
  double *vjpT;          // ... -> [B][N] d loss / d T through the optimum (the KKT backward pass, layers.py:129-141)
  int64_t B;
  int N, R, M;
  double vmax, amax, m34, tol;
  int max_iter;
  double tol_accept;  // >= tol: once met, at most 8 more Newton steps are spent on reaching tol (0: same as tol)
  double tol_tenth;      // 0.1 tol (a kernel argument: computed in the kernel it is a loop invariant parked in registers)
  int twist_min_pieces;  // chains of at least this many pieces are factored from both ends (two waves), shorter ones from one
  const int *order;      // optional [B]: workgroup w takes problem order[w] (a permutation; longest-first from a previous solve)
  // A solve in TWO launches (large batches; qp_solve_dev_impl): the first stops every problem after it_stop Newton steps and
  // parks what the iteration carries -- node states and a dozen scalars; slacks and multipliers are in global memory anyway --
  // in cont [B][NY + kIpmContScalars]; the second (resume = 1) picks the unfinished ones up, LONGEST EXPECTED FIRST (order from
  // the parked residuals).  The first launch is perfectly balanced (every problem takes the same number of steps), the second
  // starts its long problems early: the batch no longer ends with a long problem that happened to start late.
  int it_stop, resume;
  double *cont;
  const double *tab;     // k_qp_ipm_tables' output for (order, R, m34)
#ifdef ANET_IPM_PROF
  long long *prof;  // [16] cycle counters of problem 0 (tools: ANET_BUILD_FLAGS=-DANET_IPM_PROF)
#endif
};
#ifdef ANET_IPM_PROF
#define IPM_TICK(slot)                                                          \
  do {                                                                          \
    __syncthreads();                                                            \
    if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {                        \
      const long long now_ = __builtin_readcyclecounter();                      \
      a.prof[slot] += now_ - prof_t_;                                           \
      prof_t_ = now_;                                                           \
    }                                                                           \
  } while (0)
#else
#define IPM_TICK(slot) do {} while (0)
#endif

// A row of the QP as a sample sees it: (c0, c1, c2) . (derivative `dsel` of the three axes) <= bound.  Corridor rows are
// general half-spaces on the position; the twelve box rows of a sample are +-v_ax <= vmax T, +-a_ax <= amax T^2 -- one
// axis and a sign, known at compile time once the visit is unrolled, so nothing is multiplied by their zeros.
struct IpmCorridorRow {
  double c0, c1, c2;
  static constexpr int dsel = 0;
  __device__ __forceinline__ double dot(const double (&v)[3]) const { return c0 * v[0] + c1 * v[1] + c2 * v[2]; }
  __device__ __forceinline__ void axpy(double t, double *g) const { g[0] += t * c0; g[1] += t * c1; g[2] += t * c2; }
  // w c c' into the per-sample weights: [0..5] the symmetric 3 x 3 of the corridor rows
  __device__ __forceinline__ void weights(double w, double *A) const {
    A[0] += w * c0 * c0; A[1] += w * c0 * c1; A[2] += w * c0 * c2;
    A[3] += w * c1 * c1; A[4] += w * c1 * c2; A[5] += w * c2 * c2;
  }
};
struct IpmBoxRow {
  int ax, dsel;  // dsel: 1 velocity, 2 acceleration
  double sgn;
  __device__ __forceinline__ double dot(const double (&v)[3]) const { return sgn * v[ax]; }
  __device__ __forceinline__ void axpy(double t, double *g) const { g[ax] += sgn * t; }
  // [6..8] velocity, [9..11] acceleration: diagonal weights per axis
  __device__ __forceinline__ void weights(double w, double *A) const { A[3 + dsel * 3 + ax] += w; }
};

// LDS strides chosen for the banks (64 banks of 4 bytes; a double takes two): lanes of a row pass are SAMPLES, so they read the
// basis table at 20 different j and write per-sample records at 64 different samples.  With the natural strides -- 3 D = 24
// doubles = 192 B per table row (snap), 30 doubles = 240 B per record -- those addresses fall on 4 resp. 16 bank positions
// (SQ_LDS_BANK_CONFLICT as large as SQ_ACTIVE_INST_LDS for the snap kernel, profiles/r03_qp_ipm_roofline.txt); one double of
// padding each spreads them over 32.
__host__ __device__ constexpr int ipm_ht_stride(int D) { return 3 * D + 1; }
constexpr int kIpmRecord = 31;  // 30 sums per sample + 1 pad
constexpr int kIpmContScalars = 16;  // scalars parked behind the node states of a stopped problem (IpmArgs::cont)

template <int S>
inline size_t qp_ipm_lds_bytes(int N, int R, int M) {
  constexpr int D = 2 * S, NB = 3 * D, BK = 3 * S;
  const size_t NS = (size_t)N * R;
  return sizeof(double) * ((size_t)(N + 1) * BK * 5 + (size_t)(N + 1) * BK * BK + (size_t)N * BK * BK + 3 * (size_t)N * NB +
                           (size_t)R * ipm_ht_stride(D) + 2 * D * D + (size_t)N * D + NS * kIpmRecord + (size_t)N * M * 4 + 2 * N + 32);
}

// The tables of k_qp_ipm that depend on (order, samples per piece, m34) only -- Hm = inverse of the Hermite collocation matrix
// (Gauss-Jordan on ONE thread), the basis rows ht at tau_j, the cost block Hobj in Hermite coordinates -- built ONCE per context
// and (order, res, m34) by this one-workgroup kernel and copied into LDS by every solve (0.7 of a Newton step of every workgroup
// went into rebuilding them: 98 k of the lone 8-piece problem's 1.55 M cycles).  Layout: Hm [D][D] | Hobj [D][D] | ht [R][3 D + 1].
template <int S>
__global__ void __launch_bounds__(256) k_qp_ipm_tables(double *out, int R, double m34) {
  constexpr int D = 2 * S, HS = ipm_ht_stride(D);
  const int tid = threadIdx.x, nt = blockDim.x;
  __shared__ double E[2 * D * D], Hm[D * D];
  double *Hobj = out + D * D, *ht = out + 2 * D * D;
  // ---- Hermite matrix: E[row][col] = value of derivative d of the monomial col at tau = 0 / 1; Hm = E^-1
  if (tid == 0) {
    constexpr int W = 2 * D;
    for (int r = 0; r < D; ++r)
      for (int col = 0; col < D; ++col) {
        const int d = r % S, k = D - 1 - col;
        double v = 0.0;
        if (k >= d) v = (r < S) ? (k == d ? fallf(k, d) : 0.0) : fallf(k, d);
        E[r * W + col] = v;
        E[r * W + D + col] = (r == col) ? 1.0 : 0.0;
      }
    for (int c = 0; c < D; ++c) {  // Gauss-Jordan with partial pivoting
      int pv = c;
      for (int r = c + 1; r < D; ++r)
        if (fabs(E[r * W + c]) > fabs(E[pv * W + c])) pv = r;
      for (int k = 0; k < W; ++k) { const double t = E[c * W + k]; E[c * W + k] = E[pv * W + k]; E[pv * W + k] = t; }
      const double ip = 1.0 / E[c * W + c];
      for (int k = 0; k < W; ++k) E[c * W + k] *= ip;
      for (int r = 0; r < D; ++r)
        if (r != c) {
          const double f = E[r * W + c];
          if (f != 0.0)
            for (int k = 0; k < W; ++k) E[r * W + k] -= f * E[c * W + k];
        }
    }
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) Hm[r * D + c] = E[r * W + D + c];  // Hm[col][m]
  }
  __syncthreads();
  for (int e = tid; e < R * 3 * D; e += nt) {
    const int j = e / (3 * D), d = (e / D) % 3, m = e % D;  // stored at ht[j * HS + d * D + m]
    const double tau = (double)j / (double)R;
    double v = 0.0;
    for (int col = 0; col < D; ++col) {
      const int k = D - 1 - col;
      if (k >= d) {
        double bv = fallf(k, d);
        for (int q = 0; q < k - d; ++q) bv *= tau;
        v += bv * Hm[col * D + m];
      }
    }
    ht[(size_t)j * HS + d * D + m] = v;
  }
  for (int e = tid; e < D * D; e += nt) {
    const int m = e / D, m2 = e % D;
    double v = 0.0;
    for (int c1 = 0; c1 < S; ++c1)
      for (int c2 = 0; c2 < S; ++c2) v += Hm[c1 * D + m] * qblk1<S>(c1, c2, m34) * Hm[c2 * D + m2];
    Hobj[e] = v;
  }
  for (int e = tid; e < D * D; e += nt) out[e] = Hm[e];
}

// MINB: workgroups per CU the register allocation is bounded for.  The snap kernel needs more than 256 registers to run
// without spills (one workgroup per CU); bounded to 256 (92 B of scratch) two share a CU, which pays for large batches
// (4096 x 8 pieces 24.7 -> 18.8 ms) and costs a single problem a quarter of its latency (0.82 -> 1.02 ms): the host picks.
// FUSE: the updating pass of a step also forms the per-sample sums of the NEXT step (one row pass fewer, 6 % of a lone
// problem's latency) -- at the price of registers: with two workgroups per CU it spills, so the throughput shapes visit the rows again.
template <int S, int MINB = 1, bool FUSE = (MINB == 1)>
__global__ void __launch_bounds__(256, MINB) k_qp_ipm(IpmArgs a) {
  constexpr int D = 2 * S, NB = 3 * D, BK = 3 * S;
  constexpr int HS = ipm_ht_stride(D), AS = kIpmRecord;  // padded strides of the basis table rows / per-sample records
  const int N = a.N, R = a.R, M = a.M;
  const int NS = N * R, RPS = M + 12;
  const int64_t mtot = (int64_t)NS * RPS;
  const int64_t b = a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int NY = (N + 1) * BK;
  const bool resume = a.resume != 0;
  // a caller's launch order is not checked on the host: an entry outside [0, B) is skipped (the host pre-sets the status of
  // every problem to "not run" whenever an order is given, so a problem no workgroup took says so)
  if ((uint64_t)b >= (uint64_t)a.B) return;
  if (resume && a.status[b] != 0) return;  // decided in the first launch (wave-uniform: the whole workgroup leaves)
  double *ct = a.cont ? a.cont + b * (int64_t)(NY + kIpmContScalars) : nullptr;

#ifdef ANET_IPM_PROF
  const long long prof_start_ = __builtin_readcyclecounter();
#endif
  extern __shared__ double lds[];
  double *yv = lds;                       // [NY] node states (scaled by T^d of the piece to their right)
  double *dya = yv + NY;                  // [NY] affine direction
  double *dyc = dya + NY;                 // [NY] final direction
  double *rhs = dyc + NY;                 // [NY]
  double *Dg = rhs + NY;                  // [(N+1)][BK*BK] diagonal blocks -> their Cholesky factors
  double *Of = Dg + (size_t)(N + 1) * BK * BK;  // [N][BK*BK] block (k+1,k) -> L_{k+1,k}
  double *uu = Of + (size_t)N * BK * BK;  // [N][NB] u_i of y
  double *dua = uu + (size_t)N * NB;      // [N][NB] u_i of the affine direction
  double *duc = dua + (size_t)N * NB;     // [N][NB] u_i of the final direction
  double *ht = duc + (size_t)N * NB;      // [R][3][D] Hermite basis (derivative d) at tau_j
  double *Hobj = ht + (size_t)R * HS;  // [D][D] cost block in Hermite coordinates (per axis, T = 1)
  double *Hm = Hobj + D * D;              // [D][D] c~ = Hm u
  double *sc = Hm + D * D;                // [N][D] 1 for the start half, (T_i/T_i+1)^d for the end half
  double *acc = sc + (size_t)N * D;       // [NS][30]: 0-5 Wa (sym), 6-8 Wv, 9-11 Wacc, 12-20 gamma[d][ax], 21-29 gamma_lambda
  double *hp_l = acc + (size_t)NS * AS;   // [N*M*4]
  double *Tn = hp_l + (size_t)N * M * 4;  // [N]
  double *red = Tn + N;                   // [32]
  double *qsv = red + 32;                 // [N] T_i^(1-2s)
  double *dinvd = qsv + N;                // [NY] reciprocals of the diagonal of the block Cholesky factor

  const double *Tg = a.T + b * N;
  const double *hp = a.hpolys + b * (int64_t)N * M * 4;
  const double *st = a.state + b * 18;
  double *slg = a.sl + b * mtot, *lmg = a.lam + b * mtot;

  for (int e = tid; e < N * M * 4; e += nt) hp_l[e] = hp[e];
  for (int i = tid; i < N; i += nt) {
    Tn[i] = Tg[i];
    qsv[i] = pow(Tg[i], (double)(1 - 2 * S));
  }
  // ---- tables of (order, res, m34): built once per context (k_qp_ipm_tables), copied here
  {
    const double *tb = a.tab;
    for (int e = tid; e < D * D; e += nt) {
      Hm[e] = tb[e];
      Hobj[e] = tb[D * D + e];
    }
    for (int e = tid; e < R * HS; e += nt) ht[e] = tb[2 * D * D + e];
  }
  __syncthreads();  // (Tn, qsv, the corridor rows and the tables: read by other threads from here on)
  for (int e = tid; e < N * D; e += nt) {
    const int i = e / D, m = e % D;
    double v = 1.0;
    if (m >= S && i < N - 1) v = pow(Tn[i] / Tn[i + 1], (double)(m - S));
    sc[e] = v;
  }
  // ---- starting point: boundary values pinned, interior knot positions on the chord, derivatives zero
  for (int e = tid; e < NY; e += nt) {
    const int k = e / BK, ax = (e / S) % 3, d = e % S;
    double v = 0.0;
    if (k == 0 && d < 3) v = st[ax * 3 + d] * pow(Tn[0], (double)d);
    else if (k == N && d < 3) v = st[9 + ax * 3 + d] * pow(Tn[N - 1], (double)d);
    else if (d == 0) {
      double tsum = 0.0, tk = 0.0;
      for (int i = 0; i < N; ++i) { tsum += Tn[i]; if (i < k) tk += Tn[i]; }
      const double f = tk / tsum;
      v = (1.0 - f) * st[ax * 3] + f * st[9 + ax * 3];
    }
    yv[e] = resume ? ct[e] : v;
    dya[e] = 0.0;
    dyc[e] = 0.0;
  }
  __syncthreads();
  // A value every lane holds alike (read from red[], or computed from such) moved to SGPRs: the scalars of the iteration
  // (residuals, complementarity, their marks, step lengths) live across the whole Newton loop, and as VALU results they
  // would each occupy a register pair of every lane.
  auto uni = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  auto pinned = [&](int k, int d) { return (k == 0 || k == N) && d < 3; };
  // The sample a thread owns (and with it i, j, its LDS rows and its slack / multiplier addresses) does not change from
  // one Newton step to the next, so the compiler hoists all of that out of the iteration loop and carries it through
  // every phase in registers the phases themselves need.  An opaque copy of tid per pass keeps it local.
  auto fresh_tid = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  // u_i of a node vector
  auto to_u = [&](const double *y, double *u) {
    for (int e = fresh_tid(); e < N * NB; e += nt) {
      const int i = e / NB, ax = (e % NB) / D, m = e % D;
      u[e] = (m < S) ? y[i * BK + ax * S + m] : sc[i * D + m] * y[(i + 1) * BK + ax * S + (m - S)];
    }
  };
  to_u(yv, uu);
  __syncthreads();

  // state rows (derivative d, axis ax) of a u-vector at sample (i, j)
  auto state_of = [&](const double *u, int i, int j, double (&s3)[3][3]) {
    const double *hj = ht + (size_t)j * HS, *ui = u + (size_t)i * NB;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double v = 0.0;
        for (int m = 0; m < D; ++m) v += hj[d * D + m] * ui[ax * D + m];
        s3[d][ax] = v;
      }
  };
  // Visit the live rows of a sample of piece i: fn(q, row, bound), row an IpmCorridorRow or an IpmBoxRow.  The box rows
  // are unrolled so that dsel, the axis and the sign are compile-time constants inside fn (private arrays indexed by
  // them stay in registers); all-zero corridor rows are inert padding.
  auto for_rows = [&](int i, auto &&fn) {
    for (int q = 0; q < M; ++q) {
      const double *hq = hp_l + ((size_t)i * M + q) * 4;
      const double c0 = hq[0], c1 = hq[1], c2 = hq[2];
      if (c0 == 0.0 && c1 == 0.0 && c2 == 0.0) continue;
      fn(q, IpmCorridorRow{c0, c1, c2}, hq[3]);
    }
    const double hvv = a.vmax * Tn[i], hva = a.amax * Tn[i] * Tn[i];
#pragma unroll
    for (int qq = 0; qq < 12; ++qq) {
      const int axsel = qq / 4, w4 = qq % 4, dsel = 1 + (w4 & 1);
      const double sgn = (w4 < 2) ? 1.0 : -1.0;
      fn(M + qq, IpmBoxRow{axsel, dsel, sgn}, dsel == 1 ? hvv : hva);
    }
  };
  // The same visit WITH the slack and the multiplier of each row: fn(q, row, bound, sl, lm).  The row state
  // lives in global memory (L2), one round trip is ~1-2 k cycles, and a row needs ~40 instructions: loaded where they are
  // used -- what the plain loop compiles to, and all it CAN compile to in the updating pass, whose stores may alias the
  // next loads -- the 28 rows of a sample are 28 exposed round trips and the five row passes of a Newton step are pure
  // L2 latency.  Here the loads of a group of rows (eight corridor rows, six box rows) are issued together, ahead of the
  // arithmetic of the group; STORE writes sl, lm back (the caller's fn changes them) after the group.
  auto for_rows_sl = [&](int i, int smp, auto store_tag, auto &&fn) {
    constexpr bool STORE = decltype(store_tag)::value;
    double *sp = slg + smp, *lp = lmg + smp;
    // (the updating pass also forms the next iterate's sums -- 30 accumulators on top of three state sets: smaller groups)
    constexpr int GC = (STORE && FUSE) ? 4 : 8, GB = (STORE && FUSE) ? 4 : 6;
    for (int q0 = 0; q0 < M; q0 += GC) {
      double sl[GC], lm[GC];
#pragma unroll
      for (int g = 0; g < GC; ++g) {
        const int q = q0 + g < M ? q0 + g : M - 1;
        sl[g] = sp[(int64_t)q * NS];
        lm[g] = lp[(int64_t)q * NS];
      }
#pragma unroll
      for (int g = 0; g < GC; ++g) {
        const int q = q0 + g;
        if (q >= M) break;
        const double *hq = hp_l + ((size_t)i * M + q) * 4;
        const double c0 = hq[0], c1 = hq[1], c2 = hq[2];
        if (c0 == 0.0 && c1 == 0.0 && c2 == 0.0) continue;
        fn(q, IpmCorridorRow{c0, c1, c2}, hq[3], sl[g], lm[g]);
        if (STORE) {
          sp[(int64_t)q * NS] = sl[g];
          lp[(int64_t)q * NS] = lm[g];
        }
      }
    }
    const double hvv = a.vmax * Tn[i], hva = a.amax * Tn[i] * Tn[i];
#pragma unroll
    for (int g0 = 0; g0 < 12; g0 += GB) {
      double sl[GB], lm[GB];
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        sl[g] = sp[(int64_t)(M + g0 + g) * NS];
        lm[g] = lp[(int64_t)(M + g0 + g) * NS];
      }
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int qq = g0 + g, axsel = qq / 4, w4 = qq % 4, dsel = 1 + (w4 & 1);
        const double sgn = (w4 < 2) ? 1.0 : -1.0;
        fn(M + qq, IpmBoxRow{axsel, dsel, sgn}, dsel == 1 ? hvv : hva, sl[g], lm[g]);
        if (STORE) {
          sp[(int64_t)(M + qq) * NS] = sl[g];
          lp[(int64_t)(M + qq) * NS] = lm[g];
        }
      }
    }
  };
  using RowsLoad = std::integral_constant<bool, false>;
  using RowsUpdate = std::integral_constant<bool, true>;
  // (wave part on the DPP path -- in-row scans and row broadcasts, ~20 VALU instructions -- where the __shfl_xor butterfly was
  //  twelve dependent ds_bpermute round trips per reduction, ten reductions per Newton step)
  // Sums are DETERMINISTIC: each wave stores its partial sum in a slot of its own (red[16 + 4 k + wave] for the four sums of a
  // step, k = sum_index(slot)) and red_sum() adds the four in a fixed order -- with atomicAdd the order of the waves' additions,
  // i.e. the last bit of mu, changed from launch to launch (and with the order in which workgroups are started).  A maximum is
  // exact in any order.
  auto sum_index = [](int slot) { return slot == 0 ? 0 : (slot == 4 ? 1 : (slot == 5 ? 2 : 3)); };  // slots 0, 4, 5, 10
  auto block_reduce = [&](double v, int slot, bool is_min) {  // red[] must have been zeroed before a barrier
    v = is_min ? wave_min_f64(v) : wave_sum<63>(v);
    if ((tid & 63) == 0) {
      if (is_min) atomic_max_pos(&red[slot], 1.0 / fmax(v, 1e-300));  // min of positives via max of reciprocals
      else red[16 + 4 * sum_index(slot) + (tid >> 6)] = v;
    }
  };
  auto red_sum = [&](int slot) {
    const double *p4 = red + 16 + 4 * sum_index(slot);
    return (p4[0] + p4[1]) + (p4[2] + p4[3]);
  };

  // ---- initial slacks / multipliers.  The multipliers start at the scale of the cost gradient of the first iterate, lambda_0 =
  // max(1, 1e-6 |P y_0|_inf): with lambda = 1 a problem whose optimal cost is 1e8 and more (durations close to infeasibly short)
  // spends its first 20-30 steps growing them by a decade per five steps, every step short -- which is also what the
  // infeasibility rule below looks for: 12 of 34 791 random problems (tests/soak/soak_qp.py, all at durations x 0.3) were FEASIBLE
  // and called infeasible at step 30; none is now, and 512 problems at durations x 0.2 take 12.0 steps instead of 21.2.  Ordinary
  // problems (|P y_0| < 1e6: the bench's sets keep every verdict and step count) start as before; 1e-5 already costs them 0.1 step.
#ifndef ANET_IPM_LAM0_SCALE
#define ANET_IPM_LAM0_SCALE 1e-6
#endif
  auto first_iterate = [&]() -> double {  // slacks and multipliers of the first iterate; returns the number of live rows
  if (tid < 32) red[tid] = 0.0;
  __syncthreads();
  {
    double l_p = 0.0;
    for (int e = tid; e < NY; e += nt) {
      const int k = e / BK, ax = (e / S) % 3, d = e % S;
      if (pinned(k, d)) continue;
      double vp = 0.0;
      for (int side = 0; side < 2; ++side) {
        const int i = side == 0 ? k : k - 1;  // the piece that starts / ends at knot k
        if (i < 0 || i >= N) continue;
        const int m = side == 0 ? d : S + d;
        const double *ui = uu + (size_t)i * NB + ax * D;
        double g = 0.0;
        for (int m2 = 0; m2 < D; ++m2) g += Hobj[m * D + m2] * ui[m2];
        vp += sc[i * D + m] * qsv[i] * g;
      }
      l_p = fmax(l_p, fabs(vp));
    }
    block_reduce(1.0 / fmax(l_p, 1e-300), 3, true);  // (max via min of reciprocals)
  }
  __syncthreads();
  const double lam0 = uni(fmax(1.0, ANET_IPM_LAM0_SCALE * red[3]));
  __syncthreads();
  int64_t nrows_local = 0;
  for (int smp = fresh_tid(); smp < NS; smp += nt) {
    const int i = smp / R, j = smp % R;
    double s3[3][3];
    state_of(uu, i, j, s3);
    for (int q = 0; q < RPS; ++q) {  // inert rows keep s = 1, lambda = 0 and are never visited again
      slg[smp + (int64_t)q * NS] = 1.0;
      lmg[smp + (int64_t)q * NS] = 0.0;
    }
    for_rows(i, [&](int q, auto row, double hv) {
      const double gy = row.dot(s3[row.dsel]);
      slg[smp + (int64_t)q * NS] = fmax(hv - gy, 1.0);
      lmg[smp + (int64_t)q * NS] = lam0;
      ++nrows_local;
    });
  }
  if (tid < 32) red[tid] = 0.0;
  __syncthreads();
  block_reduce((double)nrows_local, 0, false);
  __syncthreads();
  const double mr = uni(fmax(red_sum(0), 1.0));
  __syncthreads();
  return mr;
  };
  const double mrows = resume ? uni(ct[NY]) : first_iterate();  // (a resumed problem: parked with the rest of its state)

  // assemble the node-space vector  out_k = sum over the pieces touching knot k of sc * (qs Hobj u + sum_j gamma_j h_j)
  // (gamma at acc offset `goff`; with_obj adds the cost gradient); pinned components are zeroed
  // (goff2 >= 0: gamma_j + w2 * gamma2_j with gamma2 at acc offset goff2; `sgn` scales the result)
  // Two adjacent threads per component: one takes the piece that starts at the knot, the other the piece that ends there (each
  // a loop over the R samples of its piece); the pair's sum by a DPP swap.  (One thread per component left 108 of 256 threads
  // walking both pieces: 9 k cycles per call at 8 pieces.)
  auto node_vector = [&](double *out, int goff, bool with_obj, const double *u, int goff2 = -1, double w2 = 0.0, double sgn = 1.0) {
    for (int e2 = fresh_tid(); e2 < 2 * ((NY + 127) / 128) * 128; e2 += nt) {  // (whole pairs: both lanes of a pair run the swap)
      const int e = e2 >> 1, side = e2 & 1;
      const bool live = e < NY;
      const int k = live ? e / BK : 0, ax = (e / S) % 3, d = e % S;
      double v = 0.0;
      const int i = side == 0 ? k : k - 1;  // piece whose start (side 0) / end (side 1) is knot k
      if (live && !pinned(k, d) && i >= 0 && i < N) {
        const int m = side == 0 ? d : S + d;
        double g = 0.0;
        if (with_obj) {
          const double qs = qsv[i];
          const double *ui = u + (size_t)i * NB + ax * D;
          for (int m2 = 0; m2 < D; ++m2) g += qs * Hobj[m * D + m2] * ui[m2];
        }
        if (goff2 < 0) {
          for (int j = 0; j < R; ++j) {
            const double *ga = acc + (size_t)(i * R + j) * AS + goff;
            const double *hj = ht + (size_t)j * HS;
            g += ga[ax] * hj[m] + ga[3 + ax] * hj[D + m] + ga[6 + ax] * hj[2 * D + m];
          }
        } else {
          for (int j = 0; j < R; ++j) {
            const double *ga = acc + (size_t)(i * R + j) * AS + goff, *gb = acc + (size_t)(i * R + j) * AS + goff2;
            const double *hj = ht + (size_t)j * HS;
            g += (ga[ax] + w2 * gb[ax]) * hj[m] + (ga[3 + ax] + w2 * gb[3 + ax]) * hj[D + m] +
                 (ga[6 + ax] + w2 * gb[6 + ax]) * hj[2 * D + m];
          }
        }
        v = sc[i * D + m] * g;
      }
      v = pair_sum(v);
      if (live && side == 0) out[e] = sgn * v;
    }
  };

  // What a Newton step needs besides the factor -- the dual residual P y + q + G'lambda with its scale |P y|, y'P y and
  // the affine right-hand side -(P y + q) - G'(lambda + w (G y - h)) -- depends on pass A only, as the Newton matrix
  // does: the two waves that have no part in the factorisation compute all of it while the chains are walked (one visit of
  // the per-sample sums for both vectors; maxima and sums into red[8..10], zeroed at the top of the step).
  auto side_vectors = [&]() {
    const int t2 = fresh_tid() - 128;
    double l_rd = 0.0, l_py = 0.0, l_obj = 0.0;
    for (int e = t2; e < NY; e += 128) {
      const int k = e / BK, ax = (e / S) % 3, d = e % S;
      double vp = 0.0, vr = 0.0, va = 0.0;
      for (int side = 0; side < 2; ++side) {
        const int i = side == 0 ? k : k - 1;  // piece whose start (side 0) / end (side 1) is knot k
        if (i < 0 || i >= N) continue;
        const int m = side == 0 ? d : S + d;
        const double qs = qsv[i];
        const double *ui = uu + (size_t)i * NB + ax * D;
        double g = 0.0;
        for (int m2 = 0; m2 < D; ++m2) g += qs * Hobj[m * D + m2] * ui[m2];
        double gr = g, ga = g;
        for (int j = 0; j < R; ++j) {
          const double *ac = acc + (size_t)(i * R + j) * AS;
          const double *hj = ht + (size_t)j * HS;
          gr += ac[21 + ax] * hj[m] + ac[24 + ax] * hj[D + m] + ac[27 + ax] * hj[2 * D + m];
          ga += ac[12 + ax] * hj[m] + ac[15 + ax] * hj[D + m] + ac[18 + ax] * hj[2 * D + m];
        }
        const double scm = sc[i * D + m];
        vp += scm * g;
        vr += scm * gr;
        va += scm * ga;
      }
      const bool pin = pinned(k, d);
      l_rd = fmax(l_rd, pin ? 0.0 : fabs(vr));
      l_py = fmax(l_py, fabs(vp));
      dya[e] = -(pin ? 0.0 : va);
    }
    for (int e = t2; e < N * NB; e += 128) {
      const int i = e / NB, ax = (e % NB) / D, m = e % D;
      const double *ui = uu + (size_t)i * NB + ax * D;
      double g = 0.0;
      for (int m2 = 0; m2 < D; ++m2) g += Hobj[m * D + m2] * ui[m2];
      l_obj += qsv[i] * ui[m] * g;
    }
    block_reduce(1.0 / fmax(l_rd, 1e-300), 8, true);
    block_reduce(1.0 / fmax(l_py, 1e-300), 9, true);  // scale of the dual residual: |P y| in node space
    block_reduce(l_obj, 10, false);                   // y'P y, the scale of the complementarity test
  };

  // TWISTED block Cholesky of (Dg, Of) in place, K = T T': the chain of knots is eliminated from BOTH ends towards the
  // middle knot PT by two waves at once -- wave 0 walks k = 0 .. PT-1 (T has the blocks L_k on and L_{k+1,k} below the
  // diagonal), wave 1 walks k = N .. PT+1 (L_k on and M_{k-1} = A_{k,k-1}' L_k^-T ABOVE the diagonal) --, then wave 0
  // factors the middle block, which takes a Schur update from either side.  No fill-in, the same flops, half the
  // sequential depth (every step of a chain depends on the previous one, and the chain is what this phase costs).
  // Of[k] ends up holding L_{k+1,k} for k < PT and M_k for k >= PT, both row-major.
  // Within a block everything is IN REGISTERS: lane r holds row r of the current block, values of other lanes arrive as
  // wave-uniform scalars through v_readlane (lane indices are compile-time constants after unrolling).  The blocks are
  // 9x9 / 12x12: going through LDS (write, read back) for each column cost ~10x more.
  auto rl = [](double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
  };
  const int PT = N >= a.twist_min_pieces ? N / 2 : N;  // PT = N: one chain, the classic order
  // Dr (row `lane` of an SPD block) -> row `lane` of its Cholesky factor; dinv = 1 / diagonal (wave-uniform)
  auto chol_rows = [&](const int lane, double (&Dr)[BK], double (&dinv)[BK]) {
#pragma unroll
    for (int c = 0; c < BK; ++c) {
      // pivot and its reciprocal from v_rsq_f64 + two Newton steps (a sqrt and a division cost ~70 instructions
      // on this sequential path, twelve times per block)
      const double dcc = fmax(rl(Dr[c], c), 1e-300);
      double rs = __builtin_amdgcn_rsq(dcc);
      rs = rs * __builtin_fma(-0.5 * dcc * rs, rs, 1.5);
      rs = rs * __builtin_fma(-0.5 * dcc * rs, rs, 1.5);
      dinv[c] = rs;
      const double piv = dcc * rs;
      Dr[c] = (lane == c) ? piv : Dr[c] * dinv[c];  // column c of L (rows > c); upper part is never read
#pragma unroll
      for (int c2 = c + 1; c2 < BK; ++c2) Dr[c2] -= Dr[c] * rl(Dr[c], c2);
    }
  };
  // D -= X X' for BK x BK blocks in LDS (X row-major), by the whole wave on the matrix cores: v_mfma_f64_16x16x4_f64 takes
  // A[i = lane & 15][k = lane >> 4] and B[k = lane >> 4][j = lane & 15] -- for X X' the same register, X[lane & 15][4 ks + (lane >> 4)],
  // zero outside the block -- and leaves C[row = (lane >> 4) + 4 r][col = lane & 15] in its r-th result; every entry of D is
  // touched by exactly one (lane, r).  Three instructions and three LDS reads per lane instead of BK^2 FMAs with BK^2
  // broadcast reads on BK of the 64 lanes -- on the one path of a Newton step that is sequential in earnest.
  // (LDS instructions of one wave execute in order: the row loads that follow see the update.)
  typedef double d4_t __attribute__((ext_vector_type(4)));
  auto schur_mfma = [&](const int lane, const double *X, double *Dk) {
    const int li = lane & 15, lk = lane >> 4;
    d4_t c4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < (BK + 3) / 4; ++ks) {
      const int k = 4 * ks + lk;
      const double x = (li < BK && k < BK) ? X[li * BK + k] : 0.0;
      c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c4, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = lk + 4 * r;
      if (row < BK && li < BK) Dk[row * BK + li] -= c4[r];
    }
  };
  auto twisted_factor = [&](auto &&side) {
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));  // (keeps the per-lane LDS addresses of this phase from being hoisted out of the Newton loop)
    const bool act = lane < BK;
    const int dir = wv == 0 ? 1 : -1, kfrom = wv == 0 ? 0 : N;
#pragma nounroll
    for (int ph = 0; ph < 2; ++ph) {
      if (ph == 1) __syncthreads();
      if (wv >= 2) {
        if (ph == 0) side();
        continue;
      }
      if (ph == 1 && wv == 1) continue;
      const int kbeg = ph == 0 ? kfrom : PT, kend = ph == 0 ? PT : PT + 1;  // (the chain of phase 1: knot PT, one step of +1)
      const int stepk = ph == 0 ? dir : 1;
#pragma nounroll
      for (int k = kbeg; k != kend; k += stepk) {
        double *Dk = Dg + (size_t)k * BK * BK;
        double Dr[BK], dinv[BK];
        // Schur updates: from the knot eliminated before this one in its chain; the middle knot from both sides
        const int src0 = ph == 0 ? (k == kfrom ? -1 : (dir > 0 ? k - 1 : k)) : (PT > 0 ? PT - 1 : -1);
        const int src1 = ph == 0 ? -1 : (PT < N ? PT : -1);
#pragma nounroll
        for (int u = 0; u < 2; ++u) {
          const int src = u == 0 ? src0 : src1;
          if (src < 0) continue;
          schur_mfma(lane, Of + (size_t)src * BK * BK, Dk);
        }
#pragma unroll
        for (int c = 0; c < BK; ++c) Dr[c] = act ? Dk[lane * BK + c] : (c == 0 ? 1.0 : 0.0);
        chol_rows(lane, Dr, dinv);
        // ---- L_k -> L_k^-1 right here, in registers: lane j = column j of the inverse by a right-looking forward
        // substitution on e_j; the entries of L (lane c holds row c) arrive as wave-uniform scalars through v_readlane.
        // The substitutions of a step walk every knot four times and with L_k each visit is a triangular solve of BK
        // dependent steps; with L_k^-1 it is a block-vector product (twisted_solve).  The inverse also gives the block
        // towards the next knot as a PRODUCT instead of a second substitution (below).  Nothing else reads the factors.
#ifndef ANET_IPM_INV_FROM_LDS
#define ANET_IPM_INV_FROM_LDS 1
#endif
        if constexpr (ANET_IPM_INV_FROM_LDS != 0) {  // L through LDS: the rows are stored, its entries come back as broadcasts
          if (act) {
#pragma unroll
            for (int c = 0; c < BK; ++c) Dk[lane * BK + c] = Dr[c];
          }
        }
        double pcol[BK];
#pragma unroll
        for (int c = 0; c < BK; ++c) pcol[c] = (c == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < BK; ++q) {
          pcol[q] *= dinv[q];
#pragma unroll
          for (int c = q + 1; c < BK; ++c) {
            if constexpr (ANET_IPM_INV_FROM_LDS != 0) pcol[c] -= Dk[c * BK + q] * pcol[q];
            else pcol[c] -= rl(Dr[q], c) * pcol[q];  // L[c][q] = lane c's Dr[q]
          }
        }
        if (act) {
#pragma unroll
          for (int c = 0; c < BK; ++c) Dk[c * BK + lane] = pcol[c];  // column `lane` of L_k^-1 (zero above the diagonal)
        }
        if (ph == 1) continue;
        // the block towards the next knot of the chain: L_{k+1,k} = A_{k+1,k} L_k^-T, or M_{k-1} = A_{k,k-1}' L_k^-T for
        // the chain that walks down -- out[r][q] = sum_c A[r][c] (or A[c][r]) Linv[q][c]: one 16x16x4 FP64 MFMA per four
        // values of c (operands straight from LDS, A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15] = Linv[j][k]), where
        // the substitution over the columns of L_k was BK (BK - 1) / 2 dependent LDS-broadcast FMAs per knot.
        double *Lo = Of + (size_t)(dir > 0 ? k : k - 1) * BK * BK;
        const int st_c = dir > 0 ? 1 : BK, st_l = dir > 0 ? BK : 1;
        {
          const int li = lane & 15, lk = lane >> 4;
          d4_t c4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < (BK + 3) / 4; ++ks) {
            const int kk = 4 * ks + lk;
            const bool in = li < BK && kk < BK;
            const double av = in ? Lo[li * st_l + kk * st_c] : 0.0;
            const double bv = in ? Dk[li * BK + kk] : 0.0;
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c4, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            if (row < BK && li < BK) Lo[row * BK + li] = c4[r];
          }
        }
      }
    }
  };
  // x <- K^-1 x with the twisted factor (diagonal blocks inverted, see twisted_factor): T z = x from both ends to the
  // middle, T'x = z from the middle outwards.  A chain keeps the vector it carries from knot to knot IN REGISTERS (lane r =
  // component r; the other lanes' values arrive through v_readlane) and the rows of the two blocks a knot needs -- off the
  // chain: they do not depend on the vector -- are requested one knot ahead, so what a knot costs is two products of a
  // BK x BK block with a vector and nothing waits on LDS in between (the substitution with L_k itself was BK dependent
  // readlane-scale-update steps per knot and an LDS round trip for the neighbour's result: 21 k cycles per solve at 8 pieces).
  auto twisted_solve = [&](double *x) {
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));  // (keeps the per-lane LDS addresses of this phase from being hoisted out of the Newton loop)
    const bool act = lane < BK;
    const int lr = act ? lane : 0;  // (idle lanes read row / column 0 and store nothing)
    const int dir = wv == 0 ? 1 : -1, kfrom = wv == 0 ? 0 : N;
    auto matvec = [&](const double (&Mrow)[BK], const double v) {  // sum_c Mrow[c] v_c, v_c = lane c's v
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < BK; ++c) acc = __builtin_fma(Mrow[c], rl(v, c), acc);
      return acc;
    };
    // row lr of block `blk` (st = BK) or column lr (st = 1 -> element [c][lr] at c * BK + lr)
    auto load_row = [&](const double *blk, double (&dst)[BK]) {
#pragma unroll
      for (int c = 0; c < BK; ++c) dst[c] = blk[lr * BK + c];
    };
    auto load_col = [&](const double *blk, double (&dst)[BK]) {
#pragma unroll
      for (int c = 0; c < BK; ++c) dst[c] = blk[c * BK + lr];
    };
    double zmid = 0.0;  // wave 0: the middle knot's value, carried from the forward to the backward phase
    // ---- forward, phase 0: the two chains; z_k = L_k^-1 (x_k - X z_prev), X = L_{k,k-1} (wave 0: rows of Of[k-1]) or M_k (wave 1: rows of Of[k])
    double zprev = 0.0;
    if (wv < 2 && kfrom != PT) {
      double Lr[BK], Xr[BK], Ln[BK], Xn[BK];
      load_row(Dg + (size_t)kfrom * BK * BK, Lr);
#pragma unroll
      for (int c = 0; c < BK; ++c) Xr[c] = 0.0;
      double xin = act ? x[kfrom * BK + lane] : 0.0;
#pragma nounroll
      for (int k = kfrom; k != PT; k += dir) {
        const int kn = k + dir;
        const bool more = kn != PT;
        double xnext = 0.0;
        if (more) {  // the next knot's blocks and right-hand side: in flight during this knot's arithmetic
          load_row(Dg + (size_t)kn * BK * BK, Ln);
          load_row(Of + (size_t)(dir > 0 ? kn - 1 : kn) * BK * BK, Xn);
          xnext = act ? x[kn * BK + lane] : 0.0;
        }
        const double xr = xin - matvec(Xr, zprev);
        const double z = matvec(Lr, xr);
        if (act) x[k * BK + lane] = z;
        zprev = z;
        if (more) {
#pragma unroll
          for (int c = 0; c < BK; ++c) {
            Lr[c] = Ln[c];
            Xr[c] = Xn[c];
          }
          xin = xnext;
        }
      }
    }
    __syncthreads();
    // ---- forward, phase 1 and backward, phase 0: wave 0 on the middle knot (Schur terms from both neighbours)
    if (wv == 0) {
      double Lr[BK], Xr[BK];
      double xr = act ? x[PT * BK + lane] : 0.0;
      if (PT > 0) {  // own chain: z_{PT-1} is in registers
        load_row(Of + (size_t)(PT - 1) * BK * BK, Xr);
        xr -= matvec(Xr, zprev);
      }
      if (PT < N) {  // the other chain's last knot: through LDS (behind the barrier)
        load_row(Of + (size_t)PT * BK * BK, Xr);
        const double zo = act ? x[(PT + 1) * BK + lane] : 0.0;
        xr -= matvec(Xr, zo);
      }
      load_row(Dg + (size_t)PT * BK * BK, Lr);
      const double z = matvec(Lr, xr);
      load_col(Dg + (size_t)PT * BK * BK, Lr);
      zmid = matvec(Lr, z);  // x_PT = L^-T z
      if (act) x[PT * BK + lane] = zmid;
    }
    __syncthreads();
    // ---- backward, phase 1: the chains outwards; x_k = L_k^-T (z_k - X' x_next), X' = columns of Of[k] (wave 0) / Of[k-1] (wave 1)
    if (wv < 2 && kfrom != PT) {
      double Lc[BK], Xc[BK], Ln[BK], Xn[BK];
      const int k0 = PT - dir;
      double xnb = wv == 0 ? zmid : (act ? x[PT * BK + lane] : 0.0);  // the inner neighbour's value
      load_col(Dg + (size_t)k0 * BK * BK, Lc);
      load_col(Of + (size_t)(dir > 0 ? k0 : k0 - 1) * BK * BK, Xc);
      double zin = act ? x[k0 * BK + lane] : 0.0;
#pragma nounroll
      for (int k = k0; k != kfrom - dir; k -= dir) {
        const int kn = k - dir;
        const bool more = kn != kfrom - dir;
        double znext = 0.0;
        if (more) {
          load_col(Dg + (size_t)kn * BK * BK, Ln);
          load_col(Of + (size_t)(dir > 0 ? kn : kn - 1) * BK * BK, Xn);
          znext = act ? x[kn * BK + lane] : 0.0;
        }
        const double xr = zin - matvec(Xc, xnb);
        const double xo = matvec(Lc, xr);
        if (act) x[k * BK + lane] = xo;
        xnb = xo;
        if (more) {
#pragma unroll
          for (int c = 0; c < BK; ++c) {
            Lc[c] = Ln[c];
            Xc[c] = Xn[c];
          }
          zin = znext;
        }
      }
    }
  };

  // Newton matrix of the current iterate into (Dg, Of) from the per-sample weights acc[0..11] of pass A
  auto assemble_newton = [&]() {
    // Newton matrix: per piece and pair (m, m') of Hermite basis functions the twelve weighted sums over the
    // samples (six corridor 3x3 entries, three velocity, three acceleration weights) are formed once and
    // scattered to the nine axis pairs of the node blocks -- 18 LDS reads per 12 results, where one thread per
    // matrix entry needed 9 per result.
    for (int e = fresh_tid(); e < (2 * N + 1) * BK * BK; e += nt) Dg[e] = 0.0;  // Of follows Dg
    __syncthreads();
    for (int w = fresh_tid(); w < N * D * D; w += nt) {
      const int i = w / (D * D), m = (w / D) % D, m2 = w % D;
      if (m < S && m2 >= S) continue;  // upper off-diagonal block: the transpose of the stored one
      double Sa[6] = {0, 0, 0, 0, 0, 0}, Sd[3] = {0, 0, 0};
      for (int j = 0; j < R; ++j) {
        const double *as = acc + (size_t)(i * R + j) * AS;
        const double *hj = ht + (size_t)j * HS;
        const double p0 = hj[m] * hj[m2], p1 = hj[D + m] * hj[D + m2], p2 = hj[2 * D + m] * hj[2 * D + m2];
#pragma unroll
        for (int q = 0; q < 6; ++q) Sa[q] += as[q] * p0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) Sd[ax] += as[6 + ax] * p1 + as[9 + ax] * p2;
      }
      const double scale = sc[i * D + m] * sc[i * D + m2], ob = qsv[i] * Hobj[m * D + m2];
#pragma unroll
      for (int axr = 0; axr < 3; ++axr)
#pragma unroll
        for (int axc = 0; axc < 3; ++axc) {
          const int wa = axr <= axc ? (axr == 0 ? axc : (axr == 1 ? 2 + axc : 5)) : (axc == 0 ? axr : (axc == 1 ? 2 + axr : 5));
          const double v = scale * (Sa[wa] + (axr == axc ? Sd[axr] + ob : 0.0));
          if (m < S) atomicAdd(&Dg[(size_t)i * BK * BK + (axr * S + m) * BK + axc * S + m2], v);
          else if (m2 >= S) atomicAdd(&Dg[(size_t)(i + 1) * BK * BK + (axr * S + m - S) * BK + axc * S + (m2 - S)], v);
          else Of[(size_t)i * BK * BK + (axr * S + m - S) * BK + axc * S + m2] = v;
        }
    }
    __syncthreads();
    for (int e = fresh_tid(); e < (2 * N + 1) * BK * BK; e += nt) {  // pinned components, diagonal regularisation
      const int blk = e / (BK * BK), r = (e % (BK * BK)) / BK, c = e % BK;
      const bool diag = blk <= N;
      const int k = diag ? blk : blk - (N + 1);
      const int kr = diag ? k : k + 1, kc = k;
      double v = Dg[e];
      if (pinned(kr, r % S) || pinned(kc, c % S)) v = (diag && r == c) ? 1.0 : 0.0;
      if (diag && r == c) v += 1e-13 * fabs(v) + 1e-300;
      Dg[e] = v;
    }
  };

  int it = 0, status = -2, accepted_steps = 0;  // OSQP_MAX_ITER_REACHED unless decided below
  double pres = 0.0, dres = 0.0, mu = 0.0, mu0 = 0.0, pres_mark = INFINITY, mu_mark = INFINITY, alpha_win = 0.0;
  int stalled_windows = 0;
  double objn_last = 0.0, alpha_last = 1.0;
  bool stopped = false;
  if (resume) {  // what the first launch parked (below, at its it_stop-th step)
    it = (int)ct[NY + 1];
    accepted_steps = (int)ct[NY + 2];
    stalled_windows = (int)ct[NY + 3];
    mu0 = ct[NY + 4];
    pres_mark = ct[NY + 5];
    mu_mark = ct[NY + 6];
    alpha_win = ct[NY + 7];
    it = __builtin_amdgcn_readfirstlane(it);
  }
  const int it_first = it;
#ifdef ANET_IPM_PROF
  long long prof_t_ = __builtin_readcyclecounter();
  if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[13] = prof_t_ - prof_start_;  // tables, first iterate
#endif
  IPM_TICK(0);
  // ---- pass A: residuals, weights, right-hand-side pieces per sample.  A row of the iterate (gy = its value, sl, lm):
  auto row_A = [&](auto row, double hv, double gy, double sl, double lm, double *A_, double &l_mu, double &l_pres, double &l_h) {
    const double rg = gy + sl - hv, w = lm * fast_rcp1(sl), t = lm + w * (gy - hv);
    l_mu += sl * lm;
    l_pres = fmax(l_pres, fabs(rg));
    l_h = fmax(l_h, fabs(hv));
    row.weights(w, A_);
    row.axpy(t, A_ + 12 + row.dsel * 3);
    row.axpy(lm, A_ + 21 + row.dsel * 3);
  };
  // As a pass of its own for the starting point of the FUSE form (every later iterate gets it from the updating pass of
  // the step that produced it: the rows are in registers there, and their values at the new point are gy + alpha ge);
  // red[0..2] must be zero.  (The unfused form has the same pass written out at the top of its loop: called through this
  // lambda there, the bounded instantiation came out 14 % slower -- 13.2 against 11.6 ms for 4096 8-segment problems --
  // from an identical instruction mix; rocprofv3 and the cycle counters of the sections show nothing else that differs.)
  auto pass_A = [&]() {
    double l_mu = 0.0, l_pres = 0.0, l_h = 0.0;
    for (int smp = fresh_tid(); smp < NS; smp += nt) {
      const int i = smp / R, j = smp % R;
      double s3[3][3];
      state_of(uu, i, j, s3);
      double A_[30];
#pragma unroll
      for (int q = 0; q < 30; ++q) A_[q] = 0.0;
      for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double hv, double sl, double lm) {
        row_A(row, hv, row.dot(s3[row.dsel]), sl, lm, A_, l_mu, l_pres, l_h);
      });
      double *as = acc + (size_t)smp * AS;
#pragma unroll
      for (int q = 0; q < 30; ++q) as[q] = A_[q];
    }
    block_reduce(l_mu, 0, false);
    block_reduce(1.0 / fmax(l_pres, 1e-300), 1, true);  // max via min of reciprocals -> stored as max
    block_reduce(1.0 / fmax(l_h, 1e-300), 2, true);
  };
  // Second opinion on "infeasible".  {y : G y <= h} (the equalities are built into the node coordinates, pinned components are
  // constants) is EMPTY iff some lambda >= 0 has G_free' lambda = 0 and lambda' h_eff < 0, h_eff = h - G_pinned y_pinned (Farkas).
  // The multipliers of a diverging interior-point iteration tend to such a ray, and with the iterate's y at hand
  //   lambda' h_eff = lambda'(h - G y) + (G_free' lambda)' y_free        (exactly),
  // while for ANY feasible y0:  lambda' h_eff >= -|G_free' lambda|_inf |y0_free|_1.  So a multiplier vector with
  //   g := |G_free' lambda|_inf <= eps_g |lambda|_inf   and   lambda' h_eff <= -eps_h |lambda|_inf
  // proves that no feasible point exists within |y_free|_1 < (eps_h / eps_g) |lambda' h_eff| / (eps_h |lambda|_inf) ... i.e. inside
  // the radius R = -lambda' h_eff / g (OSQP's primal-infeasibility test has this form on y(k+1) - y(k), eps_prim_inf = 1e-4).
  // The window heuristic's suspicion is reported only when the multipliers pass this test with eps_g = eps_h = ANET_IPM_CERT_EPS
  // (1e-3: R >= |lambda' h_eff| / (1e-3 |lambda|_inf)); otherwise the iteration goes on and the suspicion is re-examined at the
  // next window.  Measured (profiles/r05_qp_second_opinion.txt): on the bench's sets every verdict and nearly every step count
  // is unchanged; demanding R >= |y|_1 of the iterate (a proof for every trajectory of the iterate's own size) delays the
  // verdicts of 8-piece snap problems from step 30-50 to 60-120 and the 4096-problem batch from 8.0 to 10.8 ms.
  // Uses acc slots 21..29 (sum of lambda c per sample, this iterate's pass A), dya as scratch, red[0..2] and the sum slots.
#ifndef ANET_IPM_CERT_EPS
#define ANET_IPM_CERT_EPS 1e-3
#endif
  auto certified_infeasible = [&]() -> bool {
    __syncthreads();
    if (tid < 32) red[tid] = 0.0;
    __syncthreads();
    node_vector(dya, 21, false, uu);
    double l_lh = 0.0, l_lam = 0.0;
    for (int smp = fresh_tid(); smp < NS; smp += nt) {
      const int i = smp / R, j = smp % R;
      double s3[3][3];
      state_of(uu, i, j, s3);
      for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double hv, double sl, double lm) {
        (void)q; (void)sl;
        l_lh += lm * (hv - row.dot(s3[row.dsel]));
        l_lam = fmax(l_lam, lm);
      });
    }
    __syncthreads();  // dya complete
    double l_g = 0.0, l_y = 0.0;
    for (int e = fresh_tid(); e < NY; e += nt) {
      const int k = e / BK, d = e % S;
      l_g = fmax(l_g, fabs(dya[e]));
      if (!pinned(k, d)) l_y += dya[e] * yv[e];  // (G_free' lambda)' y_free
    }
    block_reduce(l_lh, 0, false);
    block_reduce(l_y, 4, false);
    block_reduce(1.0 / fmax(l_lam, 1e-300), 1, true);
    block_reduce(1.0 / fmax(l_g, 1e-300), 2, true);
    __syncthreads();
    const double lh = uni(red_sum(0)) + uni(red_sum(4)), lam_max = uni(red[1]), g_max = uni(red[2]);  // lh = lambda' h_eff
    __syncthreads();
    if (tid < 32) red[tid] = 0.0;
    __syncthreads();
#ifdef ANET_IPM_CERT_TRACE
    if (tid == 0) printf("ipm cert problem %lld it %d: lh/lam %.3e g/lam %.3e\n", (long long)b, it, lh / lam_max, g_max / lam_max);
#endif
    return g_max <= ANET_IPM_CERT_EPS * lam_max && lh <= -ANET_IPM_CERT_EPS * lam_max;
  };
  if (tid < 32) red[tid] = 0.0;
  __syncthreads();
  if constexpr (FUSE) {
    pass_A();
    __syncthreads();
  }
  for (it = it_first; it < a.max_iter; ++it) {
    if (a.it_stop > 0 && it == a.it_stop && !resume) {  // park the iterate: node states here, slacks / multipliers where they are
      stopped = true;
      break;
    }
    if constexpr (!FUSE) {
      // ---- pass A: residuals, weights, right-hand-side pieces per sample
      if (tid < 32) red[tid] = 0.0;
      __syncthreads();
      double l_mu = 0.0, l_pres = 0.0, l_h = 0.0;
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double s3[3][3];
        state_of(uu, i, j, s3);
        double A_[30];
#pragma unroll
        for (int q = 0; q < 30; ++q) A_[q] = 0.0;
        for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double hv, double sl, double lm) {
          const double gy = row.dot(s3[row.dsel]);
          const double rg = gy + sl - hv, w = lm * fast_rcp1(sl), t = lm + w * (gy - hv);
          l_mu += sl * lm;
          l_pres = fmax(l_pres, fabs(rg));
          l_h = fmax(l_h, fabs(hv));
          row.weights(w, A_);
          row.axpy(t, A_ + 12 + row.dsel * 3);
          row.axpy(lm, A_ + 21 + row.dsel * 3);
        });
        double *as = acc + (size_t)smp * AS;
#pragma unroll
        for (int q = 0; q < 30; ++q) as[q] = A_[q];
      }
      block_reduce(l_mu, 0, false);
      block_reduce(1.0 / fmax(l_pres, 1e-300), 1, true);  // max via min of reciprocals -> stored as max
      block_reduce(1.0 / fmax(l_h, 1e-300), 2, true);
      __syncthreads();
    }
    IPM_TICK(1);
    mu = uni(red_sum(0) / mrows);
    if (it == 0) mu0 = mu;
    pres = uni(red[1] / fmax(1.0, red[2]));
    __syncthreads();
    if constexpr (FUSE)
      if (tid < 32) red[tid] = 0.0;  // (red[8..10]: the side vectors' sums; the barriers of the assembly lie in between)
    // ---- the Newton matrix and its factor; beside the factorisation the dual residual and the affine right-hand side
    assemble_newton();
    __syncthreads();
    IPM_TICK(3);
    twisted_factor(side_vectors);
    __syncthreads();
    IPM_TICK(5);
    dres = uni(red[8] / fmax(1.0, red[9]));
    const double objn = uni(red_sum(10));
    objn_last = objn;
    // (the duality gap is mu * rows: that, not mu, is what bounds the distance of the objective from the optimum)
    if (pres < a.tol && dres < a.tol && mu * mrows < a.tol * fmax(1.0, 0.5 * fabs(objn))) { status = 1; break; }
    // The backward pass asks for three digits more than a plain solve; a few percent of the problems stall above that
    // (and break down after ~150 more steps).  They are solved all the same: stop them a few steps after the accepted
    // tolerance was met.
    if (a.tol_accept > a.tol && pres < a.tol_accept && dres < a.tol_accept &&
        mu * mrows < a.tol_accept * fmax(1.0, 0.5 * fabs(objn)) && ++accepted_steps > 8) { status = 1; break; }
    if (!(mu == mu) || mu > 1e12 * fmax(mu0, 1.0)) { status = -3; break; }  // diverging: no strictly feasible point
    // ... which half of the infeasible problems only reach after ~100 steps: their primal residual creeps down by 10-25 %
    // per ten steps (5e-4 -> 2e-5 over a hundred), every step is short, and the complementarity measure GROWS from window
    // to window (0.3, 3, 10, 20, 60, ... 1e12) where a feasible problem's falls.  Call it early, but not on the primal
    // residual alone -- a feasible problem in a tight corridor can crawl for a while with short Mehrotra steps.  The
    // verdict needs TWO consecutive windows of ten steps in which no step was longer than half the way to the boundary and
    // either (a) the complementarity measure grew by more than 20 % while the primal residual (above the tolerance) lost
    // less than half -- a hard FEASIBLE problem also grows its complementarity measure at first, but its residual falls
    // by 50-90 % per window (traces: tools/qp_verdicts.py sets) --, or
    // (b) the primal residual (above 1e-4) lost less than 30 % and the complementarity measure did not halve.
    // Anything else keeps iterating and is left to the divergence test above or to the iteration limit.
#ifdef ANET_IPM_TRACE
    if (tid == 0 && blockIdx.x == ANET_IPM_TRACE && it % 5 == 0)
      printf("ipm trace it %d pres %.3e dres %.3e mu %.3e alpha_win %.3f obj %.6e\n", it, pres, dres, mu, alpha_win, objn);
#endif
    // (The "less than half" of rule (a), ANET_IPM_GROWTH_PRES.  With all multipliers starting at 1 the rule called 12 of 34 791 random
    // FEASIBLE problems infeasible at step 30 (tests/soak/qp_port_only.py); at 0.65 it was 3, at 0.75 2 -- and one infeasible problem of
    // the bench's 4096 was then told at step 90 instead of 50, which the batch waits for: 8.7 -> 9.6 ms.  The cure was the starting
    // point (lambda_0 above): none of the 34 791 since.  Left at 0.5.)
#ifndef ANET_IPM_GROWTH_PRES
#define ANET_IPM_GROWTH_PRES 0.5
#endif
    if (it % 10 == 0) {
      const bool shortsteps = it >= 20 && alpha_win < 0.5;
      const bool growth = shortsteps && pres > a.tol && pres > ANET_IPM_GROWTH_PRES * pres_mark && mu > 1.2 * mu_mark;
      const bool stall = shortsteps && pres > 1e-4 && pres > 0.7 * pres_mark && mu > 0.5 * mu_mark;
      stalled_windows = (growth || stall) ? stalled_windows + 1 : 0;
      // The windows are a HEURISTIC (1 feasible problem in ~4e5 random ones crawled like an infeasible one): what they suspect is
      // reported only with a second opinion that cannot be wrong that way -- a Farkas certificate from the multipliers.
      if (stalled_windows >= 2 && certified_infeasible()) { status = -3; break; }
      pres_mark = pres;
      mu_mark = mu;
      alpha_win = 0.0;
    }
    twisted_solve(dya);
    __syncthreads();
    IPM_TICK(6);
    to_u(dya, dua);
    if (tid < 32) red[tid] = 0.0;
    __syncthreads();
    // ---- pass B: the affine step length, the three sums of (s + a ds)'(lambda + a dlambda), AND the corrector's right-hand
    //      side.  A corrector row contributes c t with
    //        t = lambda + (lambda rg - rc) / s,  rc = s lambda + ds dlambda - mu_target
    //          = [lambda + (lambda rg - s lambda - ds dlambda) / s]  +  mu_target / s  =  t0 + mu_target / s,
    //      and only mu_target waits for the sums of this very pass: the per-sample sums of c t0 (slots 12..20) and of c / s
    //      (slots 0..8: the weights there were consumed by the assembly) are taken here and combined when the node vector
    //      is formed -- no second visit of the rows (it was a pass of its own, the third of five).
    {
      double l_ap = 0.0, l_s1 = 0.0, l_s2 = 0.0;  // l_ap: max of -ds/s, -dl/lambda = 1 / (step to the boundary)
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double s3[3][3], d3[3][3], G0[9], G1[9];
        state_of(uu, i, j, s3);
        state_of(dua, i, j, d3);
#pragma unroll
        for (int q = 0; q < 9; ++q) G0[q] = G1[q] = 0.0;
        for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double hv, double sl, double lm) {
          const double rg = row.dot(s3[row.dsel]) + sl - hv;
          const double ds = -rg - row.dot(d3[row.dsel]);
          const double isl = fast_rcp1(sl);
          const double dl = -lm - (lm * isl) * ds;
          // step to the boundary: the largest of -ds/s, -dl/lambda over the rows is 1/alpha (no division per row)
          l_ap = fmax(l_ap, fmax(-ds * isl, -dl * __builtin_amdgcn_rcp(lm)));  // (a step length: the raw reciprocal does, 1e-7)
          l_s1 += sl * dl + lm * ds;
          l_s2 += ds * dl;
          const double t0 = lm + (lm * rg - sl * lm - ds * dl) * isl;
          row.axpy(t0, G0 + row.dsel * 3);
          row.axpy(isl, G1 + row.dsel * 3);
        });
        double *as = acc + (size_t)smp * AS;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          as[12 + q] = G0[q];
          as[q] = G1[q];
        }
      }
      block_reduce(1.0 / fmax(l_ap, 1e-300), 3, true);
      block_reduce(l_s1, 4, false);
      block_reduce(l_s2, 5, false);
    }
    __syncthreads();
    IPM_TICK(7);
    const double a_aff = fmin(1.0, red[3] > 0.0 ? 1.0 / red[3] : 1.0);
    const double mu_aff = uni((mu * mrows + a_aff * red_sum(4) + a_aff * a_aff * red_sum(5)) / mrows);
    double sigma = mu_aff / mu;
    sigma = sigma * sigma * sigma;
    // Centring target sigma mu, but never below a tenth of the complementarity the stopping test asks for: a problem that
    // has met the primal and complementarity tests while its dual residual is still a digit short would otherwise be
    // pushed to mu ~ 1e-20, where lambda / s spans forty decades, the Newton matrix is numerically singular and the dual
    // residual bounces between 1e-9 and 10 for the rest of the iteration budget (tests/golden/vjp_snap_n2.npz).
    const double mu_target = uni(fmax(sigma * mu, a.tol_tenth * fmax(1.0, 0.5 * fabs(objn)) / mrows));
    // ---- corrector right-hand side (same factor)
    node_vector(dyc, 12, true, uu, 0, mu_target, -1.0);
    __syncthreads();
    IPM_TICK(9);
    twisted_solve(dyc);
    __syncthreads();
    IPM_TICK(10);
    to_u(dyc, duc);
    if (tid < 32) red[tid] = 0.0;
    __syncthreads();
    // ---- pass D: step length of the combined direction -----------------------------------------------
    // slack / multiplier directions of the combined step for one row
    auto final_dir = [&](auto row, double hv, const double (&s3)[3][3], const double (&d3)[3][3],
                         const double (&e3)[3][3], double sl, double lm, double &ds, double &dl) {
      const double rg = row.dot(s3[row.dsel]) + sl - hv;
      const double dsa = -rg - row.dot(d3[row.dsel]);
      const double isl = fast_rcp1(sl);
      const double dla = -lm - (lm * isl) * dsa;
      const double rc = sl * lm + dsa * dla - mu_target;
      ds = -rg - row.dot(e3[row.dsel]);
      dl = (-rc - lm * ds) * isl;
    };
    {
      double l_a = 0.0;  // 1 / (step to the boundary)
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double s3[3][3], d3[3][3], e3[3][3];
        state_of(uu, i, j, s3);
        state_of(dua, i, j, d3);
        state_of(duc, i, j, e3);
        for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double hv, double sl, double lm) {
          double ds, dl;
          final_dir(row, hv, s3, d3, e3, sl, lm, ds, dl);
          l_a = fmax(l_a, fmax(-ds * __builtin_amdgcn_rcp(sl), -dl * __builtin_amdgcn_rcp(lm)));  // (raw reciprocals: 0.99 of it is taken)
        });
      }
      block_reduce(1.0 / fmax(l_a, 1e-300), 6, true);
    }
    __syncthreads();
    IPM_TICK(11);
    // Fraction of the way to the boundary.  Measured (tools/qp_step_hist.py, 4096 problems): 0.995 saves 2-3 % of the Newton steps
    // of a batch (8.63 -> 8.55 ms, 5-piece jerk 4.79 -> 4.65) and costs the lone 8-piece problem of the bench one step more
    // (0.64 -> 0.69 ms); 0.999, or a fraction growing towards 1 with the complementarity falling, saves 6 % on average -- and
    // sends one problem in a thousand into a limit cycle at the rounding floor of the Newton solve (mu ~ 1e-9, the dual
    // residual bouncing between 1e-10 and 1e-7 with period 20: slacks that small make the weights lambda / s span 18 decades).
#ifndef ANET_IPM_STEP_FRACTION
#define ANET_IPM_STEP_FRACTION 0.99
#endif
    constexpr double step_fraction = ANET_IPM_STEP_FRACTION;
    const double alpha = uni(fmin(1.0, step_fraction * (red[6] > 0.0 ? 1.0 / red[6] : 1e300)));
    alpha_win = uni(fmax(alpha_win, alpha));
    alpha_last = alpha;
    __syncthreads();
    if constexpr (!FUSE) {
      // ---- pass E: update
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double s3[3][3], d3[3][3], e3[3][3];
        state_of(uu, i, j, s3);
        state_of(dua, i, j, d3);
        state_of(duc, i, j, e3);
        for_rows_sl(i, smp, RowsUpdate{}, [&](int q, auto row, double hv, double &sl, double &lm) {
          double ds, dl;
          final_dir(row, hv, s3, d3, e3, sl, lm, ds, dl);
          sl += alpha * ds;
          lm += alpha * dl;
        });
      }
    } else
    // ---- pass E: update, and pass A of the new iterate --------------------------------------------------------
    {
      double l_mu = 0.0, l_pres = 0.0, l_h = 0.0;
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double s3[3][3], d3[3][3], e3[3][3];
        state_of(uu, i, j, s3);
        state_of(dua, i, j, d3);
        state_of(duc, i, j, e3);
        double A_[30];
#pragma unroll
        for (int q = 0; q < 30; ++q) A_[q] = 0.0;
        for_rows_sl(i, smp, RowsUpdate{}, [&](int q, auto row, double hv, double &sl, double &lm) {
          // (final_dir, with the two row values kept: the row at the new point is gy + alpha ge)
          const double gy = row.dot(s3[row.dsel]), ge = row.dot(e3[row.dsel]);
          const double rg = gy + sl - hv;
          const double dsa = -rg - row.dot(d3[row.dsel]);
          const double isl = fast_rcp1(sl);
          const double dla = -lm - (lm * isl) * dsa;
          const double rc = sl * lm + dsa * dla - mu_target;
          const double ds = -rg - ge;
          const double dl = (-rc - lm * ds) * isl;
          sl += alpha * ds;
          lm += alpha * dl;
          row_A(row, hv, gy + alpha * ge, sl, lm, A_, l_mu, l_pres, l_h);
        });
        double *as = acc + (size_t)smp * AS;
#pragma unroll
        for (int q = 0; q < 30; ++q) as[q] = A_[q];
      }
      block_reduce(l_mu, 0, false);  // (red[0..2] are zero since the barrier ahead of pass D)
      block_reduce(1.0 / fmax(l_pres, 1e-300), 1, true);
      block_reduce(1.0 / fmax(l_h, 1e-300), 2, true);
    }
    __syncthreads();
    for (int e = fresh_tid(); e < NY; e += nt) yv[e] += alpha * dyc[e];
    __syncthreads();
    to_u(yv, uu);
    __syncthreads();
    IPM_TICK(12);
  }
  __syncthreads();
  if (stopped) {
    for (int e = fresh_tid(); e < NY; e += nt) ct[e] = yv[e];
    if (tid == 0) {
      ct[NY] = mrows;
      ct[NY + 1] = (double)it;
      ct[NY + 2] = (double)accepted_steps;
      ct[NY + 3] = (double)stalled_windows;
      ct[NY + 4] = mu0;
      ct[NY + 5] = pres_mark;
      ct[NY + 6] = mu_mark;
      ct[NY + 7] = alpha_win;
      // what the order of the second launch is made from: the residuals of the last tested iterate and its step
      ct[NY + 8] = pres;
      ct[NY + 9] = dres;
      ct[NY + 10] = mu * mrows / fmax(1.0, 0.5 * fabs(objn_last));
      ct[NY + 11] = alpha_last;
      a.status[b] = 0;  // still running
      a.iters[b] = it;
    }
    return;
  }
  // ---- backward pass through the optimum (anet_qp_solve_vjp) ---------------------------------------------------
  // The reference's hook solves the dense KKT system  J [d_z; d_lambda; d_nu] = -grad  (layers.py:129-141) and stops
  // there: its z is a detached leaf.  Here the same adjoint is taken in Hermite coordinates and carried through to the
  // durations.  On the central path y(T) solves F(y, T) = grad_y L(y, lambda(y, T); T) = 0 with lambda_r = mu / s_r, so
  //   dy/dT = -K^-1 dF/dT,  K = dF/dy = P + G' diag(lambda / s) G   -- the Newton matrix of the method, at the optimum;
  // with d_y = -K^-1 g_y and d_lambda_r = w_r g_r.d_y (w = lambda / s) the chain rule collapses to the DIRECTIONAL
  // DERIVATIVE of the time gradient of the Lagrangian (the closed form below) along (d_y, d_lambda), plus the explicit
  // dependence of the coefficients on T at fixed node states (c = Hm u T^-k, the (T_i/T_i+1)^d halves, the pinned ends).
  if (a.grad_z && a.vjpT) {
    const double *gz = a.grad_z + b * (int64_t)N * NB;
    // At a finite barrier parameter lambda / s is a SOFT weight: a row with a small multiplier is only enforced with
    // stiffness lambda^2 / mu, and the adjoint comes out wrong by the inverse of that (1e-3 relative on rows with
    // multipliers of 1e-3 at the tolerance the solve stops at).  The optimum itself tells which rows it touches:
    // s lambda = mu on the central path, so a touched row has lambda ~ lambda* and s = mu / lambda*, an untouched one
    // s ~ s* and lambda = mu / s* -- at mu ~ 1e-10 the two kinds are ten orders of magnitude apart in lambda / s, and
    // lambda > s separates them (a row is misjudged only if its multiplier AND its slack are below 1e-5: degenerate,
    // where the derivative does not exist either).  The adjoint system of the QP proper has the touched rows as
    // equalities g_r.d_y = 0 and ignores the others (the J of layers.py:129-141 with its diag(lambda), diag(Gz-h)
    // blocks): they get a penalty weight (made exact by the multiplier passes below), the others none.
    double pmax = 0.0;
    for (int i = 0; i < N; ++i)
      for (int m = 0; m < D; ++m) pmax = fmax(pmax, qsv[i] * Hobj[m * D + m] * sc[i * D + m] * sc[i * D + m]);
    const double w_act = 1.0e4 * pmax;
    for (int smp = fresh_tid(); smp < NS; smp += nt) {
      const int i = smp / R;
      double A_[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) A_[q] = 0.0;
      for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double, double sl, double lm) {
        row.weights((lm > sl) ? w_act : 0.0, A_);
      });
      double *as = acc + (size_t)smp * AS;
#pragma unroll
      for (int q = 0; q < 12; ++q) as[q] = A_[q];
    }
    __syncthreads();
    assemble_newton();
    __syncthreads();
    twisted_factor([] {});
    // g_u = d loss / d u_i (duc), g_y (dyc)
    for (int e = fresh_tid(); e < N * NB; e += nt) {
      const int i = e / NB, ax = (e % NB) / D, m = e % D;
      double v = 0.0, tk = 1.0;
      const double rT = 1.0 / Tn[i];
      for (int col = D - 1; col >= 0; --col) {  // k = D-1-col = 0, 1, ..: T^-k
        v += gz[(size_t)i * NB + ax * D + col] * tk * Hm[col * D + m];
        tk *= rT;
      }
      duc[e] = v;
    }
    __syncthreads();
    for (int e = fresh_tid(); e < NY; e += nt) {
      const int k = e / BK, ax = (e / S) % 3, d = e % S;
      double v = 0.0;
      if (!pinned(k, d)) {
        if (k < N) v += sc[k * D + d] * duc[(size_t)k * NB + ax * D + d];
        if (k > 0) v += sc[(k - 1) * D + S + d] * duc[(size_t)(k - 1) * NB + ax * D + S + d];
      }
      dyc[e] = v;
    }
    // (sums over samples and pieces below: every term is stored where its thread owns the slot and added up in index order by one
    //  thread per piece -- no atomics, the gradient is the same bit for bit from launch to launch)
    for (int i = fresh_tid(); i < 3 * N; i += nt) rhs[i] = 0.0;
    for (int smp = fresh_tid(); smp < NS; smp += nt) {
      double *as = acc + (size_t)smp * AS + 12;
#pragma unroll
      for (int q = 0; q < 9; ++q) as[q] = 0.0;
      as[-12] = 0.0;  // slots 0 / 1 of the record (weights, consumed by the assembly above): this sample's sums of d_lambda
      as[-11] = 0.0;  //   over its velocity / acceleration rows
    }
    __syncthreads();
    // Method of multipliers on  min 1/2 d'P d + g_y.d  s.t.  g_r.d = 0 (touched rows):  (P + w Ga'Ga) d_k = -g_y - Ga'nu_k,
    // nu_k+1 = nu_k + w Ga d_k.  The penalty alone would need w so large that the factorisation loses the small
    // curvatures of the cost (they span five orders of magnitude: 1e-4 .. 1e-3 relative error at w = 1e8 pmax); with
    // the multiplier update a moderate w does, the constraint residual falling by ~1e-4 per pass.  Only Ga'nu per
    // sample (acc[12..20]) and the per-piece sums of nu over the box rows are carried -- what the time gradient needs
    // of d_lambda = nu.
    for (int pass = 0; pass < 4; ++pass) {
      node_vector(dya, 12, false, uu);
      __syncthreads();
      for (int e = fresh_tid(); e < NY; e += nt) dya[e] = -dyc[e] - dya[e];
      __syncthreads();
      twisted_solve(dya);
      __syncthreads();
      to_u(dya, dua);
      __syncthreads();
      for (int smp = fresh_tid(); smp < NS; smp += nt) {
        const int i = smp / R, j = smp % R;
        double d3[3][3], G_[9], bsv = 0.0, bsa = 0.0;
        state_of(dua, i, j, d3);
#pragma unroll
        for (int q = 0; q < 9; ++q) G_[q] = 0.0;
        for_rows_sl(i, smp, RowsLoad{}, [&](int q, auto row, double, double sl, double lm) {
          const double dl = ((lm > sl) ? w_act : 0.0) * row.dot(d3[row.dsel]);
          row.axpy(dl, G_ + row.dsel * 3);
          if (row.dsel == 1) bsv += dl;
          if (row.dsel == 2) bsa += dl;
        });
        double *as = acc + (size_t)smp * AS + 12;
#pragma unroll
        for (int q = 0; q < 9; ++q) as[q] += G_[q];
        as[-12] += bsv;
        as[-11] += bsa;
      }
      __syncthreads();
    }
    for (int i = fresh_tid(); i < N; i += nt) {
      double sv = 0.0, sa = 0.0;
      for (int j = 0; j < R; ++j) {
        sv += acc[(size_t)(i * R + j) * AS];
        sa += acc[(size_t)(i * R + j) * AS + 1];
      }
      rhs[N + i] = sv;
      rhs[2 * N + i] = sa;
    }
    double *cA = Dg, *cB = Dg + (size_t)N * NB;  // per-entry terms of piece i / of its right neighbour (the factors are done with)
    for (int e = fresh_tid(); e < N * NB; e += nt) {
      const int i = e / NB, ax = (e % NB) / D, m = e % D, d = m % S;
      const double *ui = uu + (size_t)i * NB + ax * D, *vi = dua + (size_t)i * NB + ax * D;
      double hu = 0.0, hv = 0.0;
      for (int m2 = 0; m2 < D; ++m2) {
        hu += Hobj[m * D + m2] * ui[m2];
        hv += Hobj[m * D + m2] * vi[m2];
      }
      double t = (double)(1 - 2 * S) * qsv[i] * ui[m] * hv / Tn[i];   // along d_y of (1-2s) J_i / T_i
      {  // explicit T^-k of the coefficients: - k c_col g_z / T_i  (col = m as a running index over the D columns)
        const int col = m, k = D - 1 - col;
        double cc = 0.0;
        for (int m2 = 0; m2 < D; ++m2) cc += Hm[col * D + m2] * ui[m2];
        t -= (double)k * cc * pow(Tn[i], (double)(-k)) * gz[(size_t)i * NB + ax * D + col] / Tn[i];
      }
      cA[e] = t;
      cB[e] = 0.0;
      if (d == 0) continue;
      double g = qsv[i] * hu, dg = qsv[i] * hv;
      for (int j = 0; j < R; ++j) {
        const double *ga = acc + (size_t)(i * R + j) * AS;
        const double *hj = ht + (size_t)j * HS;
        g += ga[21 + ax] * hj[m] + ga[24 + ax] * hj[D + m] + ga[27 + ax] * hj[2 * D + m];
        dg += ga[12 + ax] * hj[m] + ga[15 + ax] * hj[D + m] + ga[18 + ax] * hj[2 * D + m];
      }
      const double c = (dg * ui[m] + g * vi[m] + duc[e] * ui[m]) * (double)d;
      if (m >= S) {
        if (i < N - 1) {
          cA[e] = t + c / Tn[i];
          cB[e] = -c / Tn[i + 1];
        } else if (d < 3) {
          cA[e] = t + c / Tn[i];
        }
      } else if (i == 0 && d < 3) {
        cA[e] = t + c / Tn[0];
      }
    }
    __syncthreads();
    for (int i = fresh_tid(); i < N; i += nt) {
      double r = 0.0;
      for (int q = 0; q < NB; ++q) r += cA[i * NB + q];
      if (i > 0)
        for (int q = 0; q < NB; ++q) r += cB[(i - 1) * NB + q];
      a.vjpT[b * N + i] = r - (a.vmax * rhs[N + i] + 2.0 * a.amax * Tn[i] * rhs[2 * N + i]);
    }
    __syncthreads();
  }
  // ---- time gradient of the optimal cost (anet_qp_solve_time_grad): envelope theorem in these coordinates ----
  // L = sum_i q_i 1/2 u_i'H_obj u_i + lambda'(G u - h) at fixed (y, lambda) depends on T through q_i = T_i^(1-2s),
  // the end halves u_i[S+d] = (T_i/T_i+1)^d y_i+1[d], the pinned boundary values y_0[d] = ini_d T_0^d,
  // y_N[d] = fin_d T_N-1^d, and the box bounds vmax T_i, amax T_i^2.  (acc[21..29] holds G'lambda per sample
  // for the iterate the loop stopped at.)
  if (a.gradT) {
    double *cA = Dg, *cB = Dg + (size_t)N * NB;  // as in the backward pass: terms stored per entry, added up in index order
    for (int e = fresh_tid(); e < N * NB; e += nt) {
      const int i = e / NB, ax = (e % NB) / D, m = e % D, d = m % S;
      const double *ui = uu + (size_t)i * NB + ax * D;
      double hu = 0.0;
      for (int m2 = 0; m2 < D; ++m2) hu += Hobj[m * D + m2] * ui[m2];
      const double t = (double)(1 - 2 * S) * 0.5 * qsv[i] * ui[m] * hu / Tn[i];  // (1-2s) J_i / T_i
      cA[e] = t;
      cB[e] = 0.0;
      if (d == 0) continue;
      double g = qsv[i] * hu;
      for (int j = 0; j < R; ++j) {
        const double *ga = acc + (size_t)(i * R + j) * AS + 21;
        const double *hj = ht + (size_t)j * HS;
        g += ga[ax] * hj[m] + ga[3 + ax] * hj[D + m] + ga[6 + ax] * hj[2 * D + m];
      }
      const double c = g * ui[m] * (double)d;
      if (m >= S) {
        if (i < N - 1) {
          cA[e] = t + c / Tn[i];
          cB[e] = -c / Tn[i + 1];
        } else if (d < 3) {
          cA[e] = t + c / Tn[i];
        }
      } else if (i == 0 && d < 3) {
        cA[e] = t + c / Tn[0];
      }
    }
    for (int smp = fresh_tid(); smp < NS; smp += nt) {
      const int i = smp / R;
      double sv = 0.0, sa = 0.0;
      for (int qq = 0; qq < 12; ++qq) {
        const double lm = lmg[smp + (int64_t)(M + qq) * NS];
        if ((qq % 4) & 1) sa += lm;
        else sv += lm;
      }
      acc[(size_t)smp * AS + 30] = -(a.vmax * sv + 2.0 * a.amax * Tn[i] * sa);  // (the record's spare slot)
    }
    __syncthreads();
    for (int i = fresh_tid(); i < N; i += nt) {
      double r = 0.0;
      for (int q = 0; q < NB; ++q) r += cA[i * NB + q];
      if (i > 0)
        for (int q = 0; q < NB; ++q) r += cB[(i - 1) * NB + q];
      for (int j = 0; j < R; ++j) r += acc[(size_t)(i * R + j) * AS + 30];
      a.gradT[b * N + i] = r;
    }
    __syncthreads();
  }
  // ---- report: coefficients c = Hm u / T^k, objective in original units ---------------------------------
  for (int e = fresh_tid(); e < N * NB; e += nt) {
    const int i = e / NB, ax = (e % NB) / D, col = e % D, k = D - 1 - col;
    const double *ui = uu + (size_t)i * NB + ax * D;
    double v = 0.0;
    for (int m = 0; m < D; ++m) v += Hm[col * D + m] * ui[m];
    a.coeffs[b * (int64_t)N * NB + e] = v * pow(Tn[i], (double)(-k));
  }
  if (tid == 0) {
    double obj = 0.0;
    for (int i = 0; i < N; ++i) {
      const double qs = qsv[i];
      for (int ax = 0; ax < 3; ++ax) {
        const double *ui = uu + (size_t)i * NB + ax * D;
        for (int m = 0; m < D; ++m)
          for (int m2 = 0; m2 < D; ++m2) obj += 0.5 * qs * ui[m] * Hobj[m * D + m2] * ui[m2];
      }
    }
    a.obj[b] = obj;
    a.status[b] = status;
    a.iters[b] = it;
    a.res[b * 2] = pres;
    a.res[b * 2 + 1] = dres;
  }
}

// The FUSE instantiations (lone problems, small batches) live in qp_ipm_fuse_unit.hip, scheduled for ILP (2-3 % of their latency;
// the throughput shapes lose 3 % under that strategy and stay in the main translation unit).  Returns the hipError_t of the
// attribute call / launch as an int.
int launch_qp_ipm_fuse(int s, int64_t batch, size_t lds_bytes, hipStream_t st, const IpmArgs &a);

}  // namespace anet
