// MINCO kernels: coefficient solve + energy (k_minco_solve), per-piece partial gradients with the
// penalty functional (k_piece_grad), adjoint propagation (k_minco_propagate).  One lane per trajectory
// (or per (trajectory, piece)); batch-minor global layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "minco_core.h"

namespace anet {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
struct SolveArgs {
  const double *head, *tail, *wps, *T;
  double *coeffs, *energy;
  int64_t B, ld;
  int N, c;
};

constexpr int kSolveBlock = 64;

// One lane = one trajectory.  The block-tridiagonal factor is shared by the three axes and stays
// in registers; the axes are swept one after the other so only one axis' right-hand side is live.
// NEXACT: the piece count is exactly NB (compile time); NPC >= 0: c-1 is NPC (compile time).
// Both let every end-node / pinned-derivative mask fold away; the generic instantiation
// (NEXACT = false, NPC = -1) keeps them as wave-uniform selects.
template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_solve(SolveArgs a) {
  constexpr int m = S - 1, D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * kSolveBlock + threadIdx.x;
  if (b >= a.B) return;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int c = np + 1;
  const int64_t ld = a.ld;

  Factor<S, NB> F;
  if constexpr (NEXACT) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < N) F.r[i] = fast_rcp(a.T[i * ld + b]);
  } else {  // runtime piece count: all loads first, none behind a branch (as in k_minco_solve_axis)
    double tt[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) tt[i] = a.T[(int64_t)(i < N ? i : 0) * ld + b];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < N) F.r[i] = fast_rcp(tt[i]);
  }
  F.factorize(N, np);

  double etot = 0.0;
#pragma unroll 1
  for (int ax = 0; ax < 3; ++ax) {
    double P[NB + 1], hv[m], tv[m], X[NB + 1][m];
    const double *hp = a.head + (int64_t)(ax * c) * ld + b;
    const double *tp = a.tail + (int64_t)(ax * c) * ld + b;
    if constexpr (NEXACT) {
#pragma unroll
      for (int k = 0; k <= NB; ++k) {
        if (k == 0)
          P[k] = hp[0];
        else if (k < N)
          P[k] = a.wps[(int64_t)((k - 1) * 3 + ax) * ld + b];
        else if (k == N)
          P[k] = tp[0];
        else
          P[k] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < m; ++j) {
        hv[j] = (j < np) ? hp[(int64_t)(1 + j) * ld] : 0.0;
        tv[j] = (j < np) ? tp[(int64_t)(1 + j) * ld] : 0.0;
      }
    } else {
#pragma unroll
      for (int k = 0; k <= NB; ++k) {
        const double *src = (k == 0) ? hp : (k < N) ? a.wps + (int64_t)((k - 1) * 3 + ax) * ld + b : tp;
        const double v = *src;
        P[k] = (k <= N) ? v : 0.0;
      }
#pragma unroll
      for (int j = 0; j < m; ++j) {
        const int64_t row = (j < np) ? 1 + j : 0;
        const double h = hp[row * ld], t = tp[row * ld];
        hv[j] = (j < np) ? h : 0.0;
        tv[j] = (j < np) ? t : 0.0;
      }
    }
    double *cp = a.coeffs ? a.coeffs + (int64_t)(ax * D) * ld + b : nullptr;
    etot += solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int piece, int col, double v) {
      if (cp) cp[(int64_t)(piece * 3 * D + col) * ld] = v;
    });
  }
  if (a.energy) a.energy[b] = etot;
}

// Small-batch variant: one lane per (trajectory, AXIS).  The three lanes of a trajectory factorise the
// (tiny) shared system redundantly and sweep their own axis, so the dependent chain per lane is ~2.4x
// shorter and three times as many waves are in flight -- what matters when the batch is too small to
// fill the chip (a 1024-trajectory launch is 16 waves on 1024 SIMDs with the kernel above).
// Lane order: gid = 3*b + axis, so the 3 lanes of a trajectory are adjacent and the energy is a
// 3-lane shuffle sum; loads/stores of a wave fall into three 21-trajectory segments.
template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_solve_axis(SolveArgs a) {
  constexpr int m = S - 1, D = 2 * S;
  const int64_t gid = (int64_t)blockIdx.x * 63 + threadIdx.x;  // 21 trajectories x 3 axes per wave
  const int64_t b = gid / 3;
  const int ax = (int)(gid % 3);
  const bool live = threadIdx.x < 63 && b < a.B;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int c = np + 1;
  const int64_t ld = a.ld;
  const int64_t bb = live ? b : 0;  // idle lanes compute on trajectory 0 and store nothing

  // Every load is issued before the first use and none sits behind a branch on the (runtime) piece count: with one
  // wave per SIMD each load-then-use pair is a full L2 round trip of dead time, and the generic instantiations had
  // fifteen of them here.
  Factor<S, NB> F;
  double P[NB + 1], hv[m], tv[m], X[NB + 1][m], tt[NB];
  const double *hp = a.head + (int64_t)(ax * c) * ld + bb;
  const double *tp = a.tail + (int64_t)(ax * c) * ld + bb;
#pragma unroll
  for (int i = 0; i < NB; ++i) tt[i] = a.T[(int64_t)(i < N ? i : 0) * ld + bb];
#pragma unroll
  for (int k = 0; k <= NB; ++k) {
    const double *src = (k == 0) ? hp : (k < N) ? a.wps + (int64_t)((k - 1) * 3 + ax) * ld + bb : tp;
    const double v = *src;
    P[k] = (k <= N) ? v : 0.0;
  }
#pragma unroll
  for (int j = 0; j < m; ++j) {
    const int64_t row = (j < np) ? 1 + j : 0;  // (rows past c-1 belong to the next axis: read row 0 instead)
    const double h = hp[row * ld], t = tp[row * ld];
    hv[j] = (j < np) ? h : 0.0;
    tv[j] = (j < np) ? t : 0.0;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) F.r[i] = fast_rcp(tt[i]);
  F.factorize(N, np);
  double *cp = (a.coeffs && live) ? a.coeffs + (int64_t)(ax * D) * ld + bb : nullptr;
  double e = solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int piece, int col, double v) {
    if (cp) cp[(int64_t)(piece * 3 * D + col) * ld] = v;
  });
  // energy of the trajectory = sum over its three adjacent lanes
  const double e1 = __shfl_down(e, 1), e2 = __shfl_down(e, 2);
  if (a.energy && live && ax == 0) a.energy[bb] = e + e1 + e2;
}

// The same solve with the chain eliminated FROM BOTH ENDS (exact shapes with an even number of pieces): two lanes per (trajectory,
// axis).  Seen backwards in time a trajectory is a trajectory -- node k' = N - k, piece i' = N - 1 - i, every derivative of odd order
// changes its sign -- and its minimum-effort problem is the same problem: the second lane runs the same code on the reversed data for
// the second half of the chain (minco_core.h, the *_chain<TF> functions: the last node of a half is the interior node where the
// halves meet, nothing pinned there).  At the middle node the two lanes exchange what their half contributes to its diagonal
// block and right-hand side (a DPP swap of adjacent lanes, the partner's values with the signs of the reversal) and both finish
// the middle node for themselves.  Half the sequential depth, twice the waves: the shape of batches that leave most SIMDs empty
// (1024 trajectories: 9.4 -> see DESIGN 4.1).  Lane order: 6 lanes per trajectory (axis, role), ten trajectories per wave.
__device__ __forceinline__ double lane_pair_swap(double v) {  // the value of the other lane of the pair (lanes 2k, 2k+1)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int S, int NB, int NPC>
__global__ void __launch_bounds__(kSolveBlock) k_minco_solve_axis_two(SolveArgs a) {
  static_assert(NB % 2 == 0 && NB >= 2, "an even number of pieces");
  constexpr int m = S - 1, D = 2 * S, NC = NB / 2, N = NB, np = NPC, c = NPC + 1;
  const int l6 = (int)threadIdx.x % 6;
  const int role = l6 & 1, ax = l6 >> 1;
  const int64_t b = (int64_t)blockIdx.x * 10 + threadIdx.x / 6;
  const bool live = threadIdx.x < 60 && b < a.B;
  const int64_t ld = a.ld;
  const int64_t bb = live ? b : 0;  // idle lanes compute on trajectory 0 and store nothing
  Factor<S, NC> F;
  double P[NC + 1], hv[m], tv[m], X[NC + 1][m], tt[NC];
  const double *hp = a.head + (int64_t)(ax * c) * ld + bb;
  const double *tp = a.tail + (int64_t)(ax * c) * ld + bb;
#pragma unroll
  for (int i = 0; i < NC; ++i) tt[i] = a.T[(int64_t)(role ? N - 1 - i : i) * ld + bb];
#pragma unroll
  for (int k = 0; k <= NC; ++k) {
    const int kr = role ? N - k : k;
    const double *src = (kr == 0) ? hp : (kr < N) ? a.wps + (int64_t)((kr - 1) * 3 + ax) * ld + bb : tp;
    P[k] = *src;
  }
#pragma unroll
  for (int j = 0; j < m; ++j) {
    const int64_t row = (j < np) ? 1 + j : 0;
    const double h = (role ? tp : hp)[row * ld];
    hv[j] = (j < np) ? ((role && ((j + 1) & 1)) ? -h : h) : 0.0;
    tv[j] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) F.r[i] = fast_rcp(tt[i]);
  F.template factorize_chain<true>(NC, np, [&](double (&Dk)[m][m]) {
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l <= j; ++l) {
        const double o = lane_pair_swap(Dk[j][l]);
        Dk[j][l] += ((j + l) & 1) ? -o : o;
      }
  });
  double rr[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) rr[i] = launder(F.r[i]);
  sweep_forward_chain<true, S, NC>(F, NC, np, rr, X, [&](int k, double (&y)[m]) { rhs_primal_node<S, NC, true>(k, NC, np, rr, P, hv, tv, y); },
                                   [&](double (&y)[m]) {
#pragma unroll
                                     for (int l = 0; l < m; ++l) {
                                       const double o = lane_pair_swap(y[l]);
                                       y[l] += ((l + 1) & 1) ? -o : o;
                                     }
                                   });
#pragma unroll
  for (int i = 0; i < NC; ++i) rr[i] = launder(rr[i]);
  double *cp = (a.coeffs && live) ? a.coeffs + (int64_t)(ax * D) * ld + bb : nullptr;
  double e = 0.0;
  // the pieces are emitted in the trajectory's own direction: for the reversed half start and end of a piece change places and the
  // odd derivatives their sign
  sweep_backward_chain<true, S, NC>(F, NC, np, rr, X, [&](int k, const Pw<S> &p) {
    double x0[m], x1[m];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      const double u = X[k][j], w = X[k + 1][j];
      x0[j] = role ? (((j + 1) & 1) ? -w : w) : u;
      x1[j] = role ? (((j + 1) & 1) ? -u : u) : w;
    }
    const int piece = role ? N - 1 - k : k;
    e += emit_piece<S>(piece, p, role ? P[k + 1] : P[k], role ? P[k] : P[k + 1], x0, x1, [&](int, int col, double v) {
      if (cp) cp[(int64_t)(piece * 3 * D + col) * ld] = v;
    });
  }, [](double (&)[m]) {});
  // energy of the trajectory = sum over its six adjacent lanes (in a fixed order)
  const double e01 = e + lane_pair_swap(e);
  const double tot = (e01 + __shfl_down(e01, 2)) + __shfl_down(e01, 4);
  if (a.energy && live && l6 == 0) a.energy[bb] = tot;
}

// ------------------------------------------------------------------------------------------
// cost / gradient path: partial gradients per piece, then adjoint propagation per trajectory
// ------------------------------------------------------------------------------------------
struct Penalty {
  double rho, wc, wv, wa, mu, vmax, amax;
  int res, M;
};

// firi::smoothedL1 (gcopter/firi.hpp:60-84), 0 below 0.
__device__ __forceinline__ void smoothed_l1(double mu, double inv_mu, double x, double &f, double &df) {
  const double xd = x * inv_mu, sq = xd * xd, mm = __builtin_fma(-0.5, x, mu);
  double fm = mm * sq * xd, dm = sq * __builtin_fma(-0.5, xd, 3.0 * mm * inv_mu);
  const bool hi = x > mu, neg = x < 0.0;
  f = neg ? 0.0 : (hi ? x - 0.5 * mu : fm);
  df = neg ? 0.0 : (hi ? 1.0 : dm);
}

// The same function without selects (the penalty kernel evaluates this for every row x sample):
// with xc = clamp(x, 0, mu) the cubic piece gives 0 below 0 and mu/2, slope 1 at mu, so
//   f = cubic(xc) + max(x - mu, 0),  df = cubic'(xc).
__device__ __forceinline__ void smoothed_l1_clamped(double mu, double inv_mu, double x, double &f, double &df) {
  const double xc = fmin(fmax(x, 0.0), mu);
  const double xd = xc * inv_mu, sq = xd * xd, mm = __builtin_fma(-0.5, xc, mu);
  f = __builtin_fma(mm * sq, xd, fmax(x - mu, 0.0));
  df = sq * __builtin_fma(-0.5, xd, (3.0 * inv_mu) * mm);
}

// F and F' of the note in k_piece_grad: smoothed L1 of x = mu u is mu F(u), its slope F'(u)
__device__ __forceinline__ void smoothed_l1_unit(double u, double &F, double &dF) {
  const double w = fmax(u, 0.0), uc = fmin(w, 1.0), sq = uc * uc;
  F = __builtin_fma(sq * uc, __builtin_fma(-0.5, uc, 1.0), w - uc);
  dF = sq * __builtin_fma(-2.0, uc, 3.0);
}

struct PieceGradArgs {
  const double *coeffs, *T, *hpolys;
  double *gdC, *gdT, *pcost;
  int64_t B, ld;
  int N, with_energy, with_penalty;
  Penalty pp;
};

// One lane per (trajectory, piece): blockIdx.y = piece.  Writes (not accumulates) the partial
// gradients of  [with_energy] int (p^(s))^2  +  [with_penalty] J_pen  w.r.t. the piece's
// coefficients and duration.  J_pen = (T/res) sum_{j<res} [wc sum_rows phi(a.p-b) + wv sum phi(+-v-vmax)
// + wa sum phi(+-a-amax)] sampled at t = j T/res: the rows of the reference's inequality block
// (qp_solver.hpp:244-296 / min_traj_opt.py:535-613) turned into a smoothed-L1 penalty.
// Basis rows in normalised time, tab[j][d][col] = k!/(k-d)! tau_j^(k-d) (k = D-1-col, tau_j = j/res): built once per
// (order, res) into a small device buffer.  k_piece_grad reads it through a `const __restrict__` kernel argument with a
// wave-uniform index, i.e. as SCALAR loads: the values arrive in SGPRs and feed the FMAs directly, no LDS traffic and
// no vector registers for the table.
static __global__ void __launch_bounds__(256) k_build_basis_table(double *tab, int res, int D) {  // (static: this header is in two translation units)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= res * 4 * D) return;
  const int j = e / (4 * D), d = (e / D) % 4, col = e % D, k = D - 1 - col;
  const double tau = (double)j / (double)res;
  double v = 0.0;
  if (k >= d) {
    v = 1.0;
    for (int q = 0; q < d; ++q) v *= (double)(k - q);
    for (int q = 0; q < k - d; ++q) v *= tau;
  }
  tab[e] = v;
}

// SPLIT (small batches): TWO adjacent lanes per (trajectory, piece).  Lane 0 of the pair takes the energy part and
// the even row chunks, lane 1 the box rows and the odd chunks; the partial gradients are summed across the pair
// with a DPP swap.  Same work, half the dependent chain per lane and twice the waves -- what counts when the
// batch leaves one wave per SIMD.
__device__ __forceinline__ double pair_sum(double v) {  // v + the value of the other lane of the pair (lanes 2k, 2k+1)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
  return v + __hiloint2double(hi, lo);
}

// A row of the basis table (D <= 8 doubles, wave-uniform address) through a scalar load the COMPILER does not track: request()
// issues it, await() is the s_waitcnt the values may be read behind.  (Six-column rows load eight: the table's rows are 4 D apart.)
template <int D>
struct TabRow {
  static_assert(D == 4 || D == 6 || D == 8, "orders 2..4");
  static constexpr int W = D == 4 ? 4 : 8;
  typedef double vec_t __attribute__((ext_vector_type(W)));
  vec_t v;
  __device__ __forceinline__ void request(const double *row_) {
    // (the address in scalar registers whatever the compiler proved about it: the sample index of the sample-split shape
    // comes from threadIdx.x >> 6)
    const unsigned long long pa = (unsigned long long)row_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    const double *row = (const double *)(((unsigned long long)hi << 32) | lo);
    if constexpr (W == 8) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(v) : "s"(row));
    else asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(v) : "s"(row));
  }
  __device__ __forceinline__ void await() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }
};

// The penalty part of one (trajectory, piece) on one lane (of a pair: SPLIT; of a pair and a sample subset: SW > 1): adds the
// gradient of J_pen w.r.t. the piece's coefficients to gC and w.r.t. its duration to gT, sets pc = this lane's share of the
// piece's J_pen.  Shared by k_piece_grad and the one-launch small-batch evaluation (minco_fused_kernel.h): same code, same bits.
// LTAB = false: the samples j = wv, wv + SW, ... -- the same for every lane of the wave -- with the basis table read through
// scalar loads; LTAB = true: this LANE's samples j = wv, wv + jstep, ... (wv, jstep run-time values that may differ from lane to
// lane) with `tab` a copy of the table in LDS, rows [j][3][D] (position, velocity, acceleration).
template <int S, bool SPLIT, int SW, bool LTAB = false>
__device__ __forceinline__ void piece_penalty_part(const Penalty &pp_in, const double *hpolys, const int64_t ld, const int64_t b,
                                                   const int i, const int half, const int wv, const double Ti,
                                                   const double (&c)[3][2 * S], const double *__restrict__ tab,
                                                   double (&gC)[3][2 * S], double &gT, double &pc, const int jstep = SW) {
  constexpr int D = 2 * S;
  // Normalised time: with c~_k = c_k T^k the state rows at sample j depend on tau_j = j/res only,
  //   d^d p/dt^d (t_j) = T^-d sum_col c~[col] tab[j][d][col],  tab[j][d][col] = k!/(k-d)! tau_j^(k-d)
  // (table read with a wave-uniform index: scalar loads).
  //
  // Instruction diet (the kernel is bound by FP64 issue, tools/micro/fp64_peak.hip):
  //  * smoothed L1 in normalised form: with u = x/mu, F(u) = uc^3 (1 - uc/2) + max(u-1, 0), F'(u) = uc^2 (3 - 2 uc),
  //    uc = clamp(u, 0, 1), the penalty is mu F(u) and its slope F'(u); the polytope rows are divided by mu when
  //    they are loaded, the weights are applied once per sample;
  //  * only the pass that holds the box rows evaluates velocity and acceleration; the other passes need the
  //    position alone;
  //  * no jerk and no per-sample d/dt: the sample times t_j = tau_j T move with T, and since
  //    tau tab[j][d+1][col] = (k-d) tab[j][d][col] the sum over the samples of step tau_j (g_p.v + g_v.a + g_a.j)
  //    is  (1/T) sum_col c~[col] k gN[col] - Rs,  gN the gradient w.r.t. c~ that is accumulated anyway and Rs the
  //    scalar sum_j (s1.v + 2 T s2.a) over the samples with a violated box row (s1, s2: their weights in gN);
  //    with v = a1 / T, a = a2 / T^2 that is (1/T) (sum s1 a1 + 2 sum s2 a2): two accumulators, scaled once.
  const Penalty pp = pp_in;
  const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
  const double step = Ti * inv_res;
  const double rT = 1.0 / Ti, rT2 = rT * rT;
  const double wcm = pp.wc * pp.mu, wvm = pp.wv * pp.mu, wam = pp.wa * pp.mu;
  double ct[3][D];  // c~
  {
    double tk = 1.0;
#pragma unroll
    for (int col = D - 1; col >= 0; --col) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) ct[ax][col] = c[ax][col] * tk;
      tk *= Ti;
    }
  }
  double gN[3][D];  // gradient w.r.t. c~
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int col = 0; col < D; ++col) gN[ax][col] = 0.0;
  double csum = 0.0, Rs1 = 0.0, Rs2 = 0.0;  // sum of the sample costs; Rs = (Rs1 + 2 Rs2) / T, see above
  const double kv = rT * inv_mu, ka = rT2 * inv_mu, cv = pp.vmax * inv_mu, ca = pp.amax * inv_mu;
  const double K1 = step * rT * pp.wv, K2 = step * rT2 * pp.wa;
  // Polytope rows are held in registers, RC at a time, and the sample loop runs inside: re-reading
  // them from L2 for every sample (res x M x 32 B per lane) was the bottleneck of this kernel.
  constexpr int RC = 8;
  const int nchunk = hpolys ? (pp.M + RC - 1) / RC : 0;
  // chunks of this lane: all of them, or (SPLIT) every second one starting at `half`; at least one pass so
  // that the box rows are visited
  const int cstep = SPLIT ? 2 : 1;
  int npass = SPLIT ? (nchunk - half + 1) / 2 : nchunk;
  if (npass < 1 && !LTAB) npass = 1;
  if constexpr (LTAB) {
    // The one-launch kernel visits the velocity / acceleration limits in a pass of ITS OWN, ahead of the corridor passes: with the
    // basis rows in VGPRs (the sample index differs from lane to lane) the first pass's live set -- eight corridor rows, c~, the
    // gradient, three basis rows -- was ~20 values over the 256 registers and every sample paid 66 v_accvgpr moves of 516
    // instructions; the limits need neither the rows nor the position.  Same formulas as below; the sample costs of the limits are
    // added to csum here and those of the corridor rows there (a different order of the same sum).
    for (int j = wv; j < pp.res; j += jstep) {
      const double *tb = tab + (size_t)j * 3 * D;
      double t1[D], t2[D], a1[3], a2[3], worst = 0.0;
#pragma unroll
      for (int col = 0; col < D; ++col) {
        t1[col] = tb[D + col];
        t2[col] = tb[2 * D + col];
      }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double x1 = 0.0, x2 = 0.0;
#pragma unroll
        for (int col = 0; col < D; ++col) {
          x1 = __builtin_fma(ct[ax][col], t1[col], x1);
          x2 = __builtin_fma(ct[ax][col], t2[col], x2);
        }
        a1[ax] = x1;
        a2[ax] = x2;
        worst = fmax(worst, fmax(__builtin_fma(fabs(x1), kv, -cv), __builtin_fma(fabs(x2), ka, -ca)));
      }
      if (__any(worst > 0.0)) {
        double cost = 0.0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          double f, df;
          smoothed_l1_unit(__builtin_fma(fabs(a1[ax]), kv, -cv), f, df);
          cost = __builtin_fma(wvm, f, cost);
          const double s1 = K1 * copysign(df, a1[ax]);
          Rs1 = __builtin_fma(s1, a1[ax], Rs1);
          smoothed_l1_unit(__builtin_fma(fabs(a2[ax]), ka, -ca), f, df);
          cost = __builtin_fma(wam, f, cost);
          const double s2 = K2 * copysign(df, a2[ax]);
          Rs2 = __builtin_fma(s2, a2[ax], Rs2);
#pragma unroll
          for (int col = 0; col < D; ++col)
            gN[ax][col] = __builtin_fma(s2, t2[col], __builtin_fma(s1, t1[col], gN[ax][col]));
        }
        csum += cost;
      }
    }
  }
  for (int pass = 0; pass < npass; ++pass) {
    const int ch = half + cstep * pass;
    double hr[RC][4];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const int rr = ch * RC + r;
      const bool ok = hpolys && rr < pp.M;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        hr[r][q] = ok ? hpolys[(int64_t)((i * pp.M + rr) * 4 + q) * ld + b] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < RC; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) hr[r][q] *= inv_mu;
    const bool first = LTAB ? false : (SPLIT ? (half == 1 && pass == 0) : (ch == 0));  // the pass that also evaluates the box rows
    // The table rows are scalar loads, a wave waited for each where it was used (~10^2 cycles, five times per sample, the
    // two waves of a SIMD in phase) and the compiler keeps such a load next to its use: the position row of the NEXT sample
    // is requested by hand at the top of the sample (tab_row_request: the compiler does not know it is in flight; it is
    // awaited at the bottom, tab_row_await), the velocity / acceleration rows of THIS sample right behind it -- scalar
    // loads return out of order, so a wait for any is a wait for all, and the first one stands behind the eight corridor
    // rows' worth of arithmetic.
    TabRow<D> nx, r1, r2;
    if constexpr (!LTAB) {
      nx.request(tab + (size_t)wv * 4 * D);
      nx.await();
    }
    for (int j = wv; j < pp.res; j += (LTAB ? jstep : SW)) {
      const double *tb = tab + (size_t)j * (LTAB ? 3 : 4) * D;
      double t0[D];
      if constexpr (LTAB) {
#pragma unroll
        for (int col = 0; col < D; ++col) t0[col] = tb[col];
      } else {
#pragma unroll
        for (int col = 0; col < D; ++col) t0[col] = nx.v[col];
        nx.request(tab + (size_t)(j + SW < pp.res ? j + SW : j) * 4 * D);
        // (by hand as well: behind an asm statement the compiler takes the table for clobbered and loads it per lane; and in
        // every pass: a request under a condition leaves the registers undefined on the other path, which the register
        // allocator answers with vector registers)
        r1.request(tb + D);
        r2.request(tb + 2 * D);
      }
      double pos[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double acc = 0.0;
#pragma unroll
        for (int col = 0; col < D; ++col) acc = __builtin_fma(ct[ax][col], t0[col], acc);
        pos[ax] = acc;
      }
      double Fs = 0.0, G[3] = {0.0, 0.0, 0.0};  // sum of F(u) and of F'(u) a/mu over the rows
      double ur[RC];
#pragma unroll
      for (int r = 0; r < RC; ++r)
        ur[r] = __builtin_fma(hr[r][0], pos[0], __builtin_fma(hr[r][1], pos[1], __builtin_fma(hr[r][2], pos[2], -hr[r][3])));
      bool any_row = true;
      if constexpr (LTAB) {
        // (the one-launch kernel runs one wave per SIMD: every wave-uniform branch is a compare-to-branch bubble nothing fills;
        //  one test for the eight rows first -- inside the corridor of all of them the per-row tests are skipped as well)
        double um = ur[0];
#pragma unroll
        for (int r = 1; r < RC; ++r) um = fmax(um, ur[r]);
        any_row = __any(um > 0.0);
      }
      if (any_row)
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        const double u = ur[r];
        if (__any(u > 0.0)) {  // wave-uniform: inside the corridor nothing else is computed
          const double w = fmax(u, 0.0), uc = fmin(w, 1.0), sq = uc * uc;
          Fs += w - uc;
          Fs = __builtin_fma(sq * uc, __builtin_fma(-0.5, uc, 1.0), Fs);
          const double df = sq * __builtin_fma(-2.0, uc, 3.0);
          G[0] = __builtin_fma(df, hr[r][0], G[0]);
          G[1] = __builtin_fma(df, hr[r][1], G[1]);
          G[2] = __builtin_fma(df, hr[r][2], G[2]);
        }
      }
      double cost = wcm * Fs;
      if (first) {
        // velocity / acceleration limits in units of mu, straight from the normalised-time sums a1 = sum c~ tab',
        // a2 = sum c~ tab'':  u = (|a1| / T - vmax) / mu = |a1| kv - cv  (one FMA, |.| is an operand modifier)
        double a1[3], a2[3], worst = 0.0;
        double t1[D], t2[D];
        if constexpr (LTAB) {
#pragma unroll
          for (int col = 0; col < D; ++col) {
            t1[col] = tb[D + col];
            t2[col] = tb[2 * D + col];
          }
        } else {
          r1.await();
          r2.await();
#pragma unroll
          for (int col = 0; col < D; ++col) {
            t1[col] = r1.v[col];
            t2[col] = r2.v[col];
          }
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          double x1 = 0.0, x2 = 0.0;
#pragma unroll
          for (int col = 0; col < D; ++col) {
            x1 = __builtin_fma(ct[ax][col], t1[col], x1);
            x2 = __builtin_fma(ct[ax][col], t2[col], x2);
          }
          a1[ax] = x1;
          a2[ax] = x2;
          worst = fmax(worst, fmax(__builtin_fma(fabs(x1), kv, -cv), __builtin_fma(fabs(x2), ka, -ca)));
        }
        if (__any(worst > 0.0)) {  // only one of +v, -v (+a, -a) can be violated: the slope has the sign of a1 (a2)
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            double f, df;
            smoothed_l1_unit(__builtin_fma(fabs(a1[ax]), kv, -cv), f, df);
            cost = __builtin_fma(wvm, f, cost);
            const double s1 = K1 * copysign(df, a1[ax]);
            Rs1 = __builtin_fma(s1, a1[ax], Rs1);
            smoothed_l1_unit(__builtin_fma(fabs(a2[ax]), ka, -ca), f, df);
            cost = __builtin_fma(wam, f, cost);
            const double s2 = K2 * copysign(df, a2[ax]);
            Rs2 = __builtin_fma(s2, a2[ax], Rs2);
            // (the gradient of the limit rows goes into gN here, while s1 and s2 are at hand)
#pragma unroll
            for (int col = 0; col < D; ++col)
              gN[ax][col] = __builtin_fma(s2, t2[col], __builtin_fma(s1, t1[col], gN[ax][col]));
          }
        }
      }
      if (__any(cost > 0.0)) {
        csum += cost;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const double s0 = step * wcm * G[ax];
#pragma unroll
          for (int col = 0; col < D; ++col) gN[ax][col] = __builtin_fma(s0, t0[col], gN[ax][col]);
        }
      }
      if constexpr (!LTAB) nx.await();
    }
  }
  pc = step * csum;
  {  // d/dT at fixed c: the quadrature weight T/res and the sample times tau_j T
    double acc = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
      for (int col = 0; col < D; ++col)
        acc = __builtin_fma(ct[ax][col] * (double)(D - 1 - col), gN[ax][col], acc);
    gT += csum * inv_res + rT * (acc - __builtin_fma(2.0, Rs2, Rs1));
  }
  {  // d/dc = T^k d/dc~
    double tk = 1.0;
#pragma unroll
    for (int col = D - 1; col >= 0; --col) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) gC[ax][col] = __builtin_fma(gN[ax][col], tk, gC[ax][col]);
      tk *= Ti;
    }
  }
}

// The energy part of one (trajectory, piece) from its S highest-power coefficients ch: d/dc and d/dT of the piece's share of
// int (p^(S))^2, added to gC / gT:  d/dc of sum_{j,k>=S} c_j c_k f_j f_k T^(j+k-2S+1)/(j+k-2S+1) ;  d/dT = (p^(S)(T))^2
template <int S>
__device__ __forceinline__ void piece_energy_compute(const double (&ch)[3][S], const double Ti, double (&gC)[3][2 * S], double &gT) {
  constexpr int D = 2 * S;
  double tp[D];
  tp[0] = 1.0;
#pragma unroll
  for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Ti;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    double ps = 0.0;
#pragma unroll
    for (int j = S; j < D; ++j) {
      double fj = 1.0;
#pragma unroll
      for (int e = 0; e < S; ++e) fj *= (double)(j - e);
      ps = __builtin_fma(fj * tp[j - S], ch[ax][D - 1 - j], ps);
      double acc = 0.0;
#pragma unroll
      for (int k = S; k < D; ++k) {
        double fk = 1.0;
#pragma unroll
        for (int e = 0; e < S; ++e) fk *= (double)(k - e);
        acc = __builtin_fma(2.0 * fj * fk / (double)(j + k - 2 * S + 1) * tp[j + k - 2 * S + 1],
                            ch[ax][D - 1 - k], acc);
      }
      gC[ax][D - 1 - j] += acc;
    }
    gT = __builtin_fma(ps, ps, gT);
  }
}

// The energy part for a lane of k_piece_grad: the coefficients come from global memory.
template <int S>
__device__ __forceinline__ void piece_energy_part(const double *coeffs, const int64_t ld, const int64_t b, const int i,
                                                  const double Ti, const double pc, double (&gC)[3][2 * S], double &gT) {
  constexpr int D = 2 * S;
  // (the S highest-power coefficients are read again here: the penalty part above has the registers for itself)
  // ... and only here: tied to a result of the penalty part, or the scheduler issues these loads before the sample
  // loops and carries the twelve values across them in scratch
  int64_t be = b;
  asm volatile("" : "+v"(be) : "v"(pc));
  double ch[3][S];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int col = 0; col < S; ++col) ch[ax][col] = coeffs[(int64_t)((i * 3 + ax) * D + col) * ld + be];
  piece_energy_compute<S>(ch, Ti, gC, gT);
}

// SW = 4 (smaller batches still, with SPLIT): the four WAVES of a workgroup take every fourth sample of the SAME 32
// (trajectory, piece) pairs -- the sample index stays wave-uniform, so the basis table is still read with scalar
// loads -- and their partial gradients are summed through LDS by wave 0.  A quarter of the dependent chain per lane
// and four times the waves: at 4096 x 8 pieces the two-lane variant leaves one wave per SIMD.
#ifndef ANET_PG_MINB
#define ANET_PG_MINB 2
#endif
#ifndef ANET_PG_SW_MINB
#define ANET_PG_SW_MINB 1  // workgroups per CU the sample-split variant is register-bounded for
#endif
template <int S, bool SPLIT = false, int SW = 1>
__global__ void __launch_bounds__(256, SW > 1 ? ANET_PG_SW_MINB : ANET_PG_MINB) k_piece_grad(PieceGradArgs a, const double *__restrict__ tab) {
  constexpr int D = 2 * S;
  static_assert(SW == 1 || SPLIT, "the sample split builds on the two-lane variant");
  const int wv = SW > 1 ? (int)(threadIdx.x >> 6) : 0;  // which samples: j = wv, wv + SW, ...
  const int64_t gid = SW > 1 ? (int64_t)blockIdx.x * 64 + (threadIdx.x & 63) : (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int half = SPLIT ? (int)(gid & 1) : 0;
  const int64_t bq = SPLIT ? (gid >> 1) : gid;
  const bool live = bq < a.B;
  if (!SPLIT && !live) return;
  const int64_t b = live ? bq : 0;  // (SPLIT: idle pairs compute on trajectory 0 and store nothing: the DPP sum needs both lanes)
  const int i = blockIdx.y;
  const int64_t ld = a.ld;
  const double Ti = a.T[(int64_t)i * ld + b];
  double c[3][D], gC[3][D];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int col = 0; col < D; ++col) {
      c[ax][col] = a.coeffs[(int64_t)((i * 3 + ax) * D + col) * ld + b];
      gC[ax][col] = 0.0;
    }
  double gT = 0.0, pc = 0.0;
  if (a.with_penalty) piece_penalty_part<S, SPLIT, SW>(a.pp, a.hpolys, ld, b, i, half, wv, Ti, c, tab, gC, gT, pc);
  if (a.with_energy && half == 0 && wv == 0) piece_energy_part<S>(a.coeffs, ld, b, i, Ti, pc, gC, gT);
  if (SPLIT) {
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
      for (int col = 0; col < D; ++col) gC[ax][col] = pair_sum(gC[ax][col]);
    gT = pair_sum(gT);
    pc = pair_sum(pc);
    if constexpr (SW > 1) {  // sum over the waves of the workgroup (fixed order: deterministic)
      __shared__ double red[SW - 1][3 * D + 2][32];
      const int pl = (int)(threadIdx.x & 63) >> 1;
      if (wv > 0 && half == 0) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
          for (int col = 0; col < D; ++col) red[wv - 1][ax * D + col][pl] = gC[ax][col];
        red[wv - 1][3 * D][pl] = gT;
        red[wv - 1][3 * D + 1][pl] = pc;
      }
      __syncthreads();
      if (wv != 0) return;
#pragma unroll
      for (int q = 0; q < SW - 1; ++q) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
          for (int col = 0; col < D; ++col) gC[ax][col] += red[q][ax * D + col][pl];
        gT += red[q][3 * D][pl];
        pc += red[q][3 * D + 1][pl];
      }
    }
    if (!live || half != 0) return;
  }
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int col = 0; col < D; ++col) a.gdC[(int64_t)((i * 3 + ax) * D + col) * ld + b] = gC[ax][col];
  a.gdT[(int64_t)i * ld + b] = gT;
  if (a.pcost) a.pcost[(int64_t)i * ld + b] = pc;
}

// k_piece_grad lives in a translation unit of its own (piece_grad_unit.hip), compiled with the max-ILP scheduling strategy:
// its small-batch shapes are chains of dependent FP64 instructions on one or two waves per SIMD, and scheduled for ILP
// rather than for occupancy BASELINE configs[2] (4096 x 8 pieces) takes 55 instead of 60 us per evaluation; the same
// strategy costs k_minco_solve_axis a third (8.9 -> 12.2 us per launch of 1024) and the interior-point QP 3 %, so it is
// not a flag of the whole library.  shape: 0 lane per (trajectory, piece); 1 two lanes per pair; 2 two lanes per pair and
// the samples over the four waves of a workgroup.
void launch_piece_grad(int s, int shape, dim3 grid, dim3 block, hipStream_t st, const PieceGradArgs &a, const double *tab, int mx_cus = 0);

struct PropArgs {
  const double *T, *coeffs, *gdC, *gdT;
  double *gradP, *gradT;
  // optional total cost: cost = energy_in + rho sum T + sum_i pcost_i ; gradT += rho
  const double *energy_in, *pcost;
  double *cost;
  double rho;
  int64_t B, ld;
  int N, c;
  // optional chain rule for durations parametrised as T = forward_T(tau): gradT is written as dJ/dtau
  const double *tau = nullptr;
};

// One axis of the adjoint: accumulates this axis' share of dJ/dT into gT and stores its gradP rows.
// CHAIN (fully specialised small-batch instantiations): 1/T of a piece is made to depend, through an empty asm, on
// the previous step of the sweep it is used in.  Without the masks of the generic code the body is one straight-line
// block and the scheduler computes every piece's powers of 1/T up front for all three sweeps, which spills
// (~1 KB/lane); the false dependency keeps each piece's powers next to their use.
template <int S, int NB, bool CHAIN = false, bool PIPE = CHAIN>
__device__ __forceinline__ void propagate_axis(const Factor<S, NB> &F, const int N, const int np, const PropArgs &a,
                                               const int64_t b, const int ax, const double Tlast,
                                               double (&gT)[NB], const bool store) {
  constexpr int m = S - 1, D = 2 * S;
  const int64_t ld = a.ld;
  // per-axis base pointers: every later offset is wave-uniform also when ax differs per lane
  const double *ca = a.coeffs + (int64_t)(ax * D) * ld + b, *ga = a.gdC + (int64_t)(ax * D) * ld + b;
  double *gp = a.gradP + (int64_t)ax * ld + b;
  double rr[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(F.r[i]) : 0.0;
  // ---- node states are re-read from the coefficients where needed instead of being held:
  //      x_k[j] = j! c_j(piece k) for k < N; the last node by evaluating piece N-1 at its end
  // (`cb` is this axis' coefficient base pointer; the second phase passes a laundered copy so that the
  //  compiler re-reads instead of keeping every node state of the first phase alive)
  auto node_state = [&](const double *cb, int k, double (&x)[S]) {
    if (k < N) {
      double fact = 1.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        if (j > 0) fact *= (double)j;
        x[j] = fact * cb[(int64_t)(k * 3 * D + (D - 1 - j)) * ld];
      }
    } else {
      double cl[D], tp[D];
      tp[0] = 1.0;
#pragma unroll
      for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Tlast;
#pragma unroll
      for (int col = 0; col < D; ++col) cl[col] = cb[(int64_t)((N - 1) * 3 * D + col) * ld];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int p = j; p < D; ++p) {
          double f = 1.0;
#pragma unroll
          for (int e = 0; e < j; ++e) f *= (double)(p - e);
          acc = __builtin_fma(f * tp[p - j], cl[D - 1 - p], acc);
        }
        x[j] = acc;
      }
    }
  };
  double GP[NB + 1], XA[NB + 1][m];  // adjoint of the node positions / of the node derivatives
#pragma unroll
  for (int k = 0; k <= NB; ++k) {
    GP[k] = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) XA[k][l] = 0.0;
  }
  // ---- g_x = Phi' gdC and the direct dPhi/dT term
  // CHAIN: the loads of piece i+1 (its gdC, the state of node i+2) are issued before piece i is computed, so each
  // piece no longer starts with a full L2 round trip; the state of the last node is computed once.
  double xN[S], xc[S], xn1[S], xn2[S] = {}, gnext[D];
  if constexpr (PIPE) {
    node_state(ca, N, xN);
    node_state(ca, 0, xc);
    if (N > 1) node_state(ca, 1, xn1);
#pragma unroll
    for (int col = 0; col < D; ++col) gnext[col] = ga[(int64_t)col * ld];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      if constexpr (CHAIN)
        if (i > 0) asm volatile("" : "+v"(rr[i]) : "v"(gT[i - 1]));
      Pw<S> p(rr[i]);
      double gc[D];
      // low powers k < S: c_k = x_i[k]/k!
      double x0[S], x1[S];
      if constexpr (PIPE) {
#pragma unroll
        for (int col = 0; col < D; ++col) gc[col] = gnext[col];
#pragma unroll
        for (int j = 0; j < S; ++j) {
          x0[j] = xc[j];
          x1[j] = (i + 1 < N) ? xn1[j] : xN[j];
        }
        if (i + 1 < N) {
#pragma unroll
          for (int col = 0; col < D; ++col) gnext[col] = ga[(int64_t)((i + 1) * 3 * D + col) * ld];
        }
        if (i + 2 < N) node_state(ca, i + 2, xn2);
#pragma unroll
        for (int j = 0; j < S; ++j) {
          xc[j] = x1[j];
          xn1[j] = xn2[j];
        }
      } else {
#pragma unroll
        for (int col = 0; col < D; ++col) gc[col] = ga[(int64_t)(i * 3 * D + col) * ld];
        node_state(ca, i, x0);
        node_state(ca, i + 1, x1);
      }
      auto addg = [&](int node, int dg, double v) {  // node-state adjoint: slot 0 = position
        if (dg == 0) GP[node] += v;
        else XA[node][dg - 1] += v;
      };
      double fact = 1.0;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        if (k > 0) fact *= (double)k;
        addg(i, k, gc[D - 1 - k] * (1.0 / fact));
      }
      double h[S];
#pragma unroll
      for (int q = 0; q < S; ++q) h[q] = gc[S - 1 - q] * p[q];  // gc of power S+q times r^q
      double dsum = 0.0;
#pragma unroll
      for (int bb = 0; bb < 2 * S; ++bb) {
        const int dg = bb % S;
        double u = 0.0, qd = 0.0;
#pragma unroll
        for (int q = 0; q < S; ++q) {
          u = __builtin_fma(Tab<S>::BHI[q][bb], h[q], u);
          qd = __builtin_fma((double)(S + q - dg) * Tab<S>::BHI[q][bb], h[q], qd);
        }
        const double sc = p[S - dg];
        const double xb = (bb < S) ? x0[dg] : x1[dg];
        addg(bb < S ? i : i + 1, dg, u * sc);
        dsum = __builtin_fma(xb * sc, qd, dsum);
      }
      gT[i] = __builtin_fma(-p[1], dsum, gT[i]);
    }
  // ---- adjoint solve K lam = g_x|free (pinned rows 0)
  sweep_forward<S, NB>(F, N, np, rr, XA, [&](int k, double (&y)[m]) {
    if constexpr (CHAIN)
      if (k > 0) asm volatile("" : "+v"(rr[k - 1]) : "v"(XA[k - 1][0]));
#pragma unroll
    for (int l = 0; l < m; ++l) y[l] = ((k == 0 || k == N) && l < np) ? 0.0 : XA[k][l];
  });
#pragma unroll
  for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(rr[i]) : 0.0;
  const double *cb2 = ca;
  asm volatile("" : "+v"(cb2));
  // CHAIN: piece k needs the states of nodes k and k+1; node k+1 is kept from the previous step and node k-1 is
  // requested while piece k is computed
  double ya[S], yb[S], yp[S] = {};
  if constexpr (PIPE) {
#pragma unroll
    for (int j = 0; j < S; ++j) yb[j] = xN[j];
    node_state(cb2, N - 1, ya);
  }
  sweep_backward<S, NB>(F, N, np, rr, XA, [&](int k, const Pw<S> &p) {
    // (W_k lam^)[position row of node k]; the row of node k+1 is its negative
    double wl = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) {
      wl = __builtin_fma(Tab<S>::M[0][1 + l] * p[2 * S - 2 - l], XA[k][l], wl);
      wl = __builtin_fma(Tab<S>::M[0][S + 1 + l] * p[2 * S - 2 - l], XA[k + 1][l], wl);
    }
    GP[k] -= wl;
    GP[k + 1] += wl;
    // - lam^' (dW/dT) x^ = sum_ab lam_a M_ab e_ab r^(e_ab+1) x_b,  e_ab = 2S-1-deg a-deg b
    double x0[S], x1[S], xs[2 * S];
    if constexpr (PIPE) {
      if (k > 0) node_state(cb2, k - 1, yp);
#pragma unroll
      for (int j = 0; j < S; ++j) {
        x0[j] = ya[j];
        x1[j] = yb[j];
        yb[j] = ya[j];
        ya[j] = yp[j];
      }
    } else {
      node_state(cb2, k, x0);
      node_state(cb2, k + 1, x1);
    }
#pragma unroll
    for (int bb = 0; bb < 2 * S; ++bb) xs[bb] = ((bb < S) ? x0[bb % S] : x1[bb % S]) * p[S - bb % S];
    double acc = 0.0;
#pragma unroll
    for (int aa = 0; aa < 2 * S; ++aa) {
      const int da = aa % S;
      if (da == 0) continue;
      double row = 0.0;
#pragma unroll
      for (int bb = 0; bb < 2 * S; ++bb)
        row = __builtin_fma(Tab<S>::M[aa][bb] * (double)(2 * S - 1 - da - bb % S), xs[bb], row);
      const double ls = ((aa < S) ? XA[k][da - 1] : XA[k + 1][da - 1]) * p[S - da];
      acc = __builtin_fma(ls, row, acc);
    }
    gT[k] += acc;
    if constexpr (CHAIN)
      if (k > 0) asm volatile("" : "+v"(rr[k - 1]) : "v"(gT[k]));
  });
#pragma unroll
  for (int k = 1; k < NB; ++k)
    if (k < N && store) gp[(int64_t)((k - 1) * 3) * ld] = GP[k];
}

// MINCO propogateGrad: given the partial gradients (gdC, gdT) of a scalar J(c, T), return its total
// gradient w.r.t. the interior waypoints and the durations, c = c(waypoints, T) being the minimum-
// control-effort coefficients.  Adjoint of the Hermite/block-tridiagonal solve:
//   g_x = Phi' gdC (node-state adjoint), K lam = g_x|free, gradP_k = g_x[k].p - (W lam^)[p rows],
//   gradT_i = gdT_i + gdC_i.(dPhi_i/dT) x^ - lam^' (dW_i/dT) x^.
template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_propagate(PropArgs a) {
  constexpr int m = S - 1, D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * kSolveBlock + threadIdx.x;
  if (b >= a.B) return;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int64_t ld = a.ld;

  Factor<S, NB> F;
  double gT[NB];
  double tsum = 0.0, Tlast = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      const double t = a.T[i * ld + b];
      F.r[i] = fast_rcp(t);
      gT[i] = a.gdT[i * ld + b];
      tsum += t;
      if (i == N - 1) Tlast = t;
    }
  F.factorize(N, np);

#pragma unroll 1
  for (int ax = 0; ax < 3; ++ax) propagate_axis<S, NB>(F, N, np, a, b, ax, Tlast, gT, true);
  double csum = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      const double gt = gT[i] + a.rho;
      a.gradT[i * ld + b] = a.tau ? gt * dforward_T(a.tau[i * ld + b]) : gt;
      if (a.pcost) csum += a.pcost[i * ld + b];
    }
  if (a.cost) a.cost[b] = (a.energy_in ? a.energy_in[b] : 0.0) + a.rho * tsum + csum;
}

// Small-batch variant of the above, one lane per (trajectory, axis) like k_minco_solve_axis: the three
// lanes of a trajectory refactorise the shared system and propagate their own axis; dJ/dT is the
// 3-lane shuffle sum of the per-axis shares.
template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_propagate_axis(PropArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * 63 + threadIdx.x;
  const int64_t b = gid / 3;
  const int ax = (int)(gid % 3);
  const bool live = threadIdx.x < 63 && b < a.B;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int64_t ld = a.ld;
  const int64_t bb = live ? b : 0;

  Factor<S, NB> F;
  double gT[NB], tt[NB];
  double tsum = 0.0, Tlast = 0.0;
  // (all loads of a group first, then their uses: interleaved, every load is followed by a full wait)
#pragma unroll
  for (int i = 0; i < NB; ++i) tt[i] = a.T[(i < N ? i : 0) * ld + bb];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    gT[i] = 0.0;
    if (i < N) {
      const double t = tt[i];
      F.r[i] = fast_rcp(t);
      tsum += t;
      if (i == N - 1) Tlast = t;
    }
  }
  F.factorize(N, np);
  propagate_axis<S, NB, NEXACT, true>(F, N, np, a, bb, ax, Tlast, gT, live);

  double gd[NB], tu[NB], pc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int64_t o = (i < N ? i : 0) * ld + bb;
    gd[i] = a.gdT[o];
    tu[i] = a.tau ? a.tau[o] : 0.0;
    pc[i] = a.pcost ? a.pcost[o] : 0.0;
  }
  double csum = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      const double tot = gT[i] + __shfl_down(gT[i], 1) + __shfl_down(gT[i], 2);
      if (live && ax == 0) {
        const double gt = gd[i] + tot + a.rho;
        a.gradT[i * ld + bb] = a.tau ? gt * dforward_T(tu[i]) : gt;
        csum += pc[i];
      }
    }
  if (a.cost && live && ax == 0) a.cost[bb] = (a.energy_in ? a.energy_in[bb] : 0.0) + a.rho * tsum + csum;
}

// the small-batch adjoint (lane = (trajectory, axis)), in the same translation unit as k_piece_grad and for the same reason
void launch_propagate_axis(int s, const PropArgs &a, dim3 grid, dim3 block, hipStream_t st);

}  // namespace anet
