// Batched inequality-constrained QP solve replacing OSQP in QPSolver::solve
// (planner/qp_solver.hpp:299-358; OSQP release-0.6.3 is a third-party dependency that is not part of
// the reference tree).  Same problem, same algorithm family: OSQP's ADMM (Stellato et al., "OSQP: an
// operator splitting solver for quadratic programs", Math. Prog. Comp. 2020, Algorithm 1) with its
// defaults -- rho 0.1, 1e3*rho on equality rows, sigma 1e-6, alpha 1.6, eps_abs = eps_rel = 1e-3,
// max_iter 4000, termination check every 25 iterations, rho adaptation by the residual-ratio rule.
//
// MI355X-specific restatement (one 256-thread workgroup per trajectory; everything in LDS, the per-row state too
// when it fits -- one number per row instead of OSQP's z and y, see `wg` below):
//  * the ROWS are written in NORMALISED time (variables a~_k = c_k T^k, derivative rows scaled by T^d): every basis row
//    depends only on tau_j = j/res, identical for all pieces and trajectories, and the cost block is T^(1-2s) times a
//    constant matrix.  OSQP's modified Ruiz equilibration (round 5) runs on top as two diagonal scalings D (per variable)
//    and E (per row) that start from the inverse of that normalisation, i.e. from the reference's own matrices.
//  * the KKT system is reduced to  (Q + sigma I + A' R A) x~ = rhs, block tridiagonal with one dense
//    3D x 3D block per piece; block Cholesky with explicitly inverted diagonal factors lives in LDS,
//    each ADMM iteration is 4N small mat-vecs, refactorised only when rho changes.
//  * A x and A' w never form A: per sample the 3 state rows (p, v, a) are evaluated from the piece's
//    coefficients, the polytope rows reduce to 3-vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

struct AdmmParams {
  double rho, sigma, alpha, eps_abs, eps_rel;
  int max_iter, check_every, adapt_every, scaled_termination;
};

struct AdmmArgs {
  const double *state;   // [B][2][3][3]
  const double *T;       // [B][N]
  const double *hpolys;  // [B][N][M][4]
  double *z, *y;         // z: [B][m] workspace, ONE value per constraint row (see `wg` in the kernel), m = me + N*R*(M+12); y: unused
  double *coeffs;        // [B][n]  (piece, axis, highest power first) -- the reference's flatten order
  double *obj;           // [B]  1/2 z'Qz in original units (QPSolver::getObjCost)
  int *status, *iters;   // [B]  OSQP's status values: 1 = solved, -2 = max_iter reached, -3 = primal infeasible
  double *res;           // [B][2] primal / dual residual at exit (scaled problem)
  int64_t B;
  int N, R, M;
  double vmax, amax, m34;
  AdmmParams p;
  int zy_in_lds;  // the per-row state lives in LDS (it fits for <= 8-piece snap at M = 16) instead of the global workspace
  double *gradT;  // optional [B][N]: d(optimal 1/2 z'Qz)/dT_i (envelope theorem, see the end of the kernel)
};

__device__ __forceinline__ double atomic_max_pos(double *addr, double v) {  // v >= 0
  unsigned long long *a = (unsigned long long *)addr;
  unsigned long long old = atomicMax(a, (unsigned long long)__double_as_longlong(v));
  return __longlong_as_double((long long)old);
}

template <int S>
__device__ __forceinline__ double qblk1(int j, int k, double m34) {  // cost block at t = 1
  if (j > k) { const int q = j; j = k; k = q; }
  if (S == 4) {
    const double m[4][4] = {{100800, 50400, 20160, 5040}, {0, 25920, 10800, 2880}, {0, 0, 4800, m34}, {0, 0, 0, 576}};
    return m[j][k];
  }
  const double m[3][3] = {{720, 360, 120}, {0, 192, 72}, {0, 0, 36}};
  return m[j][k];
}

// k!/(k-d)! for k <= 7, d <= 3 (0 when k < d)
__device__ __forceinline__ double fallf(int k, int d) {
  constexpr double tab[8][4] = {{1, 0, 0, 0},   {1, 1, 0, 0},    {1, 2, 2, 0},    {1, 3, 6, 6},
                                {1, 4, 12, 24}, {1, 5, 20, 60}, {1, 6, 30, 120}, {1, 7, 42, 210}};
  return tab[k & 7][d & 3];
}

constexpr size_t kAdmmHpLdsBytes = 16 * 1024;  // polytope rows are copied to LDS when they take less than this

// OSQP's scaling limits (osqp/include/constants.h MIN_SCALING / MAX_SCALING) applied to a norm before its square root is taken
__device__ __forceinline__ double admm_limit_scaling(double v) { return v < 1e-4 ? 1.0 : (v > 1e4 ? 1e4 : v); }

#ifndef ANET_ADMM_RUIZ_ITERS
#define ANET_ADMM_RUIZ_ITERS 10  // OSQP's default `scaling`
#endif

// Modified Ruiz equilibration (OSQP, Stellato et al. 2020, section 5.1 / osqp/src/scaling.c scale_data), on top of the analytic
// normalisation: diagonal D (one entry per variable, LDS) and E (one per constraint row, global: AdmmArgs::y) and a cost scale c,
//   P_bar = c D P D,   A_bar = E A D,   l_bar = E l, u_bar = E u,   x = D x_bar,  z = E^-1 z_bar,  y = E y_bar / c,
// found by `scaling` passes of  delta_j = 1 / sqrt(|column j of [P_bar; A_bar]|_inf),  delta_r = 1 / sqrt(|row r of A_bar|_inf)
// followed by  c <- c / max(mean_j |column j of P_bar|_inf, |q_bar|_inf -> 1)  (q = 0 here).  The iteration then runs on the
// scaled problem with OSQP's unscaled termination test and its rho estimate from the SCALED residuals (auxil.c
// compute_rho_estimate).  A is never formed: a row of A_bar is E_r (its analytic entries) D, the column norms come from per-sample
// maxima contracted with the basis table like A'w.
template <int S>
__global__ void __launch_bounds__(256) k_qp_admm(AdmmArgs a) {
  constexpr int D = 2 * S, NB = 3 * D;
  const int N = a.N, R = a.R, M = a.M;
  const int n = NB * N;
  const int me = 3 * (6 + S * (N - 1));
  const int rows_per_sample = M + 12;
  const int64_t NS = (int64_t)N * R;
  const int64_t mtot = me + NS * rows_per_sample;
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;

  extern __shared__ double lds[];
  double *Linv = lds;                       // [N][NB*NB]  inverse of the Cholesky factor of diagonal block i
  double *Lo = Linv + (size_t)N * NB * NB;  // [N][NB*NB]  L_{i+1,i} (row = block i+1, col = block i)
  double *Hd = Lo + (size_t)N * NB * NB;    // [NB*NB] scratch
  double *x = Hd + NB * NB;                 // [n]  x_bar
  double *xt = x + n;                       // [n]  x~_bar
  double *rhs = xt + n;                     // [n]
  double *aty = rhs + n;                    // [n]  A'E y_bar (dual residual, without D) / temp
  double *ady = aty + n;                    // [n]  A'E (y+ - y) (primal infeasibility certificate, without D)
  double *xs = ady + n;                     // [n]  D x~_bar: the iterate in the normalised variables the rows are written in
  double *xns = xs + n;                     // [n]  D x_bar
  double *Dv = xns + n;                     // [n]  Ruiz' D
  double *be = Dv + n;                      // [R][3][D] basis rows at tau_j
  double *Tn = be + (size_t)R * 3 * D;      // [N]
  double *red = Tn + N;                     // [24] reductions / broadcast
  double *hp_l = red + 24;                  // [N*M*4] polytope rows (only when they fit the budget below)
  double *eqb = hp_l + ((size_t)N * M * 4 * sizeof(double) <= kAdmmHpLdsBytes ? (size_t)N * M * 4 : 0);  // [me] rhs b
  double *eqc = eqb + me;                    // [me] coefficient of the right-hand piece (continuity rows)
  double *eqs = eqc + me;                    // [me] 1/T^d: undoes the row scaling of the normalisation
  double *eqE = eqs + me;                    // [me] Ruiz' E of the equality rows
  double *gs = eqE + me;                     // [N*R][12] per-sample A'w blocks (9) / weighted Gram pieces of the factorisation (12)
  double *zy_l = gs + (size_t)NS * 12;       // [mtot] when a.zy_in_lds

  const double *Tg = a.T + b * N;
  const double *hp = a.hpolys + b * (int64_t)N * M * 4;
  const double *st = a.state + b * 18;
  // ADMM state per constraint row: OSQP keeps z and y; both are functions of the single number
  //   w = alpha (A x~) + (1-alpha) z_prev + y_prev/rho      (the argument of the projection):
  //   z = Pi(w) = min(w, u) [b for an equality row],   y = rho (w - z).
  // Storing w alone halves the row traffic and lets the state of an 8-piece snap problem sit in LDS.
  double *wg = a.zy_in_lds ? zy_l : a.z + b * mtot;
  double *Eg = a.y + b * mtot;               // Ruiz' E of the inequality rows at [me + smp + q * NS] (the first me entries unused)

  const bool hp_in_lds = (size_t)N * M * 4 * sizeof(double) <= kAdmmHpLdsBytes;
  const double *hpl = hp;
  if (hp_in_lds) {
    for (int e = tid; e < N * M * 4; e += nt) hp_l[e] = hp[e];
    hpl = hp_l;
  }
  // ---- tables -------------------------------------------------------------------------------
  for (int e = tid; e < R * 3 * D; e += nt) {
    const int j = e / (3 * D), d = (e / D) % 3, col = e % D, k = D - 1 - col;
    const double tau = (double)j / (double)R;
    double v = 0.0;
    if (k >= d) {
      v = fallf(k, d);
      for (int q = 0; q < k - d; ++q) v *= tau;
    }
    be[e] = v;
  }
  for (int i = tid; i < N; i += nt) Tn[i] = Tg[i];
  __syncthreads();
  // an equality row: which piece / axis / derivative, what kind (0 start, 1 end, 2 continuity)
  auto eq_row = [&](int r, int &i0, int &ax, int &d, int &kind) {
    if (r < 18) {
      ax = r / 6;
      const int q = r % 6;
      kind = q < 3 ? 0 : 1;
      d = q % 3;
      i0 = kind == 0 ? 0 : N - 1;
    } else {
      const int rr = r - 18;
      i0 = rr / (3 * S);
      ax = (rr / S) % 3;
      d = rr % S;
      kind = 2;
    }
  };
  auto eq_index = [&](int kind, int i0, int ax, int d) { return kind == 0 ? ax * 6 + d : (kind == 1 ? ax * 6 + 3 + d : 18 + (i0 * 3 + ax) * S + d); };
  for (int r = tid; r < me; r += nt) {  // per-row constants of the equality block (do not change per iteration)
    int i0, ax, d, kind;
    eq_row(r, i0, ax, d, kind);
    double bv = 0.0, cf = 0.0;
    if (kind != 2) bv = st[(kind == 0 ? 0 : 9) + ax * 3 + d] * pow(Tn[i0], (double)d);
    else cf = -fallf(d, d) * pow(Tn[i0] / Tn[i0 + 1], (double)d);
    eqb[r] = bv;
    eqc[r] = cf;
    eqs[r] = pow(Tn[i0], (double)(-d));
    // Ruiz starts from the REFERENCE's matrices, as OSQP does: E and D first undo the analytic normalisation (row of derivative d
    // times T^-d, variable of power k times T^k), so that the passes below see the norms OSQP's scale_data sees
    eqE[r] = ANET_ADMM_RUIZ_ITERS > 0 ? eqs[r] : 1.0;
  }
  for (int e = tid; e < n; e += nt) Dv[e] = ANET_ADMM_RUIZ_ITERS > 0 ? pow(Tn[e / NB], (double)(D - 1 - e % D)) : 1.0;
  for (int64_t e = tid; e < NS * rows_per_sample; e += nt) {
    const int64_t smp = e % NS, q = e / NS;
    double v = 1.0;
    if (ANET_ADMM_RUIZ_ITERS > 0 && q >= M) v = pow(Tn[smp / R], (double)(-(1 + ((q - M) & 1))));
    Eg[me + e] = v;
  }
  __syncthreads();
  // objective scale.  Without equilibration (ANET_ADMM_RUIZ_ITERS = 0): 1 / mean_i(T_i^(1-2s) * mean diagonal of the cost block);
  // with it OSQP's rule decides, starting from 1.
  double cobj = 1.0;
  if (ANET_ADMM_RUIZ_ITERS == 0) {
    double tr = 0.0;
    for (int j = 0; j < S; ++j) tr += qblk1<S>(j, j, a.m34);
    double acc = 0.0;
    for (int i = 0; i < N; ++i) acc += pow(Tn[i], (double)(1 - 2 * S));
    cobj = 1.0 / (acc / N * tr / S);
  }
  // ---- Ruiz equilibration ------------------------------------------------------------------------------------------------
  for (int pass = 0; pass < ANET_ADMM_RUIZ_ITERS; ++pass) {
    // column norms of [P_bar; A_bar] into rhs (atomic maxima), per-sample maxima into gs; row norms applied at the end of the pass
    for (int e = tid; e < n; e += nt) {
      const int i = e / NB, cr = e % D;
      double pn = 0.0;
      if (cr < S) {
        const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
        for (int k = 0; k < S; ++k) pn = fmax(pn, fabs(qs * qblk1<S>(cr, k, a.m34)) * Dv[e - cr + k]);
      }
      rhs[e] = pn * Dv[e];   // |column of P_bar|_inf
      aty[e] = 0.0;          // |column of A_bar|_inf / D_j, equality part (atomic maxima below)
    }
    __syncthreads();
    for (int r = tid; r < me; r += nt) {
      int i0, ax, d, kind;
      eq_row(r, i0, ax, d, kind);
      const double Er = eqE[r];
      double rn = 0.0;
      if (kind == 0) {
        const int v = ax * D + (D - 1 - d);
        rn = fallf(d, d) * Dv[v];
        atomic_max_pos(&aty[v], Er * fallf(d, d));
      } else {
        for (int col = 0; col < D; ++col) {
          const int k = D - 1 - col;
          if (k >= d) {
            const int v = i0 * NB + ax * D + col;
            rn = fmax(rn, fallf(k, d) * Dv[v]);
            atomic_max_pos(&aty[v], Er * fallf(k, d));
          }
        }
        if (kind == 2) {
          const int v = (i0 + 1) * NB + ax * D + (D - 1 - d);
          rn = fmax(rn, fabs(eqc[r]) * Dv[v]);
          atomic_max_pos(&aty[v], Er * fabs(eqc[r]));
        }
      }
      eqE[r] = Er / sqrt(admm_limit_scaling(Er * rn));  // (read by no other thread in this pass; the maxima above took the old value)
    }
    for (int smp = tid; smp < NS; smp += nt) {
      const int i = smp / R, j = smp % R;
      const double *bj = be + (size_t)j * 3 * D;
      double m3[3][3];  // [d][axis] max_col basis_d[col] D[i, axis, col]
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int axx = 0; axx < 3; ++axx) {
          double mx = 0.0;
          for (int col = 0; col < D; ++col) mx = fmax(mx, fabs(bj[d * D + col]) * Dv[i * NB + axx * D + col]);
          m3[d][axx] = mx;
        }
      double w9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // [d][axis] max over the rows of E_r |coefficient|
      const int64_t r0 = me + smp;
      for (int q = 0; q < M; ++q) {
        const double *hq = hpl + ((int64_t)i * M + q) * 4;
        const double c0 = fabs(hq[0]), c1 = fabs(hq[1]), c2 = fabs(hq[2]);
        const double Er = Eg[r0 + q * NS];
        w9[0] = fmax(w9[0], Er * c0); w9[1] = fmax(w9[1], Er * c1); w9[2] = fmax(w9[2], Er * c2);
        const double rn = fmax(c0 * m3[0][0], fmax(c1 * m3[0][1], c2 * m3[0][2]));
        Eg[r0 + q * NS] = Er / sqrt(admm_limit_scaling(Er * rn));
      }
#pragma unroll
      for (int qq = 0; qq < 12; ++qq) {
        const int axsel = qq / 4, dsel = 1 + (qq & 1);
        const double Er = Eg[r0 + (M + qq) * NS];
        w9[dsel * 3 + axsel] = fmax(w9[dsel * 3 + axsel], Er);
        Eg[r0 + (M + qq) * NS] = Er / sqrt(admm_limit_scaling(Er * m3[dsel][axsel]));
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) gs[(size_t)smp * 12 + q] = w9[q];
    }
    __syncthreads();
    for (int e = tid; e < n; e += nt) {
      const int i = e / NB, axx = (e % NB) / D, col = e % D;
      double an = aty[e];
      for (int j = 0; j < R; ++j) {
        const double *gj = gs + (size_t)(i * R + j) * 12;
        const double *bj = be + (size_t)j * 3 * D;
        an = fmax(an, fmax(gj[axx] * fabs(bj[col]), fmax(gj[3 + axx] * fabs(bj[D + col]), gj[6 + axx] * fabs(bj[2 * D + col]))));
      }
      const double cn = fmax(rhs[e], an * Dv[e]);
      xt[e] = Dv[e] / sqrt(admm_limit_scaling(cn));  // the new D_j (every norm above was taken with the old one)
    }
    __syncthreads();
    for (int e = tid; e < n; e += nt) Dv[e] = xt[e];
    if (tid < 24) red[tid] = 0.0;
    __syncthreads();
    // cost scaling: mean column norm of P_bar with the new D
    {
      double l_sum = 0.0;
      for (int e = tid; e < n; e += nt) {
        const int i = e / NB, cr = e % D;
        if (cr < S) {
          const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
          double pn = 0.0;
          for (int k = 0; k < S; ++k) pn = fmax(pn, fabs(qs * qblk1<S>(cr, k, a.m34)) * Dv[e - cr + k]);
          l_sum += pn * Dv[e];
        }
      }
      atomicAdd(&red[0], l_sum);
    }
    __syncthreads();
    {
      const double mean = admm_limit_scaling(red[0] / (double)n);
      cobj /= fmax(mean, 1.0);  // (|q|_inf = 0 -> limit_scaling -> 1)
    }
    __syncthreads();
  }

  double rho = a.p.rho;
  const double sigma = a.p.sigma, alpha = a.p.alpha;

  // ---- assemble + factorise the block tridiagonal H = P_bar + sigma I + A_bar' R A_bar (again whenever rho changes) ----------
  auto factorize = [&]() {
    const double rho_e = 1.0e3 * rho;
    // per sample: S6 = sum_q E_q^2 a_q a_q' (xx xy xz yy yz zz), wv[axis], wa[axis] = sums of E^2 over the +- rows   -> gs[smp][12]
    for (int smp = tid; smp < NS; smp += nt) {
      const int i = smp / R;
      const int64_t r0 = me + smp;
      double w12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int q = 0; q < M; ++q) {
        const double *hq = hpl + ((int64_t)i * M + q) * 4;
        const double Er = Eg[r0 + q * NS], e2 = Er * Er;
        w12[0] += e2 * hq[0] * hq[0]; w12[1] += e2 * hq[0] * hq[1]; w12[2] += e2 * hq[0] * hq[2];
        w12[3] += e2 * hq[1] * hq[1]; w12[4] += e2 * hq[1] * hq[2]; w12[5] += e2 * hq[2] * hq[2];
      }
#pragma unroll
      for (int qq = 0; qq < 12; ++qq) {
        const int axsel = qq / 4, dsel = 1 + (qq & 1);
        const double Er = Eg[r0 + (M + qq) * NS];
        w12[6 + (dsel - 1) * 3 + axsel] += Er * Er;
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) gs[(size_t)smp * 12 + q] = w12[q];
    }
    __syncthreads();
    auto sym6 = [](int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r; return lo == 0 ? hi : (lo == 1 ? 2 + hi : 5); };
    for (int i = 0; i < N; ++i) {
      const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
      const double ratio = (i > 0) ? Tn[i - 1] / Tn[i] : 0.0;
      // off-diagonal block (i, i-1) -> Lo[i-1] = Ho * Linv_{i-1}'
      if (i > 0) {
        for (int e = tid; e < NB * NB; e += nt) {
          const int r = e / NB, c = e % NB;
          // Ho[r][k] = rho_e * E^2 (continuity row of knot i-1, derivative k_r) * (-d! ratio^d) D_r * e1_d[k] D_k for the same axis
          const int axr = r / D, kr = D - 1 - r % D;
          double acc = 0.0;
          if (kr < S) {
            const double Er = eqE[eq_index(2, i - 1, axr, kr)];
            const double coef = -rho_e * Er * Er * fallf(kr, kr) * pow(ratio, (double)kr) * Dv[i * NB + r];
            const double *Li = Linv + (size_t)(i - 1) * NB * NB + (size_t)c * NB;  // row c of Linv_{i-1}
            for (int col = 0; col < D; ++col) {
              const int k = D - 1 - col;
              if (k >= kr) acc += fallf(k, kr) * Dv[(i - 1) * NB + axr * D + col] * Li[axr * D + col];
            }
            acc *= coef;
          }
          Lo[(size_t)(i - 1) * NB * NB + e] = acc;
        }
        __syncthreads();
      }
      for (int e = tid; e < NB * NB; e += nt) {
        const int r = e / NB, c = e % NB;
        const int axr = r / D, axc = c / D, cr = r % D, cc = c % D, kr = D - 1 - cr, kc = D - 1 - cc;
        double v = 0.0;
        {  // inequality rows: sum_j [S_j (x) b0 b0' + diag(wv_j) (x) b1 b1' + diag(wa_j) (x) b2 b2']
          const int s6 = sym6(axr, axc);
          double acc = 0.0;
          for (int j = 0; j < R; ++j) {
            const double *gj = gs + (size_t)(i * R + j) * 12;
            const double *bj = be + (size_t)j * 3 * D;
            acc += gj[s6] * bj[cr] * bj[cc];
            if (axr == axc) acc += gj[6 + axr] * bj[D + cr] * bj[D + cc] + gj[9 + axr] * bj[2 * D + cr] * bj[2 * D + cc];
          }
          v = rho * acc;
        }
        if (axr == axc) {
          if (cr < S && cc < S) v += qs * qblk1<S>(cr, cc, a.m34);
          double eq = 0.0;
          const int dmax = (i < N - 1) ? S : 3, kind = (i < N - 1) ? 2 : 1;  // continuity rows of knot i / end rows
          for (int d = 0; d < dmax; ++d)
            if (kr >= d && kc >= d) {
              const double Er = eqE[eq_index(kind, i, axr, d)];
              eq += Er * Er * fallf(kr, d) * fallf(kc, d);
            }
          if (i > 0 && r == c && kr < S) {  // knot i-1 rows seen from the right piece
            const double Er = eqE[eq_index(2, i - 1, axr, kr)];
            const double cf = fallf(kr, kr) * pow(ratio, (double)kr);
            eq += Er * Er * cf * cf;
          }
          if (i == 0 && r == c && kr < 3) {  // start rows
            const double Er = eqE[eq_index(0, 0, axr, kr)];
            eq += Er * Er * fallf(kr, kr) * fallf(kr, kr);
          }
          v += rho_e * eq;
        }
        v *= Dv[i * NB + r] * Dv[i * NB + c];
        if (r == c) v += sigma;
        if (i > 0) {  // Schur complement
          const double *lr = Lo + (size_t)(i - 1) * NB * NB + (size_t)r * NB;
          const double *lc = Lo + (size_t)(i - 1) * NB * NB + (size_t)c * NB;
          double acc = 0.0;
          for (int k = 0; k < NB; ++k) acc += lr[k] * lc[k];
          v -= acc;
        }
        Hd[e] = v;
      }
      __syncthreads();
      // Cholesky of Hd (lower), in place
      for (int k = 0; k < NB; ++k) {
        if (tid == 0) Hd[k * NB + k] = sqrt(Hd[k * NB + k]);
        __syncthreads();
        const double piv = Hd[k * NB + k];
        for (int r = k + 1 + tid; r < NB; r += nt) Hd[r * NB + k] /= piv;
        __syncthreads();
        for (int e = tid; e < NB * NB; e += nt) {
          const int r = e / NB, c = e % NB;
          if (c > k && r >= c) Hd[e] -= Hd[r * NB + k] * Hd[c * NB + k];
        }
        __syncthreads();
      }
      // Linv_i = L^-1 (lower): column c by forward substitution, one thread per column
      double *Li = Linv + (size_t)i * NB * NB;
      for (int c = tid; c < NB; c += nt) {
        for (int r = 0; r < NB; ++r) {
          double v = (r == c) ? 1.0 : 0.0;
          for (int k = c; k < r; ++k) v -= Hd[r * NB + k] * Li[k * NB + c];
          Li[r * NB + c] = (r >= c) ? v / Hd[r * NB + r] : 0.0;
        }
      }
      __syncthreads();
    }
  };

  // x~ = H^-1 rhs  (rhs -> xt).  Every NB-long dot product is split over 8 adjacent lanes and reduced
  // with shuffles (NB*8 <= 192 of the 256 threads work; `aty` holds the intermediate block vector).
  auto solve = [&]() {
    const int row = tid >> 3, sub = tid & 7;
    auto reduce8 = [&](double v) {
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      return v;
    };
    // forward: y_i = Linv_i (rhs_i - Lo_{i-1} y_{i-1})
    for (int i = 0; i < N; ++i) {
      double v = 0.0;
      if (row < NB && i > 0) {
        const double *lo = Lo + (size_t)(i - 1) * NB * NB + (size_t)row * NB;
        for (int k = sub; k < NB; k += 8) v += lo[k] * xt[(i - 1) * NB + k];
      }
      v = reduce8(v);
      if (row < NB && sub == 0) aty[row] = rhs[i * NB + row] - v;
      __syncthreads();
      v = 0.0;
      if (row < NB) {
        const double *li = Linv + (size_t)i * NB * NB + (size_t)row * NB;
        for (int k = sub; k <= row; k += 8) v += li[k] * aty[k];
      }
      v = reduce8(v);
      if (row < NB && sub == 0) xt[i * NB + row] = v;
      __syncthreads();
    }
    // backward: x_i = Linv_i' (y_i - Lo_i' x_{i+1})
    for (int i = N - 1; i >= 0; --i) {
      double v = 0.0;
      if (row < NB && i < N - 1) {
        const double *lo = Lo + (size_t)i * NB * NB;
        for (int k = sub; k < NB; k += 8) v += lo[k * NB + row] * xt[(i + 1) * NB + k];
      }
      v = reduce8(v);
      if (row < NB && sub == 0) aty[row] = xt[i * NB + row] - v;
      __syncthreads();
      v = 0.0;
      if (row < NB) {
        const double *li = Linv + (size_t)i * NB * NB;
        for (int k = row + sub; k < NB; k += 8) v += li[k * NB + row] * aty[k];
      }
      v = reduce8(v);
      if (row < NB && sub == 0) xt[i * NB + row] = v;
      __syncthreads();
    }
  };
  // w of every row -> A'E w into `dst` (unscaled by D), used when the right-hand side is rebuilt
  auto scatter_eq = [&](double *dst, int kind, int i0, int ax, int d, double cf2, double wv) {
    if (kind == 0) {
      atomicAdd(&dst[ax * D + (D - 1 - d)], fallf(d, d) * wv);
    } else {
      for (int col = 0; col < D; ++col) {
        const int k = D - 1 - col;
        if (k >= d) atomicAdd(&dst[i0 * NB + ax * D + col], fallf(k, d) * wv);
      }
      if (kind == 2) atomicAdd(&dst[(i0 + 1) * NB + ax * D + (D - 1 - d)], cf2 * wv);
    }
  };

  // ---- init -----------------------------------------------------------------------------------
  for (int e = tid; e < n; e += nt) { x[e] = 0.0; rhs[e] = 0.0; }
  for (int64_t e = tid; e < mtot; e += nt) wg[e] = 0.0;  // (the first iteration reads z = y = 0 regardless: cold start)
  __syncthreads();
  factorize();
  __syncthreads();

  int it = 0, status = -2;  // OSQP_MAX_ITER_REACHED unless decided below
  double rp = 0.0, rd = 0.0;
  for (it = 1; it <= a.p.max_iter; ++it) {
    const bool check = (it % a.p.check_every) == 0;
    const bool first = it == 1;  // cold start: z = y = 0 whatever the stored row state says
    const double rho_e = 1.0e3 * rho, inv_rho = 1.0 / rho;
    solve();
    // x+ = alpha x~ + (1-alpha) x ; the next right-hand side is sigma x+ + D A'E (rho z+ - y+): the A' part is collected in rhs
    for (int e = tid; e < n; e += nt) {
      const double xn = alpha * xt[e] + (1.0 - alpha) * x[e];
      x[e] = xn;
      xs[e] = Dv[e] * xt[e];
      xns[e] = Dv[e] * xn;
      rhs[e] = 0.0;
      aty[e] = 0.0;
      ady[e] = 0.0;
    }
    if (tid < 24) red[tid] = 0.0;
    __syncthreads();
    // unscaled (OSQP's termination test) and scaled (its rho estimate) maxima
    double l_rp = 0.0, l_ax = 0.0, l_z = 0.0, l_dy = 0.0, l_sup = 0.0, s_rp = 0.0, s_ax = 0.0, s_z = 0.0;
    // ---- equality rows ------------------------------------------------------------------------
    for (int r = tid; r < me; r += nt) {
      double zt = 0.0, ax_new = 0.0;
      int i0, ax, d, kind;
      eq_row(r, i0, ax, d, kind);
      const double Er = eqE[r], bval = Er * eqb[r];
      const double cf2 = eqc[r];
      if (kind == 0) {
        zt = fallf(d, d) * xs[ax * D + (D - 1 - d)];
        ax_new = fallf(d, d) * xns[ax * D + (D - 1 - d)];
      } else {
        const double *xb = xs + i0 * NB + ax * D, *xn = xns + i0 * NB + ax * D;
        for (int col = 0; col < D; ++col) {
          const int k = D - 1 - col;
          if (k >= d) { const double f = fallf(k, d); zt += f * xb[col]; ax_new += f * xn[col]; }
        }
        if (kind == 2) {
          zt += cf2 * xs[(i0 + 1) * NB + ax * D + (D - 1 - d)];
          ax_new += cf2 * xns[(i0 + 1) * NB + ax * D + (D - 1 - d)];
        }
      }
      zt *= Er;
      ax_new *= Er;
      const double wo = wg[r];
      const double zo = first ? 0.0 : bval, yo = first ? 0.0 : rho_e * (wo - bval);
      const double zr = alpha * zt + (1.0 - alpha) * zo;
      const double zn = bval;  // l = u = b
      const double wn = zr + yo / rho_e;
      const double yn = rho_e * (wn - zn);
      wg[r] = wn;
      const double w = rho_e * zn - yn;
      scatter_eq(rhs, kind, i0, ax, d, cf2, Er * w);
      if (check) {
        scatter_eq(aty, kind, i0, ax, d, cf2, Er * yn);
        // OSQP tests the UNSCALED residuals: E^-1 back to the normalised row, which was scaled by T^d when the QP was normalised
        const double un = a.p.scaled_termination ? 1.0 : eqs[r] / Er;
        l_rp = fmax(l_rp, fabs(ax_new - zn) * un);
        l_ax = fmax(l_ax, fabs(ax_new) * un);
        l_z = fmax(l_z, fabs(zn) * un);
        s_rp = fmax(s_rp, fabs(ax_new - zn));
        s_ax = fmax(s_ax, fabs(ax_new));
        s_z = fmax(s_z, fabs(zn));
        const double dy = yn - yo;  // certificate terms: l = u = b for equality rows
        scatter_eq(ady, kind, i0, ax, d, cf2, Er * dy);
        l_dy = fmax(l_dy, fabs(dy));
        l_sup += bval * dy;
      }
    }
    // ---- inequality rows, one sample (piece i, j) per thread ---------------------------------------
    for (int smp = tid; smp < NS; smp += nt) {
      const int i = smp / R, j = smp % R;
      const double *bj = be + (size_t)j * 3 * D;
      const double *xb = xs + i * NB, *xn = xns + i * NB;
      double s3[3][3], s3n[3][3];  // [d][axis] state rows of x~ and of x+
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int axx = 0; axx < 3; ++axx) {
          double acc = 0.0, accn = 0.0;
          for (int col = 0; col < D; ++col) {
            acc += bj[d * D + col] * xb[axx * D + col];
            accn += bj[d * D + col] * xn[axx * D + col];
          }
          s3[d][axx] = acc;
          s3n[d][axx] = accn;
        }
      const int64_t r0 = me + smp;            // row q of this sample lives at r0 + q*NS (coalesced across samples)
      double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gy[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      double gd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      const double Ti = Tn[i], rTi = a.p.scaled_termination ? 1.0 : 1.0 / Tn[i];
      // one ADMM row update (all of it in the scaled row: zt, axn, hv already carry E_r); returns w = rho z+ - y+ and (at
      // check iterations) y+ and dy; `un` takes a scaled residual back to the reference's row
      auto row_update = [&](double zt, double axn, double hv, double wo, double un, double &wn, double &zn, double &yn,
                            double &dy) {
        const double zo = first ? 0.0 : fmin(wo, hv), yo = first ? 0.0 : rho * (wo - zo);
        const double zr = alpha * zt + (1.0 - alpha) * zo;
        wn = zr + yo * inv_rho;
        zn = fmin(wn, hv);  // l = -inf
        yn = rho * (wn - zn);
        dy = yn - yo;
        if (check) {
          l_rp = fmax(l_rp, fabs(axn - zn) * un);  // un: 1 (corridor), 1/T (velocity), 1/T^2 (acceleration), each over E_r
          l_ax = fmax(l_ax, fabs(axn) * un);
          l_z = fmax(l_z, fabs(zn) * un);
          s_rp = fmax(s_rp, fabs(axn - zn));
          s_ax = fmax(s_ax, fabs(axn));
          s_z = fmax(s_z, fabs(zn));
          // l = -inf: only the positive part of dy can certify (a negative part makes the support +inf)
          l_dy = fmax(l_dy, fabs(dy));
          l_sup += (dy > 0.0) ? hv * dy : (dy < 0.0 ? 1e300 : 0.0);
        }
      };
      // ---- corridor rows, four at a time so that their state loads are in flight together
      for (int q0 = 0; q0 < M; q0 += 4) {
        double wo[4], Er[4], cf[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u;
          const bool ok = q < M;
          wo[u] = ok ? wg[r0 + q * NS] : 0.0;
          Er[u] = ok ? Eg[r0 + q * NS] : 1.0;
          const double *hq = hpl + ((int64_t)i * M + (ok ? q : 0)) * 4;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) cf[u][w4] = ok ? hq[w4] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u;
          if (q < M) {
            const double c0 = Er[u] * cf[u][0], c1 = Er[u] * cf[u][1], c2 = Er[u] * cf[u][2];
            const double zt = c0 * s3[0][0] + c1 * s3[0][1] + c2 * s3[0][2];
            const double axn = c0 * s3n[0][0] + c1 * s3n[0][1] + c2 * s3n[0][2];
            double wn, zn, yn, dy;
            row_update(zt, axn, Er[u] * cf[u][3], wo[u], a.p.scaled_termination ? 1.0 : 1.0 / Er[u], wn, zn, yn, dy);
            wg[r0 + q * NS] = wn;
            const double w = rho * zn - yn;
            g[0][0] += w * c0; g[0][1] += w * c1; g[0][2] += w * c2;
            if (check) {
              gy[0][0] += yn * c0; gy[0][1] += yn * c1; gy[0][2] += yn * c2;
              gd[0][0] += dy * c0; gd[0][1] += dy * c1; gd[0][2] += dy * c2;
            }
          }
        }
      }
      // ---- the 12 box rows (+v, +a, -v, -a per axis), all loads up front
      {
        double wo[12], Er[12];
#pragma unroll
        for (int qq = 0; qq < 12; ++qq) {
          wo[qq] = wg[r0 + (M + qq) * NS];
          Er[qq] = Eg[r0 + (M + qq) * NS];
        }
#pragma unroll
        for (int qq = 0; qq < 12; ++qq) {
          const int axsel = qq / 4, w4 = qq % 4, dsel = 1 + (w4 & 1);
          const double sgn = ((w4 < 2) ? 1.0 : -1.0) * Er[qq];
          const double hv = Er[qq] * ((dsel == 1) ? a.vmax * Ti : a.amax * Ti * Ti);
          double wn, zn, yn, dy;
          row_update(sgn * s3[dsel][axsel], sgn * s3n[dsel][axsel], hv, wo[qq],
                     a.p.scaled_termination ? 1.0 : ((dsel == 1) ? rTi : rTi * rTi) / Er[qq], wn, zn, yn, dy);
          wg[r0 + (M + qq) * NS] = wn;
          const double w = rho * zn - yn;
          g[dsel][axsel] += sgn * w;
          if (check) {
            gy[dsel][axsel] += sgn * yn;
            gd[dsel][axsel] += sgn * dy;
          }
        }
      }
      // A'E w of this sample: the 3x3 block g goes to LDS and is contracted with the basis table by the
      // threads that own the n outputs after the barrier (no 20-way contended atomics every iteration)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int axx = 0; axx < 3; ++axx) gs[(size_t)smp * 12 + d * 3 + axx] = g[d][axx];
      if (check)
      for (int axx = 0; axx < 3; ++axx)
        for (int col = 0; col < D; ++col) {
          {
            const double vy = gy[0][axx] * bj[col] + gy[1][axx] * bj[D + col] + gy[2][axx] * bj[2 * D + col];
            if (vy != 0.0) atomicAdd(&aty[i * NB + axx * D + col], vy);
            const double vd = gd[0][axx] * bj[col] + gd[1][axx] * bj[D + col] + gd[2][axx] * bj[2 * D + col];
            if (vd != 0.0) atomicAdd(&ady[i * NB + axx * D + col], vd);
          }
        }
    }
    if (check) {
      atomic_max_pos(&red[0], l_rp);
      atomic_max_pos(&red[1], l_ax);
      atomic_max_pos(&red[2], l_z);
      atomic_max_pos(&red[6], l_dy);
      if (l_sup != 0.0) atomicAdd(&red[7], fmin(l_sup, 1e300));
      atomic_max_pos(&red[12], s_rp);
      atomic_max_pos(&red[13], s_ax);
      atomic_max_pos(&red[14], s_z);
    }
    __syncthreads();
    for (int e = tid; e < n; e += nt) {
      const int i = e / NB, axx = (e % NB) / D, col = e % D;
      double acc = rhs[e];
      for (int j = 0; j < R; ++j) {
        const double *gj = gs + (size_t)(i * R + j) * 12;
        const double *bj = be + (size_t)j * 3 * D;
        acc += gj[axx] * bj[col] + gj[3 + axx] * bj[D + col] + gj[6 + axx] * bj[2 * D + col];
      }
      rhs[e] = sigma * x[e] + Dv[e] * acc;
    }
    __syncthreads();
    if (check) {
      // dual residual: P_bar x_bar + A_bar'y_bar = D (c P xs + A'E y_bar); norms of the two parts; back in the reference's variables
      // and objective: divide by c D_j, times T^k
      double l_rd = 0.0, l_px = 0.0, l_aty = 0.0, l_ady = 0.0, s_rd = 0.0, s_px = 0.0, s_aty = 0.0;
      for (int e = tid; e < n; e += nt) {
        const int i = e / NB, cr = e % D;
        double px = 0.0;
        if (cr < S) {
          const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
          const double *xb = xns + (e - cr);
          for (int k = 0; k < S; ++k) px += qs * qblk1<S>(cr, k, a.m34) * xb[k];
        }
        const double und = a.p.scaled_termination ? Dv[e] : pow(Tn[i], (double)(D - 1 - cr)) / cobj;
        l_rd = fmax(l_rd, fabs(px + aty[e]) * und);
        l_px = fmax(l_px, fabs(px) * und);
        l_aty = fmax(l_aty, fabs(aty[e]) * und);
        s_rd = fmax(s_rd, fabs(px + aty[e]) * Dv[e]);
        s_px = fmax(s_px, fabs(px) * Dv[e]);
        s_aty = fmax(s_aty, fabs(aty[e]) * Dv[e]);
        l_ady = fmax(l_ady, fabs(ady[e]) * Dv[e]);
      }
      atomic_max_pos(&red[3], l_rd);
      atomic_max_pos(&red[4], l_px);
      atomic_max_pos(&red[5], l_aty);
      atomic_max_pos(&red[8], l_ady);
      atomic_max_pos(&red[15], s_rd);
      atomic_max_pos(&red[16], s_px);
      atomic_max_pos(&red[17], s_aty);
      __syncthreads();
      rp = red[0];
      rd = red[3];
      const double eps_p = a.p.eps_abs + a.p.eps_rel * fmax(red[1], red[2]);
      const double eps_d = a.p.eps_abs + a.p.eps_rel * fmax(red[4], red[5]);
      if (rp <= eps_p && rd <= eps_d) {
        status = 1;
        break;
      }
      // OSQP's primal infeasibility test (osqp/src/auxil.c is_primal_infeasible, eps_prim_inf = 1e-4), on the scaled problem:
      //   ||A' dy||_inf <= eps ||dy||_inf   and   u'(dy)+ + l'(dy)- <= -eps ||dy||_inf
      {
        const double ndy = red[6];
        if (ndy > 1e-4 && red[8] <= 1e-4 * ndy && red[7] <= -1e-4 * ndy) {
          status = -3;
          break;
        }
      }
      if (a.p.adapt_every > 0 && (it % a.p.adapt_every) == 0) {
        // osqp/src/auxil.c compute_rho_estimate: the SCALED residuals, each over the larger of the norms it is made of
        const double np_ = red[12] / (fmax(red[13], red[14]) + 1e-10), nd_ = red[15] / (fmax(red[16], red[17]) + 1e-10);
        double rho_new = rho * sqrt(np_ / (nd_ + 1e-10));
        rho_new = fmin(fmax(rho_new, 1e-6), 1e6);
        if (rho_new > 5.0 * rho || rho_new < 0.2 * rho) {
          // rhs was accumulated with the old rho: rebuild it for the new one from z, y (decoded from the row
          // state with the OLD rho, re-encoded with the new one: y does not change when rho does)
          const double rho_old = rho;
          rho = rho_new;
          __syncthreads();
          factorize();
          const double rho_e2 = 1.0e3 * rho;
          for (int e = tid; e < n; e += nt) rhs[e] = 0.0;
          __syncthreads();
          for (int r = tid; r < me; r += nt) {
            int i0, ax, d, kind;
            eq_row(r, i0, ax, d, kind);
            const double Er = eqE[r];
            const double zo = Er * eqb[r], yo = 1.0e3 * rho_old * (wg[r] - zo);
            const double w = rho_e2 * zo - yo;
            wg[r] = zo + yo / rho_e2;
            scatter_eq(rhs, kind, i0, ax, d, eqc[r], Er * w);
          }
          for (int smp = tid; smp < NS; smp += nt) {
            const int i = smp / R, j = smp % R;
            const double *bj = be + (size_t)j * 3 * D;
            const int64_t r0 = me + smp;
            double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int q = 0; q < rows_per_sample; ++q) {
              const double *hq = hpl + ((int64_t)i * M + (q < M ? q : 0)) * 4;
              const int qq = q - M, w4 = qq % 4;
              const double Er = Eg[r0 + q * NS];
              const double hv = Er * ((q < M) ? hq[3] : (((w4 & 1) == 0) ? a.vmax * Tn[i] : a.amax * Tn[i] * Tn[i]));
              const double wo = wg[r0 + q * NS];
              const double zo = fmin(wo, hv), yo = rho_old * (wo - zo);
              const double w = Er * (rho * zo - yo);
              wg[r0 + q * NS] = zo + yo / rho;
              if (q < M) {
                g[0][0] += w * hq[0]; g[0][1] += w * hq[1]; g[0][2] += w * hq[2];
              } else {
                g[1 + (w4 & 1)][qq / 4] += ((w4 < 2) ? 1.0 : -1.0) * w;
              }
            }
            for (int axx = 0; axx < 3; ++axx)
              for (int col = 0; col < D; ++col) {
                const double v = g[0][axx] * bj[col] + g[1][axx] * bj[D + col] + g[2][axx] * bj[2 * D + col];
                if (v != 0.0) atomicAdd(&rhs[i * NB + axx * D + col], v);
              }
          }
          __syncthreads();
          for (int e = tid; e < n; e += nt) rhs[e] = sigma * x[e] + Dv[e] * rhs[e];
          __syncthreads();
        }
      }
    }
    __syncthreads();
  }
  if (it > a.p.max_iter) it = a.p.max_iter;
  __syncthreads();
  // the solution in the normalised variables (x = D x_bar); the multipliers of the normalised, cost-unscaled problem are E y_bar / c
  for (int e = tid; e < n; e += nt) x[e] *= Dv[e];
  __syncthreads();
  // ---- implicit time gradient of the optimal cost (layers.py:120-147) -------
  // The loss the reference back-propagates is the QP objective itself, J*(T) = min 1/2 z'Q(T)z.  For a
  // value function no KKT solve is needed: dJ*/dT_i = dL/dT_i at the optimum (envelope theorem),
  // L = J + y'(A x - u) in the variables of THIS kernel (normalised time), where only
  //   J_i = T_i^(1-2s) 1/2 x_i'Q1 x_i,  the start/end values b = state T^d,  the continuity coefficient
  //   -d!(T_i/T_i+1)^d,  and the box bounds vmax T_i, amax T_i^2
  // depend on T; the corridor rows do not.  (y = E y_bar / c: the duals of the iteration belong to the scaled problem.)
  if (a.gradT) {
    for (int i = tid; i < N; i += nt) {
      const double qs = pow(Tn[i], (double)(1 - 2 * S));
      double Ji = 0.0;
      for (int axx = 0; axx < 3; ++axx) {
        const double *xb = x + i * NB + axx * D;
        for (int j = 0; j < S; ++j)
          for (int k = 0; k < S; ++k) Ji += 0.5 * qs * qblk1<S>(j, k, a.m34) * xb[j] * xb[k];
      }
      rhs[i] = (double)(1 - 2 * S) * Ji / Tn[i];  // rhs is free now: accumulator [N]
    }
    __syncthreads();
    const double inv_c = 1.0 / cobj;
    for (int r = tid; r < me; r += nt) {
      int i0, ax, d, kind;
      eq_row(r, i0, ax, d, kind);
      if (d == 0) continue;
      const double Er = eqE[r];
      const double yr = Er * 1.0e3 * rho * (wg[r] - Er * eqb[r]) * inv_c;
      if (kind != 2) {
        atomicAdd(&rhs[i0], -yr * (double)d * eqb[r] / Tn[i0]);
      } else {
        const double t = yr * (double)d * eqc[r] * x[(i0 + 1) * NB + ax * D + (D - 1 - d)];
        atomicAdd(&rhs[i0], t / Tn[i0]);
        atomicAdd(&rhs[i0 + 1], -t / Tn[i0 + 1]);
      }
    }
    for (int smp = tid; smp < NS; smp += nt) {
      const int i = smp / R;
      const int64_t r0 = me + smp;
      double sv = 0.0, sa = 0.0;
      for (int qq = 0; qq < 12; ++qq) {
        const bool acc = (qq % 4) & 1;
        const double Er = Eg[r0 + (M + qq) * NS];
        const double hv = Er * (acc ? a.amax * Tn[i] * Tn[i] : a.vmax * Tn[i]);
        const double yv = Er * rho * fmax(wg[r0 + (M + qq) * NS] - hv, 0.0);
        if (acc) sa += yv;
        else sv += yv;
      }
      if (sv != 0.0 || sa != 0.0) atomicAdd(&rhs[i], -inv_c * (a.vmax * sv + 2.0 * a.amax * Tn[i] * sa));
    }
    __syncthreads();
    for (int i = tid; i < N; i += nt) a.gradT[b * N + i] = rhs[i];
  }
  // ---- unscale and report ---------------------------------------------------------------------
  __syncthreads();
  for (int e = tid; e < n; e += nt) {
    const int i = e / NB, k = D - 1 - e % D;
    a.coeffs[b * n + e] = x[e] * pow(Tn[i], (double)(-k));
  }
  if (tid == 0) {
    double obj = 0.0;
    for (int i = 0; i < N; ++i) {
      const double qs = pow(Tn[i], (double)(1 - 2 * S));
      for (int axx = 0; axx < 3; ++axx) {
        const double *xb = x + i * NB + axx * D;
        for (int j = 0; j < S; ++j)
          for (int k = 0; k < S; ++k) obj += 0.5 * qs * qblk1<S>(j, k, a.m34) * xb[j] * xb[k];
      }
    }
    a.obj[b] = obj;
    a.status[b] = status;
    a.iters[b] = it;
    a.res[b * 2] = rp;
    a.res[b * 2 + 1] = rd;
  }
}

// LDS bytes without / with the per-row state resident; the launcher keeps it in LDS when the larger figure fits.
template <int S>
inline size_t qp_admm_lds_bytes(int N, int R, int M, bool zy_in_lds) {
  constexpr int D = 2 * S, NB = 3 * D;
  const size_t n = (size_t)NB * N;
  const size_t hp = (size_t)N * M * 4 * sizeof(double) <= kAdmmHpLdsBytes ? (size_t)N * M * 4 : 0;
  const size_t me = (size_t)(3 * (6 + S * (N - 1)));
  const size_t mtot = me + (size_t)N * R * (M + 12);
  return sizeof(double) * ((size_t)2 * N * NB * NB + NB * NB + 8 * n + (size_t)R * 3 * D + N + 24 + hp + 4 * me + (size_t)N * R * 12 +
                           (zy_in_lds ? mtot : 0));
}

}  // namespace anet
