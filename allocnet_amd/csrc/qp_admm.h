// Batched inequality-constrained QP solve replacing OSQP in QPSolver::solve
// (planner/qp_solver.hpp:299-358; OSQP release-0.6.3 is a third-party dependency that is not part of
// the reference tree).  Same problem, same algorithm family: OSQP's ADMM (Stellato et al., "OSQP: an
// operator splitting solver for quadratic programs", Math. Prog. Comp. 2020, Algorithm 1) with its
// defaults -- rho 0.1, 1e3*rho on equality rows, sigma 1e-6, alpha 1.6, eps_abs = eps_rel = 1e-3,
// max_iter 4000, termination check every 25 iterations, rho adaptation by the residual-ratio rule.
//
// MI355X-specific restatement (one 256-thread workgroup per trajectory; everything in LDS, the per-row state too
// when it fits -- one number per row instead of OSQP's z and y, see `wg` below):
//  * the QP is posed in NORMALISED time (variables a~_k = c_k T^k, derivative rows scaled by T^d), which
//    is an analytic equilibration: every basis row depends only on tau_j = j/res, identical for all
//    pieces and trajectories (tables + their Gram matrices built once per workgroup), and the cost
//    block is T^(1-2s) times a constant matrix.  It replaces OSQP's iterative Ruiz scaling.
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

template <int S>
__global__ void __launch_bounds__(256) k_qp_admm(AdmmArgs a) {
  constexpr int D = 2 * S, NB = 3 * D;
  const int N = a.N, R = a.R, M = a.M;
  const int n = NB * N;
  const int me = 3 * (6 + S * (N - 1));
  const int rows_per_sample = M + 12;
  const int64_t mtot = me + (int64_t)N * R * rows_per_sample;
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;

  extern __shared__ double lds[];
  double *Linv = lds;                       // [N][NB*NB]  inverse of the Cholesky factor of diagonal block i
  double *Lo = Linv + (size_t)N * NB * NB;  // [N][NB*NB]  L_{i+1,i} (row = block i+1, col = block i)
  double *Hd = Lo + (size_t)N * NB * NB;    // [NB*NB] scratch
  double *x = Hd + NB * NB;                 // [n]
  double *xt = x + n;                       // [n]
  double *rhs = xt + n;                     // [n]
  double *aty = rhs + n;                    // [n]  A'y (dual residual) / temp
  double *ady = aty + n;                    // [n]  A'(y+ - y) (primal infeasibility certificate)
  double *be = ady + n;                     // [R][3][D] basis rows at tau_j
  double *G0 = be + (size_t)R * 3 * D;      // [D*D]  sum_j b0 b0'
  double *G12 = G0 + D * D;                 // [D*D]  sum_j (b1 b1' + b2 b2')
  double *Sp = G12 + D * D;                 // [N][9]  sum_q a_q a_q'
  double *Tn = Sp + (size_t)N * 9;          // [N]
  double *red = Tn + N;                     // [12] reductions / broadcast
  double *hp_l = red + 12;                  // [N*M*4] polytope rows (only when they fit the budget below)
  double *eqb = hp_l + ((size_t)N * M * 4 * sizeof(double) <= kAdmmHpLdsBytes ? (size_t)N * M * 4 : 0);  // [me] rhs b
  double *eqc = eqb + me;                    // [me] coefficient of the right-hand piece (continuity rows)
  double *eqs = eqc + me;                    // [me] 1/T^d: undoes the row scaling of the normalisation
  double *gs = eqs + me;                     // [N*R][9] per-sample A'w blocks
  double *zy_l = gs + (size_t)N * R * 9;    // [mtot] when a.zy_in_lds

  const double *Tg = a.T + b * N;
  const double *hp = a.hpolys + b * (int64_t)N * M * 4;
  const double *st = a.state + b * 18;
  // ADMM state per constraint row: OSQP keeps z and y; both are functions of the single number
  //   w = alpha (A x~) + (1-alpha) z_prev + y_prev/rho      (the argument of the projection):
  //   z = Pi(w) = min(w, u) [b for an equality row],   y = rho (w - z).
  // Storing w alone halves the row traffic and lets the state of an 8-piece snap problem sit in LDS.
  double *wg = a.zy_in_lds ? zy_l : a.z + b * mtot;

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
  for (int r = tid; r < me; r += nt) {  // per-row constants of the equality block (do not change per iteration)
    double bv = 0.0, cf = 0.0;
    if (r < 18) {
      const int ax = r / 6, q = r % 6, d = q % 3, i0 = q < 3 ? 0 : N - 1;
      bv = st[(q < 3 ? 0 : 9) + ax * 3 + d] * pow(Tn[i0], (double)d);
    } else {
      const int rr = r - 18, i0 = rr / (3 * S), d = rr % S;
      cf = -fallf(d, d) * pow(Tn[i0] / Tn[i0 + 1], (double)d);
    }
    eqb[r] = bv;
    eqc[r] = cf;
    {
      int i0, d;
      if (r < 18) { const int q = r % 6; d = q % 3; i0 = q < 3 ? 0 : N - 1; }
      else { const int rr = r - 18; i0 = rr / (3 * S); d = rr % S; }
      eqs[r] = pow(Tn[i0], (double)(-d));
    }
  }
  for (int e = tid; e < D * D; e += nt) {
    const int c1 = e / D, c2 = e % D;
    double g0 = 0.0, g12 = 0.0;
    for (int j = 0; j < R; ++j) {
      const double *bj = be + (size_t)j * 3 * D;
      g0 += bj[c1] * bj[c2];
      g12 += bj[D + c1] * bj[D + c2] + bj[2 * D + c1] * bj[2 * D + c2];
    }
    G0[e] = g0;
    G12[e] = g12;
  }
  for (int e = tid; e < N * 9; e += nt) {
    const int i = e / 9, r = (e / 3) % 3, c = e % 3;
    double s2 = 0.0;
    for (int q = 0; q < M; ++q) s2 += hp[((int64_t)i * M + q) * 4 + r] * hp[((int64_t)i * M + q) * 4 + c];
    Sp[e] = s2;
  }
  __syncthreads();
  // objective scaling: 1 / mean_i( T_i^(1-2s) * mean diag of the cost block )
  double cobj;
  {
    double tr = 0.0;
    for (int j = 0; j < S; ++j) tr += qblk1<S>(j, j, a.m34);
    double acc = 0.0;
    for (int i = 0; i < N; ++i) acc += pow(Tn[i], (double)(1 - 2 * S));
    cobj = 1.0 / (acc / N * tr / S);
  }

  double rho = a.p.rho;
  const double sigma = a.p.sigma, alpha = a.p.alpha;

  // ---- assemble + factorise the block tridiagonal H (again whenever rho changes) ----------------
  auto factorize = [&]() {
    const double rho_e = 1.0e3 * rho;
    for (int i = 0; i < N; ++i) {
      const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
      const double ratio = (i > 0) ? Tn[i - 1] / Tn[i] : 0.0;
      // off-diagonal block (i, i-1) -> Lo[i-1] = Ho * Linv_{i-1}'
      if (i > 0) {
        for (int e = tid; e < NB * NB; e += nt) {
          const int r = e / NB, c = e % NB;
          // Ho[r][k] = rho_e * sum_d (-d! ratio^d [k_r == d]) * e1_d[k] for the same axis
          const int axr = r / D, kr = D - 1 - r % D;
          double acc = 0.0;
          if (kr < S) {
            const double coef = -rho_e * fallf(kr, kr) * pow(ratio, (double)kr);
            const double *Li = Linv + (size_t)(i - 1) * NB * NB + (size_t)c * NB;  // row c of Linv_{i-1}
            for (int col = 0; col < D; ++col) {
              const int k = D - 1 - col;
              if (k >= kr) acc += fallf(k, kr) * Li[axr * D + col];
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
        double v = rho * Sp[i * 9 + axr * 3 + axc] * G0[cr * D + cc];
        if (axr == axc) {
          v += rho * 2.0 * G12[cr * D + cc];
          if (cr < S && cc < S) v += qs * qblk1<S>(cr, cc, a.m34);
          if (r == c) v += sigma;
          double eq = 0.0;
          if (i < N - 1) {  // continuity rows of knot i: e1_d e1_d'
            for (int d = 0; d < S; ++d)
              if (kr >= d && kc >= d) eq += fallf(kr, d) * fallf(kc, d);
          } else {  // end rows d < 3
            for (int d = 0; d < 3; ++d)
              if (kr >= d && kc >= d) eq += fallf(kr, d) * fallf(kc, d);
          }
          if (i > 0 && r == c && kr < S) {  // knot i-1 rows seen from the right piece
            const double cf = fallf(kr, kr) * pow(ratio, (double)kr);
            eq += cf * cf;
          }
          if (i == 0 && r == c && kr < 3) eq += fallf(kr, kr) * fallf(kr, kr);  // start rows
          v += rho_e * eq;
        }
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

  // ---- init -----------------------------------------------------------------------------------
  for (int e = tid; e < n; e += nt) { x[e] = 0.0; rhs[e] = 0.0; }
  for (int64_t e = tid; e < mtot; e += nt) wg[e] = 0.0;  // (the first iteration reads z = y = 0 regardless: cold start)
  factorize();
  __syncthreads();

  int it = 0, status = -2;  // OSQP_MAX_ITER_REACHED unless decided below
  double rp = 0.0, rd = 0.0;
  for (it = 1; it <= a.p.max_iter; ++it) {
    const bool check = (it % a.p.check_every) == 0;
    const bool first = it == 1;  // cold start: z = y = 0 whatever the stored row state says
    const double rho_e = 1.0e3 * rho, inv_rho = 1.0 / rho;
    solve();
    // x+ = alpha x~ + (1-alpha) x ; next rhs starts as sigma x+
    for (int e = tid; e < n; e += nt) {
      const double xn = alpha * xt[e] + (1.0 - alpha) * x[e];
      x[e] = xn;
      rhs[e] = sigma * xn;
      aty[e] = 0.0;
      ady[e] = 0.0;
    }
    if (tid < 12) red[tid] = 0.0;
    __syncthreads();
    double l_rp = 0.0, l_ax = 0.0, l_z = 0.0, l_dy = 0.0, l_sup = 0.0;
    // ---- equality rows ------------------------------------------------------------------------
    for (int r = tid; r < me; r += nt) {
      // row = sum over (block, axis, col) of coefficient * variable ; rhs value bval
      double zt = 0.0, ax_new = 0.0, bval = 0.0;
      int i0, ax, d, kind;  // kind 0 start, 1 end, 2 continuity
      if (r < 18) {
        ax = r / 6;
        const int q = r % 6;
        kind = q < 3 ? 0 : 1;
        d = q % 3;
        i0 = kind == 0 ? 0 : N - 1;
        bval = eqb[r];
      } else {
        const int rr = r - 18;
        i0 = rr / (3 * S);
        ax = (rr / S) % 3;
        d = rr % S;
        kind = 2;
      }
      const double cf2 = eqc[r];
      if (kind == 0) {
        zt = fallf(d, d) * xt[ax * D + (D - 1 - d)];
        ax_new = fallf(d, d) * x[ax * D + (D - 1 - d)];
      } else {
        const double *xb = xt + i0 * NB + ax * D, *xn = x + i0 * NB + ax * D;
        for (int col = 0; col < D; ++col) {
          const int k = D - 1 - col;
          if (k >= d) { const double f = fallf(k, d); zt += f * xb[col]; ax_new += f * xn[col]; }
        }
        if (kind == 2) {
          zt += cf2 * xt[(i0 + 1) * NB + ax * D + (D - 1 - d)];
          ax_new += cf2 * x[(i0 + 1) * NB + ax * D + (D - 1 - d)];
        }
      }
      const double wo = wg[r];
      const double zo = first ? 0.0 : bval, yo = first ? 0.0 : rho_e * (wo - bval);
      const double zr = alpha * zt + (1.0 - alpha) * zo;
      const double zn = bval;  // l = u = b
      const double wn = zr + yo / rho_e;
      const double yn = rho_e * (wn - zn);
      wg[r] = wn;
      const double w = rho_e * zn - yn;
      // scatter A' w (and A' y at check iterations)
      auto scatter = [&](double *dst, double wv) {
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
      scatter(rhs, w);
      if (check) {
        scatter(aty, yn);
        // OSQP tests the UNSCALED residuals: this row was scaled by T^d when the QP was normalised
        const double un = a.p.scaled_termination ? 1.0 : eqs[r];
        l_rp = fmax(l_rp, fabs(ax_new - zn) * un);
        l_ax = fmax(l_ax, fabs(ax_new) * un);
        l_z = fmax(l_z, fabs(zn) * un);
        const double dy = yn - yo;  // certificate terms: l = u = b for equality rows
        scatter(ady, dy);
        l_dy = fmax(l_dy, fabs(dy));
        l_sup += bval * dy;
      }
    }
    // ---- inequality rows, one sample (piece i, j) per thread ---------------------------------------
    for (int smp = tid; smp < N * R; smp += nt) {
      const int i = smp / R, j = smp % R;
      const double *bj = be + (size_t)j * 3 * D;
      const double *xb = xt + i * NB, *xn = x + i * NB;
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
      const int64_t NS = (int64_t)N * R;
      double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gy[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      double gd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      const double Ti = Tn[i], rTi = a.p.scaled_termination ? 1.0 : 1.0 / Tn[i];
      // one ADMM row update; returns w = rho z+ - y+ and (at check iterations) y+ and dy
      auto row_update = [&](double zt, double axn, double hv, double wo, double un, double &wn, double &zn, double &yn,
                            double &dy) {
        const double zo = first ? 0.0 : fmin(wo, hv), yo = first ? 0.0 : rho * (wo - zo);
        const double zr = alpha * zt + (1.0 - alpha) * zo;
        wn = zr + yo * inv_rho;
        zn = fmin(wn, hv);  // l = -inf
        yn = rho * (wn - zn);
        dy = yn - yo;
        if (check) {
          l_rp = fmax(l_rp, fabs(axn - zn) * un);  // un: 1 (corridor), 1/T (velocity), 1/T^2 (acceleration)
          l_ax = fmax(l_ax, fabs(axn) * un);
          l_z = fmax(l_z, fabs(zn) * un);
          // l = -inf: only the positive part of dy can certify (a negative part makes the support +inf)
          l_dy = fmax(l_dy, fabs(dy));
          l_sup += (dy > 0.0) ? hv * dy : (dy < 0.0 ? 1e300 : 0.0);
        }
      };
      // ---- corridor rows, four at a time so that their state loads are in flight together
      for (int q0 = 0; q0 < M; q0 += 4) {
        double wo[4], cf[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u;
          const bool ok = q < M;
          wo[u] = ok ? wg[r0 + q * NS] : 0.0;
          const double *hq = hpl + ((int64_t)i * M + (ok ? q : 0)) * 4;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) cf[u][w4] = ok ? hq[w4] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u;
          if (q < M) {
            const double c0 = cf[u][0], c1 = cf[u][1], c2 = cf[u][2];
            const double zt = c0 * s3[0][0] + c1 * s3[0][1] + c2 * s3[0][2];
            const double axn = c0 * s3n[0][0] + c1 * s3n[0][1] + c2 * s3n[0][2];
            double wn, zn, yn, dy;
            row_update(zt, axn, cf[u][3], wo[u], 1.0, wn, zn, yn, dy);
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
        double wo[12];
#pragma unroll
        for (int qq = 0; qq < 12; ++qq) wo[qq] = wg[r0 + (M + qq) * NS];
#pragma unroll
        for (int qq = 0; qq < 12; ++qq) {
          const int axsel = qq / 4, w4 = qq % 4, dsel = 1 + (w4 & 1);
          const double sgn = (w4 < 2) ? 1.0 : -1.0;
          const double hv = (dsel == 1) ? a.vmax * Ti : a.amax * Ti * Ti;
          double wn, zn, yn, dy;
          row_update(sgn * s3[dsel][axsel], sgn * s3n[dsel][axsel], hv, wo[qq], (dsel == 1) ? rTi : rTi * rTi, wn, zn, yn, dy);
          wg[r0 + (M + qq) * NS] = wn;
          const double w = rho * zn - yn;
          g[dsel][axsel] += sgn * w;
          if (check) {
            gy[dsel][axsel] += sgn * yn;
            gd[dsel][axsel] += sgn * dy;
          }
        }
      }
      // A'w of this sample: the 3x3 block g goes to LDS and is contracted with the basis table by the
      // threads that own the n outputs after the barrier (no 20-way contended atomics every iteration)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int axx = 0; axx < 3; ++axx) gs[(size_t)smp * 9 + d * 3 + axx] = g[d][axx];
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
    }
    __syncthreads();
    for (int e = tid; e < n; e += nt) {
      const int i = e / NB, axx = (e % NB) / D, col = e % D;
      double acc = 0.0;
      for (int j = 0; j < R; ++j) {
        const double *gj = gs + (size_t)(i * R + j) * 9;
        const double *bj = be + (size_t)j * 3 * D;
        acc += gj[axx] * bj[col] + gj[3 + axx] * bj[D + col] + gj[6 + axx] * bj[2 * D + col];
      }
      rhs[e] += acc;
    }
    __syncthreads();
    if (check) {
      // dual residual: P x + A'y ; norms of P x and A'y
      double l_rd = 0.0, l_px = 0.0, l_aty = 0.0, l_ady = 0.0;
      for (int e = tid; e < n; e += nt) {
        const int i = e / NB, cr = e % D;
        double px = 0.0;
        if (cr < S) {
          const double qs = cobj * pow(Tn[i], (double)(1 - 2 * S));
          const double *xb = x + (e - cr);
          for (int k = 0; k < S; ++k) px += qs * qblk1<S>(cr, k, a.m34) * xb[k];
        }
        const double und = a.p.scaled_termination ? 1.0 : pow(Tn[i], (double)(D - 1 - cr)) / cobj;  // back to the reference's variables and objective
        l_rd = fmax(l_rd, fabs(px + aty[e]) * und);
        l_px = fmax(l_px, fabs(px) * und);
        l_aty = fmax(l_aty, fabs(aty[e]) * und);
        l_ady = fmax(l_ady, fabs(ady[e]));
      }
      atomic_max_pos(&red[3], l_rd);
      atomic_max_pos(&red[4], l_px);
      atomic_max_pos(&red[5], l_aty);
      atomic_max_pos(&red[8], l_ady);
      __syncthreads();
      rp = red[0];
      rd = red[3];
      const double eps_p = a.p.eps_abs + a.p.eps_rel * fmax(red[1], red[2]);
      const double eps_d = a.p.eps_abs + a.p.eps_rel * fmax(red[4], red[5]);
      if (rp <= eps_p && rd <= eps_d) {
        status = 1;
        break;
      }
      // OSQP's primal infeasibility test (osqp/src/auxil.c is_primal_infeasible, eps_prim_inf = 1e-4):
      //   ||A' dy||_inf <= eps ||dy||_inf   and   u'(dy)+ + l'(dy)- <= -eps ||dy||_inf
      {
        const double ndy = red[6];
        if (ndy > 1e-4 && red[8] <= 1e-4 * ndy && red[7] <= -1e-4 * ndy) {
          status = -3;
          break;
        }
      }
      if (a.p.adapt_every > 0 && (it % a.p.adapt_every) == 0) {
        const double np_ = rp / fmax(fmax(red[1], red[2]), 1e-300), nd_ = rd / fmax(fmax(red[4], red[5]), 1e-300);
        double rho_new = rho * sqrt(np_ / fmax(nd_, 1e-300));
        rho_new = fmin(fmax(rho_new, 1e-6), 1e6);
        if (rho_new > 5.0 * rho || rho_new < 0.2 * rho) {
          // rhs was accumulated with the old rho: rebuild it for the new one from z, y (decoded from the row
          // state with the OLD rho, re-encoded with the new one: y does not change when rho does)
          const double rho_old = rho;
          rho = rho_new;
          __syncthreads();
          factorize();
          const double rho_e2 = 1.0e3 * rho;
          for (int e = tid; e < n; e += nt) rhs[e] = sigma * x[e];
          __syncthreads();
          for (int r = tid; r < me; r += nt) {
            int i0, ax, d, kind;
            if (r < 18) { ax = r / 6; const int q = r % 6; kind = q < 3 ? 0 : 1; d = q % 3; i0 = kind == 0 ? 0 : N - 1; }
            else { const int rr = r - 18; i0 = rr / (3 * S); ax = (rr / S) % 3; d = rr % S; kind = 2; }
            const double zo = eqb[r], yo = 1.0e3 * rho_old * (wg[r] - zo);
            const double w = rho_e2 * zo - yo;
            wg[r] = zo + yo / rho_e2;
            if (kind == 0) atomicAdd(&rhs[ax * D + (D - 1 - d)], fallf(d, d) * w);
            else {
              for (int col = 0; col < D; ++col) { const int k = D - 1 - col; if (k >= d) atomicAdd(&rhs[i0 * NB + ax * D + col], fallf(k, d) * w); }
              if (kind == 2) atomicAdd(&rhs[(i0 + 1) * NB + ax * D + (D - 1 - d)], eqc[r] * w);
            }
          }
          for (int smp = tid; smp < N * R; smp += nt) {
            const int i = smp / R, j = smp % R;
            const double *bj = be + (size_t)j * 3 * D;
            const int64_t r0 = me + smp, NS = (int64_t)N * R;
            double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int q = 0; q < rows_per_sample; ++q) {
              const double *hq = hpl + ((int64_t)i * M + (q < M ? q : 0)) * 4;
              const int qq = q - M, w4 = qq % 4;
              const double hv = (q < M) ? hq[3] : (((w4 & 1) == 0) ? a.vmax * Tn[i] : a.amax * Tn[i] * Tn[i]);
              const double wo = wg[r0 + q * NS];
              const double zo = fmin(wo, hv), yo = rho_old * (wo - zo);
              const double w = rho * zo - yo;
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
        }
      }
    }
    __syncthreads();
  }
  if (it > a.p.max_iter) it = a.p.max_iter;
  __syncthreads();
  // ---- implicit time gradient of the optimal cost (layers.py:120-147) -------
  // The loss the reference back-propagates is the QP objective itself, J*(T) = min 1/2 z'Q(T)z.  For a
  // value function no KKT solve is needed: dJ*/dT_i = dL/dT_i at the optimum (envelope theorem),
  // L = J + (y/cobj)'(A x - u) in the variables of THIS kernel (normalised time), where only
  //   J_i = T_i^(1-2s) 1/2 x_i'Q1 x_i,  the start/end values b = state T^d,  the continuity coefficient
  //   -d!(T_i/T_i+1)^d,  and the box bounds vmax T_i, amax T_i^2
  // depend on T; the corridor rows do not.  (y/cobj: the duals belong to the cobj-scaled objective.)
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
      if (r < 18) { ax = r / 6; const int q = r % 6; kind = q < 3 ? 0 : 1; d = q % 3; i0 = kind == 0 ? 0 : N - 1; }
      else { const int rr = r - 18; i0 = rr / (3 * S); ax = (rr / S) % 3; d = rr % S; kind = 2; }
      if (d == 0) continue;
      const double yr = 1.0e3 * rho * (wg[r] - eqb[r]) * inv_c;
      if (kind != 2) {
        atomicAdd(&rhs[i0], -yr * (double)d * eqb[r] / Tn[i0]);
      } else {
        const double t = yr * (double)d * eqc[r] * x[(i0 + 1) * NB + ax * D + (D - 1 - d)];
        atomicAdd(&rhs[i0], t / Tn[i0]);
        atomicAdd(&rhs[i0 + 1], -t / Tn[i0 + 1]);
      }
    }
    for (int smp = tid; smp < N * R; smp += nt) {
      const int i = smp / R;
      const int64_t r0 = me + smp, NS = (int64_t)N * R;
      double sv = 0.0, sa = 0.0;
      for (int qq = 0; qq < 12; ++qq) {
        const bool acc = (qq % 4) & 1;
        const double hv = acc ? a.amax * Tn[i] * Tn[i] : a.vmax * Tn[i];
        const double yv = rho * fmax(wg[r0 + (M + qq) * NS] - hv, 0.0);
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
  const size_t mtot = (size_t)(3 * (6 + S * (N - 1))) + (size_t)N * R * (M + 12);
  return sizeof(double) * ((size_t)2 * N * NB * NB + NB * NB + 5 * n + (size_t)R * 3 * D + 2 * D * D + (size_t)N * 9 + N + 12 + hp + 3 * (size_t)(3 * (6 + S * (N - 1))) +
                           (size_t)N * R * 9 + (zy_in_lds ? mtot : 0));
}

}  // namespace anet
