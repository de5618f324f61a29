// allocnet_amd: gfx950 kernels + C ABI (include/allocnet_amd.h).  Built by allocnet_amd/build.py:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch, no Eigen.  There is no CPU fallback in this library: without a device every entry
// point fails with ANET_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <new>

#include "../../include/allocnet_amd.h"
#include "minco_core.h"
#include "qp_admm.h"

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
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) F.r[i] = fast_rcp(a.T[i * ld + b]);
  F.factorize(N, np);

  double etot = 0.0;
#pragma unroll 1
  for (int ax = 0; ax < 3; ++ax) {
    double P[NB + 1], hv[m], tv[m], X[NB + 1][m];
    const double *hp = a.head + (int64_t)(ax * c) * ld + b;
    const double *tp = a.tail + (int64_t)(ax * c) * ld + b;
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
    double *cp = a.coeffs ? a.coeffs + (int64_t)(ax * D) * ld + b : nullptr;
    etot += solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int piece, int col, double v) {
      if (cp) cp[(int64_t)(piece * 3 * D + col) * ld] = v;
    });
  }
  if (a.energy) a.energy[b] = etot;
}

// Trajectory<D>::getPos/Vel/Acc/Jer: one lane per trajectory, nq queries each.  The accumulation
// order is the reference's (ascending powers, tn *= t), trajectory.hpp:75-133.
struct EvalArgs {
  const double *coeffs, *T, *tq;
  double *out;
  int64_t B, ld;
  int N, nq, deriv;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_eval(EvalArgs a) {
  constexpr int D = 2 * S, DEG = D - 1;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  const int N = a.N, d = a.deriv;
  for (int q = 0; q < a.nq; ++q) {
    double t = a.tq[(int64_t)q * ld + b];
    // locatePieceIdx (trajectory.hpp:496-514)
    int idx = 0;
    double dur = 0.0;
    for (; idx < N; ++idx) {
      dur = a.T[(int64_t)idx * ld + b];
      if (!(t > dur)) break;
      t -= dur;
    }
    if (idx == N) {
      --idx;
      t += a.T[(int64_t)idx * ld + b];
    }
    const double *cm = a.coeffs + (int64_t)(idx * 3 * D) * ld + b;
    double acc[3] = {0.0, 0.0, 0.0};
    double tn = 1.0;
    for (int i = DEG - d; i >= 0; --i) {
      const int k = DEG - i;  // power of column i
      double f = 1.0;
      for (int e = 0; e < d; ++e) f *= (double)(k - e);
      const double w = f * tn;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) acc[ax] += w * cm[(int64_t)(ax * D + i) * ld];
      tn *= t;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) a.out[(int64_t)(q * 3 + ax) * ld + b] = acc[ax];
  }
}

// Trajectory<D>::getTrajCost (trajectory.hpp:354-427).
struct CostArgs {
  const double *coeffs, *T;
  double *cost;   // [B] or nullptr
  double *gradT;  // [N][ld] or nullptr: d cost / d T_i at fixed coefficients
  int64_t B, ld;
  int N;
  double m34;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_cost(CostArgs a) {
  constexpr int D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  double energy = 0.0;
  for (int i = 0; i < a.N; ++i) {
    const double t = a.T[(int64_t)i * ld + b];
    const double t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3;
    double Q[S][S], dQ[S][S];  // cost block and its derivative w.r.t. t
    if constexpr (S == 4) {
      const double t6 = t3 * t3, t7 = t4 * t3;
      Q[0][0] = 100800 * t7; Q[0][1] = 50400 * t6; Q[0][2] = 20160 * t5; Q[0][3] = 5040 * t4;
      Q[1][1] = 25920 * t5;  Q[1][2] = 10800 * t4; Q[1][3] = 2880 * t3;
      Q[2][2] = 4800 * t3;   Q[2][3] = a.m34 * t2;
      Q[3][3] = 576 * t;
      dQ[0][0] = 7 * 100800 * t6; dQ[0][1] = 6 * 50400 * t5; dQ[0][2] = 5 * 20160 * t4; dQ[0][3] = 4 * 5040 * t3;
      dQ[1][1] = 5 * 25920 * t4;  dQ[1][2] = 4 * 10800 * t3; dQ[1][3] = 3 * 2880 * t2;
      dQ[2][2] = 3 * 4800 * t2;   dQ[2][3] = 2 * a.m34 * t;
      dQ[3][3] = 576;
    } else if constexpr (S == 3) {
      Q[0][0] = 720 * t5; Q[0][1] = 360 * t4; Q[0][2] = 120 * t3;
      Q[1][1] = 192 * t3; Q[1][2] = 72 * t2;
      Q[2][2] = 36 * t;
      dQ[0][0] = 5 * 720 * t4; dQ[0][1] = 4 * 360 * t3; dQ[0][2] = 3 * 120 * t2;
      dQ[1][1] = 3 * 192 * t2; dQ[1][2] = 2 * 72 * t;
      dQ[2][2] = 36;
    } else {
      Q[0][0] = 12 * t3; Q[0][1] = 6 * t2;
      Q[1][1] = 4 * t;
      dQ[0][0] = 36 * t2; dQ[0][1] = 12 * t;
      dQ[1][1] = 4;
    }
#pragma unroll
    for (int j = 1; j < S; ++j)
#pragma unroll
      for (int k = 0; k < j; ++k) {
        Q[j][k] = Q[k][j];
        dQ[j][k] = dQ[k][j];
      }
    double gti = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      double z[S];
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] = a.coeffs[(int64_t)((i * 3 + ax) * D + j) * ld + b];
      double acc = 0.0, dacc = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        double r = 0.0, dr = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          r += Q[j][k] * z[k];
          dr += dQ[j][k] * z[k];
        }
        acc += z[j] * r;
        dacc += z[j] * dr;
      }
      energy += 0.5 * acc;
      gti += 0.5 * dacc;
    }
    if (a.gradT) a.gradT[(int64_t)i * ld + b] = gti;
  }
  if (a.cost) a.cost[b] = energy;
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
template <int S>
__global__ void __launch_bounds__(256) k_piece_grad(PieceGradArgs a) {
  constexpr int D = 2 * S;
  extern __shared__ double tab[];  // [res][4][D] basis rows in normalised time
  if (a.with_penalty) {
    for (int e = threadIdx.x; e < a.pp.res * 4 * D; e += 256) {
      const int j = e / (4 * D), d = (e / D) % 4, col = e % D, k = D - 1 - col;
      const double tau = (double)j / (double)a.pp.res;
      double v = 0.0;
      if (k >= d) {
        v = 1.0;
        for (int q = 0; q < d; ++q) v *= (double)(k - q);
        for (int q = 0; q < k - d; ++q) v *= tau;
      }
      tab[e] = v;
    }
    __syncthreads();
  }
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
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
  if (a.with_energy) {
    // d/dc of sum_{j,k>=S} c_j c_k f_j f_k T^(j+k-2S+1)/(j+k-2S+1) ;  d/dT = (p^(S)(T))^2
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
        ps = __builtin_fma(fj * tp[j - S], c[ax][D - 1 - j], ps);
        double acc = 0.0;
#pragma unroll
        for (int k = S; k < D; ++k) {
          double fk = 1.0;
#pragma unroll
          for (int e = 0; e < S; ++e) fk *= (double)(k - e);
          acc = __builtin_fma(2.0 * fj * fk / (double)(j + k - 2 * S + 1) * tp[j + k - 2 * S + 1],
                              c[ax][D - 1 - k], acc);
        }
        gC[ax][D - 1 - j] = acc;
      }
      gT = __builtin_fma(ps, ps, gT);
    }
  }
  if (a.with_penalty) {
    // Normalised time: with c~_k = c_k T^k the state rows at sample j depend on tau_j = j/res only,
    //   d^d p/dt^d (t_j) = T^-d sum_col c~[col] tab[j][d][col],  tab[j][d][col] = k!/(k-d)! tau_j^(k-d)
    // the table is built once per block in LDS and read with a wave-uniform index (broadcast).
    const Penalty pp = a.pp;
    const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
    const double step = Ti * inv_res;
    const double rT = 1.0 / Ti, rT2 = rT * rT, rT3 = rT2 * rT;
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
    // Polytope rows are held in registers, RC at a time, and the sample loop runs inside: re-reading
    // them from L2 for every sample (res x M x 32 B per lane) was the bottleneck of this kernel.
    constexpr int RC = 8;
    const int nchunk = a.hpolys ? (pp.M + RC - 1) / RC : 0;
    for (int ch = 0; ch < (nchunk > 0 ? nchunk : 1); ++ch) {
      double hr[RC][4];
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        const int rr = ch * RC + r;
        const bool ok = a.hpolys && rr < pp.M;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          hr[r][q] = ok ? a.hpolys[(int64_t)((i * pp.M + rr) * 4 + q) * ld + b] : 0.0;
      }
      const bool first = (ch == 0);  // box rows are evaluated with the first chunk
      for (int j = 0; j < pp.res; ++j) {
        const double *tb = tab + (size_t)j * 4 * D;
        double st[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            double acc = 0.0;
#pragma unroll
            for (int col = 0; col < D; ++col) acc = __builtin_fma(ct[ax][col], tb[d * D + col], acc);
            st[d][ax] = acc * (d == 0 ? 1.0 : d == 1 ? rT : d == 2 ? rT2 : rT3);
          }
        double cost = 0.0, g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // d cost / d (p,v,a)
        bool active = false;
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const double viol =
              __builtin_fma(hr[r][0], st[0][0], __builtin_fma(hr[r][1], st[0][1], hr[r][2] * st[0][2])) - hr[r][3];
          if (__any(viol > 0.0)) {  // wave-uniform: inside the corridor nothing else is computed
            double f, df;
            smoothed_l1(pp.mu, inv_mu, viol, f, df);
            cost = __builtin_fma(pp.wc, f, cost);
            df *= pp.wc;
            g[0][0] = __builtin_fma(df, hr[r][0], g[0][0]);
            g[0][1] = __builtin_fma(df, hr[r][1], g[0][1]);
            g[0][2] = __builtin_fma(df, hr[r][2], g[0][2]);
            active = true;
          }
        }
        if (first) {
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            const double av = fabs(st[1][ax]) - pp.vmax, aa_ = fabs(st[2][ax]) - pp.amax;
            if (__any(av > 0.0)) {  // only one of +v, -v can be violated
              double f, df;
              smoothed_l1(pp.mu, inv_mu, av, f, df);
              cost = __builtin_fma(pp.wv, f, cost);
              g[1][ax] = __builtin_fma(pp.wv * (st[1][ax] < 0.0 ? -1.0 : 1.0), df, g[1][ax]);
              active = true;
            }
            if (__any(aa_ > 0.0)) {
              double f, df;
              smoothed_l1(pp.mu, inv_mu, aa_, f, df);
              cost = __builtin_fma(pp.wa, f, cost);
              g[2][ax] = __builtin_fma(pp.wa * (st[2][ax] < 0.0 ? -1.0 : 1.0), df, g[2][ax]);
              active = true;
            }
          }
        }
        if (__any(active)) {
          pc = __builtin_fma(step, cost, pc);
          double dt = 0.0;  // d cost / d t = g_p.v + g_v.a + g_a.j
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) dt = __builtin_fma(g[d][ax], st[d + 1][ax], dt);
          gT += cost * inv_res + step * dt * ((double)j * inv_res);
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            const double g0 = step * g[0][ax], g1 = step * g[1][ax] * rT, g2 = step * g[2][ax] * rT2;
#pragma unroll
            for (int col = 0; col < D; ++col) {
              double acc = g0 * tb[col];
              acc = __builtin_fma(g1, tb[D + col], acc);
              acc = __builtin_fma(g2, tb[2 * D + col], acc);
              gN[ax][col] += acc;
            }
          }
        }
      }
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
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int col = 0; col < D; ++col) a.gdC[(int64_t)((i * 3 + ax) * D + col) * ld + b] = gC[ax][col];
  a.gdT[(int64_t)i * ld + b] = gT;
  if (a.pcost) a.pcost[(int64_t)i * ld + b] = pc;
}

struct PropArgs {
  const double *T, *coeffs, *gdC, *gdT;
  double *gradP, *gradT;
  // optional total cost: cost = energy_in + rho sum T + sum_i pcost_i ; gradT += rho
  const double *energy_in, *pcost;
  double *cost;
  double rho;
  int64_t B, ld;
  int N, c;
};

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
  for (int ax = 0; ax < 3; ++ax) {
    double rr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(F.r[i]) : 0.0;
    // ---- node states from the coefficients: x_k[j] = j! c_j(piece k); last node by evaluation
    double XS[NB + 1][S], GX[NB + 1][S], XA[NB + 1][m];
#pragma unroll
    for (int k = 0; k <= NB; ++k)
#pragma unroll
      for (int j = 0; j < S; ++j) {
        XS[k][j] = 0.0;
        GX[k][j] = 0.0;
      }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (k < N) {
        double fact = 1.0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          if (j > 0) fact *= (double)j;
          XS[k][j] = fact * a.coeffs[(int64_t)((k * 3 + ax) * D + (D - 1 - j)) * ld + b];
        }
        if (k == N - 1) {
          double cl[D], tp[D];
          tp[0] = 1.0;
#pragma unroll
          for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Tlast;
#pragma unroll
          for (int col = 0; col < D; ++col) cl[col] = a.coeffs[(int64_t)((k * 3 + ax) * D + col) * ld + b];
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
            XS[k + 1][j] = acc;
          }
        }
      }
    // ---- g_x = Phi' gdC and the direct dPhi/dT term
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < N) {
        Pw<S> p(rr[i]);
        double gc[D];
#pragma unroll
        for (int col = 0; col < D; ++col) gc[col] = a.gdC[(int64_t)((i * 3 + ax) * D + col) * ld + b];
        // low powers k < S: c_k = x_i[k]/k!
        double fact = 1.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          if (k > 0) fact *= (double)k;
          GX[i][k] = __builtin_fma(gc[D - 1 - k], 1.0 / fact, GX[i][k]);
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
          const double xb = (bb < S) ? XS[i][dg] : XS[i + 1][dg];
          if (bb < S)
            GX[i][dg] = __builtin_fma(u, sc, GX[i][dg]);
          else
            GX[i + 1][dg] = __builtin_fma(u, sc, GX[i + 1][dg]);
          dsum = __builtin_fma(xb * sc, qd, dsum);
        }
        gT[i] = __builtin_fma(-p[1], dsum, gT[i]);
      }
    // ---- adjoint solve K lam = g_x|free (pinned rows 0)
    sweep_forward<S, NB>(F, N, np, rr, XA, [&](int k, double (&y)[m]) {
#pragma unroll
      for (int l = 0; l < m; ++l) y[l] = ((k == 0 || k == N) && l < np) ? 0.0 : GX[k][1 + l];
    });
#pragma unroll
    for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(rr[i]) : 0.0;
    sweep_backward<S, NB>(F, N, np, rr, XA, [&](int k, const Pw<S> &p) {
      // (W_k lam^)[position row of node k]; the row of node k+1 is its negative
      double wl = 0.0;
#pragma unroll
      for (int l = 0; l < m; ++l) {
        wl = __builtin_fma(Tab<S>::M[0][1 + l] * p[2 * S - 2 - l], XA[k][l], wl);
        wl = __builtin_fma(Tab<S>::M[0][S + 1 + l] * p[2 * S - 2 - l], XA[k + 1][l], wl);
      }
      GX[k][0] -= wl;
      GX[k + 1][0] += wl;
      // - lam^' (dW/dT) x^ = sum_ab lam_a M_ab e_ab r^(e_ab+1) x_b,  e_ab = 2S-1-deg a-deg b
      double xs[2 * S];
#pragma unroll
      for (int bb = 0; bb < 2 * S; ++bb)
        xs[bb] = ((bb < S) ? XS[k][bb % S] : XS[k + 1][bb % S]) * p[S - bb % S];
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
    });
#pragma unroll
    for (int k = 1; k < NB; ++k)
      if (k < N) a.gradP[(int64_t)((k - 1) * 3 + ax) * ld + b] = GX[k][0];
  }
  double csum = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      a.gradT[i * ld + b] = gT[i] + a.rho;
      if (a.pcost) csum += a.pcost[i * ld + b];
    }
  if (a.cost) a.cost[b] = (a.energy_in ? a.energy_in[b] : 0.0) + a.rho * tsum + csum;
}


// ------------------------------------------------------------------------------------------
// batched L-BFGS (lbfgs.hpp:276-384, 434-717) as a per-trajectory state machine
// ------------------------------------------------------------------------------------------
struct LbfgsP {
  int mem_size;
  double g_epsilon;
  int past;
  double delta;
  int max_iterations, max_linesearch;
  double min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
};
enum { DS_FX = 0, DS_STEP, DS_FINIT, DS_DGTEST, DS_DSTEST, DS_MU, DS_NU, DS_COUNT_ };
enum { IS_DONE = 0, IS_RET, IS_K, IS_END, IS_BOUND, IS_COUNT, IS_BRACKT, IS_TOUCHED, IS_EVALS, IS_PHASE, IS_COUNT_ };
enum {  // lbfgs.hpp:135-184
  LB_CONVERGENCE = 0, LB_STOP = 1, LB_CANCELED = 2,
  LBERR_INVALID_FUNCVAL = -1012, LBERR_MINIMUMSTEP = -1011, LBERR_MAXIMUMSTEP = -1010,
  LBERR_MAXIMUMLINESEARCH = -1009, LBERR_MAXIMUMITERATION = -1008, LBERR_WIDTHTOOSMALL = -1007,
  LBERR_INVALIDPARAMETERS = -1006, LBERR_INCREASEGRADIENT = -1005
};

struct LbfgsArgs {
  int n;
  int64_t B, ld;
  double *x, *g, *xp, *gp, *d, *lm_s, *lm_y, *lm_ys, *lm_alpha, *pf, *ds;
  const double *feval;
  int *is;
  LbfgsP p;
  int *n_active;
  int64_t vs, ps;  // internal vectors (xp, gp, d, lm_s, lm_y): element i of problem b at [i*vs + b*ps]
};

// One lane per problem.  Every launch consumes ONE objective evaluation (f = feval[b], gradient in
// g, both taken at the point currently in x) and leaves in x the next point to evaluate.  The
// control flow per problem is lbfgs_optimize's: phase 0 = the initial evaluation, phase 1 = inside
// line_search_lewisoverton.  Finished problems are untouched (x, g hold the result).
__global__ void __launch_bounds__(64) k_lbfgs_update(LbfgsArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  int *is = a.is + b;
  if (is[IS_DONE * ld]) return;
  double *ds = a.ds + b;
  const int n = a.n, m = a.p.mem_size;
  const LbfgsP &P = a.p;
  double *x = a.x + b, *g = a.g + b, *xp = a.xp + b, *gp = a.gp + b, *d = a.d + b;
  const double f = a.feval[b];
  is[IS_EVALS * ld] += 1;
  double fx = ds[DS_FX * ld];
  double step = ds[DS_STEP * ld];
  int k = is[IS_K * ld];
  bool start_ls = false;
  int finish = 0x7fffffff;  // sentinel: keep running

  auto conv_test = [&]() {
    double gn = 0.0, xn = 0.0;
    for (int i = 0; i < n; ++i) {
      gn = fmax(gn, fabs(g[i * ld]));
      xn = fmax(xn, fabs(x[i * ld]));
    }
    return gn / fmax(1.0, xn) < P.g_epsilon;
  };

  if (is[IS_PHASE * ld] == 0) {
    fx = f;
    a.pf[b] = fx;
    double dd = 0.0;
    for (int i = 0; i < n; ++i) {
      const double gi = g[i * ld];
      d[i * ld] = -gi;
      dd = __builtin_fma(gi, gi, dd);
    }
    if (conv_test()) {
      finish = LB_CONVERGENCE;
    } else {
      step = 1.0 / sqrt(dd);
      k = 1;
      is[IS_END * ld] = 0;
      is[IS_BOUND * ld] = 0;
      is[IS_PHASE * ld] = 1;
      start_ls = true;
    }
  } else {
    // ---- one trial of line_search_lewisoverton (lbfgs.hpp:307-383)
    const double finit = ds[DS_FINIT * ld], dgtest = ds[DS_DGTEST * ld], dstest = ds[DS_DSTEST * ld];
    double mu = ds[DS_MU * ld], nu = ds[DS_NU * ld];
    int count = is[IS_COUNT * ld] + 1, brackt = is[IS_BRACKT * ld], touched = is[IS_TOUCHED * ld];
    bool success = false;
    int err = 0;
    if (isinf(f) || isnan(f)) {
      err = LBERR_INVALID_FUNCVAL;
    } else {
      if (f > finit + step * dgtest) {
        nu = step;
        brackt = 1;
      } else {
        double dg = 0.0;
        for (int i = 0; i < n; ++i) dg = __builtin_fma(g[i * ld], d[i * ld], dg);
        if (dg < dstest)
          mu = step;
        else
          success = true;
      }
      if (!success) {
        if (P.max_linesearch <= count) {
          err = LBERR_MAXIMUMLINESEARCH;
        } else if (brackt && (nu - mu) < P.machine_prec * nu) {
          err = LBERR_WIDTHTOOSMALL;
        } else {
          step = brackt ? 0.5 * (mu + nu) : step * 2.0;
          if (step < P.min_step) {
            err = LBERR_MINIMUMSTEP;
          } else if (step > P.max_step) {
            if (touched) {
              err = LBERR_MAXIMUMSTEP;
            } else {
              touched = 1;
              step = P.max_step;
            }
          }
        }
      }
    }
    if (err) {
      // revert to the previous point; the reported f stays the last trial's (lbfgs.hpp:570-577,713)
      for (int i = 0; i < n; ++i) {
        x[i * ld] = xp[i * ld];
        g[i * ld] = gp[i * ld];
      }
      fx = f;
      finish = err;
    } else if (!success) {
      for (int i = 0; i < n; ++i) x[i * ld] = __builtin_fma(step, d[i * ld], xp[i * ld]);
      ds[DS_MU * ld] = mu;
      ds[DS_NU * ld] = nu;
      is[IS_COUNT * ld] = count;
      is[IS_BRACKT * ld] = brackt;
      is[IS_TOUCHED * ld] = touched;
    } else {
      // ---- accepted step (lbfgs.hpp:579-709)
      fx = f;
      if (conv_test()) {
        finish = LB_CONVERGENCE;
      } else {
        if (0 < P.past) {
          if (P.past <= k) {
            const double rate = fabs(a.pf[(int64_t)(k % P.past) * ld + b] - fx) / fmax(1.0, fabs(fx));
            if (rate < P.delta) finish = LB_STOP;
          }
          if (finish == 0x7fffffff) a.pf[(int64_t)(k % P.past) * ld + b] = fx;
        }
        if (finish == 0x7fffffff && P.max_iterations != 0 && P.max_iterations <= k) finish = LBERR_MAXIMUMITERATION;
        if (finish == 0x7fffffff) {
          ++k;
          int end = is[IS_END * ld], bound = is[IS_BOUND * ld];
          double *se = a.lm_s + (int64_t)end * n * ld + b, *ye = a.lm_y + (int64_t)end * n * ld + b;
          double ys = 0.0, yy = 0.0, ss = 0.0, gpgp = 0.0;
          for (int i = 0; i < n; ++i) {
            const double si = x[i * ld] - xp[i * ld], yi = g[i * ld] - gp[i * ld], gpi = gp[i * ld];
            se[i * ld] = si;
            ye[i * ld] = yi;
            ys = __builtin_fma(yi, si, ys);
            yy = __builtin_fma(yi, yi, yy);
            ss = __builtin_fma(si, si, ss);
            gpgp = __builtin_fma(gpi, gpi, gpgp);
            d[i * ld] = -g[i * ld];
          }
          a.lm_ys[(int64_t)end * ld + b] = ys;
          const double cau = ss * sqrt(gpgp) * P.cautious_factor;
          if (ys > cau) {
            ++bound;
            bound = m < bound ? m : bound;
            end = (end + 1) % m;
            int j = end;
            for (int it = 0; it < bound; ++it) {
              j = (j + m - 1) % m;
              const double *sj = a.lm_s + (int64_t)j * n * ld + b, *yj = a.lm_y + (int64_t)j * n * ld + b;
              double sd = 0.0;
              for (int i = 0; i < n; ++i) sd = __builtin_fma(sj[i * ld], d[i * ld], sd);
              const double al = sd / a.lm_ys[(int64_t)j * ld + b];
              a.lm_alpha[(int64_t)j * ld + b] = al;
              for (int i = 0; i < n; ++i) d[i * ld] = __builtin_fma(-al, yj[i * ld], d[i * ld]);
            }
            const double sc = ys / yy;
            for (int i = 0; i < n; ++i) d[i * ld] *= sc;
            for (int it = 0; it < bound; ++it) {
              const double *sj = a.lm_s + (int64_t)j * n * ld + b, *yj = a.lm_y + (int64_t)j * n * ld + b;
              double yd = 0.0;
              for (int i = 0; i < n; ++i) yd = __builtin_fma(yj[i * ld], d[i * ld], yd);
              const double beta = yd / a.lm_ys[(int64_t)j * ld + b];
              const double cf = a.lm_alpha[(int64_t)j * ld + b] - beta;
              for (int i = 0; i < n; ++i) d[i * ld] = __builtin_fma(cf, sj[i * ld], d[i * ld]);
              j = (j + 1) % m;
            }
          }
          is[IS_END * ld] = end;
          is[IS_BOUND * ld] = bound;
          step = 1.0;
          start_ls = true;
        }
      }
    }
  }
  if (start_ls) {
    // ---- entry of line_search_lewisoverton (lbfgs.hpp:287-305) for the new direction
    double dginit = 0.0;
    for (int i = 0; i < n; ++i) {
      const double xi = x[i * ld], gi = g[i * ld];
      xp[i * ld] = xi;
      gp[i * ld] = gi;
      dginit = __builtin_fma(gi, d[i * ld], dginit);
    }
    if (!(step > 0.0)) {
      finish = LBERR_INVALIDPARAMETERS;
    } else if (0.0 < dginit) {
      finish = LBERR_INCREASEGRADIENT;
    } else {
      ds[DS_FINIT * ld] = fx;
      ds[DS_DGTEST * ld] = P.f_dec_coeff * dginit;
      ds[DS_DSTEST * ld] = P.s_curv_coeff * dginit;
      ds[DS_MU * ld] = 0.0;
      ds[DS_NU * ld] = P.max_step;
      is[IS_COUNT * ld] = 0;
      is[IS_BRACKT * ld] = 0;
      is[IS_TOUCHED * ld] = 0;
      for (int i = 0; i < n; ++i) x[i * ld] = __builtin_fma(step, d[i * ld], xp[i * ld]);
    }
  }
  ds[DS_FX * ld] = fx;
  ds[DS_STEP * ld] = step;
  is[IS_K * ld] = k;
  if (finish != 0x7fffffff) {
    is[IS_DONE * ld] = 1;
    is[IS_RET * ld] = finish;
  } else if (a.n_active) {
    atomicAdd(a.n_active, 1);
  }
}


__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// Same state machine, ONE WAVE per problem: the n variables are spread over the 64 lanes, every dot
// product / norm is a wavefront shuffle reduction, scalars are computed redundantly by all lanes
// (no divergence: a wave holds one problem).  Used for small batches, where one lane per problem
// leaves the chip idle and serialises ~16 n-long dependent loops per accepted step.
__global__ void __launch_bounds__(64) k_lbfgs_update_wave(LbfgsArgs a) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t ld = a.ld;
  int *is = a.is + b;
  if (is[IS_DONE * ld]) return;
  double *ds = a.ds + b;
  const int n = a.n, m = a.p.mem_size;
  const LbfgsP &P = a.p;
  const int64_t vs = a.vs, ps = a.ps;
  double *x = a.x + b, *g = a.g + b;                       // batch-minor (shared with the objective)
  double *xp = a.xp + b * ps, *gp = a.gp + b * ps, *d = a.d + b * ps;
  const double f = a.feval[b];
  double fx = ds[DS_FX * ld];
  double step = ds[DS_STEP * ld];
  int k = is[IS_K * ld];
  int evals = is[IS_EVALS * ld] + 1;
  int end = is[IS_END * ld], bound = is[IS_BOUND * ld], phase = is[IS_PHASE * ld];
  int count = is[IS_COUNT * ld], brackt = is[IS_BRACKT * ld], touched = is[IS_TOUCHED * ld];
  double finit = ds[DS_FINIT * ld], dgtest = ds[DS_DGTEST * ld], dstest = ds[DS_DSTEST * ld];
  double mu = ds[DS_MU * ld], nu = ds[DS_NU * ld];
  bool start_ls = false;
  int finish = 0x7fffffff;

  auto conv_test = [&]() {
    double gn = 0.0, xn = 0.0;
    for (int i = lane; i < n; i += 64) {
      gn = fmax(gn, fabs(g[i * ld]));
      xn = fmax(xn, fabs(x[i * ld]));
    }
    gn = wave_max(gn);
    xn = wave_max(xn);
    return gn / fmax(1.0, xn) < P.g_epsilon;
  };

  if (phase == 0) {
    fx = f;
    if (lane == 0) a.pf[b] = fx;
    double dd = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double gi = g[i * ld];
      d[i * vs] = -gi;
      dd = __builtin_fma(gi, gi, dd);
    }
    dd = wave_sum(dd);
    if (conv_test()) {
      finish = LB_CONVERGENCE;
    } else {
      step = 1.0 / sqrt(dd);
      k = 1;
      end = 0;
      bound = 0;
      phase = 1;
      start_ls = true;
    }
  } else {
    ++count;
    bool success = false;
    int err = 0;
    if (isinf(f) || isnan(f)) {
      err = LBERR_INVALID_FUNCVAL;
    } else {
      if (f > finit + step * dgtest) {
        nu = step;
        brackt = 1;
      } else {
        double dg = 0.0;
        for (int i = lane; i < n; i += 64) dg = __builtin_fma(g[i * ld], d[i * vs], dg);
        dg = wave_sum(dg);
        if (dg < dstest)
          mu = step;
        else
          success = true;
      }
      if (!success) {
        if (P.max_linesearch <= count) {
          err = LBERR_MAXIMUMLINESEARCH;
        } else if (brackt && (nu - mu) < P.machine_prec * nu) {
          err = LBERR_WIDTHTOOSMALL;
        } else {
          step = brackt ? 0.5 * (mu + nu) : step * 2.0;
          if (step < P.min_step) {
            err = LBERR_MINIMUMSTEP;
          } else if (step > P.max_step) {
            if (touched) {
              err = LBERR_MAXIMUMSTEP;
            } else {
              touched = 1;
              step = P.max_step;
            }
          }
        }
      }
    }
    if (err) {
      for (int i = lane; i < n; i += 64) {
        x[i * ld] = xp[i * vs];
        g[i * ld] = gp[i * vs];
      }
      fx = f;
      finish = err;
    } else if (!success) {
      for (int i = lane; i < n; i += 64) x[i * ld] = __builtin_fma(step, d[i * vs], xp[i * vs]);
    } else {
      fx = f;
      if (conv_test()) {
        finish = LB_CONVERGENCE;
      } else {
        if (0 < P.past) {
          if (P.past <= k) {
            const double rate = fabs(a.pf[(int64_t)(k % P.past) * ld + b] - fx) / fmax(1.0, fabs(fx));
            if (rate < P.delta) finish = LB_STOP;
          }
          if (finish == 0x7fffffff && lane == 0) a.pf[(int64_t)(k % P.past) * ld + b] = fx;
        }
        if (finish == 0x7fffffff && P.max_iterations != 0 && P.max_iterations <= k) finish = LBERR_MAXIMUMITERATION;
        if (finish == 0x7fffffff) {
          ++k;
          double *lms = a.lm_s + b * ps * m, *lmy = a.lm_y + b * ps * m;  // [j][i] at (j*n_stride + i*vs)
          const int64_t js = (vs == 1) ? ps : (int64_t)n * vs;             // stride between history slots
          double *se = lms + (int64_t)end * js, *ye = lmy + (int64_t)end * js;
          // this lane's variable(s) live in registers for the whole two-loop recursion (n <= 128)
          double dv[2] = {0.0, 0.0};
          double ys = 0.0, yy = 0.0, ss = 0.0, gpgp = 0.0;
          int q = 0;
          for (int i = lane; i < n; i += 64, ++q) {
            const double gi = g[i * ld], gpi = gp[i * vs];
            const double si = x[i * ld] - xp[i * vs], yi = gi - gpi;
            se[i * vs] = si;
            ye[i * vs] = yi;
            ys = __builtin_fma(yi, si, ys);
            yy = __builtin_fma(yi, yi, yy);
            ss = __builtin_fma(si, si, ss);
            gpgp = __builtin_fma(gpi, gpi, gpgp);
            dv[q] = -gi;
          }
          ys = wave_sum(ys); yy = wave_sum(yy); ss = wave_sum(ss); gpgp = wave_sum(gpgp);
          if (lane == 0) a.lm_ys[(int64_t)end * ld + b] = ys;
          const double cau = ss * sqrt(gpgp) * P.cautious_factor;
          if (ys > cau) {
            ++bound;
            bound = m < bound ? m : bound;
            const int newest = end;
            end = (end + 1) % m;
            int j = end;
            double alpha = 0.0;  // lane `it` keeps alpha of the it-th visited slot (mem_size <= 64, host-checked)
            for (int it = 0; it < bound; ++it) {
              j = (j + m - 1) % m;
              const double *sj = lms + (int64_t)j * js, *yj = lmy + (int64_t)j * js;
              double sd = 0.0, yv[2] = {0.0, 0.0};
              q = 0;
              for (int i = lane; i < n; i += 64, ++q) {
                sd = __builtin_fma(sj[i * vs], dv[q], sd);
                yv[q] = yj[i * vs];
              }
              sd = wave_sum(sd);
              const double ysj = (j == newest) ? ys : a.lm_ys[(int64_t)j * ld + b];
              const double al = sd / ysj;
              alpha = (lane == it) ? al : alpha;
              dv[0] = __builtin_fma(-al, yv[0], dv[0]);
              dv[1] = __builtin_fma(-al, yv[1], dv[1]);
            }
            const double sc = ys / yy;
            dv[0] *= sc;
            dv[1] *= sc;
            for (int it = 0; it < bound; ++it) {
              const double *sj = lms + (int64_t)j * js, *yj = lmy + (int64_t)j * js;
              double yd = 0.0, sv[2] = {0.0, 0.0};
              q = 0;
              for (int i = lane; i < n; i += 64, ++q) {
                yd = __builtin_fma(yj[i * vs], dv[q], yd);
                sv[q] = sj[i * vs];
              }
              yd = wave_sum(yd);
              const double ysj = (j == newest) ? ys : a.lm_ys[(int64_t)j * ld + b];
              const double cf = __shfl(alpha, bound - 1 - it) - yd / ysj;
              dv[0] = __builtin_fma(cf, sv[0], dv[0]);
              dv[1] = __builtin_fma(cf, sv[1], dv[1]);
              j = (j + 1) % m;
            }
          }
          q = 0;
          for (int i = lane; i < n; i += 64, ++q) d[i * vs] = dv[q];
          step = 1.0;
          start_ls = true;
        }
      }
    }
  }
  if (start_ls) {
    double dginit = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double xi = x[i * ld], gi = g[i * ld];
      xp[i * vs] = xi;
      gp[i * vs] = gi;
      dginit = __builtin_fma(gi, d[i * vs], dginit);
    }
    dginit = wave_sum(dginit);
    if (!(step > 0.0)) {
      finish = LBERR_INVALIDPARAMETERS;
    } else if (0.0 < dginit) {
      finish = LBERR_INCREASEGRADIENT;
    } else {
      finit = fx;
      dgtest = P.f_dec_coeff * dginit;
      dstest = P.s_curv_coeff * dginit;
      mu = 0.0;
      nu = P.max_step;
      count = 0;
      brackt = 0;
      touched = 0;
      for (int i = lane; i < n; i += 64) x[i * ld] = __builtin_fma(step, d[i * vs], xp[i * vs]);
    }
  }
  if (lane == 0) {
    ds[DS_FX * ld] = fx; ds[DS_STEP * ld] = step; ds[DS_FINIT * ld] = finit; ds[DS_DGTEST * ld] = dgtest;
    ds[DS_DSTEST * ld] = dstest; ds[DS_MU * ld] = mu; ds[DS_NU * ld] = nu;
    is[IS_K * ld] = k; is[IS_END * ld] = end; is[IS_BOUND * ld] = bound; is[IS_PHASE * ld] = phase;
    is[IS_COUNT * ld] = count; is[IS_BRACKT * ld] = brackt; is[IS_TOUCHED * ld] = touched;
    is[IS_EVALS * ld] = evals;
    if (finish != 0x7fffffff) {
      is[IS_DONE * ld] = 1;
      is[IS_RET * ld] = finish;
    } else if (a.n_active) {
      atomicAdd(a.n_active, 1);
    }
  }
}

// firi::costMVIE (gcopter/firi.hpp:86-157): x = [p, rtd, cde], A is M x 3 column-major per problem
// (field k*M + r), the reference's optData packing (firi.hpp:186-200).
struct MvieArgs {
  const double *A, *x;
  double *f, *g;
  const int *done;
  int64_t B, ld;
  int M;
  double eps, wt;
};
__global__ void __launch_bounds__(64) k_mvie_eval(MvieArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  if (a.done && a.done[b]) return;
  const int64_t ld = a.ld;
  const double *x = a.x + b;
  double p[3], rtd[3], cde[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    p[q] = x[q * ld];
    rtd[q] = x[(3 + q) * ld];
    cde[q] = x[(6 + q) * ld];
  }
  const double L00 = rtd[0] * rtd[0] + 2.220446049250313e-16, L11 = rtd[1] * rtd[1] + 2.220446049250313e-16,
               L22 = rtd[2] * rtd[2] + 2.220446049250313e-16;
  const double L10 = cde[0], L21 = cde[1], L20 = cde[2];
  double cost = 0.0, gdp[3] = {0, 0, 0}, gdr[3] = {0, 0, 0}, gdc[3] = {0, 0, 0};
  const double inv_mu = 1.0 / a.eps;
  for (int r = 0; r < a.M; ++r) {
    const double a0 = a.A[(int64_t)r * ld + b], a1 = a.A[(int64_t)(a.M + r) * ld + b],
                 a2 = a.A[(int64_t)(2 * a.M + r) * ld + b];
    const double al0 = a0 * L00 + a1 * L10 + a2 * L20, al1 = a1 * L11 + a2 * L21, al2 = a2 * L22;
    const double nrm = sqrt(al0 * al0 + al1 * al1 + al2 * al2);
    const double viol = nrm + (a0 * p[0] + a1 * p[1] + a2 * p[2]) - 1.0;
    if (viol >= 0.0) {
      double c, dc;
      smoothed_l1(a.eps, inv_mu, viol, c, dc);
      const double inv = 1.0 / nrm;
      const double adj0 = al0 * inv, adj1 = al1 * inv, adj2 = al2 * inv;
      const double v0 = dc * a0, v1 = dc * a1, v2 = dc * a2;
      cost += c;
      gdp[0] += v0; gdp[1] += v1; gdp[2] += v2;
      gdr[0] += adj0 * v0; gdr[1] += adj1 * v1; gdr[2] += adj2 * v2;
      gdc[0] += adj0 * v1;
      gdc[1] += adj1 * v2;
      gdc[2] += adj0 * v2;
    }
  }
  cost *= a.wt;
  cost -= log(L00) + log(L11) + log(L22);
  const double Ld[3] = {L00, L11, L22};
  double *g = a.g + b;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    g[q * ld] = gdp[q] * a.wt;
    g[(3 + q) * ld] = (gdr[q] * a.wt - 1.0 / Ld[q]) * 2.0 * rtd[q];
    g[(6 + q) * ld] = gdc[q] * a.wt;
  }
  a.f[b] = cost;
}

// GCOPTER's smooth bijection R -> (0, inf) for the durations (upstream gcopter.hpp forwardT /
// backwardT; not part of the reference tree): T = tau>0 ? (tau/2+1)tau+1 : 1/((tau/2-1)tau+1).
__device__ __forceinline__ double forward_T(double tau) {
  return tau > 0.0 ? (0.5 * tau + 1.0) * tau + 1.0 : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0);
}
__device__ __forceinline__ double dforward_T(double tau) {
  if (tau > 0.0) return tau + 1.0;
  const double den = (0.5 * tau - 1.0) * tau + 1.0;
  return (1.0 - tau) / (den * den);
}
__device__ __forceinline__ double backward_T(double T) {
  return T > 1.0 ? sqrt(2.0 * T - 1.0) - 1.0 : 1.0 - sqrt(2.0 / T - 1.0);
}
struct MapArgs {
  double *x, *g;             // optimisation variables / gradient [n][ld]
  double *wps, *T;           // trajectory parameters
  const double *gradP, *gradT;
  int64_t B, ld;
  int nw, nt;                // optimised waypoint coordinates (0 or 3(N-1)), optimised durations (0 or N)
  int mode;                  // 0: params -> x (init), 1: x -> params, 2: (gradP, gradT) -> g
};
__global__ void __launch_bounds__(256) k_minco_map(MapArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  const int v = blockIdx.y;  // variable index: [0, nw) waypoint coordinates, [nw, nw+nt) durations
  if (v < a.nw) {
    const int64_t i = (int64_t)v * ld + b;
    if (a.mode == 0) a.x[i] = a.wps[i];
    else if (a.mode == 1) a.wps[i] = a.x[i];
    else a.g[i] = a.gradP[i];
  } else {
    const int64_t xi = (int64_t)v * ld + b, ti = (int64_t)(v - a.nw) * ld + b;
    if (a.mode == 0) a.x[xi] = backward_T(a.T[ti]);
    else if (a.mode == 1) a.T[ti] = forward_T(a.x[xi]);
    else a.g[xi] = a.gradT[ti] * dforward_T(a.x[xi]);
  }
}


// ------------------------------------------------------------------------------------------
// QP assembly, dense, in the reference's own shapes (qp_solver.hpp:119-296, min_traj_opt.py:377-613)
// ------------------------------------------------------------------------------------------
struct QpArgs {
  const double *state, *T, *hpolys;
  const int *rows;
  double *Q, *A, *b, *G, *h;
  int64_t B, n, me, mg;
  int N, res, M, float_time, row_order;
  double vmax, amax, m34;
};

// Row d (0 = p, 1 = v, 2 = a, 3 = j) of the monomial basis at t, column `col` (highest power first),
// with the reference's multiplication order for the powers (get_t_state, qp_solver.hpp:90-116 /
// min_traj_opt.py:300-336): t_2 = t*t, t_3 = t*t_2, t_4 = t_2*t_2, t_5 = t_2*t_3, t_6 = t_3*t_3,
// t_7 = t_4*t_3, each entry = integer coefficient * power.  F = float reproduces the C++ planner.
template <int S, class F>
__device__ __forceinline__ double basis_entry(F t, int d, int col) {
  constexpr int D = 2 * S;
  const int k = D - 1 - col;  // power of this column
  if (k < d) return 0.0;
  const F t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3, t6 = t3 * t3, t7 = t4 * t3;
  const F pw[8] = {(F)1, t, t2, t3, t4, t5, t6, t7};
  int coef = 1;
  for (int e = 0; e < d; ++e) coef *= (k - e);
  const int e = k - d;
  if (e == 0) return (double)coef;       // constant entries are written as literals in the reference
  if (coef == 1) return (double)pw[e];
  return (double)((F)coef * pw[e]);
}

// cost block entry (j,k) of piece time t (qp_solver.hpp:186-236 / min_traj_opt.py:466-508)
template <int S, class F>
__device__ __forceinline__ double cost_entry(F t, int j, int k, double m34) {
  if (j > k) { const int q = j; j = k; k = q; }
  const F t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3;
  if (S == 4) {
    const F t6 = t3 * t3, t7 = t4 * t3;
    const F m[4][4] = {{(F)100800 * t7, (F)50400 * t6, (F)20160 * t5, (F)5040 * t4},
                       {0, (F)25920 * t5, (F)10800 * t4, (F)2880 * t3},
                       {0, 0, (F)4800 * t3, (F)m34 * t2},
                       {0, 0, 0, (F)576 * t}};
    return (double)m[j][k];
  } else {
    const F m[3][3] = {{(F)720 * t5, (F)360 * t4, (F)120 * t3}, {0, (F)192 * t3, (F)72 * t2}, {0, 0, (F)36 * t}};
    return (double)m[j][k];
  }
}

template <int S, class F>
__device__ __forceinline__ F seg_time(const QpArgs &a, int64_t b, int i) {
  return (F)a.T[b * a.N + i];
}

// Q and [A | b]: one thread per element.
template <int S, class F>
__global__ void __launch_bounds__(256) k_qp_eq_obj(QpArgs a) {
  constexpr int D = 2 * S;
  const int64_t b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = a.n, me = a.me, nQ = n * n, nA = me * n;
  const int N = a.N;
  if (e < nQ) {
    const int64_t r = e / n, c = e % n;
    double v = 0.0;
    if (r / D == c / D) {  // same (piece, axis) block
      const int jr = (int)(r % D), jc = (int)(c % D);
      if (jr < S && jc < S) v = cost_entry<S, F>(seg_time<S, F>(a, b, (int)(r / (3 * D))), jr, jc, a.m34);
    }
    a.Q[b * nQ + e] = v;
  } else if (e < nQ + nA) {
    const int64_t ea = e - nQ, r = ea / n, c = ea % n;
    double v = 0.0;
    const int64_t s_num = (int64_t)(N - 1) * 3 * D;
    if (r < 18) {  // boundary rows: per axis 3 start rows then 3 end rows (qp_solver.hpp:152-162)
      const int ax = (int)(r / 6), q = (int)(r % 6);
      if (q < 3) {
        if (c >= ax * D && c < (ax + 1) * D) v = basis_entry<S, F>((F)0, q, (int)(c - ax * D));
      } else {
        const int64_t c0 = s_num + ax * D;
        if (c >= c0 && c < c0 + D) v = basis_entry<S, F>(seg_time<S, F>(a, b, N - 1), q - 3, (int)(c - c0));
      }
    } else {  // continuity rows (qp_solver.hpp:165-177): [basis(T_i) | -zero_A] per knot, per axis
      const int64_t rr = r - 18;
      const int i = (int)(rr / (3 * S)), ax = (int)((rr / S) % 3), d = (int)(rr % S);
      const int64_t c0 = (int64_t)i * 3 * D + ax * D, c1 = c0 + 3 * D;
      if (c >= c0 && c < c0 + D) v = basis_entry<S, F>(seg_time<S, F>(a, b, i), d, (int)(c - c0));
      else if (c >= c1 && c < c1 + D) v = -basis_entry<S, F>((F)0, d, (int)(c - c1));
    }
    a.A[b * nA + ea] = v;
  } else if (e < nQ + nA + me) {
    const int64_t r = e - nQ - nA;
    double v = 0.0;
    if (r < 18) {
      const int ax = (int)(r / 6), q = (int)(r % 6);
      v = a.state[b * 18 + (q < 3 ? 0 : 9) + ax * 3 + (q % 3)];
    }
    a.b[b * me + r] = v;
  }
}

// [G | h]: one thread per element of G, the thread of column 0 also writes h.
template <int S, class F>
__global__ void __launch_bounds__(256) k_qp_ineq(QpArgs a) {
  constexpr int D = 2 * S;
  const int64_t b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = a.n, mg = a.mg;
  if (e >= mg * n) return;
  const int64_t r = e / n, c = e % n;
  const int N = a.N, res = a.res;
  const int *rows = a.rows + b * N;
  // locate (piece i, sample j, local row q; box?) for this row in the requested ordering
  int i = 0, j = 0, q = 0;
  bool box = false;
  if (a.row_order == 0) {
    int64_t rr = r;
    for (i = 0; i < N; ++i) {
      const int64_t blk = (int64_t)res * (rows[i] + 12);
      if (rr < blk) break;
      rr -= blk;
    }
    j = (int)(rr / (rows[i] + 12));
    q = (int)(rr % (rows[i] + 12));
    box = q >= rows[i];
    if (box) q -= rows[i];
  } else {
    int64_t tot = 0;
    for (int p = 0; p < N; ++p) tot += rows[p];
    if (r < tot * res) {
      int64_t rr = r;
      for (i = 0; i < N; ++i) {
        const int64_t blk = (int64_t)res * rows[i];
        if (rr < blk) break;
        rr -= blk;
      }
      j = (int)(rr / rows[i]);
      q = (int)(rr % rows[i]);
    } else {
      const int64_t rr = r - tot * res;
      box = true;
      i = (int)(rr / (12 * res));
      j = (int)((rr / 12) % res);
      q = (int)(rr % 12);
    }
  }
  // sample time (qp_solver.hpp:252-263): step = T_i / res, t = step * j, j == 0 uses zero_A
  const F step = seg_time<S, F>(a, b, i) / (F)res;
  const F t = (j == 0) ? (F)0 : step * (F)j;
  const int64_t c0 = (int64_t)i * 3 * D;
  double v = 0.0, hv = 0.0;
  if (!box) {
    const double *hp = a.hpolys + ((b * N + i) * a.M + q) * 4;
    if (c >= c0 && c < c0 + 3 * D) {
      const int ax = (int)((c - c0) / D);
      v = hp[ax] * basis_entry<S, F>(t, 0, (int)((c - c0) % D));
    }
    hv = hp[3];
  } else {
    // per axis: +v, +a, -v, -a  (qp_solver.hpp:280-291, min_traj_opt.py:598-611)
    const int ax = q / 4, w = q % 4;
    const int64_t ca = c0 + ax * D;
    if (c >= ca && c < ca + D) {
      const double be = basis_entry<S, F>(t, 1 + (w & 1), (int)(c - ca));
      v = (w < 2) ? be : -be;
    }
    hv = (w & 1) ? a.amax : a.vmax;
  }
  a.G[b * mg * n + e] = v;
  if (c == 0) a.h[b * mg + r] = hv;
}

// dst[f*ld + b] = src[b*nf + f] through a padded LDS tile (both sides coalesced).
constexpr int kTile = 32;
__global__ void __launch_bounds__(kTile * 8) k_to_batch_minor(const double *__restrict__ src,
                                                              double *__restrict__ dst, int64_t B,
                                                              int64_t nf, int64_t ld) {
  __shared__ double tile[kTile][kTile + 1];
  const int64_t b0 = (int64_t)blockIdx.x * kTile, f0 = (int64_t)blockIdx.y * kTile;
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t bb = b0 + i, ff = f0 + threadIdx.x;
    if (bb < B && ff < nf) tile[i][threadIdx.x] = src[bb * nf + ff];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t ff = f0 + i, bb = b0 + threadIdx.x;
    if (bb < B && ff < nf) dst[ff * ld + bb] = tile[threadIdx.x][i];
  }
}
__global__ void __launch_bounds__(kTile * 8) k_to_traj_major(const double *__restrict__ src,
                                                             double *__restrict__ dst, int64_t B,
                                                             int64_t nf, int64_t ld) {
  __shared__ double tile[kTile][kTile + 1];
  const int64_t b0 = (int64_t)blockIdx.x * kTile, f0 = (int64_t)blockIdx.y * kTile;
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t ff = f0 + i, bb = b0 + threadIdx.x;
    if (bb < B && ff < nf) tile[i][threadIdx.x] = src[ff * ld + bb];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t bb = b0 + i, ff = f0 + threadIdx.x;
    if (bb < B && ff < nf) dst[bb * nf + ff] = tile[threadIdx.x][i];
  }
}

}  // namespace anet

// ------------------------------------------------------------------------------------------
// context + error plumbing
// ------------------------------------------------------------------------------------------
struct anet_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  // grow-only device scratch for the host (trajectory-major) entry points
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // L-BFGS completion polling: device counter + pinned host mirror
  int *d_counter = nullptr;
  int *h_counter = nullptr;
};

namespace {

thread_local std::string g_err;  // errors raised without a context

int fail(anet_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}
int hip_fail(anet_ctx *ctx, hipError_t e, const char *what) {
  return fail(ctx, ANET_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define ANET_HIP(ctx, call)                                   \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);    \
  } while (0)

int ensure_scratch(anet_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return ANET_OK;
  if (ctx->scratch) {
    hipError_t e = hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    if (e != hipSuccess) return hip_fail(ctx, e, "hipFree(scratch)");
  }
  hipError_t e = hipMalloc(&ctx->scratch, bytes);
  if (e != hipSuccess) {
    ctx->scratch = nullptr;
    return fail(ctx, ANET_ERR_NOMEM, std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
  }
  ctx->scratch_bytes = bytes;
  return ANET_OK;
}

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

template <int S>
int launch_solve(anet_ctx *ctx, const anet::SolveArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock));
  const dim3 block(anet::kSolveBlock);
  // fully specialised instantiations for the shapes the benchmarks and the reference use
  if constexpr (S == 4) {
    if (a.N == 8 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<4, 8, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    if (a.N == 8 && a.c == 4) {
      hipLaunchKernelGGL((anet::k_minco_solve<4, 8, true, 3>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<3, 16, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if (a.N <= 4)
    hipLaunchKernelGGL((anet::k_minco_solve<S, 4>), grid, block, 0, st, a);
  else if (a.N <= 8)
    hipLaunchKernelGGL((anet::k_minco_solve<S, 8>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((anet::k_minco_solve<S, 16>), grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

template <int S>
int launch_prop(anet_ctx *ctx, const anet::PropArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock));
  const dim3 block(anet::kSolveBlock);
  if constexpr (S == 4) {
    if (a.N == 8 && (a.c == 3 || a.c == 4)) {
      if (a.c == 3) hipLaunchKernelGGL((anet::k_minco_propagate<4, 8, true, 2>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((anet::k_minco_propagate<4, 8, true, 3>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_propagate<3, 16, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if (a.N <= 4)
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 4>), grid, block, 0, st, a);
  else if (a.N <= 8)
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 8>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 16>), grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}
int do_propagate(anet_ctx *ctx, int s, const anet::PropArgs &a, hipStream_t st) {
  switch (s) {
    case 2: return launch_prop<2>(ctx, a, st);
    case 3: return launch_prop<3>(ctx, a, st);
    default: return launch_prop<4>(ctx, a, st);
  }
}

// ---- L-BFGS driver ----------------------------------------------------------------------------
struct LbfgsLayout {
  int n, m, npf;
  int64_t ld;
  double *x, *g, *xp, *gp, *d, *lm_s, *lm_y, *lm_ys, *lm_alpha, *pf, *ds, *feval;
  int *is;
  static int64_t doubles(int n, int m, int npf, int64_t ld) {
    // + IS_COUNT_ int32 rows, rounded up to doubles
    return ((int64_t)n * (5 + 2 * m) + 2 * m + npf + anet::DS_COUNT_ + 1 + (anet::IS_COUNT_ + 1) / 2) * ld;
  }
  void carve(double *w) {
    x = w; g = x + (int64_t)n * ld; xp = g + (int64_t)n * ld; gp = xp + (int64_t)n * ld; d = gp + (int64_t)n * ld;
    lm_s = d + (int64_t)n * ld; lm_y = lm_s + (int64_t)m * n * ld; lm_ys = lm_y + (int64_t)m * n * ld;
    lm_alpha = lm_ys + (int64_t)m * ld; pf = lm_alpha + (int64_t)m * ld; ds = pf + (int64_t)npf * ld;
    feval = ds + (int64_t)anet::DS_COUNT_ * ld; is = (int *)(feval + ld);
  }
};

static anet::LbfgsP to_kernel_params(const anet_lbfgs_params &p) {
  return anet::LbfgsP{p.mem_size, p.g_epsilon, p.past, p.delta, p.max_iterations, p.max_linesearch,
                      p.min_step, p.max_step, p.f_dec_coeff, p.s_curv_coeff, p.cautious_factor, p.machine_prec};
}

static int ensure_counter(anet_ctx *ctx) {
  if (!ctx->d_counter) ANET_HIP(ctx, hipMalloc((void **)&ctx->d_counter, sizeof(int)));
  if (!ctx->h_counter) ANET_HIP(ctx, hipHostMalloc((void **)&ctx->h_counter, sizeof(int), hipHostMallocDefault));
  return ANET_OK;
}

// eval(): enqueue the objective at L.x -> L.feval, L.g (for all problems).  The loop advances every
// problem by one evaluation per pass and polls the number of unfinished problems every `poll` passes.
template <class Eval>
static int lbfgs_drive(anet_ctx *ctx, LbfgsLayout &L, int64_t B, const anet_lbfgs_params &prm, int max_evals,
                       hipStream_t st, Eval &&eval) {
  int rc = ensure_counter(ctx);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_ * L.ld, st));
  ANET_HIP(ctx, hipMemsetAsync(L.ds, 0, sizeof(double) * anet::DS_COUNT_ * L.ld, st));
  // small batches: one wave per problem (shuffle reductions, internal vectors problem-major);
  // large batches: one lane per problem (internal vectors batch-minor)
  const bool wave = B <= 32768 && L.n <= 128 && prm.mem_size <= 64;
  anet::LbfgsArgs a{L.n, B, L.ld, L.x, L.g, L.xp, L.gp, L.d, L.lm_s, L.lm_y, L.lm_ys, L.lm_alpha, L.pf, L.ds,
                    L.feval, L.is, to_kernel_params(prm), nullptr, wave ? 1 : L.ld, wave ? L.n : 1};
  const dim3 grid(wave ? (unsigned)B : (unsigned)((B + 63) / 64)), block(64);
  const int poll = 8;
  for (int it = 0; it < max_evals; ++it) {
    if ((rc = eval())) return rc;
    const bool check = ((it + 1) % poll == 0) || it + 1 == max_evals;
    if (check) ANET_HIP(ctx, hipMemsetAsync(ctx->d_counter, 0, sizeof(int), st));
    a.n_active = check ? ctx->d_counter : nullptr;
    if (wave)
      hipLaunchKernelGGL(anet::k_lbfgs_update_wave, grid, block, 0, st, a);
    else
      hipLaunchKernelGGL(anet::k_lbfgs_update, grid, block, 0, st, a);
    ANET_HIP(ctx, hipGetLastError());
    if (check) {
      ANET_HIP(ctx, hipMemcpyAsync(ctx->h_counter, ctx->d_counter, sizeof(int), hipMemcpyDeviceToHost, st));
      ANET_HIP(ctx, hipStreamSynchronize(st));
      if (*ctx->h_counter == 0) break;
    }
  }
  return ANET_OK;
}

// status / iters / evals rows -> caller arrays (device or host destination)
__global__ void k_lbfgs_results(const int *is, const double *ds, int64_t B, int64_t ld, int *status, int *iters,
                                int *evals, double *f) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  if (status) status[b] = is[anet::IS_DONE * ld + b] ? is[anet::IS_RET * ld + b] : ANET_LBFGS_RUNNING;
  if (iters) iters[b] = is[anet::IS_K * ld + b];
  if (evals) evals[b] = is[anet::IS_EVALS * ld + b];
  if (f) f[b] = ds[anet::DS_FX * ld + b];
}
}  // namespace

extern "C" {

int anet_abi_version(void) { return ANET_ABI_VERSION; }

int anet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int anet_create(int device, anet_ctx **out) {
  if (!out) return fail(nullptr, ANET_ERR_INVALID, "anet_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, ANET_ERR_NODEVICE,
                "anet_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(nullptr, ANET_ERR_INVALID, "anet_create: bad device index");
  anet_ctx *ctx = new (std::nothrow) anet_ctx();
  if (!ctx) return fail(nullptr, ANET_ERR_NOMEM, "anet_create: out of host memory");
  ctx->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    int rc = hip_fail(nullptr, e, "anet_create");
    delete ctx;
    return rc;
  }
  *out = ctx;
  return ANET_OK;
}

void anet_destroy(anet_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->d_counter) (void)hipFree(ctx->d_counter);
  if (ctx->h_counter) (void)hipHostFree(ctx->h_counter);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *anet_last_error(const anet_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

void *anet_stream(anet_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int anet_synchronize(anet_ctx *ctx) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "anet_synchronize: ctx is NULL");
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int64_t anet_recommended_ld(int64_t batch) {
  int64_t ld = round_up(batch < 1 ? 1 : batch, 64);
  if (ld % 512 == 0) ld += 576;  // 512 doubles = 4 KiB: break (near-)power-of-two row strides
  return ld;
}

int anet_dev_alloc(anet_ctx *ctx, size_t n_doubles, double **out) {
  if (!ctx || !out) return fail(ctx, ANET_ERR_INVALID, "anet_dev_alloc: NULL argument");
  *out = nullptr;
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc((void **)out, sizeof(double) * (n_doubles ? n_doubles : 1));
  if (e != hipSuccess) return fail(ctx, ANET_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return ANET_OK;
}
void anet_dev_free(double *p) {
  if (p) (void)hipFree(p);
}
int anet_dev_upload(anet_ctx *ctx, double *dst_dev, const double *src_host, size_t n_doubles) {
  if (!ctx || !dst_dev || !src_host) return fail(ctx, ANET_ERR_INVALID, "anet_dev_upload: NULL argument");
  ANET_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, sizeof(double) * n_doubles, hipMemcpyHostToDevice, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}
int anet_dev_download(anet_ctx *ctx, double *dst_host, const double *src_dev, size_t n_doubles) {
  if (!ctx || !dst_host || !src_dev) return fail(ctx, ANET_ERR_INVALID, "anet_dev_download: NULL argument");
  ANET_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, sizeof(double) * n_doubles, hipMemcpyDeviceToHost, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int anet_to_batch_minor_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                            const double *src, double *dst, void *stream) {
  if (!ctx || !src || !dst || batch < 0 || nfield < 0 || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_to_batch_minor_dev: bad argument");
  if (batch == 0 || nfield == 0) return ANET_OK;
  dim3 grid((unsigned)((batch + anet::kTile - 1) / anet::kTile),
            (unsigned)((nfield + anet::kTile - 1) / anet::kTile));
  hipLaunchKernelGGL(anet::k_to_batch_minor, grid, dim3(anet::kTile, 8), 0, (hipStream_t)stream, src,
                     dst, batch, nfield, ld);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_to_traj_major_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                           const double *src, double *dst, void *stream) {
  if (!ctx || !src || !dst || batch < 0 || nfield < 0 || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_to_traj_major_dev: bad argument");
  if (batch == 0 || nfield == 0) return ANET_OK;
  dim3 grid((unsigned)((batch + anet::kTile - 1) / anet::kTile),
            (unsigned)((nfield + anet::kTile - 1) / anet::kTile));
  hipLaunchKernelGGL(anet::k_to_traj_major, grid, dim3(anet::kTile, 8), 0, (hipStream_t)stream, src,
                     dst, batch, nfield, ld);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

static int check_solve_args(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s < 2 || s > 4) return fail(ctx, ANET_ERR_INVALID, "order s must be 2, 3 or 4");
  if (c < 1 || c > s) return fail(ctx, ANET_ERR_INVALID, "boundary derivative count c must be in [1, s]");
  if (n_pieces < 1 || n_pieces > ANET_MAX_PIECES)
    return fail(ctx, ANET_ERR_INVALID, "piece count must be in [1, ANET_MAX_PIECES]");
  if (batch < 0) return fail(ctx, ANET_ERR_INVALID, "negative batch");
  return ANET_OK;
}

int anet_minco_solve_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head, const double *tail, const double *wps, const double *T,
                         double *coeffs, double *energy, void *stream) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps) || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve_dev: NULL input or ld < batch");
  anet::SolveArgs a{head, tail, wps, T, coeffs, energy, batch, ld, n_pieces, c};
  hipStream_t st = (hipStream_t)stream;
  switch (s) {
    case 2: return launch_solve<2>(ctx, a, st);
    case 3: return launch_solve<3>(ctx, a, st);
    default: return launch_solve<4>(ctx, a, st);
  }
}

int anet_minco_solve(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, const double *wps, const double *T, double *coeffs,
                     double *energy) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps))
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve: NULL input");
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const int N = n_pieces, D = 2 * s;
  const int64_t ld = anet_recommended_ld(batch);
  const int64_t n_in = 3 * c * 2 + (int64_t)(N - 1) * 3 + N;  // fields in per trajectory
  const int64_t n_co = (int64_t)N * 3 * D;
  const int64_t n_stage = (n_in > n_co ? n_in : n_co);
  // scratch: [stage: batch*n_stage][soa_in: n_in*ld][soa_co: n_co*ld][energy: ld]
  const size_t bytes = sizeof(double) * (size_t)(batch * n_stage + (n_in + n_co + 1) * ld);
  rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  double *stage = (double *)ctx->scratch;
  double *s_head = stage + batch * n_stage;
  double *s_tail = s_head + 3 * c * ld;
  double *s_wps = s_tail + 3 * c * ld;
  double *s_T = s_wps + (int64_t)(N - 1) * 3 * ld;
  double *s_co = s_T + (int64_t)N * ld;
  double *s_en = s_co + n_co * ld;
  hipStream_t st = ctx->stream;
  struct In { const double *h; double *d; int64_t nf; } ins[4] = {
      {head, s_head, 3 * c}, {tail, s_tail, 3 * c}, {wps, s_wps, (int64_t)(N - 1) * 3}, {T, s_T, N}};
  for (auto &in : ins) {
    if (in.nf == 0) continue;
    ANET_HIP(ctx, hipMemcpyAsync(stage, in.h, sizeof(double) * batch * in.nf, hipMemcpyHostToDevice, st));
    rc = anet_to_batch_minor_dev(ctx, batch, in.nf, ld, stage, in.d, st);
    if (rc) return rc;
  }
  rc = anet_minco_solve_dev(ctx, s, c, N, batch, ld, s_head, s_tail, s_wps, s_T, coeffs ? s_co : nullptr,
                            s_en, st);
  if (rc) return rc;
  if (coeffs) {
    rc = anet_to_traj_major_dev(ctx, batch, n_co, ld, s_co, stage, st);
    if (rc) return rc;
    ANET_HIP(ctx, hipMemcpyAsync(coeffs, stage, sizeof(double) * batch * n_co, hipMemcpyDeviceToHost, st));
  }
  if (energy)
    ANET_HIP(ctx, hipMemcpyAsync(energy, s_en, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_traj_eval_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, int nq, const double *tq, int deriv,
                       double *out, void *stream) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (deriv < 0 || deriv > 3 || nq < 0) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval: deriv in [0,3], nq >= 0");
  if (batch == 0 || nq == 0) return ANET_OK;
  if (!coeffs || !T || !tq || !out || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval_dev: NULL pointer or ld < batch");
  anet::EvalArgs a{coeffs, T, tq, out, batch, ld, n_pieces, nq, deriv};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_eval<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_eval<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_eval<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_traj_cost_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, double m34, double *cost, void *stream) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !cost || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_dev: NULL pointer or ld < batch");
  anet::CostArgs a{coeffs, T, cost, nullptr, batch, ld, n_pieces, m34};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_cost<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_cost<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_cost<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

// Host (trajectory-major) wrappers: stage -> batch-minor -> kernel -> back.
namespace {
struct Stager {
  anet_ctx *ctx;
  int64_t batch, ld;
  double *stage;   // batch * max_fields doubles
  double *cursor;  // next free batch-minor region
  int upload(const double *host, int64_t nf, double **dev) {
    *dev = cursor;
    cursor += nf * ld;
    if (nf == 0) return ANET_OK;
    hipError_t e = hipMemcpyAsync(stage, host, sizeof(double) * batch * nf, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(H2D)");
    return anet_to_batch_minor_dev(ctx, batch, nf, ld, stage, *dev, ctx->stream);
  }
  double *reserve(int64_t nf) {
    double *p = cursor;
    cursor += nf * ld;
    return p;
  }
  int download(const double *dev, int64_t nf, double *host) {
    int rc = anet_to_traj_major_dev(ctx, batch, nf, ld, dev, stage, ctx->stream);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(host, stage, sizeof(double) * batch * nf, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(D2H)");
    // the staging buffer is reused by the next transfer
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize");
    return ANET_OK;
  }
};
int make_stager(anet_ctx *ctx, int64_t batch, int64_t max_field, int64_t total_fields, Stager *st) {
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t ld = anet_recommended_ld(batch);
  int rc = ensure_scratch(ctx, sizeof(double) * (size_t)(batch * max_field + total_fields * ld));
  if (rc) return rc;
  st->ctx = ctx; st->batch = batch; st->ld = ld;
  st->stage = (double *)ctx->scratch;
  st->cursor = st->stage + batch * max_field;
  return ANET_OK;
}
}  // namespace

int anet_traj_eval(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, int nq, const double *tq, int deriv, double *out) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0 || nq <= 0) return nq < 0 ? fail(ctx, ANET_ERR_INVALID, "nq < 0") : ANET_OK;
  if (!coeffs || !T || !tq || !out) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  const int64_t mx = nco > 3 * (int64_t)nq ? nco : 3 * (int64_t)nq;
  Stager st;
  rc = make_stager(ctx, batch, mx, nco + n_pieces + nq + 3 * (int64_t)nq, &st);
  if (rc) return rc;
  double *d_co, *d_T, *d_tq;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  if ((rc = st.upload(tq, nq, &d_tq))) return rc;
  double *d_out = st.reserve(3 * (int64_t)nq);
  rc = anet_traj_eval_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, nq, d_tq, deriv, d_out, ctx->stream);
  if (rc) return rc;
  return st.download(d_out, 3 * (int64_t)nq, out);
}

int anet_traj_cost(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, double m34, double *cost) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !cost) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, nco, nco + n_pieces + 1, &st);
  if (rc) return rc;
  double *d_co, *d_T;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  double *d_cost = st.reserve(1);
  rc = anet_traj_cost_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, m34, d_cost, ctx->stream);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

// ---- cost / gradient entry points ---------------------------------------------------------------
static int check_penalty(anet_ctx *ctx, const anet_penalty *pen) {
  if (!pen) return ANET_OK;
  if (!(pen->smooth_mu > 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_penalty.smooth_mu must be > 0");
  if (pen->res < 1) return fail(ctx, ANET_ERR_INVALID, "anet_penalty.res must be >= 1");
  if (pen->poly_rows < 0 || pen->poly_rows > ANET_MAX_POLY_ROWS)
    return fail(ctx, ANET_ERR_INVALID, "anet_penalty.poly_rows must be in [0, ANET_MAX_POLY_ROWS]");
  return ANET_OK;
}

int anet_minco_partial_grads_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                                 const double *coeffs, const double *T, const double *hpolys,
                                 const anet_penalty *pen, int with_energy, double *gdC, double *gdT,
                                 double *piece_cost, void *stream) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gdC || !gdT || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_partial_grads_dev: NULL pointer or ld < batch");
  anet::PieceGradArgs a{};
  a.coeffs = coeffs; a.T = T; a.hpolys = (pen && pen->poly_rows > 0) ? hpolys : nullptr;
  a.gdC = gdC; a.gdT = gdT; a.pcost = piece_cost;
  a.B = batch; a.ld = ld; a.N = n_pieces; a.with_energy = with_energy ? 1 : 0; a.with_penalty = pen ? 1 : 0;
  if (pen) a.pp = anet::Penalty{pen->rho, pen->w_corridor, pen->w_vel, pen->w_acc, pen->smooth_mu,
                               pen->max_vel, pen->max_acc, pen->res, pen->poly_rows};
  const dim3 grid((unsigned)((batch + 255) / 256), (unsigned)n_pieces), block(256);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = pen ? sizeof(double) * (size_t)pen->res * 4 * 2 * s : 0;
  if (lds > 60 * 1024) return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_penalty.res too large for the basis table");
  if (s == 2) hipLaunchKernelGGL(anet::k_piece_grad<2>, grid, block, lds, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_piece_grad<3>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(anet::k_piece_grad<4>, grid, block, lds, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}


int anet_minco_propagate_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                  const double *T, const double *coeffs, const double *gdC,
                                  const double *gdT, double *gradP, double *gradT, void *stream) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!T || !coeffs || !gdC || !gdT || !gradT || (n_pieces > 1 && !gradP) || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_propagate_grad_dev: NULL pointer or ld < batch");
  anet::PropArgs a{T, coeffs, gdC, gdT, gradP, gradT, nullptr, nullptr, nullptr, 0.0, batch, ld, n_pieces, c};
  return do_propagate(ctx, s, a, (hipStream_t)stream);
}

int64_t anet_minco_cost_grad_workspace(int s, int n_pieces, int64_t ld) {
  // coeffs + gdC + gdT + piece cost + energy
  return ((int64_t)n_pieces * 3 * 2 * s * 2 + 2 * (int64_t)n_pieces + 1) * ld;
}

int anet_minco_cost_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                             const double *head, const double *tail, const double *wps,
                             const double *T, const double *hpolys, const anet_penalty *pen,
                             double *work, double *cost, double *gradP, double *gradT,
                             double *coeffs_out, void *stream) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!work || !cost || !gradT || (n_pieces > 1 && !gradP))
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad_dev: NULL output or workspace");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  double *w_co = coeffs_out ? coeffs_out : work;
  double *w_gdC = work + nco * ld;
  double *w_gdT = w_gdC + nco * ld;
  double *w_pc = w_gdT + (int64_t)n_pieces * ld;
  double *w_en = w_pc + (int64_t)n_pieces * ld;
  rc = anet_minco_solve_dev(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, w_co, w_en, stream);
  if (rc) return rc;
  rc = anet_minco_partial_grads_dev(ctx, s, n_pieces, batch, ld, w_co, T, hpolys, pen, 1, w_gdC, w_gdT,
                                    w_pc, stream);
  if (rc) return rc;
  anet::PropArgs a{T, w_co, w_gdC, w_gdT, gradP, gradT, w_en, pen ? w_pc : nullptr, cost,
                   pen ? pen->rho : 0.0, batch, ld, n_pieces, c};
  return do_propagate(ctx, s, a, (hipStream_t)stream);
}

int anet_minco_cost_grad(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                         const double *tail, const double *wps, const double *T, const double *hpolys,
                         const anet_penalty *pen, double *cost, double *gradP, double *gradT,
                         double *coeffs_out) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && (!wps || !gradP)) || !cost || !gradT)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad: NULL pointer");
  const int N = n_pieces;
  const int64_t nco = (int64_t)N * 3 * 2 * s;
  const int64_t M = (pen && hpolys) ? pen->poly_rows : 0;
  const int64_t nhp = (int64_t)N * M * 4;
  const int64_t n_in = 6 * (int64_t)c + (int64_t)(N - 1) * 3 + N + nhp;
  const int64_t n_out = 1 + (int64_t)(N - 1) * 3 + N + nco;
  int64_t mx = nco > nhp ? nco : nhp;
  if (mx < 3 * (int64_t)c) mx = 3 * c;
  Stager st;
  rc = make_stager(ctx, batch, mx, n_in + n_out + anet_minco_cost_grad_workspace(s, N, 1), &st);
  if (rc) return rc;
  double *d_head, *d_tail, *d_wps, *d_T, *d_hp = nullptr;
  if ((rc = st.upload(head, 3 * c, &d_head))) return rc;
  if ((rc = st.upload(tail, 3 * c, &d_tail))) return rc;
  if ((rc = st.upload(wps, (int64_t)(N - 1) * 3, &d_wps))) return rc;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  if (nhp && (rc = st.upload(hpolys, nhp, &d_hp))) return rc;
  double *d_cost = st.reserve(1), *d_gP = st.reserve((int64_t)(N - 1) * 3), *d_gT = st.reserve(N);
  double *d_co = st.reserve(nco);
  double *d_work = st.reserve(anet_minco_cost_grad_workspace(s, N, 1));
  rc = anet_minco_cost_grad_dev(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, d_hp, pen, d_work,
                                d_cost, d_gP, d_gT, d_co, ctx->stream);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
  if (N > 1 && (rc = st.download(d_gP, (int64_t)(N - 1) * 3, gradP))) return rc;
  if ((rc = st.download(d_gT, N, gradT))) return rc;
  if (coeffs_out && (rc = st.download(d_co, nco, coeffs_out))) return rc;
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

// ---- L-BFGS entry points -------------------------------------------------------------------------
void anet_lbfgs_default_params(anet_lbfgs_params *p) {
  if (!p) return;
  p->mem_size = 8; p->g_epsilon = 1.0e-5; p->past = 3; p->delta = 1.0e-6; p->max_iterations = 0;
  p->max_linesearch = 64; p->min_step = 1.0e-20; p->max_step = 1.0e+20; p->f_dec_coeff = 1.0e-4;
  p->s_curv_coeff = 0.9; p->cautious_factor = 1.0e-6; p->machine_prec = 1.0e-16;
}

int anet_lbfgs_check_params(int n, const anet_lbfgs_params *p) {
  if (!p) return -1024;
  if (n <= 0) return -1023;
  if (p->mem_size <= 0) return -1022;
  if (p->g_epsilon < 0.0) return -1021;
  if (p->past < 0) return -1020;
  if (p->delta < 0.0) return -1019;
  if (p->min_step < 0.0) return -1018;
  if (p->max_step < p->min_step) return -1017;
  if (!(p->f_dec_coeff > 0.0 && p->f_dec_coeff < 1.0)) return -1016;
  if (!(p->s_curv_coeff < 1.0 && p->s_curv_coeff > p->f_dec_coeff)) return -1015;
  if (!(p->machine_prec > 0.0)) return -1014;
  if (p->max_linesearch <= 0) return -1013;
  return 0;
}

const char *anet_lbfgs_strerror(int code) {
  switch (code) {
    case 0: return "Success: reached convergence (g_epsilon).";
    case 1: return "Success: met stopping criteria (past f decrease less than delta).";
    case 2: return "The iteration has been canceled by the monitor callback.";
    case -1024: return "Unknown error.";
    case -1023: return "Invalid number of variables specified.";
    case -1022: return "Invalid parameter lbfgs_parameter_t::mem_size specified.";
    case -1021: return "Invalid parameter lbfgs_parameter_t::g_epsilon specified.";
    case -1020: return "Invalid parameter lbfgs_parameter_t::past specified.";
    case -1019: return "Invalid parameter lbfgs_parameter_t::delta specified.";
    case -1018: return "Invalid parameter lbfgs_parameter_t::min_step specified.";
    case -1017: return "Invalid parameter lbfgs_parameter_t::max_step specified.";
    case -1016: return "Invalid parameter lbfgs_parameter_t::f_dec_coeff specified.";
    case -1015: return "Invalid parameter lbfgs_parameter_t::s_curv_coeff specified.";
    case -1014: return "Invalid parameter lbfgs_parameter_t::machine_prec specified.";
    case -1013: return "Invalid parameter lbfgs_parameter_t::max_linesearch specified.";
    case -1012: return "The function value became NaN or Inf.";
    case -1011: return "The line-search step became smaller than lbfgs_parameter_t::min_step.";
    case -1010: return "The line-search step became larger than lbfgs_parameter_t::max_step.";
    case -1009: return "Line search reaches the maximum try number, assumptions not satisfied or precision not achievable.";
    case -1008: return "The algorithm routine reaches the maximum number of iterations.";
    case -1007: return "Relative search interval width is at least lbfgs_parameter_t::machine_prec.";
    case -1006: return "A logic error (negative line-search step) occurred.";
    case -1005: return "The current search direction increases the cost function value.";
    case ANET_LBFGS_RUNNING: return "Still running: the evaluation budget (max_evals) was exhausted.";
    default: return "(unknown)";
  }
}

static int check_lbfgs(anet_ctx *ctx, int n, const anet_lbfgs_params *params, int max_evals) {
  const int code = anet_lbfgs_check_params(n, params);
  if (code) return fail(ctx, ANET_ERR_INVALID, std::string("lbfgs parameters rejected: ") + anet_lbfgs_strerror(code));
  if (max_evals <= 0) return fail(ctx, ANET_ERR_INVALID, "max_evals must be > 0");
  return ANET_OK;
}

int anet_lbfgs_mvie(anet_ctx *ctx, int64_t batch, int M, const double *A, double smooth_eps,
                    double penalty_wt, double *x, double *f, const anet_lbfgs_params *params,
                    int max_evals, int32_t *status, int32_t *iters, int32_t *evals) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  int rc = check_lbfgs(ctx, 9, params, max_evals);
  if (rc) return rc;
  if (batch < 0 || M < 1 || !(smooth_eps > 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_mvie: bad batch, M or smooth_eps");
  if (batch == 0) return ANET_OK;
  if (!A || !x) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_mvie: NULL pointer");
  const int n = 9, m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  Stager st;
  const int64_t wdoubles = LbfgsLayout::doubles(n, m, npf, 1);
  const int64_t mx = 3 * (int64_t)M > n ? 3 * (int64_t)M : n;
  rc = make_stager(ctx, batch, mx, 3 * (int64_t)M + n + wdoubles + 3, &st);
  if (rc) return rc;
  double *d_A, *d_x0;
  if ((rc = st.upload(A, 3 * (int64_t)M, &d_A))) return rc;
  if ((rc = st.upload(x, n, &d_x0))) return rc;
  LbfgsLayout L{n, m, npf, st.ld};
  L.carve(st.reserve(wdoubles));
  int *d_res = (int *)st.reserve(3);  // status, iters, evals rows (int32, ld each; 3*ld doubles is ample)
  hipStream_t s0 = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(L.x, d_x0, sizeof(double) * n * st.ld, hipMemcpyDeviceToDevice, s0));
  anet::MvieArgs ma{d_A, L.x, L.feval, L.g, L.is, batch, st.ld, M, smooth_eps, penalty_wt};
  const dim3 grid((unsigned)((batch + 63) / 64)), block(64);
  rc = lbfgs_drive(ctx, L, batch, *params, max_evals, s0, [&]() -> int {
    hipLaunchKernelGGL(anet::k_mvie_eval, grid, block, 0, s0, ma);
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  });
  if (rc) return rc;
  hipLaunchKernelGGL(k_lbfgs_results, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s0, L.is, L.ds, batch,
                     st.ld, d_res, d_res + st.ld, d_res + 2 * st.ld, L.feval);
  ANET_HIP(ctx, hipGetLastError());
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_res, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_res + st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (evals) ANET_HIP(ctx, hipMemcpyAsync(evals, d_res + 2 * st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (f) ANET_HIP(ctx, hipMemcpyAsync(f, L.feval, sizeof(double) * batch, hipMemcpyDeviceToHost, s0));
  return st.download(L.x, n, x);
}

int64_t anet_lbfgs_minco_workspace(int s, int n_pieces, int64_t ld, const anet_lbfgs_params *params) {
  if (!params || params->mem_size <= 0) return -1;
  const int n = 3 * (n_pieces - 1) + n_pieces;
  const int npf = params->past > 1 ? params->past : 1;
  // L-BFGS state + cost/grad workspace + gradP + gradT
  return LbfgsLayout::doubles(n, params->mem_size, npf, ld) + anet_minco_cost_grad_workspace(s, n_pieces, ld) +
         (int64_t)n * ld;
}

int anet_lbfgs_minco_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head, const double *tail, double *wps, double *T,
                         const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                         int opt_flags, int max_evals, double *work, double *cost, double *coeffs_out,
                         int32_t *status, int32_t *iters, int32_t *evals, void *stream) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  const int N = n_pieces;
  const int nw = (opt_flags & ANET_OPT_WAYPOINTS) ? 3 * (N - 1) : 0;
  const int nt = (opt_flags & ANET_OPT_TIMES) ? N : 0;
  const int n = nw + nt;
  if (n <= 0) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: nothing to optimise (opt_flags / N)");
  if ((rc = check_lbfgs(ctx, n, params, max_evals))) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (N > 1 && !wps) || !work || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco_dev: NULL pointer or ld < batch");
  const int m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  LbfgsLayout L{n, m, npf, ld};
  L.carve(work);
  double *w_cg = work + LbfgsLayout::doubles(n, m, npf, ld);
  double *w_gP = w_cg + anet_minco_cost_grad_workspace(s, N, ld);
  double *w_gT = w_gP + (int64_t)3 * (N - 1) * ld;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g256((unsigned)((batch + 255) / 256)), b256(256);
  anet::MapArgs mp{L.x, L.g, wps, T, w_gP, w_gT, batch, ld, nw, nt, 0};
  const dim3 gmap(g256.x, (unsigned)n);
  hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
  ANET_HIP(ctx, hipGetLastError());
  bool first = true;
  rc = lbfgs_drive(ctx, L, batch, *params, max_evals, st, [&]() -> int {
    if (!first) {
      mp.mode = 1;
      hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
      ANET_HIP(ctx, hipGetLastError());
    }
    first = false;
    int r = anet_minco_cost_grad_dev(ctx, s, c, N, batch, ld, head, tail, wps, T, hpolys, pen, w_cg, L.feval,
                                     w_gP, w_gT, nullptr, st);
    if (r) return r;
    mp.mode = 2;
    hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  });
  if (rc) return rc;
  // final parameters (x may have been reverted by a failed line search) and outputs
  mp.mode = 1;
  hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
  ANET_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_lbfgs_results, g256, b256, 0, st, L.is, L.ds, batch, ld, status, iters, evals, cost);
  ANET_HIP(ctx, hipGetLastError());
  if (coeffs_out) return anet_minco_solve_dev(ctx, s, c, N, batch, ld, head, tail, wps, T, coeffs_out, nullptr, st);
  return ANET_OK;
}

int anet_lbfgs_minco(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, double *wps, double *T, const double *hpolys,
                     const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                     int max_evals, double *cost, double *coeffs_out, int32_t *status, int32_t *iters,
                     int32_t *evals) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (!params) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: params is NULL");
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: NULL pointer");
  const int N = n_pieces;
  const int64_t nco = (int64_t)N * 3 * 2 * s;
  const int64_t M = (pen && hpolys) ? pen->poly_rows : 0;
  const int64_t nhp = (int64_t)N * M * 4;
  const int64_t wdoubles = anet_lbfgs_minco_workspace(s, N, 1, params);
  if (wdoubles < 0) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: bad lbfgs parameters");
  int64_t mx = nco > nhp ? nco : nhp;
  if (mx < 3 * (int64_t)c) mx = 3 * c;
  Stager st;
  rc = make_stager(ctx, batch, mx, 6 * (int64_t)c + 3 * (int64_t)(N - 1) + N + nhp + nco + wdoubles + 4, &st);
  if (rc) return rc;
  double *d_head, *d_tail, *d_wps, *d_T, *d_hp = nullptr;
  if ((rc = st.upload(head, 3 * c, &d_head))) return rc;
  if ((rc = st.upload(tail, 3 * c, &d_tail))) return rc;
  if ((rc = st.upload(wps, (int64_t)(N - 1) * 3, &d_wps))) return rc;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  if (nhp && (rc = st.upload(hpolys, nhp, &d_hp))) return rc;
  double *d_co = st.reserve(nco), *d_work = st.reserve(wdoubles), *d_cost = st.reserve(1);
  int *d_res = (int *)st.reserve(3);
  rc = anet_lbfgs_minco_dev(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, d_hp, pen, params, opt_flags,
                            max_evals, d_work, d_cost, coeffs_out ? d_co : nullptr, d_res, d_res + st.ld,
                            d_res + 2 * st.ld, ctx->stream);
  if (rc) return rc;
  hipStream_t s0 = ctx->stream;
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_res, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_res + st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (evals) ANET_HIP(ctx, hipMemcpyAsync(evals, d_res + 2 * st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (cost) ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, s0));
  if (N > 1 && (rc = st.download(d_wps, (int64_t)(N - 1) * 3, wps))) return rc;
  if ((rc = st.download(d_T, N, T))) return rc;
  if (coeffs_out && (rc = st.download(d_co, nco, coeffs_out))) return rc;
  ANET_HIP(ctx, hipStreamSynchronize(s0));
  return ANET_OK;
}

// ---- QP assembly entry points --------------------------------------------------------------------
int anet_qp_dims_of(int s, int n_pieces, int res, const int32_t *rows, anet_qp_dims *out) {
  if (!out || !rows || (s != 3 && s != 4) || n_pieces < 1 || res < 1) return ANET_ERR_INVALID;
  int64_t tot = 0;
  for (int i = 0; i < n_pieces; ++i) {
    if (rows[i] < 0) return ANET_ERR_INVALID;
    tot += rows[i];
  }
  out->n = (int64_t)3 * 2 * s * n_pieces;
  out->m_e = 3 * (6 + (int64_t)s * (n_pieces - 1));
  out->m_g = (int64_t)res * (tot + 12 * (int64_t)n_pieces);
  return ANET_OK;
}

int anet_qp_assemble_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M,
                         double max_vel, double max_acc, double m34, int float_time, int row_order,
                         const double *state, const double *T, const double *hpolys, const int32_t *rows,
                         double *Q, double *A, double *b, double *G, double *h, void *stream) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s != 3 && s != 4) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: order must be 3 (jerk) or 4 (snap), qp_solver.hpp:61-83");
  if (n_pieces < 1 || batch < 0 || res < 1 || M < 0 || (row_order != 0 && row_order != 1))
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || !rows || (M > 0 && !hpolys) || !Q || !A || !b || !G || !h)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: NULL pointer");
  // one shape for the whole batch: read the first trajectory's row counts (device -> host, tiny)
  std::vector<int32_t> r0((size_t)n_pieces * batch);
  ANET_HIP(ctx, hipMemcpy(r0.data(), rows, sizeof(int32_t) * n_pieces * batch, hipMemcpyDeviceToHost));
  anet_qp_dims dm;
  if (anet_qp_dims_of(s, n_pieces, res, r0.data(), &dm)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad row counts");
  for (int64_t bb = 0; bb < batch; ++bb) {
    int64_t tot = 0;
    for (int i = 0; i < n_pieces; ++i) {
      const int32_t v = r0[(size_t)bb * n_pieces + i];
      if (v < 0 || v > M) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: rows[b][i] must be in [0, M]");
      tot += v;
    }
    if ((int64_t)res * (tot + 12 * (int64_t)n_pieces) != dm.m_g)
      return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: every trajectory of a batch needs the same total polytope row count");
  }
  anet::QpArgs a{state, T, hpolys, rows, Q, A, b, G, h, batch, dm.n, dm.m_e, dm.m_g, n_pieces, res, M,
                 float_time ? 1 : 0, row_order, max_vel, max_acc, m34};
  hipStream_t st = (hipStream_t)stream;
  const int64_t ne = dm.n * dm.n + dm.m_e * dm.n + dm.m_e, ng = dm.m_g * dm.n;
  const dim3 blk(256), g1((unsigned)((ne + 255) / 256), (unsigned)batch), g2((unsigned)((ng + 255) / 256), (unsigned)batch);
  if (s == 4) {
    if (float_time) { hipLaunchKernelGGL((anet::k_qp_eq_obj<4, float>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<4, float>), g2, blk, 0, st, a); }
    else { hipLaunchKernelGGL((anet::k_qp_eq_obj<4, double>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<4, double>), g2, blk, 0, st, a); }
  } else {
    if (float_time) { hipLaunchKernelGGL((anet::k_qp_eq_obj<3, float>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<3, float>), g2, blk, 0, st, a); }
    else { hipLaunchKernelGGL((anet::k_qp_eq_obj<3, double>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<3, double>), g2, blk, 0, st, a); }
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_qp_assemble(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                     double max_acc, double m34, int float_time, int row_order, const double *state,
                     const double *T, const double *hpolys, const int32_t *rows, double *Q, double *A,
                     double *b, double *G, double *h) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if ((s != 3 && s != 4) || n_pieces < 1 || batch < 0 || res < 1 || M < 0 || !rows)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad argument");
  if (batch == 0) return ANET_OK;
  anet_qp_dims dm;
  if (anet_qp_dims_of(s, n_pieces, res, rows, &dm)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad row counts");
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_state = 18 * (size_t)batch, n_T = (size_t)n_pieces * batch, n_hp = (size_t)batch * n_pieces * M * 4;
  const size_t n_rows = ((size_t)n_pieces * batch + 1) / 2;  // int32 pairs in doubles
  const size_t nQ = (size_t)(dm.n * dm.n) * batch, nA = (size_t)(dm.m_e * dm.n) * batch, nb = (size_t)dm.m_e * batch;
  const size_t nG = (size_t)(dm.m_g * dm.n) * batch, nh = (size_t)dm.m_g * batch;
  int rc = ensure_scratch(ctx, sizeof(double) * (n_state + n_T + n_hp + n_rows + nQ + nA + nb + nG + nh + 8));
  if (rc) return rc;
  double *d_state = (double *)ctx->scratch, *d_T = d_state + n_state, *d_hp = d_T + n_T;
  int32_t *d_rows = (int32_t *)(d_hp + n_hp);
  double *d_Q = d_hp + n_hp + n_rows, *d_A = d_Q + nQ, *d_b = d_A + nA, *d_G = d_b + nb, *d_h = d_G + nG;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_state, state, sizeof(double) * n_state, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_T, T, sizeof(double) * n_T, hipMemcpyHostToDevice, st));
  if (n_hp) ANET_HIP(ctx, hipMemcpyAsync(d_hp, hpolys, sizeof(double) * n_hp, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_rows, rows, sizeof(int32_t) * n_pieces * batch, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  rc = anet_qp_assemble_dev(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, float_time, row_order, d_state,
                            d_T, d_hp, d_rows, d_Q, d_A, d_b, d_G, d_h, st);
  if (rc) return rc;
  if (Q) ANET_HIP(ctx, hipMemcpyAsync(Q, d_Q, sizeof(double) * nQ, hipMemcpyDeviceToHost, st));
  if (A) ANET_HIP(ctx, hipMemcpyAsync(A, d_A, sizeof(double) * nA, hipMemcpyDeviceToHost, st));
  if (b) ANET_HIP(ctx, hipMemcpyAsync(b, d_b, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
  if (G && nG) ANET_HIP(ctx, hipMemcpyAsync(G, d_G, sizeof(double) * nG, hipMemcpyDeviceToHost, st));
  if (h && nh) ANET_HIP(ctx, hipMemcpyAsync(h, d_h, sizeof(double) * nh, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

// ---- QP solve (ADMM) entry points ------------------------------------------------------------------
void anet_qp_default_settings(anet_qp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->max_iter = 4000; s->check_termination = 25; s->adaptive_rho_interval = 100;
}

int64_t anet_qp_solve_workspace(int s, int n_pieces, int64_t batch, int res, int M) {
  const int64_t m = 3 * (6 + (int64_t)s * (n_pieces - 1)) + (int64_t)n_pieces * res * (M + 12);
  return 2 * m * batch + 2 * batch;  // z, y, residuals
}

int anet_qp_solve_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                      double max_acc, double m34, const double *state, const double *T,
                      const double *hpolys, const anet_qp_settings *settings, double *work, double *coeffs,
                      double *obj, int32_t *status, int32_t *iters, double *residuals, void *stream) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s != 3 && s != 4) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: order must be 3 (jerk) or 4 (snap)");
  if (n_pieces < 1 || batch < 0 || res < 1 || M < 0) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || (M > 0 && !hpolys) || !work || !coeffs || !obj || !status || !iters)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: NULL pointer");
  anet_qp_settings st_;
  anet_qp_default_settings(&st_);
  if (settings) st_ = *settings;
  if (!(st_.rho > 0) || !(st_.sigma > 0) || !(st_.alpha > 0 && st_.alpha < 2) || st_.max_iter < 1 ||
      st_.check_termination < 1 || st_.eps_abs < 0 || st_.eps_rel < 0 || st_.adaptive_rho_interval < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad settings");
  const size_t lds = (s == 4) ? anet::qp_admm_lds_bytes<4>(n_pieces, res) : anet::qp_admm_lds_bytes<3>(n_pieces, res);
  if (lds > 160 * 1024)
    return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_qp_solve: the block factor of this many pieces does not fit the 160 KB LDS");
  const int64_t m = 3 * (6 + (int64_t)s * (n_pieces - 1)) + (int64_t)n_pieces * res * (M + 12);
  int adapt = st_.adaptive_rho_interval;
  if (adapt > 0) adapt = (adapt + st_.check_termination - 1) / st_.check_termination * st_.check_termination;
  anet::AdmmArgs a{state, T, hpolys, work, work + m * batch, coeffs, obj, status, iters,
                   residuals ? residuals : work + 2 * m * batch, batch, n_pieces, res, M, max_vel, max_acc, m34,
                   anet::AdmmParams{st_.rho, st_.sigma, st_.alpha, st_.eps_abs, st_.eps_rel, st_.max_iter,
                                    st_.check_termination, adapt}};
  hipStream_t st = (hipStream_t)stream;
  if (s == 4) {
    ANET_HIP(ctx, hipFuncSetAttribute((const void *)anet::k_qp_admm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((anet::k_qp_admm<4>), dim3((unsigned)batch), dim3(256), lds, st, a);
  } else {
    ANET_HIP(ctx, hipFuncSetAttribute((const void *)anet::k_qp_admm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((anet::k_qp_admm<3>), dim3((unsigned)batch), dim3(256), lds, st, a);
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_qp_solve(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                  double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                  const anet_qp_settings *settings, double *coeffs, double *obj, int32_t *status,
                  int32_t *iters, double *residuals) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if ((s != 3 && s != 4) || n_pieces < 1 || batch < 0 || res < 1 || M < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || (M > 0 && !hpolys) || !coeffs) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: NULL pointer");
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)3 * 2 * s * n_pieces;
  const size_t n_state = 18 * (size_t)batch, n_T = (size_t)n_pieces * batch, n_hp = (size_t)batch * n_pieces * M * 4;
  const size_t n_work = (size_t)anet_qp_solve_workspace(s, n_pieces, batch, res, M);
  const size_t n_int = (size_t)batch;  // 2 int32 arrays fit in `batch` doubles
  int rc = ensure_scratch(ctx, sizeof(double) * (n_state + n_T + n_hp + n_work + n * batch + 3 * batch + n_int + 8));
  if (rc) return rc;
  double *d_state = (double *)ctx->scratch, *d_T = d_state + n_state, *d_hp = d_T + n_T, *d_work = d_hp + n_hp;
  double *d_co = d_work + n_work, *d_obj = d_co + n * batch, *d_res = d_obj + batch;
  int32_t *d_status = (int32_t *)(d_res + 2 * batch), *d_iters = d_status + batch;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_state, state, sizeof(double) * n_state, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_T, T, sizeof(double) * n_T, hipMemcpyHostToDevice, st));
  if (n_hp) ANET_HIP(ctx, hipMemcpyAsync(d_hp, hpolys, sizeof(double) * n_hp, hipMemcpyHostToDevice, st));
  rc = anet_qp_solve_dev(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, d_state, d_T, d_hp, settings, d_work,
                         d_co, d_obj, d_status, d_iters, d_res, st);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(coeffs, d_co, sizeof(double) * n * batch, hipMemcpyDeviceToHost, st));
  if (obj) ANET_HIP(ctx, hipMemcpyAsync(obj, d_obj, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_status, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, st));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_iters, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, st));
  if (residuals) ANET_HIP(ctx, hipMemcpyAsync(residuals, d_res, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_traj_cost_grad_T_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                              const double *coeffs, const double *T, double m34, double *gradT, void *stream) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gradT || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_grad_T_dev: NULL pointer or ld < batch");
  anet::CostArgs a{coeffs, T, nullptr, gradT, batch, ld, n_pieces, m34};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_cost<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_cost<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_cost<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_traj_cost_grad_T(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                          const double *T, double m34, double *gradT) {
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gradT) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_grad_T: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, nco, nco + 2 * (int64_t)n_pieces, &st);
  if (rc) return rc;
  double *d_co, *d_T;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  double *d_g = st.reserve(n_pieces);
  rc = anet_traj_cost_grad_T_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, m34, d_g, ctx->stream);
  if (rc) return rc;
  return st.download(d_g, n_pieces, gradT);
}

}  // extern "C"
