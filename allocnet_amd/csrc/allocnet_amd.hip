// allocnet_amd: gfx950 kernels + C ABI (include/allocnet_amd.h).  Built by allocnet_amd/build.py:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch, no Eigen.  There is no CPU fallback in this library: without a device every entry
// point fails with ANET_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <new>

#include "../../include/allocnet_amd.h"
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
  double *cost;
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
    double Q[S][S];
    if constexpr (S == 4) {
      const double t6 = t3 * t3, t7 = t4 * t3;
      Q[0][0] = 100800 * t7; Q[0][1] = 50400 * t6; Q[0][2] = 20160 * t5; Q[0][3] = 5040 * t4;
      Q[1][1] = 25920 * t5;  Q[1][2] = 10800 * t4; Q[1][3] = 2880 * t3;
      Q[2][2] = 4800 * t3;   Q[2][3] = a.m34 * t2;
      Q[3][3] = 576 * t;
    } else if constexpr (S == 3) {
      Q[0][0] = 720 * t5; Q[0][1] = 360 * t4; Q[0][2] = 120 * t3;
      Q[1][1] = 192 * t3; Q[1][2] = 72 * t2;
      Q[2][2] = 36 * t;
    } else {
      Q[0][0] = 12 * t3; Q[0][1] = 6 * t2;
      Q[1][1] = 4 * t;
    }
#pragma unroll
    for (int j = 1; j < S; ++j)
#pragma unroll
      for (int k = 0; k < j; ++k) Q[j][k] = Q[k][j];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      double z[S];
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] = a.coeffs[(int64_t)((i * 3 + ax) * D + j) * ld + b];
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) r += Q[j][k] * z[k];
        acc += z[j] * r;
      }
      energy += 0.5 * acc;
    }
  }
  a.cost[b] = energy;
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
    const Penalty pp = a.pp;
    const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
    const double step = Ti * inv_res;
    for (int j = 0; j < pp.res; ++j) {
      const double t = (double)j * step;
      double tp[D];
      tp[0] = 1.0;
#pragma unroll
      for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * t;
      double be[4][D];  // basis rows p,v,a,j at t (highest power first)
#pragma unroll
      for (int col = 0; col < D; ++col) {
        const int k = D - 1 - col;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          double f = 1.0;
#pragma unroll
          for (int e = 0; e < d; ++e) f *= (double)(k - e);
          be[d][col] = (k >= d) ? f * tp[k >= d ? k - d : 0] : 0.0;
        }
      }
      double st[4][3];
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          double acc = 0.0;
#pragma unroll
          for (int col = 0; col < D; ++col) acc = __builtin_fma(c[ax][col], be[d][col], acc);
          st[d][ax] = acc;
        }
      double cost = 0.0, g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // d cost / d (p,v,a)
      if (a.hpolys) {
        const double *hp = a.hpolys + (int64_t)(i * pp.M * 4) * ld + b;
        for (int r = 0; r < pp.M; ++r) {
          const double a0 = hp[(int64_t)(r * 4 + 0) * ld], a1 = hp[(int64_t)(r * 4 + 1) * ld];
          const double a2 = hp[(int64_t)(r * 4 + 2) * ld], bb = hp[(int64_t)(r * 4 + 3) * ld];
          const double viol = __builtin_fma(a0, st[0][0], __builtin_fma(a1, st[0][1], a2 * st[0][2])) - bb;
          double f, df;
          smoothed_l1(pp.mu, inv_mu, viol, f, df);
          cost = __builtin_fma(pp.wc, f, cost);
          df *= pp.wc;
          g[0][0] = __builtin_fma(df, a0, g[0][0]);
          g[0][1] = __builtin_fma(df, a1, g[0][1]);
          g[0][2] = __builtin_fma(df, a2, g[0][2]);
        }
      }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          const double sgn = sg ? -1.0 : 1.0;
          double f, df;
          smoothed_l1(pp.mu, inv_mu, sgn * st[1][ax] - pp.vmax, f, df);
          cost = __builtin_fma(pp.wv, f, cost);
          g[1][ax] = __builtin_fma(pp.wv * sgn, df, g[1][ax]);
          smoothed_l1(pp.mu, inv_mu, sgn * st[2][ax] - pp.amax, f, df);
          cost = __builtin_fma(pp.wa, f, cost);
          g[2][ax] = __builtin_fma(pp.wa * sgn, df, g[2][ax]);
        }
      }
      pc = __builtin_fma(step, cost, pc);
      double dt = 0.0;  // d cost / d t = g_p.v + g_v.a + g_a.j
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) dt = __builtin_fma(g[d][ax], st[d + 1][ax], dt);
      gT += cost * inv_res + step * dt * ((double)j * inv_res);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax)
#pragma unroll
        for (int col = 0; col < D; ++col) {
          double acc = g[0][ax] * be[0][col];
          acc = __builtin_fma(g[1][ax], be[1][col], acc);
          acc = __builtin_fma(g[2][ax], be[2][col], acc);
          gC[ax][col] = __builtin_fma(step, acc, gC[ax][col]);
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
  const int64_t ld = round_up(batch, 64);
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
  anet::CostArgs a{coeffs, T, cost, batch, ld, n_pieces, m34};
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
  const int64_t ld = round_up(batch, 64);
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
  if (s == 2) hipLaunchKernelGGL(anet::k_piece_grad<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_piece_grad<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_piece_grad<4>, grid, block, 0, st, a);
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

}  // extern "C"
