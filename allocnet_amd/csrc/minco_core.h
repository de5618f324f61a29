// Per-lane MINCO machinery for gfx950: one trajectory per lane, batch-minor (SoA) global layout
// so that every global access of a wave is one contiguous 512-B segment.
//
// Algorithm (derived; `minco.hpp` is not in the reference tree -- SURVEY.md section 0):
// in normalised time each piece is a Hermite interpolant of its two end states
// x_k = (p, p', .., p^(s-1)) and its control effort is  r^(2s-1) u' M u  with a CONSTANT integer
// matrix M (minco_tables.h), r = 1/T, u = diag(T^deg)[x_k; x_k+1].  Minimising over the free
// node derivatives gives a symmetric positive definite block-tridiagonal system with (s-1)x(s-1)
// blocks, shared by the three axes.  It is factorised once per trajectory (block LDL^T, no
// pivoting needed, no square roots) and each axis is a forward/backward sweep.  This is ~4x
// fewer FP64 operations than the 2sN x 2sN banded collocation LU and needs no pivoting, which
// is what lets one lane own one trajectory with the factor resident in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "minco_tables.h"

namespace anet {

// 1/d to full double precision: v_rcp_f64 + two Newton steps (no div_scale/div_fixup chain).
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}

// 1/d to ~1e-14: v_rcp_f64 + ONE Newton step -- for quantities that only steer an iteration (interior-point weights and
// step directions: the iteration tests its true residuals), not for results.
__device__ __forceinline__ double fast_rcp1(double d) {
  double x = __builtin_amdgcn_rcp(d);
  return __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
}

// Optimisation barrier.  The sweeps below deliberately RECOMPUTE the cheap per-piece quantities
// (powers of r, the coupling block Y = L^-1 Ko) instead of keeping them: left alone the compiler
// would CSE/hoist them across the sweeps and hold ~150 doubles live, which spills.  Laundering
// r through an empty asm makes each sweep's copy a distinct value.
__device__ __forceinline__ double launder(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int S>
struct Pw {  // r^1 .. r^(2S-1)
  double v[2 * S];
  __device__ __forceinline__ explicit Pw(double r) {
    v[0] = 1.0;
    v[1] = r;
#pragma unroll
    for (int e = 2; e < 2 * S; ++e) v[e] = v[e - 1] * r;
  }
  __device__ __forceinline__ double operator[](int e) const { return v[e]; }
};

// Block-tridiagonal SPD factor, one trajectory.  Node k in [0, N]; unknown j in [0, m) is the
// derivative of order j+1 at that node.  `np` = c-1 derivatives are pinned at nodes 0 and N.
//
// Coupling block of piece i (unknowns of node i x unknowns of node i+1):
//   Ko(j,l) = M[1+j][S+1+l] r^(2S-3-j-l) = a_j * M[1+j][S+1+l] * b_l,  a_j = r^(S-2-j), b_l = r^(S-1-l)
// so products with Ko are a scale, an m x m product with CONSTANTS, and a scale.
template <int S, int NB>
struct Factor {
  static constexpr int m = S - 1;
  static constexpr int nl = m * (m - 1) / 2;
  double L[NB + 1][nl > 0 ? nl : 1];  // strict lower of the unit-lower LDL^T factor, row-major
  double dinv[NB + 1][m];
  double r[NB];  // 1/T per piece

  __device__ __forceinline__ static int li(int i, int j) { return i * (i - 1) / 2 + j; }  // i>j

  // masked constant of Ko: pinned rows (node i = 0) / columns (node i+1 = N) are zero
  // (TF, "tail free": the chain is the HALF of a trajectory that is eliminated from both ends -- its last node is the interior
  //  node where the halves meet, nothing is pinned there; minco_fused_kernel.h)
  template <bool TF = false>
  __device__ __forceinline__ static double ko_const(int i, int N, int np, int j, int l) {
    return ((i == 0 && j < np) || (!TF && i == N - 1 && l < np)) ? 0.0 : Tab<S>::M[1 + j][S + 1 + l];
  }
  // out[l] -= sum_j Ko_i(j,l) v[j]      (Ko' v)
  template <bool TF = false>
  __device__ __forceinline__ static void sub_KoT(int i, int N, int np, const Pw<S> &p, const double (&v)[m],
                                                 double (&out)[m]) {
    double sv[m];
#pragma unroll
    for (int j = 0; j < m; ++j) sv[j] = v[j] * p[S - 2 - j];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) acc = __builtin_fma(ko_const<TF>(i, N, np, j, l), sv[j], acc);
      out[l] = __builtin_fma(-acc, p[S - 1 - l], out[l]);
    }
  }
  // out[j] = sum_l Ko_i(j,l) v[l]       (Ko v)
  template <bool TF = false>
  __device__ __forceinline__ static void mul_Ko(int i, int N, int np, const Pw<S> &p, const double (&v)[m],
                                                double (&out)[m]) {
    double sv[m];
#pragma unroll
    for (int l = 0; l < m; ++l) sv[l] = v[l] * p[S - 1 - l];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int l = 0; l < m; ++l) acc = __builtin_fma(ko_const<TF>(i, N, np, j, l), sv[l], acc);
      out[j] = acc * p[S - 2 - j];
    }
  }
  // v <- L_k^-1 v   (unit lower)
  __device__ __forceinline__ void solve_L(int k, double (&v)[m]) const {
#pragma unroll
    for (int i = 1; i < m; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) v[i] = __builtin_fma(-L[k][li(i, j)], v[j], v[i]);
  }
  // v <- L_k^-T v
  __device__ __forceinline__ void solve_LT(int k, double (&v)[m]) const {
#pragma unroll
    for (int i = m - 2; i >= 0; --i)
#pragma unroll
      for (int j = i + 1; j < m; ++j) v[i] = __builtin_fma(-L[k][li(j, i)], v[j], v[i]);
  }

  __device__ __forceinline__ void factorize(int N, int np) {
    factorize_chain<false>(N, np, [](double (&)[m][m]) {});
  }
  // TF: node N is free and its block is completed by `meet(Dk)` (lower triangle) before it is factored
  template <bool TF, class Meet>
  __device__ __forceinline__ void factorize_chain(int N, int np, Meet &&meet) {
    // lower triangle of the current diagonal block (symmetric): Dk[j][l], l <= j
    double Dk[m][m];
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l < m; ++l) Dk[j][l] = 0.0;
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      if (k <= N) {
        // --- assemble the diagonal block of node k (Schur part from node k-1 already in Dk)
        if (k < N) {
          Pw<S> p(r[k]);
#pragma unroll
          for (int j = 0; j < m; ++j)
#pragma unroll
            for (int l = 0; l <= j; ++l)
              Dk[j][l] = __builtin_fma(Tab<S>::M[1 + j][1 + l], p[2 * S - 3 - j - l], Dk[j][l]);
        }
        if (k > 0) {
          Pw<S> p(r[k - 1]);
#pragma unroll
          for (int j = 0; j < m; ++j)
#pragma unroll
            for (int l = 0; l <= j; ++l)
              Dk[j][l] =
                  __builtin_fma(Tab<S>::M[S + 1 + j][S + 1 + l], p[2 * S - 3 - j - l], Dk[j][l]);
        }
        if (k == 0 || (!TF && k == N)) {
#pragma unroll
          for (int j = 0; j < m; ++j)
#pragma unroll
            for (int l = 0; l <= j; ++l)
              if (j < np || l < np) Dk[j][l] = (j == l) ? 1.0 : 0.0;
        }
        if (TF && k == N) meet(Dk);
        // --- LDL^T of the m x m block
        double d[m];
#pragma unroll
        for (int j = 0; j < m; ++j) {
          double dj = Dk[j][j];
          double ld_[m];  // L[j][q] * d[q]
#pragma unroll
          for (int q = 0; q < j; ++q) {
            ld_[q] = L[k][li(j, q)] * d[q];
            dj = __builtin_fma(-ld_[q], L[k][li(j, q)], dj);
          }
          d[j] = dj;
          dinv[k][j] = fast_rcp(dj);
#pragma unroll
          for (int i = j + 1; i < m; ++i) {
            double v = Dk[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) v = __builtin_fma(-L[k][li(i, q)], ld_[q], v);
            L[k][li(i, j)] = v * dinv[k][j];
          }
        }
        // --- Schur complement seed for node k+1:  -Ko' D_k^-1 Ko = -Y' diag(dinv) Y,  Y = L^-1 Ko
        if (k < N) {
          Pw<S> p(r[k]);
          double Y[m][m], Z[m][m];
#pragma unroll
          for (int l = 0; l < m; ++l) {
            double col[m];
#pragma unroll
            for (int j = 0; j < m; ++j) col[j] = ko_const<TF>(k, N, np, j, l) * p[2 * S - 3 - j - l];
            solve_L(k, col);
#pragma unroll
            for (int j = 0; j < m; ++j) {
              Y[j][l] = col[j];
              Z[j][l] = col[j] * dinv[k][j];
            }
          }
#pragma unroll
          for (int a = 0; a < m; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
              double acc = 0.0;
#pragma unroll
              for (int j = 0; j < m; ++j) acc = __builtin_fma(-Y[j][a], Z[j][b], acc);
              Dk[a][b] = acc;
            }
        }
      }
    }
  }
};

// ---- sweeps of the block-tridiagonal system for ONE right-hand side (one axis) -------------------
// forward : X <- w,   y_k = rhs_k - Ko_{k-1}' D_{k-1}^-1 y_{k-1},  w_k = L_k^-1 y_k
//           `rhs(k, y)` fills the right-hand side of node k just before it is eliminated (fused so
//           the powers of r it needs are shared with the elimination step).
// backward: X <- x,   x_k = L_k^-T dinv (w_k - L_k^-1 Ko_k x_{k+1});  after(k, p) runs for every piece
//           k (k < N) as soon as X[k] and X[k+1] are final, with p = powers of r_k.
// TF: node N is free; `meet(y)` completes its right-hand side (after this chain's own Schur term) before L_N^-1 is applied
template <bool TF, int S, int NB, class Rhs, class Meet>
__device__ __forceinline__ void sweep_forward_chain(const Factor<S, NB> &F, int N, int np,
                                                    const double (&rr)[NB], double (&X)[NB + 1][S - 1],
                                                    Rhs &&rhs, Meet &&meet) {
  constexpr int m = S - 1;
#pragma unroll
  for (int k = 0; k <= NB; ++k) {
    if (k <= N) {
      double y[m];
      rhs(k, y);
      if (k > 0) {
        Pw<S> p(rr[k - 1]);
        double v[m];  // D_{k-1}^-1 y_{k-1} = L^-T (dinv . w_{k-1})
#pragma unroll
        for (int j = 0; j < m; ++j) v[j] = X[k - 1][j] * F.dinv[k - 1][j];
        F.solve_LT(k - 1, v);
        Factor<S, NB>::template sub_KoT<TF>(k - 1, N, np, p, v, y);
      }
      if (TF && k == N) meet(y);
      F.solve_L(k, y);
#pragma unroll
      for (int l = 0; l < m; ++l) X[k][l] = y[l];
    }
  }
}

// (TF: `mid(x)` sees -- and may replace -- the solution of node N, the node where the halves of a trajectory meet)
template <bool TF, int S, int NB, class After, class Mid>
__device__ __forceinline__ void sweep_backward_chain(const Factor<S, NB> &F, int N, int np,
                                                     const double (&rr)[NB], double (&X)[NB + 1][S - 1],
                                                     After &&after, Mid &&mid) {
  constexpr int m = S - 1;
#pragma unroll
  for (int k = NB; k >= 0; --k) {
    if (k <= N) {
      double x[m];
#pragma unroll
      for (int l = 0; l < m; ++l) x[l] = X[k][l];
      if (k < N) {
        Pw<S> p(rr[k]);
        double t[m];
        Factor<S, NB>::template mul_Ko<TF>(k, N, np, p, X[k + 1], t);
        F.solve_L(k, t);
#pragma unroll
        for (int l = 0; l < m; ++l) x[l] -= t[l];
      }
#pragma unroll
      for (int l = 0; l < m; ++l) x[l] *= F.dinv[k][l];
      F.solve_LT(k, x);
      if (TF && k == N) mid(x);
#pragma unroll
      for (int l = 0; l < m; ++l) X[k][l] = x[l];
      if (k < N) {
        Pw<S> p(rr[k]);
        after(k, p);
      }
    }
  }
}

template <int S, int NB, class Rhs>
__device__ __forceinline__ void sweep_forward(const Factor<S, NB> &F, int N, int np,
                                              const double (&rr)[NB], double (&X)[NB + 1][S - 1],
                                              Rhs &&rhs) {
  sweep_forward_chain<false, S, NB>(F, N, np, rr, X, rhs, [](double (&)[S - 1]) {});
}

template <int S, int NB, class After>
__device__ __forceinline__ void sweep_backward(const Factor<S, NB> &F, int N, int np,
                                               const double (&rr)[NB], double (&X)[NB + 1][S - 1],
                                               After &&after) {
  sweep_backward_chain<false, S, NB>(F, N, np, rr, X, after, [](double (&)[S - 1]) {});
}

// Right-hand side of the primal problem for one axis: stationarity rows moved to the right,
//   rhs_free = -sum W[free, known] x_known   (known = node positions and pinned end derivatives).
//   P[k]: node positions, hv/tv: pinned head/tail derivatives (orders 1..np).
template <int S, int NB, bool TF = false>
__device__ __forceinline__ void rhs_primal_node(int k, int N, int np, const double (&rr)[NB],
                                                const double (&P)[NB + 1], const double (&hv)[S - 1],
                                                const double (&tv)[S - 1], double (&y)[S - 1]) {
  constexpr int m = S - 1;
  {
    {
#pragma unroll
      for (int l = 0; l < m; ++l) y[l] = 0.0;
      if (k < N) {  // piece k, node k is its start
        Pw<S> p(rr[k]);
        const double dl = P[k + 1] - P[k];
#pragma unroll
        for (int l = 0; l < m; ++l) y[l] = Tab<S>::M[1 + l][0] * p[2 * S - 2 - l] * dl;
        if (!TF && k == N - 1) {  // pinned tail derivatives couple into node N-1 through piece N-1
#pragma unroll
          for (int l = 0; l < m; ++l)
#pragma unroll
            for (int j = 0; j < m; ++j)
              if (j < np)
                y[l] = __builtin_fma(-Tab<S>::M[1 + l][S + 1 + j] * p[2 * S - 3 - l - j], tv[j], y[l]);
        }
      }
      if (k > 0) {  // piece k-1, node k is its end
        Pw<S> p(rr[k - 1]);
        const double dl = P[k] - P[k - 1];
#pragma unroll
        for (int l = 0; l < m; ++l)
          y[l] = __builtin_fma(Tab<S>::M[S + 1 + l][0] * p[2 * S - 2 - l], dl, y[l]);
        if (k == 1) {  // pinned head derivatives couple into node 1 through piece 0
#pragma unroll
          for (int l = 0; l < m; ++l)
#pragma unroll
            for (int j = 0; j < m; ++j)
              if (j < np)
                y[l] = __builtin_fma(-Tab<S>::M[1 + j][S + 1 + l] * p[2 * S - 3 - l - j], hv[j], y[l]);
        }
      }
      if (k == 0 || (!TF && k == N)) {
        // unknowns of an end node also see that node's own pinned derivatives
        if (k == 0) {
          Pw<S> p(rr[0]);
#pragma unroll
          for (int l = 0; l < m; ++l)
#pragma unroll
            for (int j = 0; j < m; ++j)
              if (j < np && l >= np)
                y[l] = __builtin_fma(-Tab<S>::M[1 + l][1 + j] * p[2 * S - 3 - l - j], hv[j], y[l]);
        }
        if (k == N && k > 0) {
          Pw<S> p(rr[k > 0 ? k - 1 : 0]);
#pragma unroll
          for (int l = 0; l < m; ++l)
#pragma unroll
            for (int j = 0; j < m; ++j)
              if (j < np && l >= np)
                y[l] = __builtin_fma(-Tab<S>::M[S + 1 + l][S + 1 + j] * p[2 * S - 3 - l - j], tv[j],
                                     y[l]);
        }
#pragma unroll
        for (int l = 0; l < m; ++l)
          if (l < np) y[l] = (k == 0) ? hv[l] : tv[l];
      }
    }
  }
}

// Coefficients of piece k from its end states, highest power first, and the piece's share of
// int (p^(s))^2:  v_b = x_b r^(S-1-deg b);  g_i = sum_b BHI[i][b] v_b;  c_{S+i} = r^(i+1) g_i;
// piece energy = r g' QB g.  Only non-negative powers of r (p = powers of r_k).
template <int S, class Emit>
__device__ __forceinline__ double emit_piece(int k, const Pw<S> &p, double P0, double P1,
                                             const double (&x0)[S - 1], const double (&x1)[S - 1],
                                             Emit &&emit) {
  double v[2 * S];
  v[0] = P0 * p[S - 1];
  v[S] = P1 * p[S - 1];
#pragma unroll
  for (int j = 1; j < S; ++j) {
    v[j] = x0[j - 1] * p[S - 1 - j];
    v[S + j] = x1[j - 1] * p[S - 1 - j];
  }
  double g[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    // BHI[i][0] = -BHI[i][S]: use the position difference
    double acc = Tab<S>::BHI[i][S] * (v[S] - v[0]);
#pragma unroll
    for (int b = 1; b < S; ++b) {
      acc = __builtin_fma(Tab<S>::BHI[i][b], v[b], acc);
      acc = __builtin_fma(Tab<S>::BHI[i][S + b], v[S + b], acc);
    }
    g[i] = acc;
  }
  double e = 0.0;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    double t = Tab<S>::QB[i][i] * g[i];
#pragma unroll
    for (int j = i + 1; j < S; ++j) t = __builtin_fma(2.0 * Tab<S>::QB[i][j], g[j], t);
    e = __builtin_fma(t, g[i], e);
  }
  double fact = 1.0;
  emit(k, 2 * S - 1, P0);
#pragma unroll
  for (int j = 1; j < S; ++j) {
    fact *= (double)j;
    emit(k, 2 * S - 1 - j, x0[j - 1] * (1.0 / fact));
  }
#pragma unroll
  for (int i = 0; i < S; ++i) emit(k, S - 1 - i, g[i] * p[i + 1]);
  return e * p[1];
}

// One axis of the primal solve: rhs, forward, backward; emits the coefficients piece by piece and
// returns the axis' share of int (p^(s))^2.
template <int S, int NB, class Emit>
__device__ __forceinline__ double solve_axis(const Factor<S, NB> &F, int N, int np,
                                             const double (&P)[NB + 1], const double (&hv)[S - 1],
                                             const double (&tv)[S - 1], double (&X)[NB + 1][S - 1],
                                             Emit &&emit) {
  double rr[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(F.r[i]) : 0.0;
  sweep_forward<S, NB>(F, N, np, rr, X, [&](int k, double (&y)[S - 1]) {
    rhs_primal_node<S, NB>(k, N, np, rr, P, hv, tv, y);
  });
#pragma unroll
  for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(rr[i]) : 0.0;
  double energy = 0.0;
  sweep_backward<S, NB>(F, N, np, rr, X, [&](int k, const Pw<S> &p) {
    energy += emit_piece<S>(k, p, P[k], P[k + 1], X[k], X[k + 1], emit);
  });
  return energy;
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

}  // namespace anet
