// Wide-duration-spread path of the MINCO coefficient solve.
//
// The fast kernels (minco_kernels.h) solve the REDUCED system: node derivatives as unknowns, a Gram form whose
// condition number is about the square of the collocation system's.  With durations spread over a factor of a few
// hundred inside one trajectory that costs the coefficients more digits than the north star's 1e-6 allows (DESIGN.md
// section 2: 5e-4 at a spread of 10^3 for snap), and neither pivoting inside the blocks nor iterative refinement against
// the continuity jumps recovers them -- the residual of the reduced system cannot be evaluated more accurately than
// cond(K) eps either (tests/prototypes, round 2).  The classic formulation can: ONE 2sN x 2sN banded collocation system
// in the monomial basis (boundary rows; per interior knot a waypoint row and 2s-1 continuity rows), banded LU with
// partial pivoting, shared by the three axes -- upstream GCOPTER's BandedSystem shape (minco.hpp is not in the reference
// tree) with the row exchanges it leaves out.
//
// One 64-thread workgroup per trajectory, the band (2sN rows x (3 (3s-1) + 1) columns) and the three right-hand sides
// in LDS.  Only trajectories whose max T / min T exceeds `min_spread` are touched; the others keep what the fast
// kernel wrote.  This path is some 10^3 times slower per trajectory and is meant for the few that need it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

struct DenseSolveArgs {
  const double *head, *tail, *wps, *T;
  double *coeffs, *energy;
  int64_t B, ld;
  int N, c;
  double min_spread;  // <= 1: every trajectory
};

template <int S>
inline size_t minco_dense_lds_bytes(int N) {
  const int n = 2 * S * N, W = 3 * (3 * S - 1) + 1;
  return sizeof(double) * ((size_t)n * W + 3 * (size_t)n + 64);
}

__device__ __forceinline__ double falling_d(int k, int j) {
  double r = 1.0;
  for (int i = 0; i < j; ++i) r *= (double)(k - i);
  return r;
}

template <int S>
__global__ void __launch_bounds__(64) k_minco_solve_dense(DenseSolveArgs a) {
  constexpr int D = 2 * S, kl = 3 * S - 1, ku = 3 * S - 1, W = 2 * kl + ku + 1;
  const int64_t b = blockIdx.x, ld = a.ld;
  const int N = a.N, c = a.c, n = D * N, tid = threadIdx.x;
  extern __shared__ double sm[];
  double *Ab = sm;                       // band: M[i][k] at Ab[i * W + (k - i + kl)], i - kl <= k <= i + kl + ku
  double *R = Ab + (size_t)n * W;        // [n][3]
  double *Ts = R + 3 * (size_t)n;        // [N] durations, then scratch
  __shared__ int piv;
  __shared__ double red[64];
  if (tid < N) Ts[tid] = a.T[(int64_t)tid * ld + b];
  __syncthreads();
  {
    double tmin = Ts[0], tmax = Ts[0];
    for (int i = 1; i < N; ++i) {
      tmin = fmin(tmin, Ts[i]);
      tmax = fmax(tmax, Ts[i]);
    }
    if (!(tmax > a.min_spread * tmin)) return;  // (the whole workgroup takes the same branch)
  }
  for (int e = tid; e < n * W; e += 64) Ab[e] = 0.0;
  for (int e = tid; e < 3 * n; e += 64) R[e] = 0.0;
  __syncthreads();
  auto put = [&](int r, int col0, int k, double v) { Ab[(size_t)r * W + (col0 + k - r + kl)] = v; };
  // j-th derivative of the ascending monomial basis at t, column k
  auto dval = [&](double t, int j, int k) {
    if (k < j) return 0.0;
    double tp = 1.0;
    for (int q = 0; q < k - j; ++q) tp *= t;
    return falling_d(k, j) * tp;
  };
  // rows: [0, S) head; per interior knot i = 1..N-1: S + (i-1) * 2S + {0: waypoint, 1 + j: continuity of order j};
  // the last S rows: tail.  One thread per (row, column of the piece block).
  for (int e = tid; e < n * D; e += 64) {
    const int r = e / D, k = e % D;
    if (r < S) {
      const int j = r;
      put(r, 0, k, dval(0.0, j < c ? j : 2 * S - 1 - j, k));
    } else if (r >= n - S) {
      const int j = r - (n - S);
      put(r, (N - 1) * D, k, dval(Ts[N - 1], j < c ? j : 2 * S - 1 - j, k));
    } else {
      const int q = r - S, i = 1 + q / D, w = q % D;  // interior knot i, row w of its group
      if (w == 0) {
        put(r, (i - 1) * D, k, dval(Ts[i - 1], 0, k));
      } else {
        put(r, (i - 1) * D, k, dval(Ts[i - 1], w - 1, k));
        put(r, i * D, k, -dval(0.0, w - 1, k));
      }
    }
  }
  for (int e = tid; e < 3 * n; e += 64) {
    const int r = e / 3, ax = e % 3;
    double v = 0.0;
    if (r < S) {
      if (r < c) v = a.head[(int64_t)(ax * c + r) * ld + b];
    } else if (r >= n - S) {
      const int j = r - (n - S);
      if (j < c) v = a.tail[(int64_t)(ax * c + j) * ld + b];
    } else {
      const int q = r - S, i = 1 + q / D, w = q % D;
      if (w == 0) v = a.wps[(int64_t)((i - 1) * 3 + ax) * ld + b];
    }
    R[e] = v;
  }
  __syncthreads();
  // banded LU with partial pivoting (fill-in stays inside kl + ku above the diagonal)
  for (int j = 0; j < n; ++j) {
    const int rmax = j + kl < n - 1 ? j + kl : n - 1;
    const int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
    if (tid == 0) {
      int p = j;
      double best = fabs(Ab[(size_t)j * W + kl]);
      for (int i = j + 1; i <= rmax; ++i) {
        const double v = fabs(Ab[(size_t)i * W + (j - i + kl)]);
        if (v > best) { best = v; p = i; }
      }
      piv = p;
    }
    __syncthreads();
    const int p = piv;
    if (p != j) {
      for (int k = j + tid; k <= cmax; k += 64) {
        double *x = &Ab[(size_t)j * W + (k - j + kl)], *y = &Ab[(size_t)p * W + (k - p + kl)];
        const double t = *x; *x = *y; *y = t;
      }
      if (tid < 3) { const double t = R[j * 3 + tid]; R[j * 3 + tid] = R[p * 3 + tid]; R[p * 3 + tid] = t; }
      __syncthreads();
    }
    const double inv = 1.0 / Ab[(size_t)j * W + kl];
    const int nr = rmax - j, nc = cmax - j;        // rows below, columns to the right
    // the multipliers first (they sit in the column that is being eliminated), then the trailing update
    for (int i = tid; i < nr; i += 64) red[i] = Ab[(size_t)(j + 1 + i) * W + (j - (j + 1 + i) + kl)] * inv;
    __syncthreads();
    for (int e = tid; e < nr * (nc + 3); e += 64) {
      const int i = j + 1 + e / (nc + 3), q = e % (nc + 3);
      const double f = red[i - j - 1];
      if (q < nc) {
        const int k = j + 1 + q;
        Ab[(size_t)i * W + (k - i + kl)] -= f * Ab[(size_t)j * W + (k - j + kl)];
      } else {
        R[i * 3 + (q - nc)] -= f * R[j * 3 + (q - nc)];
      }
    }
    __syncthreads();
  }
  if (tid < 3) {
    for (int j = n - 1; j >= 0; --j) {
      const int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
      double v = R[j * 3 + tid];
      for (int k = j + 1; k <= cmax; ++k) v -= Ab[(size_t)j * W + (k - j + kl)] * R[k * 3 + tid];
      R[j * 3 + tid] = v / Ab[(size_t)j * W + kl];
    }
  }
  __syncthreads();
  // unpack: ascending -> highest power first; energy by exact integration of (p^(s))^2
  if (a.coeffs)
    for (int e = tid; e < 3 * n; e += 64) {
      const int i = e / (3 * D), ax = (e / D) % 3, col = e % D, k = D - 1 - col;
      a.coeffs[(int64_t)((i * 3 + ax) * D + col) * ld + b] = R[(i * D + k) * 3 + ax];
    }
  double el = 0.0;
  for (int e = tid; e < 3 * N; e += 64) {
    const int i = e / 3, ax = e % 3;
    double tp[2 * S];
    tp[0] = 1.0;
    for (int k = 1; k < 2 * S; ++k) tp[k] = tp[k - 1] * Ts[i];
    for (int j = S; j < D; ++j)
      for (int k = S; k < D; ++k)
        el += falling_d(j, S) * falling_d(k, S) / (double)(j + k - 2 * S + 1) * tp[j + k - 2 * S + 1] *
              R[(i * D + j) * 3 + ax] * R[(i * D + k) * 3 + ax];
  }
  red[tid] = el;
  __syncthreads();
  if (tid == 0 && a.energy) {
    double e = 0.0;
    for (int q = 0; q < 64; ++q) e += red[q];
    a.energy[b] = e;
  }
}

}  // namespace anet
