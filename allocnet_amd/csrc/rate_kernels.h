// Piece<D>::getMaxVelRate / getMaxAccRate (gcopter/trajectory.hpp:177-273) batched: the maximum of
// ||v(t)|| (or ||a(t)||) over each piece.  The reference squares the normalised-time derivative
// polynomial, differentiates it and isolates the real roots of that derivative in [0,1] with Sturm
// sequences (gcopter/root_finder.hpp:762-1113).  On the GPU one lane owns one (trajectory, piece); the
// roots are isolated by Bernstein-basis subdivision (the variation-diminishing property bounds the number
// of roots of an interval by the sign changes of its Bernstein coefficients, de Casteljau splits an
// interval exactly) and refined by bisection.  Same candidates {0, roots, 1}, same maximum.
// checkMaxVelRate/checkMaxAccRate (:275-314, "no root of ||.||^2 - max^2 in (0,1) and both ends below")
// is that maximum compared with the bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

struct RateArgs {
  const double *coeffs, *T;
  double *out;  // [N][ld] max rate per piece
  int64_t B, ld;
  int N, which;  // which: 1 velocity, 2 acceleration
};

template <int S>
__global__ void __launch_bounds__(64) k_piece_max_rate(RateArgs a) {
  constexpr int D = 2 * S, DEG = D - 1;  // position polynomial degree
  constexpr int MAXM = DEG - 1;          // degree of the velocity polynomial
  constexpr int MAXN = 2 * MAXM - 1;     // degree of d/dtau ||v||^2
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int i = blockIdx.y;
  const int64_t ld = a.ld;
  const double Ti = a.T[(int64_t)i * ld + b];
  const int dsel = a.which;      // 1 or 2
  const int m = DEG - dsel;      // degree of the derivative polynomial in tau
  // u[ax][k]: ascending coefficients of d^dsel P / dtau^dsel, P(tau) = sum c_k T^k tau^k
  double u[3][MAXM + 1];
  {
    double tk[DEG + 1];
    tk[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= DEG; ++k) tk[k] = tk[k - 1] * Ti;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
      for (int k = 0; k <= MAXM; ++k) {
        const int p = k + dsel;  // power of tau in P
        double v = 0.0;
        if (p <= DEG) {
          double f = 1.0;
          for (int e = 0; e < dsel; ++e) f *= (double)(p - e);
          v = f * tk[p] * a.coeffs[(int64_t)((i * 3 + ax) * D + (DEG - p)) * ld + b];
        }
        u[ax][k] = v;
      }
  }
  // q = sum_ax u_ax^2 (degree 2m), ascending
  double q[2 * MAXM + 1];
#pragma unroll
  for (int k = 0; k <= 2 * MAXM; ++k) q[k] = 0.0;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int j = 0; j <= MAXM; ++j)
#pragma unroll
      for (int k = 0; k <= MAXM; ++k) q[j + k] = __builtin_fma(u[ax][j], u[ax][k], q[j + k]);
  auto evalq = [&](double t) {
    double v = 0.0;
#pragma unroll
    for (int k = 2 * MAXM; k >= 0; --k) v = __builtin_fma(v, t, q[k]);
    return v;
  };
  const int n = 2 * m - 1;  // degree of dq
  double best = fmax(evalq(0.0), evalq(1.0));
  double dq[MAXN + 1];
  double nrm = 0.0;
#pragma unroll
  for (int k = 0; k <= MAXN; ++k) {
    dq[k] = (k + 1 <= 2 * MAXM) ? (double)(k + 1) * q[k + 1] : 0.0;
    nrm += dq[k] * dq[k];
  }
  double result;
  if (nrm < 2.220446049250313e-16 || n < 1) {  // DBL_EPSILON: constant rate (trajectory.hpp:189-192)
    result = evalq(0.0);
  } else {
    // Bernstein coefficients of dq on [0,1], degree n: bz[i] = sum_{k<=i} C(i,k)/C(n,k) dq[k]
    double stk[24][MAXN + 1];  // subdivision stack (private memory)
    double slo[24], shi[24];
    int sp = 0;
    {
      for (int ii = 0; ii <= n; ++ii) {
        double acc = 0.0, cik = 1.0, cnk = 1.0;  // C(ii,k), C(n,k)
        for (int k = 0; k <= ii; ++k) {
          acc += cik / cnk * dq[k];
          cik = cik * (double)(ii - k) / (double)(k + 1);
          cnk = cnk * (double)(n - k) / (double)(k + 1);
        }
        stk[0][ii] = acc;
      }
      slo[0] = 0.0;
      shi[0] = 1.0;
      sp = 1;
    }
    int guard = 0;
    while (sp > 0 && guard < 4096) {
      ++guard;
      --sp;
      double bz[MAXN + 1];
      for (int ii = 0; ii <= n; ++ii) bz[ii] = stk[sp][ii];
      const double lo = slo[sp], hi = shi[sp];
      int var = 0, last = 0, first = 0;
      for (int ii = 0; ii <= n; ++ii) {
        const int sg = (bz[ii] > 0.0) - (bz[ii] < 0.0);
        if (sg != 0) {
          if (first == 0) first = sg;
          if (last != 0 && sg != last) ++var;
          last = sg;
        }
      }
      if (var == 0) continue;
      if (var == 1) {
        // exactly one root: bisection on dq; the sign on the left is that of the first non-zero Bernstein
        // coefficient (dq itself may vanish at the end point, e.g. a rest-to-rest piece at tau = 0)
        auto evald = [&](double t) {
          double v = 0.0;
          for (int k = n; k >= 0; --k) v = __builtin_fma(v, t, dq[k]);
          return v;
        };
        double l = lo, h = hi;
        for (int it = 0; it < 60; ++it) {
          const double mid = 0.5 * (l + h), fm = evald(mid);
          const int sm = (fm > 0.0) - (fm < 0.0);
          if (sm == first) l = mid; else h = mid;
        }
        best = fmax(best, evalq(0.5 * (l + h)));
        continue;
      }
      if (hi - lo < 1e-13 || sp >= 22) {  // unresolvable cluster of roots: its location is known well enough
        best = fmax(best, evalq(0.5 * (lo + hi)));
        continue;
      }
      // de Casteljau split at the midpoint: left = diagonal, right = last row
      double left[MAXN + 1], work[MAXN + 1];
      for (int ii = 0; ii <= n; ++ii) work[ii] = bz[ii];
      left[0] = work[0];
      for (int lev = 1; lev <= n; ++lev) {
        for (int ii = 0; ii <= n - lev; ++ii) work[ii] = 0.5 * (work[ii] + work[ii + 1]);
        left[lev] = work[0];
      }
      // after the loop work[0..0] is the split point value; rebuild the right half explicitly
      double right[MAXN + 1];
      for (int ii = 0; ii <= n; ++ii) work[ii] = bz[ii];
      right[n] = work[n];
      for (int lev = 1; lev <= n; ++lev) {
        for (int ii = 0; ii <= n - lev; ++ii) work[ii] = 0.5 * (work[ii] + work[ii + 1]);
        right[n - lev] = work[n - lev];
      }
      const double mid = 0.5 * (lo + hi);
      for (int ii = 0; ii <= n; ++ii) { stk[sp][ii] = left[ii]; stk[sp + 1][ii] = right[ii]; }
      slo[sp] = lo; shi[sp] = mid;
      slo[sp + 1] = mid; shi[sp + 1] = hi;
      sp += 2;
    }
    result = best;
  }
  // q = T^(2 dsel) ||d^dsel p/dt^dsel||^2
  double sc = 1.0 / Ti;
  if (dsel == 2) sc *= sc;
  a.out[(int64_t)i * ld + b] = sqrt(fmax(result, 0.0)) * sc;
}

}  // namespace anet
