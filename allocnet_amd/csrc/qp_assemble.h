// Dense QP assembly kernels in the reference's shapes and row orders.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

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

}  // namespace anet
