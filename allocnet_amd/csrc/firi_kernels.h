// Batched FIRI (fast iterative regional inflation): firi::firi and firi::maxVolInsEllipsoid
// (src/planner/include/gcopter/firi.hpp:159-416; call site sfc_gen.hpp:116-186).  Per corridor:
//   repeat `iterations` times:  polytope for the current ellipsoid (k_firi_planes)  ->  maximum-volume
//   inscribed ellipsoid of that polytope (k_firi_mvie_setup, the batched L-BFGS on costMVIE,
//   k_firi_mvie_finish).
// One workgroup per corridor for the point-cloud stages (the greedy plane selection is a sequence of
// block-wide masked arg-min reductions over the obstacle points), one lane per corridor for the 3x3
// algebra.  All arrays problem-major except the MVIE rows A, which go straight into the batch-minor
// layout k_mvie_eval reads.
//
// Two third-party pieces of the reference are replaced by equivalent exact computations:
//   sdlp::linprog<4> (Seidel's randomised LP, gcopter/sdlp.hpp) -> parallel enumeration of the 4-subsets of
//     constraints (the Chebyshev-centre LP attains its optimum at a vertex; C(nH,4) independent 4x4
//     solves spread over the workgroup, pruned by the best depth found so far);
//   Eigen::JacobiSVD of the lower-triangular 3x3 factor -> cyclic Jacobi on L L'.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

// ellipsoid state per corridor: R (row-major 9), p (3), r (3), interior point of the last LP (3)
constexpr int kFiriEll = 18;

struct FiriArgs {
  const double *bd;    // [B][Mb][4]  rows h.[x;1] <= 0
  const double *pc;    // [B][Np][3]
  const int *npts;     // [B]
  const double *a, *b; // [B][3]
  double *ell;         // [B][kFiriEll]
  double *fpc;         // [B][Np][4] scratch: forward-transformed points + their tangent distance
  int *flag;           // [B][Np] scratch
  double *hpoly;       // [B][H][4]
  int *nh, *ok;        // [B]  ok: 1 running / done (2: an MVIE optimisation hit its budget), 0 a or b outside bd, -1 more than H rows needed
  int64_t B;
  int Mb, Np, H;
  double eps;
  const int *iters = nullptr;  // optional [B]: passes of corridor b (it keeps the polytope of its last pass); nullptr: all passes
  int pass = 0;                // the pass this launch belongs to
};

// a, b inside bd?  initial ellipsoid = unit ball at the midpoint (firi.hpp:279-293)
__global__ void __launch_bounds__(64) k_firi_init(FiriArgs g) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= g.B) return;
  const double *pa = g.a + b * 3, *pb = g.b + b * 3;
  double mx = -1e300;
  for (int r = 0; r < g.Mb; ++r) {
    const double *h = g.bd + (b * g.Mb + r) * 4;
    mx = fmax(mx, h[0] * pa[0] + h[1] * pa[1] + h[2] * pa[2] + h[3]);
    mx = fmax(mx, h[0] * pb[0] + h[1] * pb[1] + h[2] * pb[2] + h[3]);
  }
  g.ok[b] = mx > 0.0 ? 0 : 1;
  g.nh[b] = 0;
  double *e = g.ell + b * kFiriEll;
  for (int i = 0; i < 9; ++i) e[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    e[9 + i] = 0.5 * (pa[i] + pb[i]);
    e[12 + i] = 1.0;
    e[15 + i] = 0.0;
  }
}

__device__ __forceinline__ void argmin_pair(double &d, int &j, double od, int oj) {
  if (od < d || (od == d && oj < j)) { d = od; j = oj; }   // first minimum, like the reference's scan with a strict >
}

// The loop body of firi::firi (firi.hpp:297-405) for the current ellipsoid.
__global__ void __launch_bounds__(256) k_firi_planes(FiriArgs g) {
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ double s_row[4], s_red_d[4], s_fw[9], s_p[3], s_fa[3], s_fb[3];
  __shared__ int s_red_j[4], s_state[4];  // [0] completed, [1] nH, [2] overflow
  if (g.ok[b] < 1) return;
  if (g.iters && g.pass >= max(g.iters[b], 1)) return;  // this corridor has had its passes (counts below 1 count as 1)
  const int N = min(max(g.npts[b], 0), g.Np), M = g.Mb;  // counts beyond the padded capacity are clamped
  const double *E = g.ell + b * kFiriEll;
  const double eps = g.eps;
  double R[9], p[3], r[3];
  for (int i = 0; i < 9; ++i) R[i] = E[i];
  for (int i = 0; i < 3; ++i) { p[i] = E[9 + i]; r[i] = E[12 + i]; }
  // forward = diag(1/r) R',  backward = R diag(r)
  double fw[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) fw[i * 3 + j] = R[j * 3 + i] / r[i];
  const double *pa = g.a + b * 3, *pb = g.b + b * 3;
  double fa[3], fb[3];
  for (int i = 0; i < 3; ++i) {
    fa[i] = fw[i * 3] * (pa[0] - p[0]) + fw[i * 3 + 1] * (pa[1] - p[1]) + fw[i * 3 + 2] * (pa[2] - p[2]);
    fb[i] = fw[i * 3] * (pb[0] - p[0]) + fw[i * 3 + 1] * (pb[1] - p[1]) + fw[i * 3 + 2] * (pb[2] - p[2]);
  }
  double *fpc = g.fpc + b * (int64_t)g.Np * 4;
  int *flag = g.flag + b * (int64_t)g.Np;
  const double *pc = g.pc + b * (int64_t)g.Np * 3;

  // tangent plane of point q (forward space) that keeps a and b inside: firi.hpp:312-341
  auto tangent = [&](const double q[3], double t[4], double &dist) {
    dist = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    t[3] = -dist;
    for (int i = 0; i < 3; ++i) t[i] = q[i] / dist;
    auto toward = [&](const double *f) {
      double d[3] = {q[0] - f[0], q[1] - f[1], q[2] - f[2]};
      const double s = (d[0] * f[0] + d[1] * f[1] + d[2] * f[2]) / (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      for (int i = 0; i < 3; ++i) t[i] = f[i] - s * d[i];
      dist = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      t[3] = -dist;
      for (int i = 0; i < 3; ++i) t[i] /= dist;
    };
    if (t[0] * fa[0] + t[1] * fa[1] + t[2] * fa[2] + t[3] > eps) toward(fa);
    if (t[0] * fb[0] + t[1] * fb[1] + t[2] * fb[2] + t[3] > eps) toward(fb);
    if (t[0] * fa[0] + t[1] * fa[1] + t[2] * fa[2] + t[3] > eps) {
      const double u[3] = {fa[0] - q[0], fa[1] - q[1], fa[2] - q[2]}, v[3] = {fb[0] - q[0], fb[1] - q[1], fb[2] - q[2]};
      double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      for (int i = 0; i < 3; ++i) t[i] = n[i] / nn;
      t[3] = -(t[0] * fa[0] + t[1] * fa[1] + t[2] * fa[2]);
      const double sg = t[3] > 0.0 ? -1.0 : 1.0;
      for (int i = 0; i < 4; ++i) t[i] *= sg;
    }
  };

  // forward points, tangent distances, flags; first arg-min over the points
  double bd_ = 1e300;
  int bj = 0x7fffffff;
  for (int j = tid; j < N; j += 256) {
    double q[3], t[4], dist;
    const double d0 = pc[j * 3] - p[0], d1 = pc[j * 3 + 1] - p[1], d2 = pc[j * 3 + 2] - p[2];
    for (int i = 0; i < 3; ++i) q[i] = fw[i * 3] * d0 + fw[i * 3 + 1] * d1 + fw[i * 3 + 2] * d2;
    tangent(q, t, dist);
    fpc[j * 4] = q[0]; fpc[j * 4 + 1] = q[1]; fpc[j * 4 + 2] = q[2]; fpc[j * 4 + 3] = dist;
    flag[j] = 1;
    argmin_pair(bd_, bj, dist, j);
  }
  auto block_argmin = [&](double d, int j, double &od, int &oj) {
    for (int o = 32; o > 0; o >>= 1) {
      const double d2 = __shfl_xor(d, o);
      const int j2 = __shfl_xor(j, o);
      argmin_pair(d, j, d2, j2);
    }
    __syncthreads();  // previous readers of s_red are done
    if ((tid & 63) == 0) { s_red_d[tid >> 6] = d; s_red_j[tid >> 6] = j; }
    __syncthreads();
    od = s_red_d[0]; oj = s_red_j[0];
    for (int w = 1; w < 4; ++w) argmin_pair(od, oj, s_red_d[w], s_red_j[w]);
  };
  double minR;
  int pcMin;
  block_argmin(bd_, bj, minR, pcMin);
  if (N == 0) minR = INFINITY;

  // boundary planes in forward space (M is small: every thread keeps its own copy of the scan state)
  double *hp = g.hpoly + b * (int64_t)g.H * 4;
  unsigned long long bdmask = (M >= 64) ? ~0ull : ((1ull << M) - 1ull);
  auto bd_forward = [&](int j, double fB[3], double &fD) {
    const double *h = g.bd + (b * M + j) * 4;
    for (int c = 0; c < 3; ++c) fB[c] = h[0] * R[0 * 3 + c] * r[c] + h[1] * R[1 * 3 + c] * r[c] + h[2] * R[2 * 3 + c] * r[c];
    fD = h[3] + h[0] * p[0] + h[1] * p[1] + h[2] * p[2];
  };
  auto bd_dist = [&](int j) {
    double fB[3], fD;
    bd_forward(j, fB, fD);
    return fabs(fD) / sqrt(fB[0] * fB[0] + fB[1] * fB[1] + fB[2] * fB[2]);
  };
  int bdMin = 0;
  double minD = INFINITY;
  for (int j = 0; j < M; ++j) {
    const double d = bd_dist(j);
    if (d < minD) { minD = d; bdMin = j; }
  }
  int nH = 0;
  bool completed = false, overflow = false;
  for (int it = 0; !completed && it < M + N; ++it) {
    if (nH >= g.H) { overflow = true; break; }
    double row[4];
    if (minD < minR) {
      double fB[3], fD;
      bd_forward(bdMin, fB, fD);
      row[0] = fB[0]; row[1] = fB[1]; row[2] = fB[2]; row[3] = fD;
      bdmask &= ~(1ull << bdMin);
    } else {
      // the owner of the point recomputes its tangent plane and publishes it
      __syncthreads();
      if (tid == (pcMin & 255)) {
        double q[3] = {fpc[pcMin * 4], fpc[pcMin * 4 + 1], fpc[pcMin * 4 + 2]}, t[4], dist;
        tangent(q, t, dist);
        for (int i = 0; i < 4; ++i) s_row[i] = t[i];
        flag[pcMin] = 0;
      }
      __syncthreads();
      for (int i = 0; i < 4; ++i) row[i] = s_row[i];
    }
    // hPoly row = forwardH * forward, offset moved back to world coordinates (firi.hpp:393-398)
    if (tid == 0) {
      double h3[3];
      for (int c = 0; c < 3; ++c) h3[c] = row[0] * fw[c] + row[1] * fw[3 + c] + row[2] * fw[6 + c];
      hp[nH * 4] = h3[0]; hp[nH * 4 + 1] = h3[1]; hp[nH * 4 + 2] = h3[2];
      hp[nH * 4 + 3] = row[3] - (h3[0] * p[0] + h3[1] * p[1] + h3[2] * p[2]);
    }
    completed = true;
    minD = INFINITY;
    for (int j = 0; j < M; ++j)
      if (bdmask >> j & 1ull) {
        completed = false;
        const double d = bd_dist(j);
        if (minD > d) { minD = d; bdMin = j; }
      }
    double ld_ = 1e300;
    int lj = 0x7fffffff;
    for (int j = tid; j < N; j += 256)
      if (flag[j]) {
        const double v = row[0] * fpc[j * 4] + row[1] * fpc[j * 4 + 1] + row[2] * fpc[j * 4 + 2] + row[3];
        if (v > -eps) flag[j] = 0;
        else argmin_pair(ld_, lj, fpc[j * 4 + 3], j);
      }
    block_argmin(ld_, lj, minR, pcMin);
    if (pcMin == 0x7fffffff) minR = INFINITY;
    else completed = false;
    ++nH;
  }
  for (int e = nH * 4 + tid; e < g.H * 4; e += 256) hp[e] = 0.0;  // rows of an earlier, larger pass
  if (tid == 0) {
    g.nh[b] = nH;
    if (overflow) g.ok[b] = -1;
  }
}

struct FiriMvieArgs {
  const double *hpoly;  // [B][H][4]
  const int *nh;
  int *ok;              // set to 2 where an MVIE optimisation was cut off by its evaluation budget
  double *ell;          // [B][kFiriEll]
  double *A;            // [3*H][ld] batch-minor rows for k_mvie_eval (zero rows = inactive)
  double *x;            // [9][ld]   L-BFGS variables
  int *is_done, *is_ret;  // L-BFGS state rows: problems without an interior point are marked finished
  int *mvie_ok;         // [B]
  int64_t B, ld;
  int H;
  const int *iters = nullptr;  // as in FiriArgs: no ellipsoid after a corridor's last pass
  int pass = 0;
};

// max t  s.t.  n_q.x + t <= b_q  for the nH rows (n_q, b_q) in LDS (`sm`, four doubles per row): the 4-variable LP of
// sdlp::linprog<4> as firi::maxVolInsEllipsoid (firi.hpp:166-180), geo_utils::findInterior and geo_utils::overlap
// (geo_utils.hpp:43-85) pose it.  A linear programme attains its optimum at a vertex, so the 256 threads of the
// workgroup enumerate the C(nH,4) row subsets (pairs (i, j) over the threads, (k, l) inside), solve each 4 x 4 system
// by a 3 x 3 adjugate, keep the best t seen and scan feasibility only for improving candidates; exact and
// deterministic.  Every thread returns the optimum (-inf when no vertex is feasible); bounded polytopes assumed.
__device__ __forceinline__ void lp_deepest_point(const double *sm, const int nH, const int tid, double (*s_best)[5],
                                                 double &best, double (&bx)[3]) {
  best = -INFINITY;
  bx[0] = bx[1] = bx[2] = 0.0;
  {
    const int P = nH * (nH - 1) / 2;
    for (int pr = tid; pr < P; pr += 256) {
      // unrank the pair (i < j)
      int i = 0, rem = pr;
      while (rem >= nH - 1 - i) { rem -= nH - 1 - i; ++i; }
      const int j = i + 1 + rem;
      for (int k = j + 1; k < nH; ++k)
        for (int l = k + 1; l < nH; ++l) {
          // rows (n_q, 1).(x, t) = b_q for q in {i,j,k,l}: subtract row i, solve the 3x3 for x, then t
          const double *r0 = sm + i * 4, *r1 = sm + j * 4, *r2 = sm + k * 4, *r3 = sm + l * 4;
          const double m[3][3] = {{r1[0] - r0[0], r1[1] - r0[1], r1[2] - r0[2]},
                                  {r2[0] - r0[0], r2[1] - r0[1], r2[2] - r0[2]},
                                  {r3[0] - r0[0], r3[1] - r0[1], r3[2] - r0[2]}};
          const double v[3] = {r1[3] - r0[3], r2[3] - r0[3], r3[3] - r0[3]};
          // x = adj(m) v / det(m)
          const double i00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], i01 = m[0][2] * m[2][1] - m[0][1] * m[2][2],
                       i02 = m[0][1] * m[1][2] - m[0][2] * m[1][1];
          const double i10 = m[1][2] * m[2][0] - m[1][0] * m[2][2], i11 = m[0][0] * m[2][2] - m[0][2] * m[2][0],
                       i12 = m[0][2] * m[1][0] - m[0][0] * m[1][2];
          const double i20 = m[1][0] * m[2][1] - m[1][1] * m[2][0], i21 = m[0][1] * m[2][0] - m[0][0] * m[2][1],
                       i22 = m[0][0] * m[1][1] - m[0][1] * m[1][0];
          const double det = m[0][0] * i00 + m[0][1] * i10 + m[0][2] * i20;
          if (!(fabs(det) > 1e-12)) continue;
          const double id = 1.0 / det;
          double x[3];
          x[0] = (i00 * v[0] + i01 * v[1] + i02 * v[2]) * id;
          x[1] = (i10 * v[0] + i11 * v[1] + i12 * v[2]) * id;
          x[2] = (i20 * v[0] + i21 * v[1] + i22 * v[2]) * id;
          const double t = r0[3] - (r0[0] * x[0] + r0[1] * x[1] + r0[2] * x[2]);
          if (!(t > best)) continue;  // cannot improve: skip the feasibility scan
          bool feas = true;
          for (int q = 0; q < nH && feas; ++q) {
            const double *rq = sm + q * 4;
            feas = rq[0] * x[0] + rq[1] * x[1] + rq[2] * x[2] + t <= rq[3] + 1e-9 * fmax(1.0, fabs(rq[3]));
          }
          if (feas) { best = t; bx[0] = x[0]; bx[1] = x[1]; bx[2] = x[2]; }
        }
    }
  }
  // block arg-max of the depth
  for (int o = 32; o > 0; o >>= 1) {
    const double t2 = __shfl_xor(best, o), x0 = __shfl_xor(bx[0], o), x1 = __shfl_xor(bx[1], o), x2 = __shfl_xor(bx[2], o);
    if (t2 > best) { best = t2; bx[0] = x0; bx[1] = x1; bx[2] = x2; }
  }
  if ((tid & 63) == 0) {
    s_best[tid >> 6][0] = best; s_best[tid >> 6][1] = bx[0]; s_best[tid >> 6][2] = bx[1]; s_best[tid >> 6][3] = bx[2];
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w)
    if (s_best[w][0] > best) { best = s_best[w][0]; bx[0] = s_best[w][1]; bx[1] = s_best[w][2]; bx[2] = s_best[w][3]; }
}

// geo_utils::findInterior / geo_utils::overlap (geo_utils.hpp:43-85), batched: depth[b] = max t s.t. h.x + t <= -h3 over
// the non-zero rows of polytope b (rows h0 x + h1 y + h2 z + h3 <= 0, GCOPTER's raw form; all-zero rows are padding),
// with the rows normalised first (findInterior) or as they are (overlap), and the point that attains it.
// -inf: empty.  sfc_gen::shortCut (sfc_gen.hpp:188-226) is this test on pairs of stacked polytopes.
struct DepthArgs {
  const double *hpoly;  // [B][H][4]
  double *depth;        // [B]
  double *point;        // [B][3] or nullptr
  int64_t B;
  int H, normalise;
  int only_nan;         // k_polytope_depth: only the polytopes whose depth is NaN (left over by the simplex kernel)
};
// The same LP by an active-set (simplex) ascent, ONE LANE per polytope: a few dozen steps of O(rows) each, where the
// enumeration below costs C(rows, 4) candidates (2.3e5 at the 50 rows of two stacked corridor polytopes: milliseconds per
// workgroup).  From v = (x, t) feasible with the constraints W tight: move along p = the projection of the objective onto
// the null space of W until the next constraint blocks, add it; at p = 0 the multipliers of G_W' lam = c decide --
// all >= 0: optimal; else drop the most negative one.  The result is CERTIFIED before it is returned (every row
// satisfied, multipliers non-negative, p = 0); anything else -- iteration cap at a degenerate vertex, a singular Gram
// matrix -- leaves NaN in depth[b], which k_polytope_depth takes as its cue to enumerate that polytope.
// row(i, gi, bi) -> false for a padding row, else g_i = (gi, 1) and the right-hand side bi.  Leaves the point in v and returns
// the certified optimum, +inf (unbounded), -inf (no rows) or NaN (not certified: enumerate).
template <typename RowFn>
__device__ __forceinline__ double lp_ascent(RowFn row, const int H, double (&v)[4]) {
  v[0] = v[1] = v[2] = 0.0;
  v[3] = INFINITY;
  // start: x = 0, t = the smallest right-hand side; that row is tight
  int W[4] = {-1, -1, -1, -1}, nW = 0, nrows = 0;
  for (int i = 0; i < H; ++i) {
    double gi[3], bi;
    if (!row(i, gi, bi)) continue;
    ++nrows;
    if (bi < v[3]) { v[3] = bi; W[0] = i; nW = 1; }
  }
  double depth = NAN;
  if (nrows == 0) {
    depth = -INFINITY;  // nothing but padding (k_polytope_depth's convention)
  } else {
    const int cap = 64 + 4 * nrows;
    for (int it = 0; it < cap; ++it) {
      // Gram matrix of the active rows, its Cholesky factor, lam = (G G')^-1 G c with G c = (1, .., 1)
      double Gw[4][4], bw[4], L[4][4], lam[4];
      for (int k = 0; k < 4; ++k) {
        Gw[k][0] = Gw[k][1] = Gw[k][2] = 0.0; Gw[k][3] = 1.0; bw[k] = 0.0;
        if (k < nW) { double gi[3]; row(W[k], gi, bw[k]); Gw[k][0] = gi[0]; Gw[k][1] = gi[1]; Gw[k][2] = gi[2]; }
      }
      bool singular = false;
      for (int r = 0; r < nW && !singular; ++r)
        for (int c = 0; c <= r; ++c) {
          double acc = Gw[r][0] * Gw[c][0] + Gw[r][1] * Gw[c][1] + Gw[r][2] * Gw[c][2] + 1.0;
          for (int q = 0; q < c; ++q) acc -= L[r][q] * L[c][q];
          if (r == c) {
            const double nr2 = Gw[r][0] * Gw[r][0] + Gw[r][1] * Gw[r][1] + Gw[r][2] * Gw[r][2] + 1.0;
            if (!(acc > 1e-13 * nr2)) { singular = true; break; }
            L[r][r] = sqrt(acc);
          } else {
            L[r][c] = acc / L[c][c];
          }
        }
      if (singular) break;
      for (int r = 0; r < nW; ++r) {
        double acc = 1.0;
        for (int q = 0; q < r; ++q) acc -= L[r][q] * lam[q];
        lam[r] = acc / L[r][r];
      }
      for (int r = nW - 1; r >= 0; --r) {
        double acc = lam[r];
        for (int q = r + 1; q < nW; ++q) acc -= L[q][r] * lam[q];
        lam[r] = acc / L[r][r];
      }
      double p[4] = {0.0, 0.0, 0.0, 1.0};
      for (int k = 0; k < nW; ++k)
        for (int q = 0; q < 4; ++q) p[q] -= lam[k] * Gw[k][q];
      const double pn2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
      if (nW < 4 && pn2 > 1e-24) {
        // ratio test: the first row to become tight along p
        double alpha = INFINITY;
        int blk = -1;
        for (int i = 0; i < H; ++i) {
          if (i == W[0] || i == W[1] || i == W[2] || i == W[3]) continue;
          double gi[3], bi;
          if (!row(i, gi, bi)) continue;
          const double den = gi[0] * p[0] + gi[1] * p[1] + gi[2] * p[2] + p[3];
          if (den > 1e-14 * sqrt(pn2 * (gi[0] * gi[0] + gi[1] * gi[1] + gi[2] * gi[2] + 1.0))) {
            const double a = fmax(0.0, bi - (gi[0] * v[0] + gi[1] * v[1] + gi[2] * v[2] + v[3])) / den;
            if (a < alpha) { alpha = a; blk = i; }
          }
        }
        if (blk < 0) { depth = INFINITY; break; }  // unbounded: nothing blocks the ascent
        for (int q = 0; q < 4; ++q) v[q] += alpha * p[q];
        W[nW++] = blk;
      } else {
        int worst = -1;
        double lmin = -1e-12;
        for (int k = 0; k < nW; ++k)
          if (lam[k] < lmin) { lmin = lam[k]; worst = k; }
        if (worst < 0) {  // candidate optimum: certify it against every row
          bool ok = pn2 <= 1e-18;
          if (nW == 4) {  // a vertex: re-solve it from its four tight rows (the steps accumulate rounding), v = G'(G G')^-1 b
            double mu[4];
            for (int r = 0; r < 4; ++r) {
              double acc = bw[r];
              for (int q = 0; q < r; ++q) acc -= L[r][q] * mu[q];
              mu[r] = acc / L[r][r];
            }
            for (int r = 3; r >= 0; --r) {
              double acc = mu[r];
              for (int q = r + 1; q < 4; ++q) acc -= L[q][r] * mu[q];
              mu[r] = acc / L[r][r];
            }
            for (int q = 0; q < 4; ++q) v[q] = mu[0] * Gw[0][q] + mu[1] * Gw[1][q] + mu[2] * Gw[2][q] + mu[3] * Gw[3][q];
          }
          for (int i = 0; i < H && ok; ++i) {
            double gi[3], bi;
            if (!row(i, gi, bi)) continue;
            const double r = gi[0] * v[0] + gi[1] * v[1] + gi[2] * v[2] + v[3] - bi;
            ok = r <= 1e-9 * fmax(1.0, fabs(bi));
          }
          if (ok) depth = v[3];
          break;
        }
        for (int k = worst; k + 1 < nW; ++k) W[k] = W[k + 1];
        W[--nW] = -1;
      }
    }
  }
  return depth;
}

__global__ void __launch_bounds__(64) k_polytope_depth_simplex(DepthArgs g) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= g.B) return;
  const double *hp = g.hpoly + b * g.H * 4;
  auto row = [&](int i, double (&gi)[3], double &bi) -> bool {
    const double h0 = hp[i * 4], h1 = hp[i * 4 + 1], h2 = hp[i * 4 + 2], h3 = hp[i * 4 + 3];
    if (h0 == 0.0 && h1 == 0.0 && h2 == 0.0) return false;
    const double sc = g.normalise ? 1.0 / sqrt(h0 * h0 + h1 * h1 + h2 * h2) : 1.0;
    gi[0] = h0 * sc; gi[1] = h1 * sc; gi[2] = h2 * sc; bi = -h3 * sc;
    return true;
  };
  double v[4];
  const double depth = lp_ascent(row, g.H, v);
  g.depth[b] = depth;
  if (g.point) { g.point[b * 3] = v[0]; g.point[b * 3 + 1] = v[1]; g.point[b * 3 + 2] = v[2]; }
}

__global__ void __launch_bounds__(256) k_polytope_depth(DepthArgs g) {
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  if (g.only_nan && !isnan(g.depth[b])) return;  // certified by k_polytope_depth_simplex
  extern __shared__ double sm[];  // [H][4] compacted rows
  __shared__ double s_best[4][5];
  __shared__ int s_n;
  if (tid == 0) {  // compact the non-zero rows (H <= a few dozen: serial is fine and keeps the row order)
    int n = 0;
    for (int r = 0; r < g.H; ++r) {
      const double *h = g.hpoly + (b * g.H + r) * 4;
      if (h[0] == 0.0 && h[1] == 0.0 && h[2] == 0.0) continue;
      const double sc = g.normalise ? 1.0 / sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]) : 1.0;
      sm[n * 4] = h[0] * sc; sm[n * 4 + 1] = h[1] * sc; sm[n * 4 + 2] = h[2] * sc; sm[n * 4 + 3] = -h[3] * sc;
      ++n;
    }
    s_n = n;
  }
  __syncthreads();
  double best, bx[3];
  lp_deepest_point(sm, s_n, tid, s_best, best, bx);
  if (tid == 0) {
    g.depth[b] = best;
    if (g.point) { g.point[b * 3] = bx[0]; g.point[b * 3 + 1] = bx[1]; g.point[b * 3 + 2] = bx[2]; }
  }
}

// firi.hpp:166-222: deepest interior point (Chebyshev centre), rows normalised about it, x0 from (R, p, r)
__global__ void __launch_bounds__(256) k_firi_mvie_setup(FiriMvieArgs g) {
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  extern __shared__ double sm[];  // [H][4]: unit normals + offsets
  __shared__ double s_best[4][5];
  const int H = g.H, nH = g.nh[b];
  const bool live = g.ok[b] >= 1 && nH >= 4 && !(g.iters && g.pass + 1 >= max(g.iters[b], 1));
  for (int r = tid; r < nH; r += 256) {
    const double *h = g.hpoly + (b * H + r) * 4;
    const double nrm = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    sm[r * 4] = h[0] / nrm; sm[r * 4 + 1] = h[1] / nrm; sm[r * 4 + 2] = h[2] / nrm;
    sm[r * 4 + 3] = -h[3] / nrm;
  }
  __syncthreads();
  // deepest interior point: the certified ascent on one lane; the vertex enumeration only for what it leaves uncertified
  double best = -INFINITY, bx[3] = {0.0, 0.0, 0.0};
  if (live && tid == 0) {
    auto row = [&](int i, double (&gi)[3], double &bi) -> bool {
      gi[0] = sm[i * 4]; gi[1] = sm[i * 4 + 1]; gi[2] = sm[i * 4 + 2]; bi = sm[i * 4 + 3];
      return true;
    };
    double v[4];
    const double d = lp_ascent(row, nH, v);
    s_best[0][0] = d; s_best[0][1] = v[0]; s_best[0][2] = v[1]; s_best[0][3] = v[2];
  }
  __syncthreads();
  bool certified = false;
  if (live) {
    best = s_best[0][0]; bx[0] = s_best[0][1]; bx[1] = s_best[0][2]; bx[2] = s_best[0][3];
    certified = !isnan(best);
  }
  __syncthreads();  // (s_best is reused by the enumeration)
  if (live && !certified) lp_deepest_point(sm, nH, tid, s_best, best, bx);
  const bool okk = live && best > 0.0 && !isinf(best);
  const int64_t ld = g.ld;
  // MVIE rows about the interior point; zero rows beyond nH (and everywhere for skipped problems)
  for (int r = tid; r < H; r += 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (okk && r < nH) {
      const double *rq = sm + r * 4;
      const double den = rq[3] - (rq[0] * bx[0] + rq[1] * bx[1] + rq[2] * bx[2]);
      a0 = rq[0] / den; a1 = rq[1] / den; a2 = rq[2] / den;
    }
    g.A[(int64_t)r * ld + b] = a0;
    g.A[(int64_t)(H + r) * ld + b] = a1;
    g.A[(int64_t)(2 * H + r) * ld + b] = a2;
  }
  if (tid == 0) {
    double *E = g.ell + b * kFiriEll;
    g.mvie_ok[b] = okk ? 1 : 0;
    double x[9] = {0, 0, 0, 1, 1, 1, 0, 0, 0};
    if (okk) {
      double Q[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double acc = 0.0;
          for (int k = 0; k < 3; ++k) acc += E[i * 3 + k] * (E[12 + k] * E[12 + k]) * E[j * 3 + k];
          Q[i * 3 + j] = acc;
        }
      // chol3d (firi.hpp:45-58)
      const double L00 = sqrt(Q[0]);
      const double L10 = 0.5 * (Q[1] + Q[3]) / L00;
      const double L11 = sqrt(Q[4] - L10 * L10);
      const double L20 = 0.5 * (Q[2] + Q[6]) / L00;
      const double L21 = (0.5 * (Q[5] + Q[7]) - L20 * L10) / L11;
      const double L22 = sqrt(Q[8] - L20 * L20 - L21 * L21);
      for (int i = 0; i < 3; ++i) {
        x[i] = E[9 + i] - bx[i];
        E[15 + i] = bx[i];
      }
      x[3] = sqrt(L00); x[4] = sqrt(L11); x[5] = sqrt(L22);
      x[6] = L10; x[7] = L21; x[8] = L20;
    } else {
      g.is_done[b] = 1;   // maxVolInsEllipsoid returns false before optimising (firi.hpp:182-185): ellipsoid kept
      g.is_ret[b] = 0;
    }
    for (int i = 0; i < 9; ++i) g.x[(int64_t)i * ld + b] = x[i];
  }
}

// firi.hpp:235-265: centre, rotation and radii from the optimiser's variables (SVD of the 3x3 factor)
__global__ void __launch_bounds__(64) k_firi_mvie_finish(FiriMvieArgs g) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= g.B || !g.mvie_ok[b]) return;
  const int64_t ld = g.ld;
  // The reference optimises without an evaluation budget (max_iterations = 0, firi.hpp:212-227).  A corridor whose
  // optimisation was still running at the budget continues from the unfinished iterate, like the reference continues
  // after a negative return code (firi.hpp:229-232), and says so in its ok flag.
  if (!g.is_done[b]) g.ok[b] = 2;
  double x[9];
  for (int i = 0; i < 9; ++i) x[i] = g.x[(int64_t)i * ld + b];
  double *E = g.ell + b * kFiriEll;
  const double L[3][3] = {{x[3] * x[3], 0.0, 0.0}, {x[6], x[4] * x[4], 0.0}, {x[8], x[7], x[5] * x[5]}};
  // L = U S V'  ->  L L' = U S^2 U': cyclic Jacobi on the symmetric 3x3
  double S[3][3], U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i][j] = L[i][0] * L[j][0] + L[i][1] * L[j][1] + L[i][2] * L[j][2];
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
    if (off <= 1e-300 || off <= 1e-17 * (fabs(S[0][0]) + fabs(S[1][1]) + fabs(S[2][2]))) break;
    for (int pq = 0; pq < 3; ++pq) {
      const int pI = pq == 2 ? 1 : 0, qI = pq == 0 ? 1 : 2;
      if (S[pI][qI] == 0.0) continue;
      const double th = (S[qI][qI] - S[pI][pI]) / (2.0 * S[pI][qI]);
      const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) {  // S <- S J
        const double skp = S[k][pI], skq = S[k][qI];
        S[k][pI] = c * skp - s * skq;
        S[k][qI] = s * skp + c * skq;
      }
      for (int k = 0; k < 3; ++k) {  // S <- J' S
        const double spk = S[pI][k], sqk = S[qI][k];
        S[pI][k] = c * spk - s * sqk;
        S[qI][k] = s * spk + c * sqk;
      }
      for (int k = 0; k < 3; ++k) {
        const double ukp = U[k][pI], ukq = U[k][qI];
        U[k][pI] = c * ukp - s * ukq;
        U[k][qI] = s * ukp + c * ukq;
      }
    }
  }
  // singular values descending (JacobiSVD's order), columns of U permuted with them
  int ord[3] = {0, 1, 2};
  double ev[3] = {S[0][0], S[1][1], S[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (ev[ord[j]] < ev[ord[j + 1]]) { const int q = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = q; }
  double Uo[3][3], sv[3];
  for (int c = 0; c < 3; ++c) {
    sv[c] = sqrt(fmax(ev[ord[c]], 0.0));
    for (int k = 0; k < 3; ++k) Uo[k][c] = U[k][ord[c]];
  }
  const double det = Uo[0][0] * (Uo[1][1] * Uo[2][2] - Uo[1][2] * Uo[2][1]) - Uo[0][1] * (Uo[1][0] * Uo[2][2] - Uo[1][2] * Uo[2][0]) +
                     Uo[0][2] * (Uo[1][0] * Uo[2][1] - Uo[1][1] * Uo[2][0]);
  const int c0 = det < 0.0 ? 1 : 0, c1 = det < 0.0 ? 0 : 1;  // firi.hpp:250-259
  for (int k = 0; k < 3; ++k) {
    E[k * 3 + 0] = Uo[k][c0];
    E[k * 3 + 1] = Uo[k][c1];
    E[k * 3 + 2] = Uo[k][2];
  }
  E[12] = sv[c0]; E[13] = sv[c1]; E[14] = sv[2];
  for (int i = 0; i < 3; ++i) E[9 + i] = x[i] + E[15 + i];
}

}  // namespace anet
