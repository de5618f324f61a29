/*
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY -- the inequality QP the reference hands to OSQP
 * (src/planner/include/planner/qp_solver.hpp:119-358: objective blocks :186-236, equality rows :150-183 / :240-243,
 * corridor and box rows :244-296), solved on the host by the STRUCTURED algorithm the GPU kernel uses, restated in
 * scalar C: a block-tridiagonal interior point in Hermite node coordinates.  Only tests/ and bench.py's cpu_baseline
 * leg may load this (bench: `kind: "port"`, the like-for-like CPU figure beside k_qp_ipm; the dense numpy / LAPACK
 * interior point of oracle/qp_np.py stays the high-accuracy oracle and is reported beside it as `dense_numpy`).
 *
 * Formulation.  Unknowns: the node states x_k = (p, p', .., p^(s-1)) of the three axes at the N+1 knots.  A piece is the
 * Hermite interpolant of its two nodes, so the reference's equality block -- C^(s-1) continuity at the interior knots
 * and the start / end position-velocity-acceleration rows -- holds by construction (the pinned components are
 * constants): only the inequality rows are left, and every one of them touches two neighbouring knots.  With
 * v~_b = x_b T^deg(b) (node state in normalised time) a piece has c~ = Phi~ v~ (Phi~ = inverse of the constant 2s x 2s
 * Hermite collocation matrix, computed here by elimination -- not taken from the kernels' tables) and
 *   objective_i = 1/2 T^(1-2s) sum_ax v~' (Phi~_hi' Q_1 Phi~_hi) v~        Q_1 = the reference's cost block at t = 1 (m34 kept)
 *   position at tau_j = j / res :  sum_b H_0[j][b] v~_b;  T v = sum_b H_1[j][b] v~_b;  T^2 a = sum_b H_2[j][b] v~_b
 * so the Newton matrix P + G' W G is SPD block tridiagonal (blocks of 3s) and is assembled from per-sample 3 x 3 /
 * per-axis weights; Mehrotra predictor-corrector (Nocedal & Wright Alg. 16.4), both solves of a step with one banded
 * Cholesky factor.  Pinned by tests/test_qp_port_cpu.py against oracle/qp_np.py on the reference-assembled fixtures.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QMAX_S 4
#define QMAX_D 8

static double fall(int k, int d) { double v = 1.0; for (int q = 0; q < d; ++q) v *= (double)(k - q); return v; }

/* cost block at t = 1 on the s highest coefficients, highest power first (qp_solver.hpp:186-236) */
static void cost_block1(int s, double m34, double Q1[QMAX_S][QMAX_S]) {
  if (s == 4) {
    const double q[4][4] = {{100800, 50400, 20160, 5040}, {50400, 25920, 10800, 2880}, {20160, 10800, 4800, m34}, {5040, 2880, m34, 576}};
    memcpy(Q1, q, sizeof(q));
  } else {
    const double q[3][3] = {{720, 360, 120}, {360, 192, 72}, {120, 72, 36}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Q1[i][j] = q[i][j];
  }
}

typedef struct {
  int s, D, R;
  double Phi[QMAX_D][QMAX_D];  /* c~[col] = sum_b Phi[col][b] v~[b], col = highest power first */
  double Pn[QMAX_D][QMAX_D];   /* Phi_hi' Q1 Phi_hi */
  double *H;                   /* [R][3][D]: Hermite basis and its first two derivatives at tau_j */
} qp_tables;

static int build_tables(int s, int R, double m34, qp_tables *t) {
  const int D = 2 * s;
  t->s = s; t->D = D; t->R = R;
  double E[QMAX_D][2 * QMAX_D];
  for (int e = 0; e < D; ++e) {
    const int d = e % s;
    const double tau = e < s ? 0.0 : 1.0;
    for (int col = 0; col < D; ++col) {
      const int k = D - 1 - col;
      E[e][col] = k >= d ? fall(k, d) * pow(tau, k - d) : 0.0;
      E[e][D + col] = (e == col) ? 1.0 : 0.0;
    }
  }
  for (int c = 0; c < D; ++c) { /* Gauss-Jordan with partial pivoting: [E | I] -> [I | E^-1] */
    int piv = c;
    for (int r = c + 1; r < D; ++r) if (fabs(E[r][c]) > fabs(E[piv][c])) piv = r;
    if (E[piv][c] == 0.0) return -1;
    if (piv != c) for (int j = 0; j < 2 * D; ++j) { double tmp = E[c][j]; E[c][j] = E[piv][j]; E[piv][j] = tmp; }
    const double inv = 1.0 / E[c][c];
    for (int j = 0; j < 2 * D; ++j) E[c][j] *= inv;
    for (int r = 0; r < D; ++r) if (r != c) { const double f = E[r][c]; if (f != 0.0) for (int j = 0; j < 2 * D; ++j) E[r][j] -= f * E[c][j]; }
  }
  for (int col = 0; col < D; ++col) for (int b = 0; b < D; ++b) t->Phi[col][b] = E[col][D + b];
  double Q1[QMAX_S][QMAX_S];
  cost_block1(s, m34, Q1);
  for (int b = 0; b < D; ++b)
    for (int c = 0; c < D; ++c) {
      double acc = 0.0;
      for (int p = 0; p < s; ++p) for (int q = 0; q < s; ++q) acc += t->Phi[p][b] * Q1[p][q] * t->Phi[q][c];
      t->Pn[b][c] = acc;
    }
  t->H = (double *)malloc(sizeof(double) * (size_t)R * 3 * D);
  if (!t->H) return -1;
  for (int j = 0; j < R; ++j) {
    const double tau = (double)j / (double)R;
    for (int d = 0; d < 3; ++d)
      for (int b = 0; b < D; ++b) {
        double acc = 0.0;
        for (int col = 0; col < D; ++col) {
          const int k = D - 1 - col;
          if (k >= d) acc += fall(k, d) * pow(tau, k - d) * t->Phi[col][b];
        }
        t->H[((size_t)j * 3 + d) * D + b] = acc;
      }
  }
  return 0;
}

/* banded Cholesky of the symmetric n x n matrix A (row-major, lower triangle used), half bandwidth bw: A = L L' in place */
static int band_chol(double *A, int n, int bw) {
  for (int j = 0; j < n; ++j) {
    const int lo = j - bw > 0 ? j - bw : 0;
    double d = A[(size_t)j * n + j];
    for (int k = lo; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    const int hi = j + bw < n - 1 ? j + bw : n - 1;
    for (int i = j + 1; i <= hi; ++i) {
      const int l2 = i - bw > lo ? i - bw : lo;
      double v = A[(size_t)i * n + j];
      for (int k = l2; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / d;
    }
  }
  return 0;
}
static void band_solve(const double *L, int n, int bw, double *x) {
  for (int i = 0; i < n; ++i) {
    const int lo = i - bw > 0 ? i - bw : 0;
    double v = x[i];
    for (int k = lo; k < i; ++k) v -= L[(size_t)i * n + k] * x[k];
    x[i] = v / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    const int hi = i + bw < n - 1 ? i + bw : n - 1;
    double v = x[i];
    for (int k = i + 1; k <= hi; ++k) v -= L[(size_t)k * n + i] * x[k];
    x[i] = v / L[(size_t)i * n + i];
  }
}

typedef struct {
  int s, N, R, M;
  double vmax, amax, tol;
  int max_iter;
  const qp_tables *tab;
} qp_cfg;

/* One problem.  state [2][3][3] (start / end, axis, p v a), T [N], hp [N][M][4] (rows a.x <= b, zero rows = padding).
 * coeffs (may be NULL) [N][3][D] highest power first.  Returns 1 solved to tol, 2 solved to 1e-7 only (stalled at the rounding floor of
 * its Newton solve: a checker comparing optima to better than that must leave these out), -2 iteration cap (infeasible or not
 * converged), -1 failure. */
static int qp_ipm_one(const qp_cfg *cf, const double *state, const double *T, const double *hp, double *coeffs, double *obj_out,
                      int *iters_out) {
  const int s = cf->s, D = 2 * s, N = cf->N, R = cf->R, M = cf->M, BK = 3 * s, nv = (N + 1) * BK, bw = 2 * BK - 1;
  const qp_tables *tb = cf->tab;
  const int rows_per_sample = M + 12, m_all = N * R * rows_per_sample;
  double *X = (double *)calloc((size_t)nv * 6 + (size_t)nv * nv + (size_t)m_all * 6 + (size_t)N * D + (size_t)N * R * 9, sizeof(double));
  unsigned char *act = (unsigned char *)calloc((size_t)m_all + nv, 1);
  if (!X || !act) { free(X); free(act); return -1; }
  double *rd = X + nv, *dx = rd + nv, *dx2 = dx + nv, *Px = dx2 + nv, *Gl = Px + nv;
  double *Hm = Gl + nv;
  double *sl = Hm + (size_t)nv * nv, *lam = sl + m_all, *rg = lam + m_all, *hh = rg + m_all, *ds = hh + m_all, *dl = ds + m_all;
  double *sc = dl + m_all;              /* [N][D]: T^deg(b) */
  double *val = sc + (size_t)N * D;     /* [N*R][9]: position, T v, T^2 a of the three axes */
  unsigned char *pinned = act + m_all;
  /* x index of (node k, axis ax, derivative d) */
#define XI(k, ax, d) (((k) * 3 + (ax)) * s + (d))
  for (int ax = 0; ax < 3; ++ax)
    for (int d = 0; d < 3 && d < s; ++d) {
      X[XI(0, ax, d)] = state[(0 * 3 + ax) * 3 + d];
      X[XI(N, ax, d)] = state[(1 * 3 + ax) * 3 + d];
      pinned[XI(0, ax, d)] = pinned[XI(N, ax, d)] = 1;
    }
  for (int k = 1; k < N; ++k) /* start: positions on the chord, derivatives zero */
    for (int ax = 0; ax < 3; ++ax) X[XI(k, ax, 0)] = state[ax * 3] + (state[(3 + ax) * 3] - state[ax * 3]) * (double)k / (double)N;
  for (int i = 0; i < N; ++i) {
    double tp = 1.0;
    for (int d = 0; d < s; ++d) { sc[i * D + d] = sc[i * D + s + d] = tp; tp *= T[i]; }
  }
  /* rows: bounds and which ones exist */
  int m = 0;
  double hs = 1.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < R; ++j) {
      const int base = (i * R + j) * rows_per_sample;
      for (int r = 0; r < M; ++r) {
        const double *row = hp + ((size_t)i * M + r) * 4;
        act[base + r] = (row[0] != 0.0 || row[1] != 0.0 || row[2] != 0.0);
        hh[base + r] = row[3];
      }
      for (int q = 0; q < 12; ++q) { /* +v, -v per axis, then +a, -a per axis */
        act[base + M + q] = 1;
        hh[base + M + q] = q < 6 ? cf->vmax * T[i] : cf->amax * T[i] * T[i];
      }
      for (int r = 0; r < rows_per_sample; ++r) if (act[base + r]) { ++m; if (fabs(hh[base + r]) > hs) hs = fabs(hh[base + r]); }
    }
  /* G x for every row from the sample values */
#define SAMPLE_VALUES()                                                                                   \
  for (int i = 0; i < N; ++i)                                                                             \
    for (int j = 0; j < R; ++j) {                                                                         \
      double *v = val + ((size_t)i * R + j) * 9;                                                          \
      for (int d = 0; d < 3; ++d)                                                                         \
        for (int ax = 0; ax < 3; ++ax) {                                                                  \
          const double *h = tb->H + ((size_t)j * 3 + d) * D;                                              \
          double acc = 0.0;                                                                               \
          for (int b = 0; b < D; ++b) acc += h[b] * sc[i * D + b] * XV[XI(i + (b >= s), ax, b % s)];       \
          v[d * 3 + ax] = acc;                                                                            \
        }                                                                                                 \
    }
#define ROW_VALUE(i, j, r, v) ((r) < M ? hp[((size_t)(i) * M + (r)) * 4] * (v)[0] + hp[((size_t)(i) * M + (r)) * 4 + 1] * (v)[1] + hp[((size_t)(i) * M + (r)) * 4 + 2] * (v)[2] \
                                       : (((r) - M) & 1 ? -1.0 : 1.0) * (v)[(((r) - M) < 6 ? 3 : 6) + (((r) - M) % 6) / 2])
  {
    const double *XV = X;
    SAMPLE_VALUES();
  }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < R; ++j) {
      const double *v = val + ((size_t)i * R + j) * 9;
      const int base = (i * R + j) * rows_per_sample;
      for (int r = 0; r < rows_per_sample; ++r)
        if (act[base + r]) {
          const double gx = ROW_VALUE(i, j, r, v);
          sl[base + r] = fmax(hh[base + r] - gx, 1.0);
          lam[base + r] = 1.0;
        }
    }
  int it, status = -2;
  double best = INFINITY;
  int best_it = 0;
  for (it = 0; it < cf->max_iter; ++it) {
    /* ---- residuals */
    {
      const double *XV = X;
      SAMPLE_VALUES();
    }
    memset(Px, 0, sizeof(double) * nv);
    memset(Gl, 0, sizeof(double) * nv);
    double xPx = 0.0;
    for (int i = 0; i < N; ++i) { /* P x = T^(1-2s) diag(sc) Pn diag(sc) x per axis */
      const double w = pow(T[i], 1 - 2 * s);
      for (int ax = 0; ax < 3; ++ax) {
        double vt[QMAX_D];
        for (int b = 0; b < D; ++b) vt[b] = sc[i * D + b] * X[XI(i + (b >= s), ax, b % s)];
        for (int b = 0; b < D; ++b) {
          double acc = 0.0;
          for (int c = 0; c < D; ++c) acc += tb->Pn[b][c] * vt[c];
          Px[XI(i + (b >= s), ax, b % s)] += w * sc[i * D + b] * acc;
          xPx += w * vt[b] * acc;
        }
      }
    }
    double mu = 0.0, rgmax = 0.0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < R; ++j) {
        const double *v = val + ((size_t)i * R + j) * 9;
        const int base = (i * R + j) * rows_per_sample;
        double gsum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* sum_r lam_r (row as a weight on position / T v / T^2 a per axis) */
        for (int r = 0; r < rows_per_sample; ++r)
          if (act[base + r]) {
            const double gx = ROW_VALUE(i, j, r, v);
            rg[base + r] = gx + sl[base + r] - hh[base + r];
            if (fabs(rg[base + r]) > rgmax) rgmax = fabs(rg[base + r]);
            mu += sl[base + r] * lam[base + r];
            if (r < M) { const double *row = hp + ((size_t)i * M + r) * 4; gsum[0] += lam[base + r] * row[0]; gsum[1] += lam[base + r] * row[1]; gsum[2] += lam[base + r] * row[2]; }
            else { const int q = r - M; gsum[(q < 6 ? 3 : 6) + (q % 6) / 2] += (q & 1 ? -1.0 : 1.0) * lam[base + r]; }
          }
        for (int d = 0; d < 3; ++d)
          for (int ax = 0; ax < 3; ++ax) {
            const double g = gsum[d * 3 + ax];
            if (g == 0.0) continue;
            const double *h = tb->H + ((size_t)j * 3 + d) * D;
            for (int b = 0; b < D; ++b) Gl[XI(i + (b >= s), ax, b % s)] += g * h[b] * sc[i * D + b];
          }
      }
    mu /= (double)m;
    double rdmax = 0.0, pmax = 1.0, gmax = 1.0;
    for (int q = 0; q < nv; ++q) {
      rd[q] = pinned[q] ? 0.0 : Px[q] + Gl[q];
      if (fabs(rd[q]) > rdmax) rdmax = fabs(rd[q]);
      if (!pinned[q]) { if (fabs(Px[q]) > pmax) pmax = fabs(Px[q]); if (fabs(Gl[q]) > gmax) gmax = fabs(Gl[q]); }
    }
    const double merit = fmax(fmax(rdmax / fmax(pmax, gmax), rgmax / hs), mu / fmax(1.0, fabs(xPx)));
    if (merit <= cf->tol) { status = 1; break; }
    if (merit < best) { best = merit; best_it = it; }
    else if (it - best_it >= 15) break;
    if (!isfinite(merit)) break;
    /* ---- Newton matrix P + G' W G (lower triangle; pinned rows / columns = identity) */
    memset(Hm, 0, sizeof(double) * (size_t)nv * nv);
    for (int i = 0; i < N; ++i) {
      const double w = pow(T[i], 1 - 2 * s);
      for (int ax = 0; ax < 3; ++ax)
        for (int b = 0; b < D; ++b)
          for (int c = 0; c < D; ++c) {
            const int p = XI(i + (b >= s), ax, b % s), q = XI(i + (c >= s), ax, c % s);
            if (p >= q) Hm[(size_t)p * nv + q] += w * sc[i * D + b] * tb->Pn[b][c] * sc[i * D + c];
          }
      for (int j = 0; j < R; ++j) {
        const int base = (i * R + j) * rows_per_sample;
        double A3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, wv[3] = {0, 0, 0}, wa[3] = {0, 0, 0};
        for (int r = 0; r < rows_per_sample; ++r)
          if (act[base + r]) {
            const double w_r = lam[base + r] / sl[base + r];
            if (r < M) {
              const double *row = hp + ((size_t)i * M + r) * 4;
              for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) A3[p][q] += w_r * row[p] * row[q];
            } else {
              const int q = r - M;
              (q < 6 ? wv : wa)[(q % 6) / 2] += w_r;
            }
          }
        const double *h0 = tb->H + ((size_t)j * 3 + 0) * D, *h1 = h0 + D, *h2 = h1 + D;
        for (int ax = 0; ax < 3; ++ax)
          for (int ay = 0; ay < 3; ++ay)
            for (int b = 0; b < D; ++b)
              for (int c = 0; c < D; ++c) {
                const int p = XI(i + (b >= s), ax, b % s), q = XI(i + (c >= s), ay, c % s);
                if (p < q) continue;
                double v = A3[ax][ay] * h0[b] * h0[c];
                if (ax == ay) v += wv[ax] * h1[b] * h1[c] + wa[ax] * h2[b] * h2[c];
                Hm[(size_t)p * nv + q] += v * sc[i * D + b] * sc[i * D + c];
              }
      }
    }
    for (int q = 0; q < nv; ++q)
      if (pinned[q]) {
        for (int p = 0; p < nv; ++p) Hm[(size_t)q * nv + p] = Hm[(size_t)p * nv + q] = 0.0;
        Hm[(size_t)q * nv + q] = 1.0;
      }
    { /* a tiny diagonal shift keeps the factor alive when slacks collapse (oracle/qp_np.py regularises the same way) */
      double dmax = 1.0;
      for (int q = 0; q < nv; ++q) if (Hm[(size_t)q * nv + q] > dmax) dmax = Hm[(size_t)q * nv + q];
      for (int q = 0; q < nv; ++q) if (!pinned[q]) Hm[(size_t)q * nv + q] += 1e-14 * dmax;
    }
    if (band_chol(Hm, nv, bw)) { status = -1; break; }
    /* ---- predictor and corrector with the one factor (oracle/qp_np.py: step(rc)) */
    double sig_mu = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      /* rhs = -rd - G'((lam rg - rc) / s), rc = s lam (+ ds dl - sigma mu) */
      double *dd = pass ? dx2 : dx;
      for (int q = 0; q < nv; ++q) dd[q] = -rd[q];
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < R; ++j) {
          const int base = (i * R + j) * rows_per_sample;
          double gsum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int r = 0; r < rows_per_sample; ++r)
            if (act[base + r]) {
              const double rc = sl[base + r] * lam[base + r] + (pass ? ds[base + r] * dl[base + r] - sig_mu : 0.0);
              const double t = -(lam[base + r] * rg[base + r] - rc) / sl[base + r];
              if (r < M) { const double *row = hp + ((size_t)i * M + r) * 4; gsum[0] += t * row[0]; gsum[1] += t * row[1]; gsum[2] += t * row[2]; }
              else { const int q = r - M; gsum[(q < 6 ? 3 : 6) + (q % 6) / 2] += (q & 1 ? -1.0 : 1.0) * t; }
            }
          for (int d = 0; d < 3; ++d)
            for (int ax = 0; ax < 3; ++ax) {
              const double g = gsum[d * 3 + ax];
              if (g == 0.0) continue;
              const double *h = tb->H + ((size_t)j * 3 + d) * D;
              for (int b = 0; b < D; ++b) dd[XI(i + (b >= s), ax, b % s)] += g * h[b] * sc[i * D + b];
            }
        }
      for (int q = 0; q < nv; ++q) if (pinned[q]) dd[q] = 0.0;
      band_solve(Hm, nv, bw, dd);
      /* ds = -rg - G dx ; dl = (-rc - lam ds) / s ; step lengths */
      {
        const double *XV = dd;
        SAMPLE_VALUES();
      }
      double as = 1.0, al = 1.0, num = 0.0;
      const double frac = pass ? 0.995 : 1.0;
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < R; ++j) {
          const double *v = val + ((size_t)i * R + j) * 9;
          const int base = (i * R + j) * rows_per_sample;
          for (int r = 0; r < rows_per_sample; ++r)
            if (act[base + r]) {
              const double gdx = ROW_VALUE(i, j, r, v);
              const double rc = sl[base + r] * lam[base + r] + (pass ? ds[base + r] * dl[base + r] - sig_mu : 0.0);
              const double nds = -rg[base + r] - gdx;
              const double ndl = (-rc - lam[base + r] * nds) / sl[base + r];
              ds[base + r] = nds; /* (the predictor's ds dl of THIS row went into rc above: overwriting is safe) */
              dl[base + r] = ndl;
              if (nds < 0.0) { const double q = -frac * sl[base + r] / nds; if (q < as) as = q; }
              if (ndl < 0.0) { const double q = -frac * lam[base + r] / ndl; if (q < al) al = q; }
            }
        }
      const double a = fmin(1.0, fmin(as, al));
      if (!pass) {
        for (int q = 0; q < m_all; ++q) if (act[q]) num += (sl[q] + a * ds[q]) * (lam[q] + a * dl[q]);
        const double mu_aff = num / (double)m, ratio = mu_aff / mu;
        sig_mu = ratio * ratio * ratio * mu;
      } else {
        for (int q = 0; q < nv; ++q) if (!pinned[q]) X[q] += a * dd[q];
        for (int q = 0; q < m_all; ++q) if (act[q]) { sl[q] += a * ds[q]; lam[q] += a * dl[q]; }
      }
    }
  }
  /* objective and coefficients from the node states */
  double obj = 0.0;
  for (int i = 0; i < N; ++i) {
    const double w = pow(T[i], 1 - 2 * s);
    for (int ax = 0; ax < 3; ++ax) {
      double vt[QMAX_D];
      for (int b = 0; b < D; ++b) vt[b] = sc[i * D + b] * X[XI(i + (b >= s), ax, b % s)];
      for (int b = 0; b < D; ++b) for (int c = 0; c < D; ++c) obj += 0.5 * w * vt[b] * tb->Pn[b][c] * vt[c];
      if (coeffs) {
        double tk = 1.0; /* c_k = c~_k / T^k, col = D-1-k */
        for (int col = D - 1; col >= 0; --col) {
          double acc = 0.0;
          for (int b = 0; b < D; ++b) acc += tb->Phi[col][b] * vt[b];
          coeffs[((size_t)i * 3 + ax) * D + col] = acc / tk;
          tk *= T[i];
        }
      }
    }
  }
  if (status == -2 && best <= 1e-7) status = 2; /* stalled at its rounding floor above tol: solved, to 1e-7 only (oracle/qp_np.py accepts the same) */
  *obj_out = obj;
  *iters_out = it;
  free(X);
  free(act);
  return status;
#undef XI
#undef SAMPLE_VALUES
#undef ROW_VALUE
}

typedef struct {
  qp_cfg cf;
  int64_t B, lo, hi;
  const double *state, *T, *hp;
  double *coeffs, *obj;
  int *status, *iters;
} qp_job;

static void *qp_worker(void *arg) {
  qp_job *j = (qp_job *)arg;
  const int N = j->cf.N, M = j->cf.M, D = 2 * j->cf.s;
  for (int64_t b = j->lo; b < j->hi; ++b)
    j->status[b] = qp_ipm_one(&j->cf, j->state + b * 18, j->T + b * N, j->hp + b * (int64_t)N * M * 4,
                              j->coeffs ? j->coeffs + b * (int64_t)N * 3 * D : NULL, j->obj + b, j->iters + b);
  return NULL;
}

/* Batch driver, one problem per task, contiguous ranges per thread.  state [B][2][3][3], T [B][N], hpolys [B][N][M][4];
 * coeffs [B][N][3][2s] or NULL, obj [B], status [B] (1 solved, 2 solved to 1e-7 only, -2 not converged / infeasible, -1 failure), iters [B]. */
int oracle_qp_ipm_batch(int s, int N, int R, int M, int64_t B, const double *state, const double *T, const double *hpolys,
                        double vmax, double amax, double m34, double tol, int max_iter, double *coeffs, double *obj,
                        int *status, int *iters, int nthreads) {
  if ((s != 3 && s != 4) || N < 1 || R < 1 || M < 0 || B < 0) return -1;
  qp_tables tab;
  if (build_tables(s, R, m34, &tab)) return -1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if ((int64_t)nthreads > B) nthreads = B > 0 ? (int)B : 1;
  pthread_t th[256];
  qp_job jobs[256];
  for (int t = 0; t < nthreads; ++t) {
    qp_job *j = &jobs[t];
    j->cf.s = s; j->cf.N = N; j->cf.R = R; j->cf.M = M; j->cf.vmax = vmax; j->cf.amax = amax; j->cf.tol = tol;
    j->cf.max_iter = max_iter; j->cf.tab = &tab;
    j->B = B; j->lo = B * t / nthreads; j->hi = B * (t + 1) / nthreads;
    j->state = state; j->T = T; j->hp = hpolys; j->coeffs = coeffs; j->obj = obj; j->status = status; j->iters = iters;
    if (nthreads == 1) qp_worker(j);
    else if (pthread_create(&th[t], NULL, qp_worker, j)) { for (int q = 0; q < t; ++q) pthread_join(th[q], NULL); free(tab.H); return -1; }
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(tab.H);
  return 0;
}
