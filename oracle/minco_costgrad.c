/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the trajectory cost + analytic gradient and of the
 * L-BFGS loop around it, for (1) parity checks at sizes the numpy oracle is too slow for and (2) bench.py's
 * cpu_baseline legs of BASELINE configs[2] / configs[3].  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.
 *
 * What it restates
 *   oracle_minco_cost_grad : J(P, T) = int (p^(s))^2 + rho sum T + J_pen and its total gradient w.r.t. the interior
 *     waypoints and the durations, the classic MINCO way: ONE 2sN x 2sN banded collocation system M c = b factorised by
 *     banded LU with partial pivoting (as oracle_minco_solve), partial gradients dJ/dc, dJ/dT, then the adjoint
 *     M' lam = dJ/dc through the SAME factors, gradP_k = lam[waypoint row k], gradT_i = dJ/dT_i - lam'(dM/dT_i)c.
 *     That is the shape of upstream GCOPTER's minco.hpp (setParameters / getEnergyPartialGradBy* / propogateGrad with
 *     BandedSystem::solve / solveAdj), which is NOT in /root/reference: PARITY UNPINNED against
 *     the reference itself; pinned against oracle/minco_np.py (dense numpy adjoint, tests/test_oracle_cpu.py) and by
 *     finite differences.  J_pen is the smoothed-L1 (firi.hpp:60-84) penalty on exactly the rows of the reference's
 *     inequality block (qp_solver.hpp:244-296 / min_traj_opt.py:535-613): corridor rows a.p <= b and +-v, +-a box rows
 *     at t = j T_i / res, quadrature weight T_i / res.
 *   oracle_lbfgs_minco_batch : oracle_lbfgs_optimize (lbfgs.hpp:434-717 restated in lbfgs_oracle.c) driving that
 *     objective in the variables x = [waypoints, tau], T = forward_T(tau), one problem per task, tasks pulled by
 *     `nthreads` host threads from a shared counter.
 *
 * Layout: trajectory-major.  head/tail 3 x c, wps (N-1) x 3, T N, hpolys N x M x 4 (rows a.x <= b, zero rows = padding),
 * coeffs N x 3 x D highest power first.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define MAXN 16
#define MAXDIM (8 * MAXN)

typedef struct {
  double rho, wc, wv, wa, mu, vmax, amax;
  int res, M;
} oracle_penalty;

typedef struct {
  int mem_size;
  double g_epsilon;
  int past;
  double delta;
  int max_iterations;
  int max_linesearch;
  double min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
} oracle_lbfgs_param;
typedef double (*oracle_eval_t)(void *instance, const double *x, double *g, int n);
typedef double (*oracle_stepbound_t)(void *instance, const double *xp, const double *d, int n);
int oracle_lbfgs_optimize_sb(int n, double *x, double *f, oracle_eval_t eval, oracle_stepbound_t stepbound, void *inst,
                             const oracle_lbfgs_param *param, int *iters, int *evals_out);

/* k!/(k-j)! for k < 8, j <= 8 (0 when j > k) */
static const double FALL[8][9] = {
    {1, 0, 0, 0, 0, 0, 0, 0, 0},          {1, 1, 0, 0, 0, 0, 0, 0, 0},           {1, 2, 2, 0, 0, 0, 0, 0, 0},
    {1, 3, 6, 6, 0, 0, 0, 0, 0},          {1, 4, 12, 24, 24, 0, 0, 0, 0},        {1, 5, 20, 60, 120, 120, 0, 0, 0},
    {1, 6, 30, 120, 360, 720, 720, 0, 0}, {1, 7, 42, 210, 840, 2520, 5040, 5040, 0}};
static inline double falling(int k, int j) { return FALL[k][j]; }

/* j-th derivative of the ascending monomial basis at t */
static void drow(int D, double t, int j, double *row) {
  for (int k = 0; k < D; ++k) row[k] = 0.0;
  double tp = 1.0;
  for (int k = j; k < D; ++k) {
    row[k] = falling(k, j) * tp;
    tp *= t;
  }
}

/* firi::smoothedL1 (firi.hpp:60-84); 0 below 0 (the reference returns false there and the caller skips the term) */
static inline void sl1(double mu, double x, double *f, double *df) {
  if (x < 0.0) {
    *f = 0.0; *df = 0.0;
  } else if (x > mu) {
    *f = x - 0.5 * mu; *df = 1.0;
  } else {
    const double xd = x / mu, sq = xd * xd, mm = mu - 0.5 * x;
    *f = mm * sq * xd;
    *df = sq * (-0.5 * xd + 3.0 * mm / mu);
  }
}

typedef struct {
  double M[MAXDIM][MAXDIM]; /* LU factors: U on and above the diagonal, the multipliers of L below */
  double R[MAXDIM][3];      /* right-hand sides -> ascending coefficients; later the adjoint */
  double G[MAXDIM][3];
  int piv[MAXDIM];
} ws_t;

/* cost, gradP (N-1) x 3, gradT N of one trajectory; coeffs (may be NULL) N x 3 x D.  Returns 0, or < 0 on a bad
 * argument / singular system. */
int oracle_minco_cost_grad(int s, int c, int N, const double *head, const double *tail, const double *wps, const double *T,
                           const double *hpolys, const oracle_penalty *pp, double *cost, double *gradP, double *gradT,
                           double *coeffs) {
  const int D = 2 * s, n = D * N;
  const int kl = 3 * s - 1, ku = 3 * s - 1;
  if (s < 2 || s > 4 || c < 1 || c > s || N < 1 || N > MAXN) return -1;
  static __thread ws_t *w = NULL;
  if (!w) w = (ws_t *)malloc(sizeof(ws_t));
  for (int i = 0; i < n; ++i) {
    memset(w->M[i], 0, sizeof(double) * n);
    w->R[i][0] = w->R[i][1] = w->R[i][2] = 0.0;
  }
  double row[8];
  int r = 0;
  for (int j = 0; j < s; ++j, ++r) {
    if (j < c) {
      drow(D, 0.0, j, row);
      for (int a = 0; a < 3; ++a) w->R[r][a] = head[a * c + j];
    } else {
      drow(D, 0.0, 2 * s - 1 - j, row);
    }
    memcpy(&w->M[r][0], row, sizeof(double) * D);
  }
  for (int i = 1; i < N; ++i) {
    const int cl = (i - 1) * D, cr = i * D;
    drow(D, T[i - 1], 0, row);
    memcpy(&w->M[r][cl], row, sizeof(double) * D);
    for (int a = 0; a < 3; ++a) w->R[r][a] = wps[(i - 1) * 3 + a];
    ++r;
    for (int j = 0; j < 2 * s - 1; ++j, ++r) {
      drow(D, T[i - 1], j, row);
      memcpy(&w->M[r][cl], row, sizeof(double) * D);
      drow(D, 0.0, j, row);
      for (int k = 0; k < D; ++k) w->M[r][cr + k] = -row[k];
    }
  }
  {
    const int cl = (N - 1) * D;
    for (int j = 0; j < s; ++j, ++r) {
      if (j < c) {
        drow(D, T[N - 1], j, row);
        for (int a = 0; a < 3; ++a) w->R[r][a] = tail[a * c + j];
      } else {
        drow(D, T[N - 1], 2 * s - 1 - j, row);
      }
      memcpy(&w->M[r][cl], row, sizeof(double) * D);
    }
  }
  /* banded LU, partial pivoting; row interchanges applied to the trailing part only and replayed on the right-hand
   * sides (the multipliers stay where they were computed) */
  for (int j = 0; j < n; ++j) {
    const int rmax = j + kl < n - 1 ? j + kl : n - 1;
    const int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
    int p = j;
    double best = fabs(w->M[j][j]);
    for (int i = j + 1; i <= rmax; ++i)
      if (fabs(w->M[i][j]) > best) { best = fabs(w->M[i][j]); p = i; }
    if (best == 0.0) return -2;
    w->piv[j] = p;
    if (p != j) {
      for (int k = j; k <= cmax; ++k) { double t = w->M[j][k]; w->M[j][k] = w->M[p][k]; w->M[p][k] = t; }
      for (int a = 0; a < 3; ++a) { double t = w->R[j][a]; w->R[j][a] = w->R[p][a]; w->R[p][a] = t; }
    }
    const double inv = 1.0 / w->M[j][j];
    for (int i = j + 1; i <= rmax; ++i) {
      const double f = w->M[i][j] * inv;
      w->M[i][j] = f;
      if (f == 0.0) continue;
      for (int k = j + 1; k <= cmax; ++k) w->M[i][k] -= f * w->M[j][k];
      w->R[i][0] -= f * w->R[j][0]; w->R[i][1] -= f * w->R[j][1]; w->R[i][2] -= f * w->R[j][2];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    const int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
    for (int a = 0; a < 3; ++a) {
      double v = w->R[j][a];
      for (int k = j + 1; k <= cmax; ++k) v -= w->M[j][k] * w->R[k][a];
      w->R[j][a] = v / w->M[j][j];
    }
  }
  /* ---- cost and partial gradients (ascending coefficient order, G = dJ/dc) ---- */
  double J = 0.0;
  double gdT[MAXN];
  for (int i = 0; i < N; ++i) {
    const double Ti = T[i];
    double tp[8];
    tp[0] = 1.0;
    for (int k = 1; k < 8; ++k) tp[k] = tp[k - 1] * Ti;
    double gT = pp ? pp->rho : 0.0;
    J += (pp ? pp->rho : 0.0) * Ti;
    for (int a = 0; a < 3; ++a) {
      for (int k = 0; k < D; ++k) {
        w->G[i * D + k][a] = 0.0;
        if (coeffs) coeffs[(i * 3 + a) * D + (D - 1 - k)] = w->R[i * D + k][a];
      }
      /* energy: sum_{j,k>=s} f_j f_k T^(j+k-2s+1)/(j+k-2s+1) c_j c_k ; d/dT = (p^(s)(T))^2 */
      double ps = 0.0;
      for (int j = s; j < D; ++j) {
        const double fj = falling(j, s);
        ps += fj * tp[j - s] * w->R[i * D + j][a];
        double acc = 0.0;
        for (int k = s; k < D; ++k)
          acc += fj * falling(k, s) / (double)(j + k - 2 * s + 1) * tp[j + k - 2 * s + 1] * w->R[i * D + k][a];
        J += acc * w->R[i * D + j][a];
        w->G[i * D + j][a] += 2.0 * acc;
      }
      gT += ps * ps;
    }
    if (pp && pp->res > 0) {
      const double step = Ti / (double)pp->res;
      const double *hp = hpolys ? hpolys + (size_t)i * pp->M * 4 : NULL;
      for (int js = 0; js < pp->res; ++js) {
        const double t = js * step;
        double b0[8], b1[8], b2[8], b3[8];
        drow(D, t, 0, b0); drow(D, t, 1, b1); drow(D, t, 2, b2); drow(D, t, 3, b3);
        double p[3], v[3], ac[3], jr[3];
        for (int a = 0; a < 3; ++a) {
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          for (int k = 0; k < D; ++k) {
            const double ck = w->R[i * D + k][a];
            s0 += ck * b0[k]; s1 += ck * b1[k]; s2 += ck * b2[k]; s3 += ck * b3[k];
          }
          p[a] = s0; v[a] = s1; ac[a] = s2; jr[a] = s3;
        }
        double cst = 0.0, gp[3] = {0, 0, 0}, gv[3] = {0, 0, 0}, ga[3] = {0, 0, 0};
        if (hp)
          for (int q = 0; q < pp->M; ++q) {
            const double *h = hp + q * 4;
            double f, df;
            sl1(pp->mu, h[0] * p[0] + h[1] * p[1] + h[2] * p[2] - h[3], &f, &df);
            cst += pp->wc * f;
            gp[0] += pp->wc * df * h[0]; gp[1] += pp->wc * df * h[1]; gp[2] += pp->wc * df * h[2];
          }
        for (int a = 0; a < 3; ++a)
          for (int sg = 0; sg < 2; ++sg) {
            const double sgn = sg ? -1.0 : 1.0;
            double f, df;
            sl1(pp->mu, sgn * v[a] - pp->vmax, &f, &df);
            cst += pp->wv * f; gv[a] += pp->wv * sgn * df;
            sl1(pp->mu, sgn * ac[a] - pp->amax, &f, &df);
            cst += pp->wa * f; ga[a] += pp->wa * sgn * df;
          }
        if (cst == 0.0) continue;
        J += step * cst;
        double dot = 0.0;
        for (int a = 0; a < 3; ++a) {
          for (int k = 0; k < D; ++k) w->G[i * D + k][a] += step * (gp[a] * b0[k] + gv[a] * b1[k] + ga[a] * b2[k]);
          dot += gp[a] * v[a] + gv[a] * ac[a] + ga[a] * jr[a];
        }
        gT += cst / (double)pp->res + step * dot * ((double)js / (double)pp->res);
      }
    }
    gdT[i] = gT;
  }
  if (cost) *cost = J;
  if (!gradP && !gradT) return 0;
  /* ---- adjoint  M' lam = G  through the factors:  U' y = G,  then L' and the interchanges in reverse ---- */
  for (int j = 0; j < n; ++j) {
    const int k0 = j - kl - ku > 0 ? j - kl - ku : 0;
    for (int a = 0; a < 3; ++a) {
      double v = w->G[j][a];
      for (int k = k0; k < j; ++k) v -= w->M[k][j] * w->G[k][a];
      w->G[j][a] = v / w->M[j][j];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    const int rmax = j + kl < n - 1 ? j + kl : n - 1;
    for (int a = 0; a < 3; ++a) {
      double v = w->G[j][a];
      for (int i = j + 1; i <= rmax; ++i) v -= w->M[i][j] * w->G[i][a];
      w->G[j][a] = v;
    }
    const int p = w->piv[j];
    if (p != j)
      for (int a = 0; a < 3; ++a) { double t = w->G[j][a]; w->G[j][a] = w->G[p][a]; w->G[p][a] = t; }
  }
  /* rows of M that depend on T_i evaluate piece i at T_i; d/dT of the j-th derivative row is the (j+1)-th */
  r = s;
  for (int i = 1; i <= N; ++i) {
    const int pi = i - 1;
    int djs[8], nd;
    if (i < N) {
      if (gradP)
        for (int a = 0; a < 3; ++a) gradP[(i - 1) * 3 + a] = w->G[r][a];
      djs[0] = 0;
      for (int j = 0; j < 2 * s - 1; ++j) djs[1 + j] = j;
      nd = 2 * s;
    } else {
      for (int j = 0; j < s; ++j) djs[j] = j < c ? j : 2 * s - 1 - j;
      nd = s;
    }
    for (int q = 0; q < nd; ++q) {
      drow(D, T[pi], djs[q] + 1, row);
      for (int a = 0; a < 3; ++a) {
        double val = 0.0;
        for (int k = 0; k < D; ++k) val += row[k] * w->R[pi * D + k][a];
        gdT[pi] -= w->G[r + q][a] * val;
      }
    }
    r += nd;
  }
  if (gradT)
    for (int i = 0; i < N; ++i) gradT[i] = gdT[i];
  return 0;
}

/* ---- the duration map of the optimiser's variables: T = forward_T(tau) (the quadratic / reciprocal-quadratic map of
 * DESIGN.md section 5; C2, T > 0 for every tau) ---- */
static double fwd_T(double tau) { return tau > 0.0 ? (0.5 * tau + 1.0) * tau + 1.0 : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
static double dfwd_T(double tau) {
  if (tau > 0.0) return tau + 1.0;
  const double den = (0.5 * tau - 1.0) * tau + 1.0;
  return (1.0 - tau) / (den * den);
}
static double bwd_T(double T) { return T > 1.0 ? sqrt(2.0 * T - 1.0) - 1.0 : 1.0 - sqrt(2.0 / T - 1.0); }

typedef struct {
  int s, c, N;
  const double *head, *tail, *hpolys;
  const oracle_penalty *pp;
  int rc;
  double tau_min; /* backward_T(minimum duration) of the step bound */
} obj_t;

/* lbfgs_stepbound_t (lbfgs.hpp:221-224) for a minimum duration: the largest step along d that keeps every duration
 * variable at or above tau_min = backward_T(T_min), i.e. every T_i >= T_min;  1 / max_i (-d_i / (tau_i - tau_min)) over the
 * duration variables that move down (the form the kernel evaluates: one wave maximum and one division). */
static double obj_stepbound(void *inst, const double *xp, const double *d, int n) {
  obj_t *o = (obj_t *)inst;
  const int nw = 3 * (o->N - 1);
  double worst = 0.0;
  for (int i = nw; i < n; ++i)
    if (d[i] < 0.0) {
      const double room = xp[i] - o->tau_min;
      const double q = -d[i] / (room > 1e-300 ? room : 1e-300);
      if (q > worst) worst = q;
    }
  return worst > 0.0 ? 1.0 / worst : INFINITY;
}

/* x = [waypoints (N-1) x 3, tau N] */
static double obj_eval(void *inst, const double *x, double *g, int n) {
  obj_t *o = (obj_t *)inst;
  const int nw = 3 * (o->N - 1);
  double T[MAXN], gT[MAXN], f = 0.0;
  (void)n;
  for (int i = 0; i < o->N; ++i) T[i] = fwd_T(x[nw + i]);
  int rc = oracle_minco_cost_grad(o->s, o->c, o->N, o->head, o->tail, x, T, o->hpolys, o->pp, &f, g, gT, NULL);
  if (rc) o->rc = rc;
  for (int i = 0; i < o->N; ++i) g[nw + i] = gT[i] * dfwd_T(x[nw + i]);
  return f;
}

typedef struct {
  int s, c, N, mode; /* mode 0: one cost + gradient evaluation per trajectory; 1: L-BFGS to its own stop */
  int64_t B;
  const double *head, *tail, *hpolys;
  double *wps, *T;
  const oracle_penalty *pp;
  const oracle_lbfgs_param *param;
  double *cost, *gradP, *gradT;
  int *status, *iters, *evals;
  int64_t *next;
  pthread_mutex_t *mu;
  int rc;
  double min_duration; /* > 0: L-BFGS with the minimum-duration step bound */
} batch_t;

static void *batch_run(void *arg) {
  batch_t *j = (batch_t *)arg;
  const int N = j->N, c = j->c, nw = 3 * (N - 1);
  const int M = j->pp ? j->pp->M : 0;
  for (;;) {
    pthread_mutex_lock(j->mu);
    const int64_t b0 = *j->next;
    const int64_t chunk = j->mode ? 1 : 64;
    *j->next = b0 + chunk;
    pthread_mutex_unlock(j->mu);
    if (b0 >= j->B) break;
    const int64_t b1 = b0 + chunk < j->B ? b0 + chunk : j->B;
    for (int64_t b = b0; b < b1; ++b) {
      const double *hp = j->hpolys ? j->hpolys + (size_t)b * N * M * 4 : NULL;
      if (j->mode == 0) {
        int rc = oracle_minco_cost_grad(j->s, c, N, j->head + b * 3 * c, j->tail + b * 3 * c, j->wps + b * nw, j->T + b * N,
                                        hp, j->pp, j->cost + b, j->gradP ? j->gradP + b * nw : NULL,
                                        j->gradT ? j->gradT + b * N : NULL, NULL);
        if (rc) j->rc = rc;
      } else {
        double x[3 * MAXN + MAXN];
        obj_t o = {j->s, c, N, j->head + b * 3 * c, j->tail + b * 3 * c, hp, j->pp, 0,
                   j->min_duration > 0.0 ? bwd_T(j->min_duration) : 0.0};
        memcpy(x, j->wps + b * nw, sizeof(double) * nw);
        for (int i = 0; i < N; ++i) x[nw + i] = bwd_T(j->T[b * N + i]);
        double f = 0.0;
        int it = 0, ev = 0;
        const int ret = oracle_lbfgs_optimize_sb(nw + N, x, &f, obj_eval, j->min_duration > 0.0 ? obj_stepbound : NULL, &o,
                                                 j->param, &it, &ev);
        if (o.rc) j->rc = o.rc;
        memcpy(j->wps + b * nw, x, sizeof(double) * nw);
        for (int i = 0; i < N; ++i) j->T[b * N + i] = fwd_T(x[nw + i]);
        j->cost[b] = f;
        if (j->status) j->status[b] = ret;
        if (j->iters) j->iters[b] = it;
        if (j->evals) j->evals[b] = ev;
      }
    }
  }
  return NULL;
}

static int run_batch(batch_t *proto, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 512) nthreads = 512;
  int64_t next = 0;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  batch_t *jobs = (batch_t *)malloc(sizeof(batch_t) * nthreads);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = *proto;
    jobs[t].next = &next;
    jobs[t].mu = &mu;
    jobs[t].rc = 0;
  }
  if (nthreads == 1) {
    batch_run(&jobs[0]);
  } else {
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_run, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; ++t)
    if (jobs[t].rc) rc = jobs[t].rc;
  free(jobs);
  free(th);
  return rc;
}

/* One cost + gradient evaluation of each of B trajectories on `nthreads` host threads. */
int oracle_minco_cost_grad_batch(int s, int c, int N, int64_t B, const double *head, const double *tail, const double *wps,
                                 const double *T, const double *hpolys, const oracle_penalty *pp, double *cost,
                                 double *gradP, double *gradT, int nthreads) {
  batch_t j;
  memset(&j, 0, sizeof(j));
  j.s = s; j.c = c; j.N = N; j.mode = 0; j.B = B;
  j.head = head; j.tail = tail; j.hpolys = hpolys; j.wps = (double *)wps; j.T = (double *)T; j.pp = pp;
  j.cost = cost; j.gradP = gradP; j.gradT = gradT;
  return run_batch(&j, nthreads);
}

/* L-BFGS (lbfgs.hpp:434-717 as restated by oracle_lbfgs_optimize) on each of B trajectories; wps and T are updated in
 * place, cost / status / iters / evals per trajectory.  min_duration > 0: with lbfgs_optimize's proc_stepbound set to the
 * minimum-duration bound above (lbfgs.hpp:557-565). */
int oracle_lbfgs_minco_batch(int s, int c, int N, int64_t B, const double *head, const double *tail, double *wps, double *T,
                             const double *hpolys, const oracle_penalty *pp, const oracle_lbfgs_param *param, double *cost,
                             int *status, int *iters, int *evals, int nthreads, double min_duration) {
  batch_t j;
  memset(&j, 0, sizeof(j));
  j.s = s; j.c = c; j.N = N; j.mode = 1; j.B = B;
  j.head = head; j.tail = tail; j.hpolys = hpolys; j.wps = wps; j.T = T; j.pp = pp; j.param = param;
  j.cost = cost; j.status = status; j.iters = iters; j.evals = evals;
  j.min_duration = min_duration;
  return run_batch(&j, nthreads);
}
