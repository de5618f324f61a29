/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's L-BFGS and of its only
 * objective.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 *   oracle_lbfgs_optimize : lbfgs::lbfgs_optimize + line_search_lewisoverton,
 *                           src/planner/include/gcopter/lbfgs.hpp:276-384, 434-717 -- same control
 *                           flow, same parameter validation order, same return codes (:135-184),
 *                           including the quirk that on a failed line search x and g are reverted
 *                           but the reported f is the last trial's value (the line search writes
 *                           through the caller's fx, :570-577, :713).
 *   oracle_cost_mvie      : firi::costMVIE + smoothedL1, gcopter/firi.hpp:60-157.
 *
 * PARITY STATUS: the reference cannot be compiled here (lbfgs.hpp/firi.hpp need Eigen, which the
 * image lacks, and no stand-in may be written for it) and holds no tests or golden vectors for
 * this path -> PARITY UNPINNED against reference outputs.  Pinned by known-answer problems
 * (quadratic, Rosenbrock: tests/test_lbfgs_cpu.py) and by finite differences of costMVIE.
 * Dot products are plain left-to-right sums (Eigen's vectorised order is not reproducible).
 *
 * ROUNDING CONVENTION: the reference is built with -O3 and no -march (src/planner/CMakeLists.txt:4-6): baseline x86-64
 * has no fused multiply-add, so every a + b * c of lbfgs.hpp is a multiply and an add.  This file is therefore compiled
 * with -ffp-contract=off (oracle/Makefile, oracle/cbind.py): no expression here is fused, whatever -march says.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int mem_size;
  double g_epsilon;
  int past;
  double delta;
  int max_iterations;
  int max_linesearch;
  double min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
} oracle_lbfgs_param;

enum {
  LBFGS_CONVERGENCE = 0, LBFGS_STOP, LBFGS_CANCELED,
  LBFGSERR_UNKNOWNERROR = -1024, LBFGSERR_INVALID_N, LBFGSERR_INVALID_MEMSIZE, LBFGSERR_INVALID_GEPSILON,
  LBFGSERR_INVALID_TESTPERIOD, LBFGSERR_INVALID_DELTA, LBFGSERR_INVALID_MINSTEP, LBFGSERR_INVALID_MAXSTEP,
  LBFGSERR_INVALID_FDECCOEFF, LBFGSERR_INVALID_SCURVCOEFF, LBFGSERR_INVALID_MACHINEPREC,
  LBFGSERR_INVALID_MAXLINESEARCH, LBFGSERR_INVALID_FUNCVAL, LBFGSERR_MINIMUMSTEP, LBFGSERR_MAXIMUMSTEP,
  LBFGSERR_MAXIMUMLINESEARCH, LBFGSERR_MAXIMUMITERATION, LBFGSERR_WIDTHTOOSMALL,
  LBFGSERR_INVALIDPARAMETERS, LBFGSERR_INCREASEGRADIENT
};

typedef double (*oracle_eval_t)(void *instance, const double *x, double *g, int n);

void oracle_lbfgs_default_param(oracle_lbfgs_param *p) { /* lbfgs.hpp:15-129 */
  p->mem_size = 8; p->g_epsilon = 1.0e-5; p->past = 3; p->delta = 1.0e-6; p->max_iterations = 0;
  p->max_linesearch = 64; p->min_step = 1.0e-20; p->max_step = 1.0e+20; p->f_dec_coeff = 1.0e-4;
  p->s_curv_coeff = 0.9; p->cautious_factor = 1.0e-6; p->machine_prec = 1.0e-16;
}

static double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static double maxabs(const double *a, int n) { double m = 0; for (int i = 0; i < n; ++i) { double v = fabs(a[i]); if (v > m) m = v; } return m; }

/* lbfgs.hpp:276-384 */
static int line_search(int n, double *x, double *f, double *g, double *stp, const double *s, const double *xp,
                       const double *gp, double stpmin, double stpmax, oracle_eval_t eval, void *inst,
                       const oracle_lbfgs_param *param, int *evals) {
  int count = 0, brackt = 0, touched = 0;
  double finit, dginit, dgtest, dstest, mu = 0.0, nu = stpmax;
  if (!(*stp > 0.0)) return LBFGSERR_INVALIDPARAMETERS;
  dginit = dot(gp, s, n);
  if (0.0 < dginit) return LBFGSERR_INCREASEGRADIENT;
  finit = *f;
  dgtest = param->f_dec_coeff * dginit;
  dstest = param->s_curv_coeff * dginit;
  for (;;) {
    for (int i = 0; i < n; ++i) x[i] = xp[i] + *stp * s[i]; /* lbfgs.hpp:308; two roundings: this file is compiled with -ffp-contract=off */
    *f = eval(inst, x, g, n);
    ++count; ++*evals;
    if (isinf(*f) || isnan(*f)) return LBFGSERR_INVALID_FUNCVAL;
    if (*f > finit + *stp * dgtest) { nu = *stp; brackt = 1; }
    else {
      if (dot(g, s, n) < dstest) mu = *stp;
      else return count;
    }
    if (param->max_linesearch <= count) return LBFGSERR_MAXIMUMLINESEARCH;
    if (brackt && (nu - mu) < param->machine_prec * nu) return LBFGSERR_WIDTHTOOSMALL;
    if (brackt) *stp = 0.5 * (mu + nu); else *stp *= 2.0;
    if (*stp < stpmin) return LBFGSERR_MINIMUMSTEP;
    if (*stp > stpmax) {
      if (touched) return LBFGSERR_MAXIMUMSTEP;
      touched = 1; *stp = stpmax;
    }
  }
}

typedef double (*oracle_stepbound_t)(void *instance, const double *xp, const double *d, int n);

/* lbfgs.hpp:434-717.  iters/evals (may be NULL) receive the iteration count k and the number of
 * objective evaluations.  stepbound (may be NULL): lbfgs_stepbound_t, applied as lbfgs.hpp:557-565 does. */
int oracle_lbfgs_optimize_sb(int n, double *x, double *f, oracle_eval_t eval, oracle_stepbound_t stepbound, void *inst,
                             const oracle_lbfgs_param *param, int *iters, int *evals_out);
int oracle_lbfgs_optimize(int n, double *x, double *f, oracle_eval_t eval, void *inst,
                          const oracle_lbfgs_param *param, int *iters, int *evals_out) {
  return oracle_lbfgs_optimize_sb(n, x, f, eval, NULL, inst, param, iters, evals_out);
}
typedef int (*oracle_progress_t)(void *instance, const double *x, const double *g, double fx, double step, int k, int ls, int n);
int oracle_lbfgs_optimize_full(int n, double *x, double *f, oracle_eval_t eval, oracle_stepbound_t stepbound,
                               oracle_progress_t progress, void *inst, const oracle_lbfgs_param *param, int *iters, int *evals_out);
int oracle_lbfgs_optimize_sb(int n, double *x, double *f, oracle_eval_t eval, oracle_stepbound_t stepbound, void *inst,
                             const oracle_lbfgs_param *param, int *iters, int *evals_out) {
  return oracle_lbfgs_optimize_full(n, x, f, eval, stepbound, NULL, inst, param, iters, evals_out);
}
/* ... and with lbfgs_progress_t (lbfgs.hpp:226-246), called after every successful line search, before the convergence test
 * (lbfgs.hpp:580-587): a non-zero return ends the run with LBFGS_CANCELED. */
int oracle_lbfgs_optimize_full(int n, double *x, double *f, oracle_eval_t eval, oracle_stepbound_t stepbound,
                               oracle_progress_t progress, void *inst, const oracle_lbfgs_param *param, int *iters, int *evals_out) {
  int ret, i, j, k = 0, ls, end, bound, evals = 0;
  double step, fx, ys, yy, gnorm_inf, xnorm_inf, beta, rate, cau;
  const int m = param->mem_size;
  if (n <= 0) return LBFGSERR_INVALID_N;
  if (m <= 0) return LBFGSERR_INVALID_MEMSIZE;
  if (param->g_epsilon < 0.0) return LBFGSERR_INVALID_GEPSILON;
  if (param->past < 0) return LBFGSERR_INVALID_TESTPERIOD;
  if (param->delta < 0.0) return LBFGSERR_INVALID_DELTA;
  if (param->min_step < 0.0) return LBFGSERR_INVALID_MINSTEP;
  if (param->max_step < param->min_step) return LBFGSERR_INVALID_MAXSTEP;
  if (!(param->f_dec_coeff > 0.0 && param->f_dec_coeff < 1.0)) return LBFGSERR_INVALID_FDECCOEFF;
  if (!(param->s_curv_coeff < 1.0 && param->s_curv_coeff > param->f_dec_coeff)) return LBFGSERR_INVALID_SCURVCOEFF;
  if (!(param->machine_prec > 0.0)) return LBFGSERR_INVALID_MACHINEPREC;
  if (param->max_linesearch <= 0) return LBFGSERR_INVALID_MAXLINESEARCH;

  const int npf = param->past > 1 ? param->past : 1;
  double *buf = (double *)calloc((size_t)(4 * n + npf + 2 * m + 2 * (size_t)n * m), sizeof(double));
  double *xp = buf, *g = xp + n, *gp = g + n, *d = gp + n, *pf = d + n;
  double *lm_alpha = pf + npf, *lm_ys = lm_alpha + m, *lm_s = lm_ys + m, *lm_y = lm_s + (size_t)n * m;

  fx = eval(inst, x, g, n); ++evals;
  pf[0] = fx;
  for (i = 0; i < n; ++i) d[i] = -g[i];
  gnorm_inf = maxabs(g, n); xnorm_inf = maxabs(x, n);
  if (gnorm_inf / fmax(1.0, xnorm_inf) < param->g_epsilon) {
    ret = LBFGS_CONVERGENCE;
  } else {
    step = 1.0 / sqrt(dot(d, d, n));
    k = 1; end = 0; bound = 0;
    for (;;) {
      memcpy(xp, x, sizeof(double) * n); memcpy(gp, g, sizeof(double) * n);
      double step_max = param->max_step;
      if (stepbound) { /* lbfgs.hpp:557-565 */
        step_max = stepbound(inst, xp, d, n);
        step_max = step_max < param->max_step ? step_max : param->max_step;
        step = step < step_max ? step : 0.5 * step_max;
      }
      ls = line_search(n, x, &fx, g, &step, d, xp, gp, param->min_step, step_max, eval, inst, param, &evals);
      if (ls < 0) { memcpy(x, xp, sizeof(double) * n); memcpy(g, gp, sizeof(double) * n); ret = ls; break; }
      if (progress && progress(inst, x, g, fx, step, k, ls, n)) { ret = LBFGS_CANCELED; break; } /* lbfgs.hpp:580-587 */
      gnorm_inf = maxabs(g, n); xnorm_inf = maxabs(x, n);
      if (gnorm_inf / fmax(1.0, xnorm_inf) < param->g_epsilon) { ret = LBFGS_CONVERGENCE; break; }
      if (0 < param->past) {
        if (param->past <= k) {
          rate = fabs(pf[k % param->past] - fx) / fmax(1.0, fabs(fx));
          if (rate < param->delta) { ret = LBFGS_STOP; break; }
        }
        pf[k % param->past] = fx;
      }
      if (param->max_iterations != 0 && param->max_iterations <= k) { ret = LBFGSERR_MAXIMUMITERATION; break; }
      ++k;
      double *se = lm_s + (size_t)end * n, *ye = lm_y + (size_t)end * n;
      for (i = 0; i < n; ++i) { se[i] = x[i] - xp[i]; ye[i] = g[i] - gp[i]; }
      ys = dot(ye, se, n); yy = dot(ye, ye, n);
      lm_ys[end] = ys;
      for (i = 0; i < n; ++i) d[i] = -g[i];
      cau = dot(se, se, n) * sqrt(dot(gp, gp, n)) * param->cautious_factor;
      if (ys > cau) {
        ++bound; bound = m < bound ? m : bound;
        end = (end + 1) % m;
        j = end;
        for (i = 0; i < bound; ++i) {
          j = (j + m - 1) % m;
          lm_alpha[j] = dot(lm_s + (size_t)j * n, d, n) / lm_ys[j];
          for (int q = 0; q < n; ++q) d[q] += (-lm_alpha[j]) * lm_y[(size_t)j * n + q];
        }
        for (int q = 0; q < n; ++q) d[q] *= ys / yy;
        for (i = 0; i < bound; ++i) {
          beta = dot(lm_y + (size_t)j * n, d, n) / lm_ys[j];
          for (int q = 0; q < n; ++q) d[q] += (lm_alpha[j] - beta) * lm_s[(size_t)j * n + q];
          j = (j + 1) % m;
        }
      }
      step = 1.0;
    }
  }
  *f = fx;
  if (iters) *iters = k;
  if (evals_out) *evals_out = evals;
  free(buf);
  return ret;
}

/* firi::smoothedL1, firi.hpp:60-84 */
static int smoothed_l1(double mu, double x, double *f, double *df) {
  if (x < 0.0) return 0;
  if (x > mu) { *f = x - 0.5 * mu; *df = 1.0; return 1; }
  const double xdmu = x / mu, sqrxdmu = xdmu * xdmu, mumxd2 = mu - 0.5 * x;
  *f = mumxd2 * sqrxdmu * xdmu;
  *df = sqrxdmu * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
  return 1;
}

/* firi::costMVIE, firi.hpp:86-157.  data = {M, smoothEps, penaltyWt, A[M x 3] COLUMN-major} as the
 * reference packs it (Eigen::Map<const MatrixX3d>(pA, M, 3) is column-major). */
typedef struct { int M; double eps, wt; const double *A; } oracle_mvie_data;

double oracle_cost_mvie(void *data, const double *x, double *grad, int n) {
  (void)n;
  const oracle_mvie_data *dd = (const oracle_mvie_data *)data;
  const int M = dd->M;
  const double *A = dd->A;
  const double *p = x, *rtd = x + 3, *cde = x + 6;
  double *gdp = grad, *gdrtd = grad + 3, *gdcde = grad + 6;
  double cost = 0;
  for (int i = 0; i < 9; ++i) grad[i] = 0.0;
  double L[3][3] = {{rtd[0] * rtd[0] + DBL_EPSILON, 0, 0},
                    {cde[0], rtd[1] * rtd[1] + DBL_EPSILON, 0},
                    {cde[2], cde[1], rtd[2] * rtd[2] + DBL_EPSILON}};
  for (int i = 0; i < M; ++i) {
    const double a[3] = {A[i], A[M + i], A[2 * M + i]};
    const double AL[3] = {a[0] * L[0][0] + a[1] * L[1][0] + a[2] * L[2][0], a[1] * L[1][1] + a[2] * L[2][1],
                          a[2] * L[2][2]};
    const double nrm = sqrt(AL[0] * AL[0] + AL[1] * AL[1] + AL[2] * AL[2]);
    const double viol = nrm + (a[0] * p[0] + a[1] * p[1] + a[2] * p[2]) - 1.0;
    double c, dc;
    if (smoothed_l1(dd->eps, viol, &c, &dc)) {
      const double adj[3] = {AL[0] / nrm, AL[1] / nrm, AL[2] / nrm};
      const double vec[3] = {dc * a[0], dc * a[1], dc * a[2]};
      cost += c;
      for (int q = 0; q < 3; ++q) { gdp[q] += vec[q]; gdrtd[q] += adj[q] * vec[q]; }
      gdcde[0] += adj[0] * vec[1];
      gdcde[1] += adj[1] * vec[2];
      gdcde[2] += adj[0] * vec[2];
    }
  }
  cost *= dd->wt;
  for (int q = 0; q < 3; ++q) { gdp[q] *= dd->wt; gdrtd[q] *= dd->wt; gdcde[q] *= dd->wt; }
  cost -= log(L[0][0]) + log(L[1][1]) + log(L[2][2]);
  for (int q = 0; q < 3; ++q) { gdrtd[q] -= 1.0 / L[q][q]; gdrtd[q] *= 2.0 * rtd[q]; }
  return cost;
}

/* L-BFGS on costMVIE for one problem (the reference's only call site, firi.hpp:207-227). */
int oracle_lbfgs_mvie(int M, const double *A_colmajor, double eps, double wt, double *x9, double *f,
                      const oracle_lbfgs_param *param, int *iters, int *evals) {
  oracle_mvie_data d = {M, eps, wt, A_colmajor};
  return oracle_lbfgs_optimize(9, x9, f, oracle_cost_mvie, &d, param, iters, evals);
}
