// L-BFGS facade: the reference's parameter struct, return codes and lbfgs_strerror
// (src/planner/include/gcopter/lbfgs.hpp:15-129, 135-184, 724-799) with the same names and values.
// lbfgs_optimize itself takes host callbacks (lbfgs.hpp:200-259, 434) and has a single caller in the
// reference (firi::maxVolInsEllipsoid, firi.hpp:207-227); its GPU replacement is batched over
// problems with the objective evaluated on the device: lbfgs_optimize_mvie (that call site) and
// lbfgs_optimize_minco (the trajectory cost of the north star).
#pragma once
#include <stdexcept>
#include <string.h>
#include <vector>

#include "core.hpp"

namespace firi {
// firi::costMVIE (gcopter/firi.hpp:86-157).  In this build the objective is evaluated on the device inside
// lbfgs::lbfgs_optimize (k_mvie_eval / k_lbfgs_mvie_persistent); the function exists so that the reference's call
// expression `lbfgs_optimize(x, minCost, &costMVIE, nullptr, nullptr, optData, paramsMVIE)` (firi.hpp:221-227) keeps
// compiling -- its address is the tag the overload below dispatches on.  There is no CPU evaluation to fall back to.
template <class V>
inline double costMVIE(void *, const V &, V &) {
  throw std::logic_error("firi::costMVIE is evaluated on the device: pass it to lbfgs::lbfgs_optimize");
}
}  // namespace firi

namespace lbfgs {

namespace detail {
template <class T> struct same { typedef T type; };
}
// The callback types of lbfgs.hpp:200-259, on any VectorXd-like V ((i) access, size()).
template <class V> using lbfgs_evaluate_t = double (*)(void *instance, const typename detail::same<V>::type &x, typename detail::same<V>::type &g);
template <class V> using lbfgs_stepbound_t = double (*)(void *instance, const typename detail::same<V>::type &xp, const typename detail::same<V>::type &d);
template <class V> using lbfgs_progress_t = int (*)(void *instance, const typename detail::same<V>::type &x, const typename detail::same<V>::type &g,
                                                    const double fx, const double step, const int k, const int ls);

struct lbfgs_parameter_t {
  int mem_size = 8;
  double g_epsilon = 1.0e-5;
  int past = 3;
  double delta = 1.0e-6;
  int max_iterations = 0;
  int max_linesearch = 64;
  double min_step = 1.0e-20;
  double max_step = 1.0e+20;
  double f_dec_coeff = 1.0e-4;
  double s_curv_coeff = 0.9;
  double cautious_factor = 1.0e-6;
  double machine_prec = 1.0e-16;
};

enum {
  LBFGS_CONVERGENCE = 0,
  LBFGS_STOP,
  LBFGS_CANCELED,
  LBFGSERR_UNKNOWNERROR = -1024,
  LBFGSERR_INVALID_N,
  LBFGSERR_INVALID_MEMSIZE,
  LBFGSERR_INVALID_GEPSILON,
  LBFGSERR_INVALID_TESTPERIOD,
  LBFGSERR_INVALID_DELTA,
  LBFGSERR_INVALID_MINSTEP,
  LBFGSERR_INVALID_MAXSTEP,
  LBFGSERR_INVALID_FDECCOEFF,
  LBFGSERR_INVALID_SCURVCOEFF,
  LBFGSERR_INVALID_MACHINEPREC,
  LBFGSERR_INVALID_MAXLINESEARCH,
  LBFGSERR_INVALID_FUNCVAL,
  LBFGSERR_MINIMUMSTEP,
  LBFGSERR_MAXIMUMSTEP,
  LBFGSERR_MAXIMUMLINESEARCH,
  LBFGSERR_MAXIMUMITERATION,
  LBFGSERR_WIDTHTOOSMALL,
  LBFGSERR_INVALIDPARAMETERS,
  LBFGSERR_INCREASEGRADIENT,
};

inline anet_lbfgs_params to_c(const lbfgs_parameter_t &p) {
  anet_lbfgs_params q;
  q.mem_size = p.mem_size; q.g_epsilon = p.g_epsilon; q.past = p.past; q.delta = p.delta;
  q.max_iterations = p.max_iterations; q.max_linesearch = p.max_linesearch; q.min_step = p.min_step;
  q.max_step = p.max_step; q.f_dec_coeff = p.f_dec_coeff; q.s_curv_coeff = p.s_curv_coeff;
  q.cautious_factor = p.cautious_factor; q.machine_prec = p.machine_prec;
  return q;
}

inline const char *lbfgs_strerror(const int err) { return anet_lbfgs_strerror(err); }

// Batched replacement of the reference's only lbfgs_optimize call (firi.hpp:221-227): per problem b,
// A[b] is the M x 3 matrix column-major exactly as firi packs optData (firi.hpp:186-200), x[b] the 9
// variables (in/out).  Returns per-problem lbfgs_optimize return codes.
inline std::vector<int> lbfgs_optimize_mvie(int batch, int M, const std::vector<double> &A, double smoothEps,
                                            double penaltyWt, std::vector<double> &x, std::vector<double> &minCost,
                                            const lbfgs_parameter_t &param, int max_evals = 20000) {
  std::vector<int> status(batch), iters(batch), evals(batch);
  minCost.assign(batch, 0.0);
  anet_lbfgs_params q = to_c(param);
  anet::Context &ctx = anet::Context::thread_default();
  ctx.check(anet_lbfgs_mvie(ctx.get(), batch, M, A.data(), smoothEps, penaltyWt, x.data(), minCost.data(), &q,
                            max_evals, status.data(), iters.data(), evals.data()));
  return status;
}

// lbfgs::lbfgs_optimize (lbfgs.hpp:434-440) for the reference's own call (firi.hpp:221-227): objective
// &firi::costMVIE, no step bound, no progress monitor, `instance` = firi's optData blob {int M; double smoothEps,
// penaltyWt; double A[3 M] column-major} (firi.hpp:186-200).  Runs anet_lbfgs_mvie with a batch of one; x and f are
// updated like the reference's, the return value is its return code.  Any other HOST-evaluated objective is refused: there
// is no CPU L-BFGS in this library.  Batched device objectives: lbfgs_optimize_mvie above, anet_lbfgs_minco, and -- any
// objective the caller can evaluate on the device -- lbfgs_optimize_batched below (anet_lbfgs_optimize_dev).  The step-bound
// mechanism itself (lbfgs.hpp:557-565) is there for the MINCO objective as a built-in bound, a minimum duration:
// anet_lbfgs_minco_bounded[_dev](..., min_duration, ...); the progress monitor's one effect (lbfgs.hpp:580-587: a non-zero
// return cancels the run) as a word the caller owns: anet_set_cancel_flag.
template <class V>
inline int lbfgs_optimize(V &x, double &f, lbfgs_evaluate_t<V> proc_evaluate, lbfgs_stepbound_t<V> proc_stepbound,
                          lbfgs_progress_t<V> proc_progress, void *instance, const lbfgs_parameter_t &param) {
  if (proc_evaluate != static_cast<lbfgs_evaluate_t<V>>(&firi::costMVIE<V>))
    throw std::invalid_argument("lbfgs_optimize: only firi::costMVIE is available as a host-named objective (device evaluation)");
  if (proc_stepbound || proc_progress)
    throw std::invalid_argument("lbfgs_optimize: step-bound / progress callbacks are not supported (the optimisation runs on the device)");
  if ((int)x.size() != 9) return LBFGSERR_INVALID_N;
  int M = 0;
  double eps = 0.0, wt = 0.0;
  const unsigned char *blob = static_cast<const unsigned char *>(instance);
  memcpy(&M, blob, sizeof(int));                       // (the doubles follow the int unaligned, as firi packs them)
  memcpy(&eps, blob + sizeof(int), sizeof(double));
  memcpy(&wt, blob + sizeof(int) + sizeof(double), sizeof(double));
  std::vector<double> A((size_t)3 * M), xs(9), cost;
  memcpy(A.data(), blob + sizeof(int) + 2 * sizeof(double), sizeof(double) * 3 * M);
  for (int i = 0; i < 9; ++i) xs[i] = x(i);
  const std::vector<int> ret = lbfgs_optimize_mvie(1, M, A, eps, wt, xs, cost, param);
  for (int i = 0; i < 9; ++i) x(i) = xs[i];
  f = cost[0];
  return ret[0];
}

// lbfgs_optimize for a batch of problems whose objective the caller evaluates ON THE DEVICE: proc_evaluate (the C signature
// anet_lbfgs_evaluate_t: instance, x [n][ld], f [batch], g [n][ld], batch, ld, n, stream) enqueues one evaluation of the whole
// batch on `stream`.  x (device, batch-minor) is the start point in and the result out, f / g are device buffers of the
// caller's; returns lbfgs_optimize's return code per problem (anet_lbfgs_optimize_dev).
inline std::vector<int> lbfgs_optimize_batched(int n, int64_t batch, int64_t ld, double *x_dev, double *f_dev, double *g_dev,
                                               anet_lbfgs_evaluate_t proc_evaluate, void *instance,
                                               const lbfgs_parameter_t &param, int max_evals = 20000, void *stream = nullptr) {
  anet_lbfgs_params q = to_c(param);
  anet::Context &ctx = anet::Context::thread_default();
  const int64_t nwork = anet_lbfgs_workspace(n, ld, &q);
  double *work = nullptr, *st3 = nullptr;  // st3: the int32 status row in a buffer of doubles
  ctx.check(anet_dev_alloc(ctx.get(), (size_t)nwork, &work));
  ctx.check(anet_dev_alloc(ctx.get(), (size_t)(ld + 1) / 2, &st3));
  std::vector<int32_t> status32((size_t)(ld + 2));
  const int rc = anet_lbfgs_optimize_dev(ctx.get(), n, batch, ld, x_dev, f_dev, g_dev, proc_evaluate, instance, &q, max_evals, n,
                                         0.0, work, (int32_t *)st3, nullptr, nullptr, stream ? stream : anet_stream(ctx.get()));
  if (rc == ANET_OK) anet_dev_download(ctx.get(), (double *)status32.data(), st3, (size_t)(ld + 1) / 2);
  anet_dev_free(work);
  anet_dev_free(st3);
  std::vector<int> status(status32.begin(), status32.begin() + batch);
  ctx.check(rc);
  return status;
}

}  // namespace lbfgs
