// L-BFGS facade: the reference's parameter struct, return codes and lbfgs_strerror
// (src/planner/include/gcopter/lbfgs.hpp:15-129, 135-184, 724-799) with the same names and values.
// lbfgs_optimize itself takes host callbacks (lbfgs.hpp:200-259, 434) and has a single caller in the
// reference (firi::maxVolInsEllipsoid, firi.hpp:207-227): that call runs on the device as a whole
// (lbfgs_optimize_mvie), any other objective through the caller's host callbacks with the optimiser's
// vectors on the device (anet_lbfgs_optimize_host); batched device objectives: lbfgs_optimize_mvie,
// anet_lbfgs_minco (the trajectory cost of the north star), lbfgs_optimize_batched.
#pragma once
#include <stdexcept>
#include <string.h>
#include <type_traits>
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

namespace detail {
// The reference's callbacks take Eigen vectors; the C ABI hands over plain arrays: a V of the right size is built around each
// call (V needs a size constructor, size() and (i) access -- Eigen::VectorXd has them).
template <class V>
struct HostCallbacks {
  lbfgs_evaluate_t<V> evaluate;
  lbfgs_stepbound_t<V> stepbound;
  lbfgs_progress_t<V> progress;
  void *instance;
  static V sized(int n, std::true_type) { return V(n); }
  static V sized(int, std::false_type) {
    throw std::invalid_argument("lbfgs_optimize: host callbacks need a vector type with a size constructor (Eigen::VectorXd has one)");
  }
  static V wrap(const double *p, int n) {
    V v = sized(n, std::is_constructible<V, int>());
    for (int i = 0; i < n; ++i) v(i) = p[i];
    return v;
  }
  static double c_evaluate(void *self, const double *x, double *g, int n) {
    HostCallbacks *h = static_cast<HostCallbacks *>(self);
    const V xv = wrap(x, n);
    V gv = wrap(x, n);
    for (int i = 0; i < n; ++i) gv(i) = 0.0;
    const double f = h->evaluate(h->instance, xv, gv);
    for (int i = 0; i < n; ++i) g[i] = gv(i);
    return f;
  }
  static double c_stepbound(void *self, const double *xp, const double *d, int n) {
    HostCallbacks *h = static_cast<HostCallbacks *>(self);
    return h->stepbound(h->instance, wrap(xp, n), wrap(d, n));
  }
  static int c_progress(void *self, const double *x, const double *g, double fx, double step, int k, int ls, int n) {
    HostCallbacks *h = static_cast<HostCallbacks *>(self);
    return h->progress(h->instance, wrap(x, n), wrap(g, n), fx, step, k, ls);
  }
};
}  // namespace detail

// lbfgs::lbfgs_optimize (lbfgs.hpp:434-440), source-compatible: x and f are updated like the reference's, the return value is
// its return code.
//  * The reference's own call (firi.hpp:221-227) -- objective &firi::costMVIE, no step bound, no progress monitor, `instance` =
//    firi's optData blob {int M; double smoothEps, penaltyWt; double A[3 M] column-major} (firi.hpp:186-200) -- runs entirely on
//    the device (anet_lbfgs_mvie with a batch of one).
//  * Any other objective is evaluated by the caller's HOST callbacks exactly where lbfgs_optimize calls them -- proc_evaluate per
//    trial point, proc_stepbound at the entry of every line search (lbfgs.hpp:557-565), proc_progress after every successful one
//    (lbfgs.hpp:580-587; non-zero cancels) -- while the optimiser's vectors and arithmetic stay on the device
//    (anet_lbfgs_optimize_host: the library has no CPU optimiser; one PCIe round trip per evaluation).
//  Batched device objectives: lbfgs_optimize_mvie above, anet_lbfgs_minco, lbfgs_optimize_batched below.
template <class V>
inline int lbfgs_optimize(V &x, double &f, lbfgs_evaluate_t<V> proc_evaluate, lbfgs_stepbound_t<V> proc_stepbound,
                          lbfgs_progress_t<V> proc_progress, void *instance, const lbfgs_parameter_t &param) {
  if (proc_evaluate != static_cast<lbfgs_evaluate_t<V>>(&firi::costMVIE<V>) || proc_stepbound || proc_progress) {
    if (proc_evaluate == static_cast<lbfgs_evaluate_t<V>>(&firi::costMVIE<V>))
      throw std::invalid_argument("lbfgs_optimize: firi::costMVIE is evaluated on the device and takes no host step-bound / progress callbacks");
    if (!proc_evaluate) throw std::invalid_argument("lbfgs_optimize: proc_evaluate is null");
    const int n = (int)x.size();
    detail::HostCallbacks<V> cb{proc_evaluate, proc_stepbound, proc_progress, instance};
    std::vector<double> xs((size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) xs[i] = x(i);
    anet_lbfgs_params q = to_c(param);
    int32_t ret = LBFGSERR_UNKNOWNERROR;
    double fx = 0.0;
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_lbfgs_optimize_host(ctx.get(), n, xs.data(), &fx, &detail::HostCallbacks<V>::c_evaluate,
                                       proc_stepbound ? &detail::HostCallbacks<V>::c_stepbound : nullptr,
                                       proc_progress ? &detail::HostCallbacks<V>::c_progress : nullptr, &cb, &q, &ret, nullptr,
                                       nullptr));
    if (ret > LBFGSERR_INVALID_MAXLINESEARCH || ret >= 0) {  // (a parameter error leaves x and f untouched, as the reference does)
      for (int i = 0; i < n; ++i) x(i) = xs[i];
      f = fx;
    }
    return ret;
  }
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
  // one persistent workspace of the context (grown, never freed between calls): the optimiser's state, then the int32 status row
  double *work = ctx.workspace((size_t)nwork + (size_t)(ld + 1) / 2);
  double *st3 = work + nwork;
  std::vector<int32_t> status32((size_t)(ld + 2));
  const int rc = anet_lbfgs_optimize_dev(ctx.get(), n, batch, ld, x_dev, f_dev, g_dev, proc_evaluate, instance, &q, max_evals, n,
                                         0.0, work, (int32_t *)st3, nullptr, nullptr, stream ? stream : anet_stream(ctx.get()));
  if (rc == ANET_OK) anet_dev_download(ctx.get(), (double *)status32.data(), st3, (size_t)(ld + 1) / 2);
  std::vector<int> status(status32.begin(), status32.begin() + batch);
  ctx.check(rc);
  return status;
}

}  // namespace lbfgs
