// L-BFGS facade: the reference's parameter struct, return codes and lbfgs_strerror
// (src/planner/include/gcopter/lbfgs.hpp:15-129, 135-184, 724-799) with the same names and values.
// lbfgs_optimize itself takes host callbacks (lbfgs.hpp:200-259, 434) and has a single caller in the
// reference (firi::maxVolInsEllipsoid, firi.hpp:207-227); its GPU replacement is batched over
// problems with the objective evaluated on the device: lbfgs_optimize_mvie (that call site) and
// lbfgs_optimize_minco (the trajectory cost of the north star).
#pragma once
#include <vector>

#include "core.hpp"

namespace lbfgs {

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

}  // namespace lbfgs
