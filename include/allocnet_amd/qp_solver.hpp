// Source-compatible stand-in for the reference's planner/qp_solver.hpp (QPConfig, QPSolver:
// src/planner/include/planner/qp_solver.hpp:14-26, 28-366) as LearningPlanner uses it
// (planner/learning_planner.hpp:21,30,36,196): QPSolver(QPConfig), setOrder(int&),
// solve(iniPVA, finPVA, hPolys, times, qp_solution) -> bool, getObjCost().
//
// The QP is the reference's own (same variables, equality rows, cost blocks, corridor and box rows
// sampled at ConstRes points per piece); it is solved on the MI355X behind anet_qp_solve -- by the batched
// interior-point kernel (k_qp_ipm, the default method; the OSQP-style ADMM kernel is opt-in through
// anet_qp_settings.method) -- instead of by OsqpEigen/OSQP on the CPU.  Acceptance rule
// as in the reference (:334-352): status Solved and -0.01 <= objective (read as float) <= 5000.
// Matrix arguments are duck-typed ((r,c) / (i) access, rows(), resize(n)): Eigen types work unchanged.
#pragma once
#include <cstdio>
#include <memory>
#include <vector>

#include "core.hpp"

struct QPConfig {
  // for other planners
  double MaxVelBox, MaxAccBox;
  int ConstRes;

  QPConfig(double maxVelBox = 4.0, double maxAccBox = 6.0, int constRes = 20)  // config/planner.yaml:17-21
      : MaxVelBox(maxVelBox), MaxAccBox(maxAccBox), ConstRes(constRes) {}
  // ros::NodeHandle-like objects (anything with getParam(name, value)), as in the reference (:19-25)
  template <class NodeHandle, class = decltype(std::declval<const NodeHandle &>().getParam("", std::declval<double &>()))>
  explicit QPConfig(const NodeHandle &nh_priv) : MaxVelBox(4.0), MaxAccBox(6.0), ConstRes(20) {
    nh_priv.getParam("MaxVelBox", MaxVelBox);
    nh_priv.getParam("MaxAccBox", MaxAccBox);
    nh_priv.getParam("ConstRes", ConstRes);
  }
};

class QPSolver {
 private:
  QPConfig config;
  int order_ = 3;
  double obj_cost_ = -1;
  int last_status_ = 0, last_iters_ = 0;
  std::vector<double> time_grad_;
  int method_ = ANET_QP_METHOD_INTERIOR_POINT;
  double m34_ = 1400.0;  // the reference's snap-block constant (qp_solver.hpp:212)

 public:
  QPSolver(const QPConfig &conf) : config(conf) {}

  inline void setOrder(int &order) { order_ = order; }
  inline void setOrder(const int &order) { order_ = order; }
  inline double getObjCost() { return obj_cost_; }
  inline int getIterations() const { return last_iters_; }
  // Extension (not in the reference): d(getObjCost())/d(times(i)) of the last successful solve, the
  // derivative of the optimal cost through the inequality QP (anet_qp_solve_time_grad).
  inline const std::vector<double> &getTimeGrad() const { return time_grad_; }
  // Extension: ANET_QP_METHOD_INTERIOR_POINT (default: the optimum to 1e-6 in ~10 Newton steps; returns `true` for
  // every problem either method can solve) or ANET_QP_METHOD_ADMM (OSQP's algorithm and tolerances; without OSQP's Ruiz
  // equilibration 4-6 % of feasible 8-piece snap problems end at max_iter, i.e. a failed plan for the caller).
  inline void setMethod(int method) { method_ = method; }

  template <typename MatA, typename MatB, typename Poly, typename Times, typename Sol>
  inline bool solve(const MatA &iniPVA,  // 3*3
                    const MatB &finPVA, const std::vector<Poly> &hPolys, const Times &times, Sol &qp_solution) {
    const int seg = (int)hPolys.size();
    int M = 1;
    for (int i = 0; i < seg; ++i) M = hPolys[i].rows() > M ? (int)hPolys[i].rows() : M;
    std::vector<double> state(18), T(seg), hp((size_t)seg * M * 4, 0.0);
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < 3; ++j) {
        state[a * 3 + j] = iniPVA(a, j);
        state[9 + a * 3 + j] = finPVA(a, j);
      }
    for (int i = 0; i < seg; ++i) {
      T[i] = (double)times(i);
      for (int r = 0; r < (int)hPolys[i].rows(); ++r)
        for (int k = 0; k < 4; ++k) hp[((size_t)i * M + r) * 4 + k] = hPolys[i](r, k);
    }
    const int D = 2 * order_;
    std::vector<double> co((size_t)seg * 3 * D);
    double obj = 0.0;
    int32_t status = 0, iters = 0;
    anet::Context &ctx = anet::Context::thread_default();
    time_grad_.assign(seg, 0.0);
    anet_qp_settings st;
    anet_qp_default_settings(&st);
    st.method = method_;
    ctx.check(anet_qp_solve_time_grad(ctx.get(), order_, seg, 1, config.ConstRes, M, config.MaxVelBox, config.MaxAccBox,
                                      m34_, state.data(), T.data(), hp.data(), &st, co.data(), &obj, &status, &iters,
                                      nullptr, time_grad_.data()));
    last_status_ = status;
    last_iters_ = iters;
    const float result = (float)obj;
    if (result > 5000 || result < -0.01) {
      std::printf("[QP solver]: cannot solve the problem\n");
      return false;
    }
    if (status != ANET_QP_SOLVED) {
      std::printf("[QP solver]: solver failed \n");
      return false;
    }
    qp_solution.resize((long)co.size());
    for (size_t i = 0; i < co.size(); ++i) qp_solution((long)i) = co[i];
    obj_cost_ = obj;
    return true;
  }

 public:
  typedef std::unique_ptr<QPSolver> Ptr;
};
