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
  int method_ = ANET_QP_METHOD_INTERIOR_POINT;
  double m34_ = 1400.0;  // the reference's snap-block constant (qp_solver.hpp:212)
  // The time-gradient epilogue (an extension: the reference's caller never asks for it) runs inside solve() only once
  // getTimeGrad() has been asked for; the first request after a plain solve re-solves the remembered problem with it.
  mutable bool want_time_grad_ = false, have_time_grad_ = false;
  mutable std::vector<double> time_grad_;
  std::vector<double> last_state_, last_T_, last_hp_;
  int last_M_ = 0;

  inline int run(bool with_time_grad, int seg, int M, const double *state, const double *T, const double *hp, double *co,
                 double *obj, int32_t *status, int32_t *iters, double *tg) const {
    anet::Context &ctx = anet::Context::thread_default();
    anet_qp_settings st;
    anet_qp_default_settings(&st);
    st.method = method_;
    if (with_time_grad)
      return anet_qp_solve_time_grad(ctx.get(), order_, seg, 1, config.ConstRes, M, config.MaxVelBox, config.MaxAccBox, m34_,
                                     state, T, hp, &st, co, obj, status, iters, nullptr, tg);
    return anet_qp_solve(ctx.get(), order_, seg, 1, config.ConstRes, M, config.MaxVelBox, config.MaxAccBox, m34_, state, T, hp,
                         &st, co, obj, status, iters, nullptr);
  }

 public:
  QPSolver(const QPConfig &conf) : config(conf) {}

  inline void setOrder(int &order) { order_ = order; }
  inline void setOrder(const int &order) { order_ = order; }
  inline double getObjCost() { return obj_cost_; }
  inline int getIterations() const { return last_iters_; }

  // The reference's get_t_state<T> (qp_solver.hpp:88-116): the order_ x 2*order_ matrix whose row k holds the k-th
  // derivative of the monomial basis (t^(d-1), ..., t, 1) at t, evaluated in T's arithmetic (the planner calls it with
  // float sample times, :252) with the reference's multiplication tree for the powers (t2 = t*t, t3 = t*t2, t4 = t2*t2,
  // t5 = t2*t3, t6 = t3*t3, t7 = t4*t3), so a float argument gives the float entries the reference's rows have.  Host
  // arithmetic: 24 / 32 numbers; the device assembles the same rows inside anet_qp_assemble.  `.as<Eigen::MatrixXd>()`
  // converts.
  template <typename T>
  inline anet::MatrixX get_t_state(const T &t) const {
    const int d = 2 * order_;
    T p[8];
    p[0] = T(1);
    p[1] = t;
    p[2] = t * t;
    p[3] = t * p[2];
    p[4] = p[2] * p[2];
    p[5] = p[2] * p[3];
    p[6] = p[3] * p[3];
    p[7] = p[4] * p[3];
    anet::MatrixX A(order_, d);
    for (int k = 0; k < order_; ++k)
      for (int j = 0; j < d; ++j) {
        const int e = d - 1 - j;  // the power of column j
        if (e < k) continue;
        int ff = 1;  // e (e - 1) ... (e - k + 1)
        for (int q = 0; q < k; ++q) ff *= e - q;
        A(k, j) = e == k ? (double)ff : (double)(T(ff) * p[e - k]);
      }
    return A;
  }

  // Extension (not in the reference): d(getObjCost())/d(times(i)) of the last solve, the derivative of the optimal cost
  // through the inequality QP (anet_qp_solve_time_grad).  The first request switches the epilogue on for later solves.
  inline const std::vector<double> &getTimeGrad() const {
    want_time_grad_ = true;
    if (!have_time_grad_ && !last_T_.empty()) {
      const int seg = (int)last_T_.size();
      std::vector<double> co((size_t)seg * 3 * 2 * order_);
      double obj = 0.0;
      int32_t status = 0, iters = 0;
      time_grad_.assign(seg, 0.0);
      anet::Context::thread_default().check(run(true, seg, last_M_, last_state_.data(), last_T_.data(), last_hp_.data(),
                                                co.data(), &obj, &status, &iters, time_grad_.data()));
      have_time_grad_ = true;
    }
    return time_grad_;
  }
  // Extension: ANET_QP_METHOD_INTERIOR_POINT (default: the optimum to 1e-6 in ~10 Newton steps; returns `true` for
  // every problem either method can solve) or ANET_QP_METHOD_ADMM (OSQP's algorithm, tolerances, Ruiz equilibration and
  // cost scaling; at OSQP's default max_iter 1-6 % of feasible 8-piece snap problems, by corridor set, still end
  // unsolved -- a failed plan for the caller; profiles/r05_qp_unsolved_admm.txt).
  inline void setMethod(int method) { method_ = method; }

  template <typename MatA, typename MatB, typename Poly, typename Times, typename Sol>
  inline bool solve(const MatA &iniPVA,  // 3*3
                    const MatB &finPVA, const std::vector<Poly> &hPolys, const Times &times, Sol &qp_solution) {
    const int seg = (int)hPolys.size();
    int M = 1;
    for (int i = 0; i < seg; ++i) M = hPolys[i].rows() > M ? (int)hPolys[i].rows() : M;
    std::vector<double> &state = last_state_, &T = last_T_, &hp = last_hp_;
    state.assign(18, 0.0);
    T.assign(seg, 0.0);
    hp.assign((size_t)seg * M * 4, 0.0);
    last_M_ = M;
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
    time_grad_.assign(seg, 0.0);
    have_time_grad_ = want_time_grad_;
    anet::Context::thread_default().check(run(want_time_grad_, seg, M, state.data(), T.data(), hp.data(), co.data(), &obj,
                                              &status, &iters, time_grad_.data()));
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
