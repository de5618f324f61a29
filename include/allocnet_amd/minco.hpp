// MINCO_S{2,3,4}NU facade.  `minco.hpp` is NOT part of the reference tree; the
// north star asks for its upstream GCOPTER surface -- setConditions / setParameters / getTrajectory
// / getEnergy / getCoeffs / getEnergyPartialGradByCoeffs / getEnergyPartialGradByTimes /
// propogateGrad -- in front of the HIP kernels.  Order numbering is the reference's: S3 = min-jerk
// (degree 5, Trajectory<5>), S4 = min-snap (degree 7, Trajectory<7>)
// (config/planner.yaml:23, planner/learning_planner.hpp:203-233).
//
// One object = one trajectory (like upstream); for batches call the C ABI (anet_minco_*_dev) directly.
// Matrix arguments are duck-typed ((r,c) access): Eigen::Matrix<double,3,S>, Eigen::Matrix3Xd, ... or
// anet::Matrix.  Coefficients are exposed in the reference's layout: per piece 3 x 2S, highest
// power first.
#pragma once
#include <vector>

#include "core.hpp"
#include "trajectory.hpp"

namespace minco {

template <int S>
class MINCO_SNU {
 public:
  static constexpr int D = 2 * S;

  // headState / tailState: 3 x c, columns p, v, a[, j]; c = bcCols (<= S): c = 3 is the reference's
  // PVA convention (qp_solver.hpp:37,152-158), c = S the classic MINCO convention.
  template <class M1, class M2>
  inline void setConditions(const M1 &headState, const M2 &tailState, const int &pieceNum, int bcCols = S) {
    N = pieceNum;
    c = bcCols;
    head.assign(3 * c, 0.0);
    tail.assign(3 * c, 0.0);
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < c; ++j) {
        head[a * c + j] = headState(a, j);
        tail[a * c + j] = tailState(a, j);
      }
    T.assign(N, 0.0);
    wps.assign(3 * (size_t)(N > 1 ? N - 1 : 0), 0.0);
    coeffs.assign((size_t)N * 3 * D, 0.0);
  }

  // inPs: 3 x (N-1) interior waypoints, ts: N durations.
  template <class MP, class VT>
  inline void setParameters(const MP &inPs, const VT &ts) {
    for (int i = 0; i < N; ++i) T[i] = ts(i);
    for (int k = 0; k + 1 < N; ++k)
      for (int a = 0; a < 3; ++a) wps[(size_t)k * 3 + a] = inPs(a, k);
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_minco_solve(ctx.get(), S, c, N, 1, head.data(), tail.data(), wps.data(), T.data(),
                               coeffs.data(), &energy));
  }

  inline void getTrajectory(Trajectory<D - 1> &traj) const {
    traj.clear();
    traj.reserve(N);
    for (int i = 0; i < N; ++i) {
      anet::Matrix<3, D> cm;
      for (int a = 0; a < 3; ++a)
        for (int k = 0; k < D; ++k) cm(a, k) = coeffs[((size_t)i * 3 + a) * D + k];
      traj.emplace_back(T[i], cm);
    }
  }

  // Extension (no upstream counterpart; upstream's use is a loop over setParameters / getEnergy): the cost
  // int (p^(S))^2 + rho * sum T of K candidate time allocations of THIS problem (conditions and waypoints as last set),
  // one launch.  candidateTs: K x N, row k = the durations of candidate k.
  template <class MT>
  inline void sampleTimeAllocations(const MT &candidateTs, const int K, const double rho, std::vector<double> &costs) const {
    std::vector<double> Ts((size_t)K * N);
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < N; ++i) Ts[(size_t)k * N + i] = candidateTs(k, i);
    costs.assign(K, 0.0);
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_minco_sample_costs(ctx.get(), S, c, N, K, head.data(), tail.data(), wps.data(), Ts.data(), rho, costs.data()));
  }

  inline void getEnergy(double &e) const { e = energy; }  // int (p^(S))^2 dt, all axes
  inline double getEnergy() const { return energy; }
  // piece-major, axis, coefficient (highest power first): the reference's flatten order
  inline const std::vector<double> &getCoeffs() const { return coeffs; }

  // partial gradients of the energy, coefficients treated as independent (same layout as getCoeffs)
  inline void getEnergyPartialGradByCoeffs(std::vector<double> &gdC) const {
    std::vector<double> gdT;
    partials(gdC, gdT);
  }
  inline void getEnergyPartialGradByTimes(std::vector<double> &gdT) const {
    std::vector<double> gdC;
    partials(gdC, gdT);
  }

  // total gradient w.r.t. interior waypoints (gradByPoints: (N-1) x 3, waypoint-major) and durations
  inline void propogateGrad(const std::vector<double> &partialGradByCoeffs,
                            const std::vector<double> &partialGradByTimes, std::vector<double> &gradByPoints,
                            std::vector<double> &gradByTimes) const {
    gradByPoints.assign(3 * (size_t)(N > 1 ? N - 1 : 0), 0.0);
    gradByTimes.assign(N, 0.0);
    // the host entry points work on trajectory-major arrays; a batch of one is its own batch-minor form
    anet::Context &ctx = anet::Context::thread_default();
    DevBuf co(ctx, coeffs), tt(ctx, T), gc(ctx, partialGradByCoeffs), gt(ctx, partialGradByTimes);
    DevBuf gp(ctx, gradByPoints.size() ? gradByPoints.size() : 1), gT(ctx, N);
    ctx.check(anet_minco_propagate_grad_dev(ctx.get(), S, c, N, 1, 1, tt.p, co.p, gc.p, gt.p, gp.p, gT.p,
                                            anet_stream(ctx.get())));
    gp.download(ctx, gradByPoints);
    gT.download(ctx, gradByTimes);
  }

 private:
  // minimal device buffer for the single-trajectory calls above (ld = 1 makes both layouts coincide)
  struct DevBuf {
    double *p = nullptr;
    size_t n = 0;
    DevBuf(anet::Context &ctx, size_t count) : n(count) { ctx.check(anet_dev_alloc(ctx.get(), n, &p)); }
    DevBuf(anet::Context &ctx, const std::vector<double> &h) : n(h.size() ? h.size() : 1) {
      ctx.check(anet_dev_alloc(ctx.get(), n, &p));
      if (!h.empty()) ctx.check(anet_dev_upload(ctx.get(), p, h.data(), h.size()));
    }
    ~DevBuf() { anet_dev_free(p); }
    void download(anet::Context &ctx, std::vector<double> &h) {
      if (!h.empty()) ctx.check(anet_dev_download(ctx.get(), h.data(), p, h.size()));
    }
  };
  void partials(std::vector<double> &gdC, std::vector<double> &gdT) const {
    gdC.assign(coeffs.size(), 0.0);
    gdT.assign(N, 0.0);
    anet::Context &ctx = anet::Context::thread_default();
    DevBuf co(ctx, coeffs), tt(ctx, T), gc(ctx, coeffs.size()), gt(ctx, N);
    ctx.check(anet_minco_partial_grads_dev(ctx.get(), S, N, 1, 1, co.p, tt.p, nullptr, nullptr, 1, gc.p, gt.p,
                                           nullptr, anet_stream(ctx.get())));
    gc.download(ctx, gdC);
    gt.download(ctx, gdT);
  }

  int N = 0, c = S;
  std::vector<double> head, tail, wps, T, coeffs;
  double energy = 0.0;
};

typedef MINCO_SNU<2> MINCO_S2NU;
typedef MINCO_SNU<3> MINCO_S3NU;
typedef MINCO_SNU<4> MINCO_S4NU;

}  // namespace minco
