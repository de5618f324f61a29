// Source-compatible stand-in for the reference's gcopter/trajectory.hpp (Piece<D>, Trajectory<D>,
// src/planner/include/gcopter/trajectory.hpp:37-645) whose polynomial evaluation and control-effort
// cost run on the MI355X through the C ABI (anet_traj_eval, anet_traj_cost).
//
// Same public surface and semantics: coefficients 3 x (D+1), highest power first; getPos/getVel/
// getAcc/getJer with the reference's piece location (including the clamp to the last piece);
// getTrajCost(order) with the reference's constants (m_34 = 1400 by default).  Return types are
// anet::Vec3 / anet::Matrix which convert to and from Eigen types by duck typing (core.hpp).
// getMax{Vel,Acc}Rate / checkMax{Vel,Acc}Rate (trajectory.hpp:177-314, 576-630) are provided through
// anet_traj_max_rate (Bernstein-subdivision root isolation instead of Sturm sequences, same extrema).
#pragma once
#include <algorithm>
#include <vector>

#include "core.hpp"

template <int D>
class Piece {
 public:
  typedef anet::Matrix<3, D + 1> CoefficientMat;

 private:
  double duration = 0.0;
  CoefficientMat coeffMat;

 public:
  Piece() = default;
  template <class M>
  Piece(double dur, const M &cMat) : duration(dur), coeffMat(cMat) {}

  inline int getDim() const { return 3; }
  inline int getDegree() const { return D; }
  inline double getDuration() const { return duration; }
  inline const CoefficientMat &getCoeffMat() const { return coeffMat; }

  // one-piece evaluation: a single-piece trajectory with an unbounded duration (no location step)
  inline anet::Vec3 eval(double t, int deriv) const {
    static_assert((D + 1) % 2 == 0, "degree must be 2s-1");
    const double T = 1.0e300;
    anet::Vec3 out;
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_traj_eval(ctx.get(), (D + 1) / 2, 1, 1, coeffMat.data(), &T, 1, &t, deriv, out.v));
    return out;
  }
  inline anet::Vec3 getPos(const double &t) const { return eval(t, 0); }
  inline anet::Vec3 getVel(const double &t) const { return eval(t, 1); }
  inline anet::Vec3 getAcc(const double &t) const { return eval(t, 2); }
  inline anet::Vec3 getJer(const double &t) const { return eval(t, 3); }

  // Piece::normalizePosCoeffMat / normalizeVelCoeffMat / normalizeAccCoeffMat (trajectory.hpp:135-171): the coefficient matrices
  // of position, velocity and acceleration in normalised time (3 x (D + 1), 3 x D, 3 x (D - 1); highest power first)
  typedef anet::Matrix<3, D> VelCoefficientMat;
  typedef anet::Matrix<3, D - 1> AccCoefficientMat;
  // One piece = 3 x (D + 1 - d) multiplications: done here on the host with the statements of k_piece_normalize
  // (csrc/traj_kernels.h: running product t *= duration from the constant column up, factor * coefficient * t) -- a GPU round
  // trip would cost ~20 us for 24 products.  Batches of pieces: anet_piece_normalized_coeffs[_dev].
  template <class M>
  inline M normalized(int deriv) const {
    M out;
    const int W = D + 1 - deriv;
    double t = 1.0;
    for (int e = 0; e < deriv; ++e) t *= duration;
    for (int i = W - 1; i >= 0; --i) {
      const int k = D - i;
      double f = 1.0;
      for (int e = 0; e < deriv; ++e) f *= (double)(k - e);
      for (int ax = 0; ax < 3; ++ax) out(ax, i) = f * coeffMat(ax, i) * t;
      t *= duration;
    }
    return out;
  }
  inline CoefficientMat normalizePosCoeffMat() const { return normalized<CoefficientMat>(0); }
  inline VelCoefficientMat normalizeVelCoeffMat() const { return normalized<VelCoefficientMat>(1); }
  inline AccCoefficientMat normalizeAccCoeffMat() const { return normalized<AccCoefficientMat>(2); }

  inline double maxRate(int which) const {
    double r = 0.0;
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_traj_max_rate(ctx.get(), (D + 1) / 2, 1, 1, coeffMat.data(), &duration, which, &r));
    return r;
  }
  inline double getMaxVelRate() const { return maxRate(1); }
  inline double getMaxAccRate() const { return maxRate(2); }
  inline bool checkMaxVelRate(const double &maxVelRate) const { return getMaxVelRate() < maxVelRate; }
  inline bool checkMaxAccRate(const double &maxAccRate) const { return getMaxAccRate() < maxAccRate; }
};

template <int D>
class Trajectory {
 private:
  typedef std::vector<Piece<D>> Pieces;
  Pieces pieces;

  void flatten(std::vector<double> &co, std::vector<double> &T) const {
    const int N = getPieceNum();
    co.resize((size_t)N * 3 * (D + 1));
    T.resize(N);
    for (int i = 0; i < N; ++i) {
      T[i] = pieces[i].getDuration();
      std::copy(pieces[i].getCoeffMat().data(), pieces[i].getCoeffMat().data() + 3 * (D + 1),
                co.begin() + (size_t)i * 3 * (D + 1));
    }
  }

 public:
  Trajectory() = default;
  template <class CMat>
  Trajectory(const std::vector<double> &durs, const std::vector<CMat> &cMats) {
    int N = (int)std::min(durs.size(), cMats.size());
    pieces.reserve(N);
    for (int i = 0; i < N; i++) pieces.emplace_back(durs[i], cMats[i]);
  }

  inline int getPieceNum() const { return (int)pieces.size(); }
  inline std::vector<double> getDurations() const {
    std::vector<double> d(pieces.size());
    for (size_t i = 0; i < pieces.size(); ++i) d[i] = pieces[i].getDuration();
    return d;
  }
  inline double getTotalDuration() const {
    double t = 0.0;
    for (const auto &p : pieces) t += p.getDuration();
    return t;
  }

  // Trajectory::getTrajCost (trajectory.hpp:354-427).  m34 = 1400 is the reference's constant.
  inline double getTrajCost(int order, double m34 = 1400.0) const {
    if (order * 2 != D + 1) throw anet::Error(ANET_ERR_INVALID, "getTrajCost: order must be (D+1)/2");
    std::vector<double> co, T;
    flatten(co, T);
    double cost = 0.0;
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_traj_cost(ctx.get(), order, getPieceNum(), 1, co.data(), T.data(), m34, &cost));
    return cost;
  }

  inline const Piece<D> &operator[](int i) const { return pieces[i]; }
  inline Piece<D> &operator[](int i) { return pieces[i]; }
  inline void clear(void) { pieces.clear(); }
  inline typename Pieces::const_iterator begin() const { return pieces.begin(); }
  inline typename Pieces::const_iterator end() const { return pieces.end(); }
  inline typename Pieces::iterator begin() { return pieces.begin(); }
  inline typename Pieces::iterator end() { return pieces.end(); }
  inline void reserve(const int &n) { pieces.reserve(n); }
  inline void emplace_back(const Piece<D> &piece) { pieces.emplace_back(piece); }
  template <class CMat>
  inline void emplace_back(const double &dur, const CMat &cMat) { pieces.emplace_back(dur, cMat); }
  inline void append(const Trajectory<D> &traj) { pieces.insert(pieces.end(), traj.begin(), traj.end()); }

  // trajectory.hpp:496-514 (t is modified in place, like the reference)
  inline int locatePieceIdx(double &t) const {
    int N = getPieceNum();
    int idx;
    double dur;
    for (idx = 0; idx < N && t > (dur = pieces[idx].getDuration()); idx++) t -= dur;
    if (idx == N) {
      idx--;
      t += pieces[idx].getDuration();
    }
    return idx;
  }

  // batched evaluation of one trajectory at many times (one kernel launch)
  inline std::vector<anet::Vec3> evaluate(const std::vector<double> &ts, int deriv) const {
    std::vector<double> co, T;
    flatten(co, T);
    std::vector<double> out(ts.size() * 3);
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_traj_eval(ctx.get(), (D + 1) / 2, getPieceNum(), 1, co.data(), T.data(), (int)ts.size(),
                             ts.data(), deriv, out.data()));
    std::vector<anet::Vec3> r(ts.size());
    for (size_t i = 0; i < ts.size(); ++i) r[i] = anet::Vec3(out[3 * i], out[3 * i + 1], out[3 * i + 2]);
    return r;
  }
  inline anet::Vec3 getPos(double t) const { return evaluate({t}, 0)[0]; }
  inline anet::Vec3 getVel(double t) const { return evaluate({t}, 1)[0]; }
  inline anet::Vec3 getAcc(double t) const { return evaluate({t}, 2)[0]; }
  inline anet::Vec3 getJer(double t) const { return evaluate({t}, 3)[0]; }

  inline anet::Vec3 getJuncPos(int juncIdx) const {
    if (juncIdx != getPieceNum()) return pieces[juncIdx].getCoeffMat().col(D);
    return pieces[juncIdx - 1].getPos(pieces[juncIdx - 1].getDuration());
  }
  inline anet::Vec3 getJuncVel(int juncIdx) const {
    if (juncIdx != getPieceNum()) return pieces[juncIdx].getCoeffMat().col(D - 1);
    return pieces[juncIdx - 1].getVel(pieces[juncIdx - 1].getDuration());
  }
  inline anet::Vec3 getJuncAcc(int juncIdx) const {
    if (juncIdx != getPieceNum()) {
      anet::Vec3 a = pieces[juncIdx].getCoeffMat().col(D - 2);
      return anet::Vec3(2.0 * a(0), 2.0 * a(1), 2.0 * a(2));
    }
    return pieces[juncIdx - 1].getAcc(pieces[juncIdx - 1].getDuration());
  }
  // trajectory.hpp:576-630: maximum over the pieces (one kernel launch for the whole trajectory)
  inline double maxRate(int which) const {
    std::vector<double> co, T;
    flatten(co, T);
    std::vector<double> r(T.size());
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_traj_max_rate(ctx.get(), (D + 1) / 2, getPieceNum(), 1, co.data(), T.data(), which, r.data()));
    double m = 0.0;
    for (double v : r) m = v > m ? v : m;
    return m;
  }
  inline double getMaxVelRate() const { return maxRate(1); }
  inline double getMaxAccRate() const { return maxRate(2); }
  inline bool checkMaxVelRate(const double &maxVelRate) const { return getMaxVelRate() < maxVelRate; }
  inline bool checkMaxAccRate(const double &maxAccRate) const { return getMaxAccRate() < maxAccRate; }

  // 3 x (N+1) junction positions (trajectory.hpp:440-450), column k = junction k
  inline std::vector<anet::Vec3> getPositions() const {
    int N = getPieceNum();
    std::vector<anet::Vec3> p(N + 1);
    for (int i = 0; i < N; i++) p[i] = pieces[i].getCoeffMat().col(D);
    p[N] = pieces[N - 1].getPos(pieces[N - 1].getDuration());
    return p;
  }
};
