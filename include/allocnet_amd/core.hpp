// allocnet_amd C++ facade, common pieces: RAII context, error -> exception, tiny fixed-size
// matrices that interoperate with Eigen BY DUCK TYPING (any type with operator()(r,c) / (i) and,
// for conversions back, a (x,y,z) or default constructor) so that <Eigen/Eigen> is not needed to
// build against this header and the reference's Eigen-typed call sites still compile.
#pragma once
#include <array>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../allocnet_amd.h"

namespace anet {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// One context per host thread per device (same re-entrancy contract as the reference's QPSolver,
// planner/qp_solver.hpp:43-45).
class Context {
 public:
  explicit Context(int device = 0) {
    int rc = anet_create(device, &h_);
    if (rc != ANET_OK) throw Error(rc, anet_last_error(nullptr));
  }
  ~Context() {
    if (ws_) anet_dev_free(ws_);
    anet_destroy(h_);
  }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  anet_ctx *get() const { return h_; }
  void check(int rc) const {
    if (rc != ANET_OK) throw Error(rc, anet_last_error(h_));
  }
  static Context &thread_default() {
    static thread_local Context ctx(0);
    return ctx;
  }
  // Grow-only device workspace of this context for the facade calls that need one (lbfgs_optimize_batched): allocated on first
  // use, kept across calls (a receding-horizon caller pays no hipMalloc / hipFree per plan), released with the context.  The
  // caller must not hold it across another facade call that asks for one.
  double *workspace(size_t n_doubles) {
    if (n_doubles > ws_n_) {
      if (ws_) anet_dev_free(ws_);
      ws_ = nullptr;
      ws_n_ = 0;
      check(anet_dev_alloc(h_, n_doubles, &ws_));
      ws_n_ = n_doubles;
    }
    return ws_;
  }

 private:
  anet_ctx *h_ = nullptr;
  double *ws_ = nullptr;
  size_t ws_n_ = 0;
};

struct Vec3 {
  double v[3] = {0.0, 0.0, 0.0};
  Vec3() = default;
  Vec3(double x, double y, double z) : v{x, y, z} {}
  template <class V, class = decltype(std::declval<const V &>()(0))>
  Vec3(const V &o) : v{o(0), o(1), o(2)} {}
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  const double *data() const { return v; }
  double squaredNorm() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
  // e.g. Eigen::Vector3d p = traj.getPos(t);
  template <class V, class = typename std::enable_if<std::is_constructible<V, double, double, double>::value &&
                                                     !std::is_same<V, Vec3>::value>::type>
  operator V() const {
    return V(v[0], v[1], v[2]);
  }
};

// Row-major R x C block of doubles.
template <int R, int C>
struct Matrix {
  std::array<double, (size_t)R * C> a{};
  Matrix() = default;
  template <class M, class = decltype(std::declval<const M &>()(0, 0))>
  Matrix(const M &m) {
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) a[(size_t)r * C + c] = m(r, c);
  }
  static Matrix Zero() { return Matrix(); }
  double &operator()(int r, int c) { return a[(size_t)r * C + c]; }
  double operator()(int r, int c) const { return a[(size_t)r * C + c]; }
  Vec3 col(int c) const {
    static_assert(R == 3, "col() is defined for 3-row matrices");
    return Vec3((*this)(0, c), (*this)(1, c), (*this)(2, c));
  }
  constexpr int rows() const { return R; }
  constexpr int cols() const { return C; }
  const double *data() const { return a.data(); }
  double *data() { return a.data(); }
  // fill any matrix type M with (r,c) access and a default constructor of the right size
  template <class M>
  M as() const {
    M m;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) m(r, c) = (*this)(r, c);
    return m;
  }
};

// Row-major rows x cols block of doubles with run-time shape (what the reference returns as Eigen::MatrixXd).
struct MatrixX {
  int r_ = 0, c_ = 0;
  std::vector<double> a;
  MatrixX() = default;
  MatrixX(int r, int c) : r_(r), c_(c), a((size_t)r * c, 0.0) {}
  double &operator()(int r, int c) { return a[(size_t)r * c_ + c]; }
  double operator()(int r, int c) const { return a[(size_t)r * c_ + c]; }
  int rows() const { return r_; }
  int cols() const { return c_; }
  const double *data() const { return a.data(); }
  // fill any matrix type M constructible from (rows, cols) with (r,c) access (Eigen::MatrixXd among them)
  template <class M>
  M as() const {
    M m(r_, c_);
    for (int r = 0; r < r_; ++r)
      for (int c = 0; c < c_; ++c) m(r, c) = (*this)(r, c);
    return m;
  }
};

}  // namespace anet
