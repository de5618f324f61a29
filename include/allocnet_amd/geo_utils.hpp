// Source-compatible stand-ins for the polytope tests of the reference's geo_utils
// (src/planner/include/gcopter/geo_utils.hpp:43-111):
//     bool geo_utils::findInterior(hPoly, interior);
//     bool geo_utils::overlap(hPoly0, hPoly1, eps = 1.0e-6);
//     bool geo_utils::overlapPt(hPoly0, hPoly1, inner_pt, eps = 1.0e-6);     (geo_utils.hpp:88-111: the same test + its point)
// Both are the 4-variable linear programme  max t  s.t.  n.x + t <= -h3  the reference hands to sdlp::linprog<4>;
// here it runs on the MI355X behind anet_polytope_depth (batched; one polytope per call from this header,
// sfc_gen::shortCut in sfc_gen.hpp sends all its pairs at once).  Rows h of a polytope: h.[x;1] <= 0.
// Matrix arguments are duck-typed ((r,c) access, rows()); `interior` needs (i) access: Eigen types work unchanged.
#pragma once
#include <cmath>
#include <vector>

#include "core.hpp"

namespace geo_utils {

namespace detail {
template <typename Poly>
inline void append_rows(const Poly &h, std::vector<double> &out) {
  const int m = (int)h.rows();
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < 4; ++c) out.push_back(h(r, c));
}
}  // namespace detail

template <typename Poly, typename V3>
inline bool findInterior(const Poly &hPoly, V3 &interior) {
  std::vector<double> rows;
  detail::append_rows(hPoly, rows);
  const int m = (int)(rows.size() / 4);
  double depth = -INFINITY, pt[3] = {0.0, 0.0, 0.0};
  anet::Context &ctx = anet::Context::thread_default();
  ctx.check(anet_polytope_depth(ctx.get(), 1, m > 0 ? m : 1, rows.data(), 1, &depth, pt));
  for (int c = 0; c < 3; ++c) interior(c) = pt[c];
  return depth > 0.0 && !std::isinf(depth);
}

template <typename Poly0, typename Poly1>
inline bool overlap(const Poly0 &hPoly0, const Poly1 &hPoly1, const double eps = 1.0e-6) {
  std::vector<double> rows;
  detail::append_rows(hPoly0, rows);
  detail::append_rows(hPoly1, rows);
  const int m = (int)(rows.size() / 4);
  double depth = -INFINITY;
  anet::Context &ctx = anet::Context::thread_default();
  ctx.check(anet_polytope_depth(ctx.get(), 1, m > 0 ? m : 1, rows.data(), 0, &depth, nullptr));
  return depth > eps && !std::isinf(depth);
}

template <typename Poly0, typename Poly1, typename V3>
inline bool overlapPt(const Poly0 &hPoly0, const Poly1 &hPoly1, V3 &inner_pt, const double eps = 1.0e-6) {
  std::vector<double> rows;
  detail::append_rows(hPoly0, rows);
  detail::append_rows(hPoly1, rows);
  const int m = (int)(rows.size() / 4);
  double depth = -INFINITY, pt[3] = {0.0, 0.0, 0.0};
  anet::Context &ctx = anet::Context::thread_default();
  ctx.check(anet_polytope_depth(ctx.get(), 1, m > 0 ? m : 1, rows.data(), 0, &depth, pt));
  for (int c = 0; c < 3; ++c) inner_pt(c) = pt[c];
  return depth > eps && !std::isinf(depth);
}

}  // namespace geo_utils
