// Source-compatible stand-ins for the corridor half of the reference's sfc_gen
// (src/planner/include/gcopter/sfc_gen.hpp:116-226), as LearningPlanner calls it (learning_planner.hpp:274-283):
//     sfc_gen::convexCover(path, points, lowCorner, highCorner, progress, range, hpolys, eps = 1.0e-6);
//     sfc_gen::shortCut(hpolys);
// convexCover's segments are independent, so all its firi::firi calls go to the MI355X as ONE anet_firi batch (plus
// one more for the gap polytopes); shortCut's overlap tests of all pairs go as ONE anet_polytope_depth batch and the
// walk over the answers stays on the host.  sfc_gen::planPath (OMPL's RRT*) is out of scope.
// Vector / matrix arguments are duck-typed: points need (i) access, polytopes (r,c), rows() and resize(r,c).
#pragma once
#include <algorithm>
#include <cmath>
#include <deque>
#include <vector>

#include "firi.hpp"
#include "geo_utils.hpp"

namespace sfc_gen {

template <typename V3, typename Poly>
inline void convexCover(const std::vector<V3> &path, const std::vector<V3> &points, const V3 &lowCorner,
                        const V3 &highCorner, const double &progress, const double &range, std::vector<Poly> &hpolys,
                        const double eps = 1.0e-6) {
  hpolys.clear();
  const int n = (int)path.size();
  if (n < 2) return;
  // the walk (sfc_gen.hpp:135-148): segments (a, b) of at most `progress`
  std::vector<double> A, Bv;
  double b[3] = {path[0](0), path[0](1), path[0](2)};
  for (int i = 1; i < n;) {
    const double a[3] = {b[0], b[1], b[2]};
    const double d[3] = {path[i](0) - a[0], path[i](1) - a[1], path[i](2) - a[2]};
    const double nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nrm > progress) {
      for (int c = 0; c < 3; ++c) b[c] = d[c] / nrm * progress + a[c];
    } else {
      for (int c = 0; c < 3; ++c) b[c] = path[i](c);
      i++;
    }
    A.insert(A.end(), a, a + 3);
    Bv.insert(Bv.end(), b, b + 3);
  }
  const int S = (int)(A.size() / 3);
  // bounding boxes and the obstacle points inside each (sfc_gen.hpp:150-165)
  std::vector<double> bd((size_t)S * 24, 0.0);
  std::vector<std::vector<double>> sel(S);
  size_t Np = 1;
  for (int k = 0; k < S; ++k) {
    double *q = bd.data() + (size_t)k * 24;
    for (int ax = 0; ax < 3; ++ax) {
      const double a = A[(size_t)k * 3 + ax], bb = Bv[(size_t)k * 3 + ax];
      q[(2 * ax) * 4 + ax] = 1.0;
      q[(2 * ax) * 4 + 3] = -std::fmin(std::fmax(a, bb) + range, highCorner(ax));
      q[(2 * ax + 1) * 4 + ax] = -1.0;
      q[(2 * ax + 1) * 4 + 3] = +std::fmax(std::fmin(a, bb) - range, lowCorner(ax));
    }
    for (const V3 &p : points) {
      bool inside = true;
      for (int r = 0; r < 6 && inside; ++r)
        inside = q[r * 4] * p(0) + q[r * 4 + 1] * p(1) + q[r * 4 + 2] * p(2) + q[r * 4 + 3] < 0.0;
      if (inside)
        for (int c = 0; c < 3; ++c) sel[k].push_back(p(c));
    }
    if (sel[k].size() / 3 > Np) Np = sel[k].size() / 3;
  }
  std::vector<double> pc((size_t)S * Np * 3, 0.0);
  std::vector<int32_t> npts(S);
  for (int k = 0; k < S; ++k) {
    npts[k] = (int32_t)(sel[k].size() / 3);
    for (size_t j = 0; j < sel[k].size(); ++j) pc[(size_t)k * Np * 3 + j] = sel[k][j];
  }
  anet::Context &ctx = anet::Context::thread_default();
  // ONE batch (anet_firi_var): the S segments with firi::firi's default pass count and, speculatively, the S - 1 gap
  // polytopes firi::firi(bd, pc, a, a, gap, 1) of sfc_gen.hpp:171-179 -- one pass each, i.e. one planes kernel; whether a
  // gap polytope is used depends on the segments' results, but computing all of them costs less than a second call.
  const int Bn = 2 * S - 1;
  std::vector<double> sbd((size_t)Bn * 24), spc((size_t)Bn * Np * 3), sa((size_t)Bn * 3), sb((size_t)Bn * 3);
  std::vector<int32_t> sn(Bn), its(Bn), nh(Bn), ok(Bn);
  anet_firi_params prm;
  anet_firi_default_params(&prm);
  for (int q = 0; q < Bn; ++q) {
    const bool gap = q >= S;
    const int k = gap ? q - S + 1 : q;
    std::copy(bd.begin() + (size_t)k * 24, bd.begin() + (size_t)(k + 1) * 24, sbd.begin() + (size_t)q * 24);
    std::copy(pc.begin() + (size_t)k * Np * 3, pc.begin() + (size_t)(k + 1) * Np * 3, spc.begin() + (size_t)q * Np * 3);
    sn[q] = npts[k];
    its[q] = gap ? 1 : prm.iterations;
    for (int c = 0; c < 3; ++c) {
      sa[(size_t)q * 3 + c] = A[(size_t)k * 3 + c];
      sb[(size_t)q * 3 + c] = gap ? A[(size_t)k * 3 + c] : Bv[(size_t)k * 3 + c];
    }
  }
  int cap = 64;
  std::vector<double> hp;
  for (;;) {  // the reference's hPoly can have up to 6 + Np rows; grow on demand
    hp.assign((size_t)Bn * cap * 4, 0.0);
    ctx.check(anet_firi_var(ctx.get(), Bn, 6, (int)Np, cap, sbd.data(), spc.data(), sn.data(), sa.data(), sb.data(), its.data(),
                            &prm, hp.data(), nh.data(), ok.data(), nullptr));
    bool overflow = false;
    for (int q = 0; q < Bn; ++q) overflow |= ok[q] == -1;
    if (!overflow || cap >= 6 + (int)Np) break;
    cap = cap * 4 < 6 + (int)Np ? cap * 4 : 6 + (int)Np;
  }
  auto poly_of = [&](int q) {
    return ok[q] >= 1 ? std::vector<double>(hp.begin() + (size_t)q * cap * 4, hp.begin() + ((size_t)q * cap + nh[q]) * 4)
                      : std::vector<double>();
  };
  std::vector<std::vector<double>> mainp(S), gapp;
  for (int k = 0; k < S; ++k) mainp[k] = poly_of(k);
  // gap polytopes (sfc_gen.hpp:167-179): where the junction point touches 3 or more faces of the two neighbours
  auto touching = [&](const std::vector<double> &h, const double *a) {
    int cnt = 0;
    for (size_t r = 0; r + 3 < h.size(); r += 4) cnt += h[r] * a[0] + h[r + 1] * a[1] + h[r + 2] * a[2] + h[r + 3] > -eps;
    return cnt;
  };
  std::vector<int> need;
  for (int k = 1; k < S; ++k)
    if (3 <= touching(mainp[k], &A[(size_t)k * 3]) + touching(mainp[k - 1], &A[(size_t)k * 3])) {
      need.push_back(k);
      gapp.push_back(poly_of(S + k - 1));
    }
  auto emit = [&](const std::vector<double> &h) {
    Poly P;
    const int m = (int)(h.size() / 4);
    P.resize(m, 4);
    for (int r = 0; r < m; ++r)
      for (int c = 0; c < 4; ++c) P(r, c) = h[(size_t)r * 4 + c];
    hpolys.emplace_back(P);
  };
  size_t g = 0;
  for (int k = 0; k < S; ++k) {
    if (g < need.size() && need[g] == k) emit(gapp[g++]);
    emit(mainp[k]);
  }
}

template <typename Poly>
inline void shortCut(std::vector<Poly> &hpolys) {
  std::vector<Poly> htemp = hpolys;
  if (htemp.size() == 1) htemp.insert(htemp.begin(), htemp.front());
  hpolys.clear();
  const int M = (int)htemp.size();
  if (M == 0) return;
  // overlap(htemp[i], htemp[j], 0.1) for every pair j < i - 1, as one batch (sfc_gen.hpp:204-211 asks for them lazily)
  int H = 1;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j + 1 < i; ++j) H = std::max(H, (int)(htemp[i].rows() + htemp[j].rows()));
  std::vector<double> rows, depth;
  std::vector<int> first(M, 0);  // pair (i, j) lives at first[i] + j
  int64_t P = 0;
  for (int i = 0; i < M; ++i) {
    first[i] = (int)P;
    for (int j = 0; j + 1 < i; ++j) {
      std::vector<double> r;
      geo_utils::detail::append_rows(htemp[i], r);
      geo_utils::detail::append_rows(htemp[j], r);
      r.resize((size_t)H * 4, 0.0);
      rows.insert(rows.end(), r.begin(), r.end());
      ++P;
    }
  }
  depth.assign((size_t)(P > 0 ? P : 1), 0.0);
  if (P > 0) {
    anet::Context &ctx = anet::Context::thread_default();
    ctx.check(anet_polytope_depth(ctx.get(), P, H, rows.data(), 0, depth.data(), nullptr));
  }
  std::deque<int> idices;
  idices.push_front(M - 1);
  for (int i = M - 1; i > 0;) {
    int j = 0;
    for (; j < i - 1; ++j) {
      const double d = depth[(size_t)first[i] + j];
      if (d > 0.1 && !std::isinf(d)) break;
    }
    idices.push_front(j);  // j == i - 1 when nothing earlier overlaps: consecutive polytopes always count
    i = j;
  }
  for (const auto &ele : idices) hpolys.push_back(htemp[ele]);
}

}  // namespace sfc_gen
