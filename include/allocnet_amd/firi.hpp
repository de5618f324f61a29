// Source-compatible stand-in for firi::firi of the reference (src/planner/include/gcopter/firi.hpp:268-416)
// as sfc_gen::convexCover calls it (gcopter/sfc_gen.hpp:163,176):
//     bool firi::firi(bd, pc, a, b, hPoly, iterations = 4, epsilon = 1.0e-6);
// bd  : M x 4 rows h with h.[x;1] <= 0;  pc : 3 x N obstacle points (column = point);  a, b : the segment;
// hPoly : resized to nH x 4, same raw form.  The work runs on the MI355X behind anet_firi (one corridor
// here; anet_firi itself is batched -- convexCover's segments are independent, see allocnet_amd/firi.py).
// Matrix arguments are duck-typed ((r,c) access, rows(), cols(), resize(r,c)): Eigen types work unchanged.
#pragma once
#include <stdio.h>
#include <vector>

#include "core.hpp"

namespace firi {

template <typename Bd, typename Pc, typename VA, typename VB, typename Poly>
inline bool firi(const Bd &bd, const Pc &pc, const VA &a, const VB &b, Poly &hPoly, const int iterations = 4,
                 const double epsilon = 1.0e-6) {
  const int M = (int)bd.rows(), N = (int)pc.cols();
  std::vector<double> hbd((size_t)M * 4), hpc((size_t)(N > 0 ? N : 1) * 3), ha(3), hb(3);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < 4; ++c) hbd[(size_t)r * 4 + c] = bd(r, c);
  for (int j = 0; j < N; ++j)
    for (int c = 0; c < 3; ++c) hpc[(size_t)j * 3 + c] = pc(c, j);
  for (int c = 0; c < 3; ++c) {
    ha[c] = a(c);
    hb[c] = b(c);
  }
  anet_firi_params prm;
  anet_firi_default_params(&prm);
  prm.iterations = iterations;
  prm.epsilon = epsilon;
  anet::Context &ctx = anet::Context::thread_default();
  int32_t np = N, nh = 0, ok = 0;
  int cap = 64;
  std::vector<double> hp;
  for (;;) {  // the reference's hPoly can have up to M + N rows; grow on demand
    hp.assign((size_t)cap * 4, 0.0);
    ctx.check(anet_firi(ctx.get(), 1, M, N, cap, hbd.data(), N > 0 ? hpc.data() : nullptr, N > 0 ? &np : nullptr,
                        ha.data(), hb.data(), &prm, hp.data(), &nh, &ok, nullptr));
    if (ok != -1 || cap >= M + N) break;
    cap = cap * 4 < M + N ? cap * 4 : M + N;
  }
  if (ok < 1) return false;
  if (ok == 2) printf("FIRI WARNING: an MVIE optimisation stopped at its evaluation budget\n");  // cf. firi.hpp:229-232
  hPoly.resize(nh, 4);
  for (int r = 0; r < nh; ++r)
    for (int c = 0; c < 4; ++c) hPoly(r, c) = hp[(size_t)r * 4 + c];
  return true;
}

}  // namespace firi
