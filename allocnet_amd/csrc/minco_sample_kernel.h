// Time-allocation sampling: MANY candidate duration vectors for FEW trajectory problems in one launch.
//
// The literal BASELINE configs[1] batch (1024 trajectories) is launch-bound -- 2 MB per launch -- and a sampler of time
// allocations (the north star's "batch of candidate trajectories / time-allocation samples") does not have 1024 different
// problems, it has one problem (boundary states, waypoints) and K candidate duration vectors.  Replicating the problem K
// times to use k_minco_solve would stream 3c + 3c + 3(N-1) doubles per sample that are all the same.  Here lane = sample,
// sample b belongs to problem b / group: the durations are read per sample (coalesced), the boundary states and waypoints
// per PROBLEM (a wave touches at most two problems: the loads collapse to one or two addresses), and only the cost
// energy + rho * sum T comes back -- 8 (N + 1) bytes per sample instead of 1920, so the launch is bound by its FP64 work.
#pragma once
#include "minco_kernels.h"

namespace anet {

struct SampleArgs {
  const double *head, *tail, *wps;  // per problem: [3c][ldp], [3c][ldp], [(N-1)*3][ldp]
  const double *T;                  // per sample:  [N][ld]
  double *cost;                     // [B] energy + rho * sum T
  int64_t B, ld, ldp, group;        // B = problems * group samples
  int N, c;
  double rho;
};

template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_sample(SampleArgs a) {
  constexpr int m = S - 1;
  const int64_t b = (int64_t)blockIdx.x * kSolveBlock + threadIdx.x;
  if (b >= a.B) return;
  const int64_t p = a.group > 1 ? b / a.group : b;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int c = np + 1;
  const int64_t ld = a.ld, ldp = a.ldp;

  Factor<S, NB> F;
  double tt[NB], tsum = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i) tt[i] = a.T[(int64_t)(i < N ? i : 0) * ld + b];
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) {
      F.r[i] = fast_rcp(tt[i]);
      tsum += tt[i];
    }
  F.factorize(N, np);

  double etot = 0.0;
#pragma unroll 1
  for (int ax = 0; ax < 3; ++ax) {
    double P[NB + 1], hv[m], tv[m], X[NB + 1][m];
    const double *hp = a.head + (int64_t)(ax * c) * ldp + p;
    const double *tp = a.tail + (int64_t)(ax * c) * ldp + p;
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      const double *src = (k == 0) ? hp : (k < N) ? a.wps + (int64_t)((k - 1) * 3 + ax) * ldp + p : tp;
      const double v = *src;
      P[k] = (k <= N) ? v : 0.0;
    }
#pragma unroll
    for (int j = 0; j < m; ++j) {
      const int64_t row = (j < np) ? 1 + j : 0;
      const double h = hp[row * ldp], t = tp[row * ldp];
      hv[j] = (j < np) ? h : 0.0;
      tv[j] = (j < np) ? t : 0.0;
    }
    etot += solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int, int, double) {});
  }
  a.cost[b] = __builtin_fma(a.rho, tsum, etot);
}

}  // namespace anet
