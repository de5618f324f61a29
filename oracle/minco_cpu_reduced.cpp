/*
 * CPU BASELINE ONLY -- not an oracle and not a product path.
 *
 * bench.py's `cpu_baseline` leg times the host cores on the same workload as the GPU.  The oracle's solve
 * (minco_oracle.c) is the classic banded-LU formulation: ~15 kFLOP per 8-segment snap trajectory, a 128 x 128 scratch
 * matrix cleared per solve -- a fair restatement, a weak baseline.  This file compiles the ALGORITHM OF THE KERNELS for
 * the host instead: the reduced Hermite / block-tridiagonal form of allocnet_amd/csrc/minco_core.h (~4.1 kFLOP), the very
 * header the HIP kernels are built from, with the three device-only spellings it uses mapped to their host equivalents.
 * It shares the arithmetic with the product by construction, so it proves nothing about parity (the checker is
 * minco_oracle.c / minco_np.py); it only makes the stated CPU number like-for-like.  One trajectory per loop iteration,
 * pthreads over contiguous slices of the batch, trajectory-major arrays as in oracle_minco_solve_batch.
 * Only bench.py (cpu_baseline) and tests/ may load it.
 */
#define HIP_INCLUDE_HIP_HIP_RUNTIME_H /* keep the HIP runtime header out of a host build */
#define __device__
#define __forceinline__ inline __attribute__((always_inline))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include "../allocnet_amd/csrc/minco_core.h"

namespace {

template <int S, int NB>
void solve_range(int c, int N, int64_t b0, int64_t b1, const double *head, const double *tail, const double *wps,
                 const double *T, double *coeffs, double *energy) {
  constexpr int m = S - 1, D = 2 * S;
  const int np = c - 1;
  for (int64_t b = b0; b < b1; ++b) {
    anet::Factor<S, NB> F;
    for (int i = 0; i < NB; ++i) F.r[i] = (i < N) ? 1.0 / T[b * N + i] : 0.0;
    F.factorize(N, np);
    double etot = 0.0;
    for (int ax = 0; ax < 3; ++ax) {
      double P[NB + 1], hv[m], tv[m], X[NB + 1][m];
      const double *hp = head + (b * 3 + ax) * c, *tp = tail + (b * 3 + ax) * c;
      for (int k = 0; k <= NB; ++k) P[k] = (k == 0) ? hp[0] : (k < N) ? wps[(b * (N - 1) + (k - 1)) * 3 + ax] : (k == N) ? tp[0] : 0.0;
      for (int j = 0; j < m; ++j) {
        hv[j] = (j < np) ? hp[1 + j] : 0.0;
        tv[j] = (j < np) ? tp[1 + j] : 0.0;
      }
      double *cp = coeffs ? coeffs + ((b * N) * 3 + ax) * D : nullptr;
      etot += anet::solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int piece, int col, double v) {
        if (cp) cp[(int64_t)piece * 3 * D + col] = v;
      });
    }
    if (energy) energy[b] = etot;
  }
}

struct Job {
  int s, c, N;
  int64_t b0, b1;
  const double *head, *tail, *wps, *T;
  double *coeffs, *energy;
};

void *run(void *p) {
  const Job &j = *static_cast<Job *>(p);
#define GO(S, NB) solve_range<S, NB>(j.c, j.N, j.b0, j.b1, j.head, j.tail, j.wps, j.T, j.coeffs, j.energy)
  if (j.s == 4) { if (j.N <= 8) GO(4, 8); else GO(4, 16); }
  else if (j.s == 3) { if (j.N <= 8) GO(3, 8); else GO(3, 16); }
  else { if (j.N <= 8) GO(2, 8); else GO(2, 16); }
#undef GO
  return nullptr;
}

}  // namespace

extern "C" int cpu_reduced_minco_solve_batch(int s, int c, int N, int64_t B, const double *head, const double *tail,
                                             const double *wps, const double *T, double *coeffs, double *energy,
                                             int nthreads) {
  if (s < 2 || s > 4 || c < 1 || c > s || N < 1 || N > 16) return -1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 512) nthreads = 512;
  static Job jobs[512];
  static pthread_t th[512];
  const int64_t per = (B + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; ++t) {
    const int64_t b0 = t * per, b1 = b0 + per < B ? b0 + per : B;
    if (b0 >= b1) break;
    jobs[t] = Job{s, c, N, b0, b1, head, tail, wps, T, coeffs, energy};
    ++used;
  }
  if (used == 1) run(&jobs[0]);
  else {
    for (int t = 0; t < used; ++t) pthread_create(&th[t], nullptr, run, &jobs[t]);
    for (int t = 0; t < used; ++t) pthread_join(th[t], nullptr);
  }
  return 0;
}
