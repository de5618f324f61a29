// Which throughput regime of the headline solve ("Row stride": ~65 % or ~73-75 % of 8 TB/s, fixed
// per process) does a fresh process land in, as a function of HOW the 1.5 GB coefficient buffer is allocated?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/regime_alloc.hip -I include -L allocnet_amd/lib -lallocnet_amd \
//         -Wl,-rpath,$PWD/allocnet_amd/lib -o tools/micro/regime_alloc
//   tools/micro/regime_alloc <variant>        (tools/regime_alloc_probe.sh runs every variant in fresh processes)
// variants: 0 hipMalloc (inputs first)   1 hipMalloc, output FIRST in the process   2 size rounded up to 1 GiB
//           3 hipExtMallocWithFlags(uncached)   4 hipExtMallocWithFlags(fine grained)   5 hipMallocAsync (pool)
//           6 virtual memory API, 2 MiB-granular physical chunks mapped contiguously
//           7 virtual memory API, ONE physical allocation   8 hipMallocManaged + coarse-grain advice + prefetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "allocnet_amd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("variant failed: %s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

__global__ void k_fill(double *p, size_t n, double lo, double hi, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ seed;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  p[i] = lo + (hi - lo) * (double)(x & 0xffffff) / 16777216.0;
}

static int vmm_alloc(double **out, size_t bytes, bool chunked) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t align = gran > ((size_t)2 << 20) ? gran : ((size_t)2 << 20);
  const size_t total = (bytes + align - 1) / align * align;
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, total, align, nullptr, 0));
  const size_t two_mib = (size_t)2 << 20;
  const size_t chunk = chunked ? (gran > two_mib ? gran : two_mib) : total;
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char *)va + off, chunk, 0, h, 0));
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  printf("(granularity %zu KiB, %zu chunks) ", gran >> 10, total / chunk);
  *out = (double *)va;
  return 0;
}

int main(int argc, char **argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const int64_t B = 1 << 20, ld = anet_recommended_ld(B);
  const int s = 4, c = 3, N = 8, D = 8;
  const size_t n_out = (size_t)N * 3 * D * ld;
  CK(hipSetDevice(0));
  double *co = nullptr;
  auto alloc_out = [&]() -> int {
    const size_t bytes = n_out * sizeof(double);
    switch (variant) {
      case 2: CK(hipMalloc((void **)&co, (bytes + (1ull << 30) - 1) >> 30 << 30)); break;
      case 3: CK(hipExtMallocWithFlags((void **)&co, bytes, hipDeviceMallocUncached)); break;
      case 4: CK(hipExtMallocWithFlags((void **)&co, bytes, hipDeviceMallocFinegrained)); break;
      case 5: CK(hipMallocAsync((void **)&co, bytes, 0)); CK(hipStreamSynchronize(0)); break;
      case 6: return vmm_alloc(&co, bytes, true);
      case 7: return vmm_alloc(&co, bytes, false);
      case 8:
        CK(hipMallocManaged((void **)&co, bytes));
        CK(hipMemAdvise(co, bytes, hipMemAdviseSetCoarseGrain, 0));
        CK(hipMemPrefetchAsync(co, bytes, 0, 0));
        CK(hipStreamSynchronize(0));
        break;
      default: CK(hipMalloc((void **)&co, bytes)); break;
    }
    return 0;
  };
  if (variant == 1 && alloc_out()) return 2;
  anet_ctx *ctx = nullptr;
  if (anet_create(0, &ctx)) { printf("anet_create failed\n"); return 1; }
  double *head, *tail, *wps, *T, *en;
  CK(hipMalloc((void **)&head, sizeof(double) * 3 * c * ld));
  CK(hipMalloc((void **)&tail, sizeof(double) * 3 * c * ld));
  CK(hipMalloc((void **)&wps, sizeof(double) * 3 * (N - 1) * ld));
  CK(hipMalloc((void **)&T, sizeof(double) * N * ld));
  CK(hipMalloc((void **)&en, sizeof(double) * ld));
  if (variant != 1 && alloc_out()) return 2;
  auto fill = [&](double *p, size_t n, double lo, double hi, unsigned seed) {
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, lo, hi, seed);
  };
  fill(head, (size_t)3 * c * ld, -1, 1, 1); fill(tail, (size_t)3 * c * ld, 4, 6, 2);
  fill(wps, (size_t)3 * (N - 1) * ld, 0, 5, 3); fill(T, (size_t)N * ld, 0.5, 2.0, 4);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i)
    if (anet_minco_solve_dev(ctx, s, c, N, B, ld, head, tail, wps, T, co, en, nullptr)) { printf("solve failed: %s\n", anet_last_error(ctx)); return 1; }
  CK(hipDeviceSynchronize());
  const int K = 40;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < K; ++i) anet_minco_solve_dev(ctx, s, c, N, B, ld, head, tail, wps, T, co, en, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= K;
  printf("variant %d: %.4f ms  %.1f %% of 8 TB/s\n", variant, ms, (double)B * 1920.0 / (ms * 1e-3) / 8e12 * 100.0);
  return 0;
}
