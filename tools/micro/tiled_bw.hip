// Microbenchmark: batch-minor [f][ld] vs wave-tiled [b/64][f][64] layout for the solve's traffic
// (48 loads + 192 stores of 8 B per lane), stores issued 8 at a time with arithmetic in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int F, int FIN, int WORK, bool TILED>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, long B, long ld) {
  long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const long fs_in = TILED ? 64 : ld, fs_out = TILED ? 64 : ld;
  const double* ip = TILED ? in + (long)blockIdx.x * FIN * 64 + threadIdx.x : in + b;
  double* op = TILED ? out + (long)blockIdx.x * F * 64 + threadIdx.x : out + b;
  double acc = 0;
#pragma unroll
  for (int f = 0; f < FIN; ++f) acc += ip[f * fs_in];
#pragma unroll 1
  for (int g = 0; g < F / 8; ++g) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = acc + q;
#pragma unroll 1
    for (int w = 0; w < WORK; ++w) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = __builtin_fma(v[q], 1.0000001, 0.5);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) op[(g * 8 + q) * fs_out] = v[q];
    acc += v[0];
  }
}

template <int WORK, bool TILED>
int run(const double* in, double* out, long B, long ld, const char* name) {
  constexpr int F = 192, FIN = 48;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 10; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<F, FIN, WORK, TILED>), dim3(B / 64), dim3(64), 0, 0, in, out, B, ld);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2 && ms < best) best = ms;
  }
  const double bytes = (double)(F + FIN) * 8 * B;
  printf("%-40s %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)\n", name, best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
  return 0;
}

int main() {
  const long B = 1 << 20;
  for (int trial = 0; trial < 3; ++trial) {
    double *in, *out, *junk;
    CK(hipMalloc(&junk, (size_t)(trial + 1) * 12345678));
    const long ld = B + (trial == 2 ? 576 : 0);
    CK(hipMalloc(&in, sizeof(double) * 48 * ld));
    CK(hipMalloc(&out, sizeof(double) * 192 * ld));
    CK(hipMemset(in, 0, sizeof(double) * 48 * ld));
    printf("-- trial %d (ld = B + %ld)\n", trial, ld - B);
    run<0, false>(in, out, B, ld, "[f][ld]   no work");
    run<0, true>(in, out, B, ld, "tiled     no work");
    run<25, false>(in, out, B, ld, "[f][ld]   200 FMA / 8 stores");
    run<25, true>(in, out, B, ld, "tiled     200 FMA / 8 stores");
    CK(hipFree(in)); CK(hipFree(out));
  }
  return 0;
}
