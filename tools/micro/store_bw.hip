// Microbenchmark: what HBM write bandwidth does the batch-minor access pattern reach?
// Each lane owns one trajectory and writes F doubles at [f*ld + b] (8 B/lane, 512 B/wave per
// instruction) vs the pair-interleaved variant [(f/2)*2*ld + 2*b + f%2] (16 B/lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int F, int FIN>
__global__ void __launch_bounds__(64) k_w8(const double* __restrict__ in, double* __restrict__ out, long B, long ld) {
  long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double acc = 0;
#pragma unroll
  for (int f = 0; f < FIN; ++f) acc += in[f * ld + b];
#pragma unroll
  for (int f = 0; f < F; ++f) out[f * ld + b] = acc + f;
}
template <int F, int FIN>
__global__ void __launch_bounds__(64) k_w16(const double2* __restrict__ in, double2* __restrict__ out, long B, long ld) {
  long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double acc = 0;
#pragma unroll
  for (int f = 0; f < FIN / 2; ++f) { double2 v = in[f * ld + b]; acc += v.x + v.y; }
#pragma unroll
  for (int f = 0; f < F / 2; ++f) out[f * ld + b] = make_double2(acc + f, acc - f);
}
// same but 256-thread blocks, grid-stride persistent
template <int F, int FIN>
__global__ void __launch_bounds__(256) k_w8p(const double* __restrict__ in, double* __restrict__ out, long B, long ld) {
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
    double acc = 0;
#pragma unroll
    for (int f = 0; f < FIN; ++f) acc += in[f * ld + b];
#pragma unroll
    for (int f = 0; f < F; ++f) out[f * ld + b] = acc + f;
  }
}

int main() {
  const long B = 1 << 20, ld = B;
  constexpr int F = 192, FIN = 48;
  double *in, *out;
  CK(hipMalloc(&in, sizeof(double) * FIN * ld));
  CK(hipMalloc(&out, sizeof(double) * F * ld));
  CK(hipMemset(in, 0, sizeof(double) * FIN * ld));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(F + FIN) * 8 * B;
  for (int variant = 0; variant < 3; ++variant) {
    float best = 1e9;
    for (int it = 0; it < 12; ++it) {
      CK(hipEventRecord(e0));
      if (variant == 0) hipLaunchKernelGGL((k_w8<F, FIN>), dim3(B / 64), dim3(64), 0, 0, in, out, B, ld);
      if (variant == 1) hipLaunchKernelGGL((k_w16<F, FIN>), dim3(B / 64), dim3(64), 0, 0, (const double2*)in, (double2*)out, B, ld);
      if (variant == 2) hipLaunchKernelGGL((k_w8p<F, FIN>), dim3(256 * 8), dim3(256), 0, 0, in, out, B, ld);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2 && ms < best) best = ms;
    }
    const char* names[] = {"8B/lane  [f][b]      ", "16B/lane [f/2][b][2] ", "8B/lane persistent   "};
    printf("%s best %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)\n", names[variant], best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
  }
  return 0;
}
