// Microbenchmark: the FP64 vector FMA rate the chip sustains (no MFMA), the ceiling the compute-bound kernels
// (k_piece_grad, the L-BFGS update, the one-launch L-BFGS) are reported against.  MI355X_MICROARCH.md carries no
// FP64 vector peak; the nominal figure is 256 CU x 4 SIMD x 16 lanes x 2 flop x clock (78.6 TFLOP/s at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_peak.hip -o /tmp/fp64_peak && /tmp/fp64_peak
// Rows: independent FMA chains per lane (ILP) x waves per SIMD.  One wave issues at most one instruction every four
// cycles and a dependent FMA waits for its predecessor, so the peak needs either ILP >= 2 or >= 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP>
__global__ void __launch_bounds__(64) k_fma(double *out, int iters, double a, double b) {
  double acc[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) acc[i] = (double)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += acc[i];
  if (s == 12345.678) out[0] = s;  // keep the chains alive
}

template <int ILP>
static int run(double *d_out, int waves_per_simd, int n_cu) {
  const int iters = 4096;
  const int blocks = n_cu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_fma<ILP>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 0.999999, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 1 && ms < best) best = ms;
  }
  const double flop = 2.0 * 64.0 * blocks * (double)iters * 8.0 * ILP;
  const double cyc_per_fma_wave = best * 1e-3 / ((double)iters * 8.0 * ILP * waves_per_simd);  // seconds per wave-FMA per SIMD
  printf("ILP %2d  waves/SIMD %d : %7.3f ms  %6.2f TFLOP/s  (%.2f ns per wave-instruction per SIMD)\n", ILP, waves_per_simd, best,
         flop / best / 1e9, cyc_per_fma_wave * 1e9);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s: %d CUs, clock %.0f MHz -> nominal FP64 vector FMA peak %.1f TFLOP/s\n", p.gcnArchName, p.multiProcessorCount,
         p.clockRate / 1e3, p.multiProcessorCount * 4.0 * 16.0 * 2.0 * p.clockRate * 1e3 / 1e12);
  double *d_out;
  CK(hipMalloc(&d_out, 64));
  for (int w : {1, 2, 4, 8}) {
    if (run<1>(d_out, w, p.multiProcessorCount)) return 1;
    if (run<2>(d_out, w, p.multiProcessorCount)) return 1;
    if (run<4>(d_out, w, p.multiProcessorCount)) return 1;
    if (run<16>(d_out, w, p.multiProcessorCount)) return 1;
  }
  return 0;
}
