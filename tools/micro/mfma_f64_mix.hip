// Microbenchmark: what the FP64 matrix pipe of gfx950 can take off the vector pipe -- the question behind "the constant
// basis-table contractions of the cost + gradient sample loop on MFMA" (VERDICT round 5, next-round item 2).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_f64_mix.hip -o /tmp/mfma_f64_mix && /tmp/mfma_f64_mix
// Per wave and iteration: NM independent v_mfma_f64_16x16x4_f64 (or v_mfma_f64_4x4x4_4b_f64) on accumulators of their own and NV
// independent v_fma_f64 chains, both in ONE instruction stream; 1 / 2 waves per SIMD (k_piece_grad runs two).  Rows:
//   MFMA alone   -> cycles per MFMA per SIMD (the matrix pipe's issue interval)
//   VALU alone   -> cycles per v_fma_f64 per SIMD
//   mixed        -> does the sum or the maximum of the two come out?  (one wave: both streams share an issue port; two waves:
//                   the other wave's VALU can run under this wave's MFMA)
// A 16x16x4 FP64 MFMA is 1024 FMAs = 16 wave-wide v_fma_f64; a 4x4x4 (4 blocks) is 256 FMAs = 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, bool SMALL>
__global__ void __launch_bounds__(64) k_mix(double *out, int iters, double a, double b) {
  d4 acc[NM > 0 ? NM : 1];
  double s1[NM > 0 ? NM : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
    acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    s1[i] = 0.0;
  }
  double v[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = (double)(threadIdx.x + i);
  const double x = a + 1e-9 * threadIdx.x, y = b + 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // interleaved by hand: one MFMA, then its share of the VALU instructions
#pragma unroll
      for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
        if constexpr (NM > 0) {
          if (i < NM) {
            if constexpr (SMALL) s1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s1[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
          }
        }
        if constexpr (NV > 0) {
          if (i < NV) v[i] = __builtin_fma(v[i], a, b);
        }
      }
      // the order the scheduler must keep: one MFMA, then NV / NM vector FMAs, NM times (mask 0x8 = MFMA, 0x2 = VALU)
      if constexpr (NM > 0 && NV > 0) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2, NV / NM, 0);
        }
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s1[i];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  if (s == 12345.678) out[0] = s;
}

template <int NM, int NV, bool SMALL>
static int run(double *d_out, int waves_per_simd, int n_cu, double clock_ghz) {
  const int iters = 2048;
  const int blocks = n_cu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NM, NV, SMALL>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 0.999999, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 1 && ms < best) best = ms;
  }
  const double per_simd_cycles = best * 1e-3 * clock_ghz * 1e9;                    // cycles the SIMD was busy
  const double n_mfma = (double)iters * 4 * NM * waves_per_simd, n_valu = (double)iters * 4 * NV * waves_per_simd;
  printf("%s  MFMA/it %2d  VALU/it %2d  waves/SIMD %d : %7.3f ms", SMALL ? " 4x4x4 " : "16x16x4", NM, NV, waves_per_simd, best);
  if (NM > 0 && NV == 0) printf("   %.1f cycles per MFMA per SIMD", per_simd_cycles / n_mfma);
  if (NV > 0 && NM == 0) printf("   %.2f cycles per v_fma_f64 per SIMD", per_simd_cycles / n_valu);
  if (NM > 0 && NV > 0) printf("   %.0f k cycles in all", per_simd_cycles / 1e3);
  printf("\n");
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate / 1e6;
  printf("%s: %d CUs, %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1e3);
  double *d_out;
  CK(hipMalloc(&d_out, 64));
  for (int w : {1, 2}) {
    if (run<4, 0, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<4, 0, true>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<0, 16, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<0, 32, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    // 16x16x4: one MFMA replaces 16 v_fma_f64; mixes of 1 MFMA : 4 / 8 / 16 VALU
    if (run<4, 16, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<4, 32, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<2, 32, false>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    // 4x4x4: one MFMA replaces 4 v_fma_f64; mixes of 1 : 2 / 4 / 8
    if (run<8, 16, true>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<8, 32, true>(d_out, w, p.multiProcessorCount, ghz)) return 1;
    if (run<4, 32, true>(d_out, w, p.multiProcessorCount, ghz)) return 1;
  }
  return 0;
}
