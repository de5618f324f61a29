// Trajectory-major <-> batch-minor transposes through padded LDS tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

// dst[f*ld + b] = src[b*nf + f] through a padded LDS tile (both sides coalesced).
constexpr int kTile = 32;
__global__ void __launch_bounds__(kTile * 8) k_to_batch_minor(const double *__restrict__ src,
                                                              double *__restrict__ dst, int64_t B,
                                                              int64_t nf, int64_t ld) {
  __shared__ double tile[kTile][kTile + 1];
  const int64_t b0 = (int64_t)blockIdx.x * kTile, f0 = (int64_t)blockIdx.y * kTile;
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t bb = b0 + i, ff = f0 + threadIdx.x;
    if (bb < B && ff < nf) tile[i][threadIdx.x] = src[bb * nf + ff];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t ff = f0 + i, bb = b0 + threadIdx.x;
    if (bb < B && ff < nf) dst[ff * ld + bb] = tile[threadIdx.x][i];
  }
}
__global__ void __launch_bounds__(kTile * 8) k_to_traj_major(const double *__restrict__ src,
                                                             double *__restrict__ dst, int64_t B,
                                                             int64_t nf, int64_t ld) {
  __shared__ double tile[kTile][kTile + 1];
  const int64_t b0 = (int64_t)blockIdx.x * kTile, f0 = (int64_t)blockIdx.y * kTile;
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t ff = f0 + i, bb = b0 + threadIdx.x;
    if (bb < B && ff < nf) tile[i][threadIdx.x] = src[ff * ld + bb];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < kTile; i += 8) {
    const int64_t bb = b0 + i, ff = f0 + threadIdx.x;
    if (bb < B && ff < nf) dst[bb * nf + ff] = tile[threadIdx.x][i];
  }
}

}  // namespace anet
