// Wave-wide reductions of gfx950 on the DPP path (shared by the L-BFGS kernels and the interior-point QP).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

// Wave-wide reductions on the DPP path: in-row inclusive scan by row_shr 1/2/4/8, row_bcast:15 /
// row_bcast:31 to chain the four rows, total read from lane 63 as a wave-uniform scalar.  About 20 VALU
// ops, where the ds_bpermute butterfly of __shfl_xor costs several hundred cycles per reduction --
// the two-loop recursion is a chain of 2*mem_size dependent reductions.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);  // masked rows / missing sources read 0
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int LANE = 63>
__device__ __forceinline__ double last_lane(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), LANE),
                          __builtin_amdgcn_readlane(__double2loint(v), LANE));
}
// LAST = 15 / 31: the caller guarantees zeros in the lanes above it, so the steps that chain the upper rows would only
// add zeros and are left out (the value is the same; a problem of nine variables saves a third of every reduction).
template <int LAST = 63>
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0x111>(v);
  v += dpp_f64<0x112>(v);
  v += dpp_f64<0x114>(v);
  v += dpp_f64<0x118>(v);
  if constexpr (LAST >= 31) v += dpp_f64<0x142, 0xa>(v);
  if constexpr (LAST >= 63) v += dpp_f64<0x143, 0xc>(v);
  return last_lane<LAST>(v);
}
// maximum of NON-NEGATIVE values (0 is the fill of masked / missing lanes)
template <int LAST = 63>
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = fmax(v, dpp_f64<0x111>(v));
  v = fmax(v, dpp_f64<0x112>(v));
  v = fmax(v, dpp_f64<0x114>(v));
  v = fmax(v, dpp_f64<0x118>(v));
  if constexpr (LAST >= 31) v = fmax(v, dpp_f64<0x142, 0xa>(v));
  if constexpr (LAST >= 63) v = fmax(v, dpp_f64<0x143, 0xc>(v));
  return last_lane<LAST>(v);
}

// minimum over the wave; lanes without a source keep their own value (bound_ctrl off, old = the value itself)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_keep_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_keep_f64<0x111>(v));
  v = fmin(v, dpp_keep_f64<0x112>(v));
  v = fmin(v, dpp_keep_f64<0x114>(v));
  v = fmin(v, dpp_keep_f64<0x118>(v));
  v = fmin(v, dpp_keep_f64<0x142, 0xa>(v));
  v = fmin(v, dpp_keep_f64<0x143, 0xc>(v));
  return last_lane<63>(v);
}

}  // namespace anet
