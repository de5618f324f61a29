// The penalty / energy-gradient kernel k_piece_grad and the small-batch adjoint k_minco_propagate_axis (minco_kernels.h) as a
// translation unit of their own: built with
// -mllvm -amdgpu-sched-strategy=max-ilp (allocnet_amd/build.py; why: the comment at launch_piece_grad's declaration).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "minco_kernels.h"
#include "minco_fused_kernel.h"
#include "piece_grad_mx.h"

namespace anet {

void launch_piece_grad(int s, int shape, dim3 grid, dim3 block, hipStream_t st, const PieceGradArgs &a, const double *tab, int mx_cus) {
  if (shape == 3) {  // large batches, res = 20, orders 3 / 4: the table contractions on the matrix instructions (piece_grad_mx.h)
    // (ANET_PGMX_DYNLDS: bytes of unused dynamic LDS per workgroup -- an occupancy probe for tools/, never set otherwise)
    static const int dyn = [] {
      const char *e = getenv("ANET_PGMX_DYNLDS");
      const int v = e ? atoi(e) : 0;
      if (v > 0) {
        (void)hipFuncSetAttribute((const void *)k_piece_grad_mx<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
        (void)hipFuncSetAttribute((const void *)k_piece_grad_mx<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
        (void)hipFuncSetAttribute((const void *)k_piece_grad_mx<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
        (void)hipFuncSetAttribute((const void *)k_piece_grad_mx<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
      }
      return v;
    }();
    // a workgroup = four waves x 16 NCS trajectories of one piece index; eight column sets per wave where that leaves >= 4 waves per
    // SIMD (shape >> 8: the device's compute units, handed down by the caller; 0 = unknown: four sets)
    const int cus = mx_cus;
    const int64_t waves8 = (a.B + 127) / 128 * (int64_t)grid.y;
    const bool eight = cus > 0 && waves8 >= (int64_t)16 * cus;
    const int ncs = eight ? 8 : 4;
    const dim3 gmx((unsigned)((a.B + 64 * ncs - 1) / (64 * ncs)), grid.y);
    if (s == 3) {
      if (eight) hipLaunchKernelGGL((k_piece_grad_mx<3, 8>), gmx, block, dyn, st, a, tab);
      else hipLaunchKernelGGL((k_piece_grad_mx<3, 4>), gmx, block, dyn, st, a, tab);
    } else {
      if (eight) hipLaunchKernelGGL((k_piece_grad_mx<4, 8>), gmx, block, dyn, st, a, tab);
      else hipLaunchKernelGGL((k_piece_grad_mx<4, 4>), gmx, block, dyn, st, a, tab);
    }
  } else if (shape == 2) {
    if (s == 2) hipLaunchKernelGGL((k_piece_grad<2, true, 4>), grid, block, 0, st, a, tab);
    else if (s == 3) hipLaunchKernelGGL((k_piece_grad<3, true, 4>), grid, block, 0, st, a, tab);
    else hipLaunchKernelGGL((k_piece_grad<4, true, 4>), grid, block, 0, st, a, tab);
  } else if (shape == 1) {
    if (s == 2) hipLaunchKernelGGL((k_piece_grad<2, true>), grid, block, 0, st, a, tab);
    else if (s == 3) hipLaunchKernelGGL((k_piece_grad<3, true>), grid, block, 0, st, a, tab);
    else hipLaunchKernelGGL((k_piece_grad<4, true>), grid, block, 0, st, a, tab);
  } else {
    if (s == 2) hipLaunchKernelGGL((k_piece_grad<2, false>), grid, block, 0, st, a, tab);
    else if (s == 3) hipLaunchKernelGGL((k_piece_grad<3, false>), grid, block, 0, st, a, tab);
    else hipLaunchKernelGGL((k_piece_grad<4, false>), grid, block, 0, st, a, tab);
  }
}

template <int S>
static void launch_propagate_axis_t(const PropArgs &a, dim3 g3, dim3 block, hipStream_t st) {
  if constexpr (S == 4) {
    if (a.N == 8 && a.c == 3) { hipLaunchKernelGGL((k_minco_propagate_axis<4, 8, true, 2>), g3, block, 0, st, a); return; }
    if (a.N == 8 && a.c == 4) { hipLaunchKernelGGL((k_minco_propagate_axis<4, 8, true, 3>), g3, block, 0, st, a); return; }
    if (a.N == 5 && a.c == 3) { hipLaunchKernelGGL((k_minco_propagate_axis<4, 5, true, 2>), g3, block, 0, st, a); return; }
  } else if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) { hipLaunchKernelGGL((k_minco_propagate_axis<3, 16, true, 2>), g3, block, 0, st, a); return; }
    if (a.N == 5 && a.c == 3) { hipLaunchKernelGGL((k_minco_propagate_axis<3, 5, true, 2>), g3, block, 0, st, a); return; }
  }
  if (a.N <= 4) hipLaunchKernelGGL((k_minco_propagate_axis<S, 4>), g3, block, 0, st, a);
  else if (a.N <= 8) hipLaunchKernelGGL((k_minco_propagate_axis<S, 8>), g3, block, 0, st, a);
  else hipLaunchKernelGGL((k_minco_propagate_axis<S, 16>), g3, block, 0, st, a);
}

void launch_propagate_axis(int s, const PropArgs &a, dim3 grid, dim3 block, hipStream_t st) {
  switch (s) {
    case 2: return launch_propagate_axis_t<2>(a, grid, block, st);
    case 3: return launch_propagate_axis_t<3>(a, grid, block, st);
    default: return launch_propagate_axis_t<4>(a, grid, block, st);
  }
}

// ---- the one-launch evaluation of small batches (minco_fused_kernel.h) ----------------------------------------------------------
template <int S, int NB, bool NEXACT, int NPC>
static void launch_fused_t(const FusedArgs &a_in, const double *tab, hipStream_t st, int cus) {
  constexpr int GM = FusedShape<NB>::G;
  FusedArgs a = a_in;
  // the largest group that still gives (nearly) every CU a workgroup (`cus` = the device's compute units: 256 on an MI355X in
  // SPX mode); what the group gives up, the samples split takes
  int G = GM;
  // (and at most 16 lane pairs per piece: their partial sums are added by ONE lane, in order)
  while (G > 1 && (a.B + G / 2 - 1) / (G / 2) <= cus && (G / 2) * NB >= 8) G /= 2;
  a.G = G;
  const dim3 grid((unsigned)((a.B + G - 1) / G));
  // Full groups of the exact shapes at 20 samples per piece: phase 2 on the matrix instructions (minco_fused_kernel.h, MX).
  // ANET_FUSED_MX=0 keeps the vector sample loop (A-B runs).
  if constexpr (NEXACT && NPC >= 0 && FusedShape<NB>::G * NB == 128) {
    static const int mx = [] { const char *e = getenv("ANET_FUSED_MX"); return e ? atoi(e) : 1; }();
    // (groups of at least two column sets of 16 pairs: G NB >= 32; a wave per column set, four waves at least)
    if (mx && G * NB >= 16 && a.pp.res == kMxRes) {
      const int waves = G * NB / 16 < 4 ? 4 : G * NB / 16;
      hipLaunchKernelGGL((k_minco_cost_grad_fused<S, NB, NEXACT, NPC, true>), grid, dim3(64 * waves), 0, st, a, tab);
      return;
    }
  }
  hipLaunchKernelGGL((k_minco_cost_grad_fused<S, NB, NEXACT, NPC>), grid, dim3(256), 0, st, a, tab);
}

int cost_grad_fused_group(int s, int n_pieces) {
  if (s != 3 && s != 4) return 0;
  if (n_pieces <= 8) return FusedShape<8>::G;
  if (s == 3 && n_pieces <= 16) return FusedShape<16>::G;
  return 0;
}

bool launch_cost_grad_fused(int s, const FusedArgs &a, const double *tab, hipStream_t st, int cus) {
  if (s == 4) {
    if (a.N == 8 && a.c == 3) launch_fused_t<4, 8, true, 2>(a, tab, st, cus);
    else if (a.N == 1) launch_fused_t<4, 1, false, -1>(a, tab, st, cus);  // (one piece: no halves to walk from both ends)
    else if (a.N <= 8) launch_fused_t<4, 8, false, -1>(a, tab, st, cus);
    else return false;
    return true;
  }
  if (s == 3) {
    if (a.N == 16 && a.c == 3) launch_fused_t<3, 16, true, 2>(a, tab, st, cus);
    else if (a.N == 1) launch_fused_t<3, 1, false, -1>(a, tab, st, cus);
    else if (a.N <= 8) launch_fused_t<3, 8, false, -1>(a, tab, st, cus);
    else if (a.N <= 16) launch_fused_t<3, 16, false, -1>(a, tab, st, cus);
    else return false;
    return true;
  }
  return false;
}

}  // namespace anet
