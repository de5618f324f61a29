// allocnet_amd: gfx950 kernels + C ABI (include/allocnet_amd.h).  Built by allocnet_amd/build.py:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch, no Eigen.  There is no CPU fallback in this library: without a device every entry
// point fails with ANET_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <new>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is dlopen()ed

#include "../../include/allocnet_amd.h"
#include "minco_core.h"
#include "minco_kernels.h"
#include "minco_fused_kernel.h"
#include "minco_sample_kernel.h"
#include "minco_dense_kernels.h"
#include "traj_kernels.h"
#include "rate_kernels.h"
#include "lbfgs_kernels.h"
#include "lbfgs_minco_persistent.h"
#include "firi_kernels.h"
#include "qp_assemble.h"
#include "qp_admm.h"
#include "qp_ipm.h"
#include "layout_kernels.h"

// ------------------------------------------------------------------------------------------
// context + error plumbing
// ------------------------------------------------------------------------------------------
struct anet_ctx {
  int device = -1;
  // compute units of the device (hipDeviceAttributeMultiprocessorCount, read once in anet_create): every launch-shape threshold
  // below was measured on the 256 CUs of an MI355X in SPX mode and is scaled by cus / 256 (per_cu()), so that a partitioned
  // compute mode (CPX: 32 CUs per logical device) or another part picks its shapes by rounds of workgroups per CU, not by literals
  int cus = 256;
  hipStream_t stream = nullptr;
  std::string err;
  // grow-only device scratch for the host (trajectory-major) entry points
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // L-BFGS completion polling: device counter + pinned host mirror
  int *d_counter = nullptr;
  int *h_counter = nullptr;          // two pinned ints (polls alternate)
  hipEvent_t poll_ev[2] = {nullptr, nullptr};
  // pinned host staging for single-trajectory calls (inputs are packed and sent with ONE copy)
  double *h_pack = nullptr;
  size_t h_pack_doubles = 0;
  // cancel word of the one-launch L-BFGS (anet_set_cancel_flag), device-visible, owned by the caller
  const int32_t *cancel_flag = nullptr;
  // RCCL communicator for the all-gather of costs
  ncclComm_t comm = nullptr;
  int comm_ranks = 0;
  // basis tables of k_piece_grad, one per (order, res) ever used: never rebuilt, never freed before anet_destroy
  // (a launch on another stream may still be reading one), built on the caller's stream
  struct BasisTable {
    int s, res;
    double *d;
    hipStream_t built_on;
    hipEvent_t ready;
  };
  std::vector<BasisTable> tabs;
  // tables of k_qp_ipm, one per (order, res, m34) ever used (csrc/qp_ipm.h k_qp_ipm_tables): same lifetime rules
  struct IpmTable {
    int s, res;
    double m34;
    double *d;
    hipStream_t built_on;
    hipEvent_t ready;
  };
  std::vector<IpmTable> ipm_tabs;
};

namespace {

thread_local std::string g_err;  // errors raised without a context

int fail(anet_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}
int hip_fail(anet_ctx *ctx, hipError_t e, const char *what) {
  return fail(ctx, ANET_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define ANET_HIP(ctx, call)                                   \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);    \
  } while (0)

// Every entry point makes the context's device current first: allocations made inside *_dev calls (counters, basis
// tables) and the launches must land on the context's GPU, not on whatever device the calling thread used last.  The
// caller's current device is put back on the way out (a multi-GPU torch process keeps allocating on ITS device), and
// nothing is switched when the context's device is current already.
struct DeviceGuard {
  int prev = -1;
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define ANET_ON_DEVICE(ctx)                                                   \
  DeviceGuard anet_device_guard_;                                             \
  do {                                                                        \
    if (!(ctx)) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");        \
    int cur_ = -1;                                                            \
    ANET_HIP(ctx, hipGetDevice(&cur_));                                       \
    if (cur_ != (ctx)->device) {                                              \
      ANET_HIP(ctx, hipSetDevice((ctx)->device));                             \
      anet_device_guard_.prev = cur_;                                         \
    }                                                                         \
  } while (0)

int ensure_scratch(anet_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return ANET_OK;
  if (ctx->scratch) {
    hipError_t e = hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    if (e != hipSuccess) return hip_fail(ctx, e, "hipFree(scratch)");
  }
  hipError_t e = hipMalloc(&ctx->scratch, bytes);
  if (e != hipSuccess) {
    ctx->scratch = nullptr;
    return fail(ctx, ANET_ERR_NOMEM, std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
  }
  ctx->scratch_bytes = bytes;
  return ANET_OK;
}

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// a batch / group-count threshold measured on 256 compute units, for this context's device
inline int64_t per_cu(const anet_ctx *ctx, int64_t at_256_cus) { return at_256_cus * (int64_t)ctx->cus / 256; }

// Per-context tables (basis rows of k_piece_grad per (order, res); tables of k_qp_ipm per (order, res, m34)) live until
// anet_destroy; their number is bounded, and a table whose build fails half-way is released, not leaked.
constexpr size_t kMaxTablesPerContext = 256;
int new_table(anet_ctx *ctx, size_t bytes, double **d, hipEvent_t *ready) {
  *d = nullptr;
  *ready = nullptr;
  hipError_t e = hipMalloc((void **)d, bytes);
  if (e != hipSuccess) {
    *d = nullptr;
    return fail(ctx, ANET_ERR_NOMEM, std::string("hipMalloc(table): ") + hipGetErrorString(e));
  }
  e = hipEventCreateWithFlags(ready, hipEventDisableTiming);
  if (e != hipSuccess) {
    (void)hipFree(*d);
    *d = nullptr;
    *ready = nullptr;
    return hip_fail(ctx, e, "hipEventCreateWithFlags(table)");
  }
  return ANET_OK;
}
void drop_table(double *d, hipEvent_t ready) {
  if (d) (void)hipFree(d);
  if (ready) (void)hipEventDestroy(ready);
}

// max T / min T inside one trajectory above which the host entry point of the coefficient solve switches to the pivoted
// collocation solve (minco_dense_kernels.h)
constexpr double kWideSpread = 50.0;

// Up to this batch the axis-parallel kernel is used (3*B/63 waves still fit the chip's 1024 SIMDs about
// once); measured crossover on MI355X in DESIGN.md section 4.
constexpr int64_t kPieceSampleSplitMaxPairs = 16384;  // (trajectory, piece) pairs up to which k_piece_grad splits the samples over waves
constexpr int64_t kAxisVariantMaxBatchDefault = 16384;
inline int64_t axis_variant_max_batch(const anet_ctx *ctx) {  // ANET_AXIS_MAX_BATCH overrides (tuning / A-B runs)
  static const int64_t v = [] {
    const char *e = getenv("ANET_AXIS_MAX_BATCH");
    return e ? (int64_t)atoll(e) : (int64_t)-1;
  }();
  return v >= 0 ? v : per_cu(ctx, kAxisVariantMaxBatchDefault);
}

template <int S>
int launch_solve(anet_ctx *ctx, const anet::SolveArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock));
  const dim3 block(anet::kSolveBlock);
  // small batches: lane per (trajectory, axis) -- three times the waves, ~2.4x shorter dependent chains
  if (a.B <= axis_variant_max_batch(ctx)) {
    const dim3 g3((unsigned)((a.B + 20) / 21));
    bool done = true;
    // (exact shapes with an even number of pieces, batches that leave SIMDs empty: the chain from both ends, two lanes per axis)
    static const int64_t two_env = [] { const char *e = getenv("ANET_AXIS_TWO_MAX_BATCH"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
    const int64_t two_max = two_env >= 0 ? two_env : per_cu(ctx, 4096);
    const dim3 g6((unsigned)((a.B + 9) / 10));
    if (a.B <= two_max && a.c == 3 && ((S == 4 && a.N == 8) || (S == 3 && a.N == 16))) {
      if constexpr (S == 4) hipLaunchKernelGGL((anet::k_minco_solve_axis_two<4, 8, 2>), g6, block, 0, st, a);
      else if constexpr (S == 3) hipLaunchKernelGGL((anet::k_minco_solve_axis_two<3, 16, 2>), g6, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    if constexpr (S == 4) {
      if (a.N == 8 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_solve_axis<4, 8, true, 2>), g3, block, 0, st, a);
      else if (a.N == 8 && a.c == 4) hipLaunchKernelGGL((anet::k_minco_solve_axis<4, 8, true, 3>), g3, block, 0, st, a);
      else if (a.N == 5 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_solve_axis<4, 5, true, 2>), g3, block, 0, st, a);
      else done = false;
    } else if constexpr (S == 3) {
      if (a.N == 16 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_solve_axis<3, 16, true, 2>), g3, block, 0, st, a);
      else if (a.N == 5 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_solve_axis<3, 5, true, 2>), g3, block, 0, st, a);
      else done = false;
    } else {
      done = false;
    }
    if (!done) {
      if (a.N <= 4) hipLaunchKernelGGL((anet::k_minco_solve_axis<S, 4>), g3, block, 0, st, a);
      else if (a.N <= 8) hipLaunchKernelGGL((anet::k_minco_solve_axis<S, 8>), g3, block, 0, st, a);
      else hipLaunchKernelGGL((anet::k_minco_solve_axis<S, 16>), g3, block, 0, st, a);
    }
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  }
  // fully specialised instantiations for the shapes the benchmarks and the reference use
  if constexpr (S == 4) {
    if (a.N == 8 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<4, 8, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    if (a.N == 8 && a.c == 4) {
      hipLaunchKernelGGL((anet::k_minco_solve<4, 8, true, 3>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    if (a.N == 5 && a.c == 3) {  // the planner's own shape: five pieces (learning_planner.hpp:179), PVA ends
      hipLaunchKernelGGL((anet::k_minco_solve<4, 5, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<3, 16, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    if (a.N == 5 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<3, 5, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if (a.N <= 4)
    hipLaunchKernelGGL((anet::k_minco_solve<S, 4>), grid, block, 0, st, a);
  else if (a.N <= 8)
    hipLaunchKernelGGL((anet::k_minco_solve<S, 8>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((anet::k_minco_solve<S, 16>), grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

template <int S>
static int launch_sample(anet_ctx *ctx, const anet::SampleArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock)), block(anet::kSolveBlock);
  bool done = true;
  if constexpr (S == 4) {
    if (a.N == 8 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_sample<4, 8, true, 2>), grid, block, 0, st, a);
    else if (a.N == 5 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_sample<4, 5, true, 2>), grid, block, 0, st, a);
    else done = false;
  } else if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_sample<3, 16, true, 2>), grid, block, 0, st, a);
    else if (a.N == 5 && a.c == 3) hipLaunchKernelGGL((anet::k_minco_sample<3, 5, true, 2>), grid, block, 0, st, a);
    else done = false;
  } else {
    done = false;
  }
  if (!done) {
    if (a.N <= 4) hipLaunchKernelGGL((anet::k_minco_sample<S, 4>), grid, block, 0, st, a);
    else if (a.N <= 8) hipLaunchKernelGGL((anet::k_minco_sample<S, 8>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((anet::k_minco_sample<S, 16>), grid, block, 0, st, a);
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

template <int S>
int launch_prop(anet_ctx *ctx, const anet::PropArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock));
  const dim3 block(anet::kSolveBlock);
  if (a.B <= axis_variant_max_batch(ctx)) {  // same small-batch split as launch_solve
    const dim3 g3((unsigned)((a.B + 20) / 21));
    anet::launch_propagate_axis(S, a, g3, block, st);  // (piece_grad_unit.hip: scheduled for ILP)
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  }
  if constexpr (S == 4) {
    if (a.N == 8 && (a.c == 3 || a.c == 4)) {
      if (a.c == 3) hipLaunchKernelGGL((anet::k_minco_propagate<4, 8, true, 2>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((anet::k_minco_propagate<4, 8, true, 3>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_propagate<3, 16, true, 2>), grid, block, 0, st, a);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
  }
  if (a.N == 5 && a.c == 3 && S >= 3) {  // the planner's shape
    if constexpr (S == 4) hipLaunchKernelGGL((anet::k_minco_propagate<4, 5, true, 2>), grid, block, 0, st, a);
    else if constexpr (S == 3) hipLaunchKernelGGL((anet::k_minco_propagate<3, 5, true, 2>), grid, block, 0, st, a);
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  }
  if (a.N <= 4)
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 4>), grid, block, 0, st, a);
  else if (a.N <= 8)
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 8>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((anet::k_minco_propagate<S, 16>), grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}
int do_propagate(anet_ctx *ctx, int s, const anet::PropArgs &a, hipStream_t st) {
  switch (s) {
    case 2: return launch_prop<2>(ctx, a, st);
    case 3: return launch_prop<3>(ctx, a, st);
    default: return launch_prop<4>(ctx, a, st);
  }
}

// ---- L-BFGS driver ----------------------------------------------------------------------------
struct LbfgsLayout {
  int n, m, npf;
  int64_t ld;
  double *x, *g, *xp, *gp, *d, *lm_s, *lm_y, *lm_ys, *lm_alpha, *pf, *ds, *feval;
  int *is;
  static int64_t doubles(int n, int m, int npf, int64_t ld) {
    // + IS_COUNT_ int32 rows, rounded up to doubles
    return ((int64_t)n * (5 + 2 * m) + 2 * m + npf + anet::DS_COUNT_ + 1 + (anet::IS_COUNT_ + 1) / 2) * ld;
  }
  void carve(double *w) {
    x = w; g = x + (int64_t)n * ld; xp = g + (int64_t)n * ld; gp = xp + (int64_t)n * ld; d = gp + (int64_t)n * ld;
    lm_s = d + (int64_t)n * ld; lm_y = lm_s + (int64_t)m * n * ld; lm_ys = lm_y + (int64_t)m * n * ld;
    lm_alpha = lm_ys + (int64_t)m * ld; pf = lm_alpha + (int64_t)m * ld; ds = pf + (int64_t)npf * ld;
    feval = ds + (int64_t)anet::DS_COUNT_ * ld; is = (int *)(feval + ld);
  }
};

static anet::LbfgsP to_kernel_params(const anet_lbfgs_params &p) {
  return anet::LbfgsP{p.mem_size, p.g_epsilon, p.past, p.delta, p.max_iterations, p.max_linesearch,
                      p.min_step, p.max_step, p.f_dec_coeff, p.s_curv_coeff, p.cautious_factor, p.machine_prec};
}

// One wave per problem at every batch size: at 131072 x 29 variables the lane-per-problem update kernel took 1.67 ms
// per tick (three times the objective evaluation), the wave kernel 0.4 ms.  The lane kernel remains for n > 128 or
// mem_size > 64.
constexpr int64_t kLbfgsWaveMaxBatchDefault = INT64_MAX;
inline int64_t lbfgs_wave_max_batch() {  // ANET_LBFGS_WAVE_MAX_BATCH overrides (tuning / A-B runs)
  static const int64_t v = [] {
    const char *e = getenv("ANET_LBFGS_WAVE_MAX_BATCH");
    return e ? (int64_t)atoll(e) : kLbfgsWaveMaxBatchDefault;
  }();
  return v;
}

static int ensure_counter(anet_ctx *ctx) {
  if (!ctx->d_counter) ANET_HIP(ctx, hipMalloc((void **)&ctx->d_counter, sizeof(int)));
  if (!ctx->h_counter) ANET_HIP(ctx, hipHostMalloc((void **)&ctx->h_counter, 2 * sizeof(int), hipHostMallocDefault));
  for (int i = 0; i < 2; ++i)
    if (!ctx->poll_ev[i]) ANET_HIP(ctx, hipEventCreateWithFlags(&ctx->poll_ev[i], hipEventDisableTiming));
  return ANET_OK;
}

// eval(): enqueue the objective at L.x -> L.feval, L.g (for all problems).  The loop advances every
// problem by one evaluation per pass and polls an "any problem still running" flag every `poll` passes.
// The poll is one group behind the enqueue: the flag of group g is read only after group g+1 is in the
// stream, so the device never idles while the host looks.  The price is up to `poll` extra passes after the
// last problem stopped; they change nothing (finished problems ignore evaluations).
template <class Eval>
static int lbfgs_drive(anet_ctx *ctx, LbfgsLayout &L, int64_t B, const anet_lbfgs_params &prm, int max_evals,
                       hipStream_t st, Eval &&eval, double *map_T = nullptr, int map_nw = 0, bool reset = true,
                       int sb_on = 0, double sb_xmin = 0.0, const int32_t *cancel = nullptr) {
  int rc = ensure_counter(ctx);
  if (rc) return rc;
  if (reset) {  // (a caller that pre-marks problems as finished resets the state itself)
    ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_ * L.ld, st));
    ANET_HIP(ctx, hipMemsetAsync(L.ds, 0, sizeof(double) * anet::DS_COUNT_ * L.ld, st));
  }
  // one wave per problem (DPP reductions, internal vectors problem-major) whenever the problem fits a wave's
  // registers; otherwise one lane per problem (internal vectors batch-minor)
  const bool wave = B <= lbfgs_wave_max_batch() && L.n <= 128 && prm.mem_size <= 64;
  anet::LbfgsArgs a{L.n, B, L.ld, L.x, L.g, L.xp, L.gp, L.d, L.lm_s, L.lm_y, L.lm_ys, L.lm_alpha, L.pf, L.ds,
                    L.feval, L.is, to_kernel_params(prm), nullptr, wave ? 1 : L.ld, wave ? L.n : 1, map_T, map_nw,
                    sb_on, map_nw, sb_xmin, (const int *)cancel};
  const dim3 grid(wave ? (unsigned)B : (unsigned)((B + 63) / 64)), block(64);
  const int poll = 8;
  int group = 0;
  for (int it = 0; it < max_evals; ++it) {
    if ((rc = eval())) return rc;
    const bool check = ((it + 1) % poll == 0) || it + 1 == max_evals;
    if (check) ANET_HIP(ctx, hipMemsetAsync(ctx->d_counter, 0, sizeof(int), st));
    a.n_active = check ? ctx->d_counter : nullptr;
    if (wave) {
      auto shape = [&](int waves, dim3 &g, dim3 &bl) {
        g = dim3((unsigned)((B + waves - 1) / waves));
        bl = dim3(64u * waves);
      };
      dim3 gw, bw;
      const bool one = L.n <= 64;  // one variable per lane: half the registers
      static const int half_ok = [] { const char *e = getenv("ANET_LBFGS_HALF_WAVE"); return e ? atoi(e) : 1; }();
      if (a.p.mem_size <= 8 && L.n <= 32 && half_ok) {  // two problems per wave
        const int waves = anet::LbfgsWaveShape<8>::kWaves;
        gw = dim3((unsigned)((B + 2 * waves - 1) / (2 * waves)));
        bw = dim3(64u * waves);
        hipLaunchKernelGGL((anet::k_lbfgs_update_wave<8, 1, true>), gw, bw, 0, st, a);
      } else if (a.p.mem_size <= 8) {
        shape(anet::LbfgsWaveShape<8>::kWaves, gw, bw);
        if (one) hipLaunchKernelGGL((anet::k_lbfgs_update_wave<8, 1>), gw, bw, 0, st, a);
        else hipLaunchKernelGGL((anet::k_lbfgs_update_wave<8, 2>), gw, bw, 0, st, a);
      } else if (a.p.mem_size <= 20) {
        shape(anet::LbfgsWaveShape<20>::kWaves, gw, bw);
        if (one) hipLaunchKernelGGL((anet::k_lbfgs_update_wave<20, 1>), gw, bw, 0, st, a);
        else hipLaunchKernelGGL((anet::k_lbfgs_update_wave<20, 2>), gw, bw, 0, st, a);
      } else {
        shape(anet::LbfgsWaveShape<0>::kWaves, gw, bw);
        if (one) hipLaunchKernelGGL((anet::k_lbfgs_update_wave<0, 1>), gw, bw, 0, st, a);
        else hipLaunchKernelGGL((anet::k_lbfgs_update_wave<0, 2>), gw, bw, 0, st, a);
      }
    } else {
      hipLaunchKernelGGL(anet::k_lbfgs_update, grid, block, 0, st, a);
    }
    ANET_HIP(ctx, hipGetLastError());
    if (check) {
      const int slot = group & 1;
      ANET_HIP(ctx, hipMemcpyAsync(ctx->h_counter + slot, ctx->d_counter, sizeof(int), hipMemcpyDeviceToHost, st));
      ANET_HIP(ctx, hipEventRecord(ctx->poll_ev[slot], st));
      if (group > 0) {
        ANET_HIP(ctx, hipEventSynchronize(ctx->poll_ev[slot ^ 1]));
        if (ctx->h_counter[slot ^ 1] == 0) break;
      }
      ++group;
    }
  }
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

// status / iters / evals rows -> caller arrays (device or host destination)
// Launch order for anet_lbfgs_minco_ordered_dev from the evaluation counts of a previous solve: a counting sort into 4096
// buckets of 16 evaluations, longest first (the order inside a bucket is whatever the atomics make it: irrelevant here).
constexpr int kOrderBuckets = 4096;
// (shift 4: evaluation counts of an L-BFGS run, up to 65535; shift 0: Newton-step counts of the interior point, up to 4095)
__device__ __forceinline__ int order_bucket(int v, int shift) {
  const int top = (kOrderBuckets << shift) - 1;
  v = v < 0 ? 0 : (v > top ? top : v);
  return kOrderBuckets - 1 - (v >> shift);  // descending
}
__global__ void k_order_hist(const int *counts, int64_t B, int *hist, int shift) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < B) atomicAdd(&hist[order_bucket(counts[b], shift)], 1);
}
__global__ void __launch_bounds__(1024) k_order_scan(int *hist) {  // exclusive scan of the 4096 bucket sizes, one workgroup
  __shared__ int part[1024];
  const int t = threadIdx.x;
  int v[4], sum = 0;
  for (int q = 0; q < 4; ++q) { v[q] = hist[4 * t + q]; sum += v[q]; }
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int add = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int q = 0; q < 4; ++q) { hist[4 * t + q] = run; run += v[q]; }
}
__global__ void k_order_scatter(const int *counts, int64_t B, int *hist, int *order, int shift) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < B) order[atomicAdd(&hist[order_bucket(counts[b], shift)], 1)] = (int)b;
}

// 1 where the durations of a trajectory spread over more than min_spread (max T > min_spread min T)
__global__ void k_spread_flags(const double *T, int64_t B, int64_t ld, int N, double min_spread, int32_t *flags) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  double lo = T[b], hi = lo;
  for (int i = 1; i < N; ++i) {
    const double t = T[(int64_t)i * ld + b];
    lo = fmin(lo, t);
    hi = fmax(hi, t);
  }
  flags[b] = hi > min_spread * lo ? 1 : 0;
}

__global__ void k_lbfgs_results(const int *is, const double *ds, int64_t B, int64_t ld, int *status, int *iters,
                                int *evals, double *f) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  if (status) status[b] = is[anet::IS_DONE * ld + b] ? is[anet::IS_RET * ld + b] : ANET_LBFGS_RUNNING;
  if (iters) iters[b] = is[anet::IS_K * ld + b];
  if (evals) evals[b] = is[anet::IS_EVALS * ld + b];
  // a problem no workgroup took (a launch order that skips it) has no cost: NaN, not whatever the workspace held
  if (f) f[b] = (is[anet::IS_DONE * ld + b] == 0 && is[anet::IS_EVALS * ld + b] == 0) ? __builtin_nan("") : ds[anet::DS_FX * ld + b];
}
}  // namespace

extern "C" {

int anet_abi_version(void) { return ANET_ABI_VERSION; }

int anet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int anet_compute_units(const anet_ctx *ctx) { return ctx ? ctx->cus : 0; }

int anet_create(int device, anet_ctx **out) {
  if (!out) return fail(nullptr, ANET_ERR_INVALID, "anet_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, ANET_ERR_NODEVICE,
                "anet_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(nullptr, ANET_ERR_INVALID, "anet_create: bad device index");
  anet_ctx *ctx = new (std::nothrow) anet_ctx();
  if (!ctx) return fail(nullptr, ANET_ERR_NOMEM, "anet_create: out of host memory");
  ctx->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess && cus > 0) ctx->cus = cus;
  }
  if (e != hipSuccess) {
    int rc = hip_fail(nullptr, e, "anet_create");
    delete ctx;
    return rc;
  }
  *out = ctx;
  return ANET_OK;
}

void anet_destroy(anet_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->comm) (void)anet_comm_destroy(ctx);
  if (ctx->d_counter) (void)hipFree(ctx->d_counter);
  for (auto &t : ctx->tabs) {
    if (t.d) (void)hipFree(t.d);
    if (t.ready) (void)hipEventDestroy(t.ready);
  }
  for (auto &t : ctx->ipm_tabs) {
    if (t.d) (void)hipFree(t.d);
    if (t.ready) (void)hipEventDestroy(t.ready);
  }
  if (ctx->h_counter) (void)hipHostFree(ctx->h_counter);
  for (int i = 0; i < 2; ++i)
    if (ctx->poll_ev[i]) (void)hipEventDestroy(ctx->poll_ev[i]);
  if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *anet_last_error(const anet_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

void *anet_stream(anet_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int anet_synchronize(anet_ctx *ctx) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "anet_synchronize: ctx is NULL");
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int64_t anet_recommended_ld(int64_t batch) {
  int64_t ld = round_up(batch < 1 ? 1 : batch, 64);
  if (ld % 512 == 0) ld += 576;  // 512 doubles = 4 KiB: break (near-)power-of-two row strides
  return ld;
}

int anet_dev_alloc(anet_ctx *ctx, size_t n_doubles, double **out) {
  if (!ctx || !out) return fail(ctx, ANET_ERR_INVALID, "anet_dev_alloc: NULL argument");
  *out = nullptr;
  ANET_ON_DEVICE(ctx);
  hipError_t e = hipMalloc((void **)out, sizeof(double) * (n_doubles ? n_doubles : 1));
  if (e != hipSuccess) return fail(ctx, ANET_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return ANET_OK;
}
void anet_dev_free(double *p) {
  if (p) (void)hipFree(p);
}
int anet_dev_upload(anet_ctx *ctx, double *dst_dev, const double *src_host, size_t n_doubles) {
  if (!ctx || !dst_dev || !src_host) return fail(ctx, ANET_ERR_INVALID, "anet_dev_upload: NULL argument");
  ANET_ON_DEVICE(ctx);
  ANET_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, sizeof(double) * n_doubles, hipMemcpyHostToDevice, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}
int anet_dev_download(anet_ctx *ctx, double *dst_host, const double *src_dev, size_t n_doubles) {
  if (!ctx || !dst_host || !src_dev) return fail(ctx, ANET_ERR_INVALID, "anet_dev_download: NULL argument");
  ANET_ON_DEVICE(ctx);
  ANET_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, sizeof(double) * n_doubles, hipMemcpyDeviceToHost, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int anet_to_batch_minor_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                            const double *src, double *dst, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (!ctx || !src || !dst || batch < 0 || nfield < 0 || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_to_batch_minor_dev: bad argument");
  if (batch == 0 || nfield == 0) return ANET_OK;
  dim3 grid((unsigned)((batch + anet::kTile - 1) / anet::kTile),
            (unsigned)((nfield + anet::kTile - 1) / anet::kTile));
  hipLaunchKernelGGL(anet::k_to_batch_minor, grid, dim3(anet::kTile, 8), 0, (hipStream_t)stream, src,
                     dst, batch, nfield, ld);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_to_traj_major_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                           const double *src, double *dst, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (!ctx || !src || !dst || batch < 0 || nfield < 0 || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_to_traj_major_dev: bad argument");
  if (batch == 0 || nfield == 0) return ANET_OK;
  dim3 grid((unsigned)((batch + anet::kTile - 1) / anet::kTile),
            (unsigned)((nfield + anet::kTile - 1) / anet::kTile));
  hipLaunchKernelGGL(anet::k_to_traj_major, grid, dim3(anet::kTile, 8), 0, (hipStream_t)stream, src,
                     dst, batch, nfield, ld);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

// (callers make the context's device current first: ANET_ON_DEVICE in front of every call)
static int check_solve_args(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s < 2 || s > 4) return fail(ctx, ANET_ERR_INVALID, "order s must be 2, 3 or 4");
  if (c < 1 || c > s) return fail(ctx, ANET_ERR_INVALID, "boundary derivative count c must be in [1, s]");
  if (n_pieces < 1 || n_pieces > ANET_MAX_PIECES)
    return fail(ctx, ANET_ERR_INVALID, "piece count must be in [1, ANET_MAX_PIECES]");
  if (batch < 0) return fail(ctx, ANET_ERR_INVALID, "negative batch");
  return ANET_OK;
}

int anet_minco_solve_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head, const double *tail, const double *wps, const double *T,
                         double *coeffs, double *energy, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps) || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve_dev: NULL input or ld < batch");
  anet::SolveArgs a{head, tail, wps, T, coeffs, energy, batch, ld, n_pieces, c};
  hipStream_t st = (hipStream_t)stream;
  switch (s) {
    case 2: return launch_solve<2>(ctx, a, st);
    case 3: return launch_solve<3>(ctx, a, st);
    default: return launch_solve<4>(ctx, a, st);
  }
}

int anet_minco_sample_costs_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t problems, int64_t samples_per_problem,
                                int64_t ld, int64_t ldp, const double *head, const double *tail, const double *wps,
                                const double *T, double rho, double *cost, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (problems < 0 || samples_per_problem < 1) return fail(ctx, ANET_ERR_INVALID, "anet_minco_sample_costs: problems >= 0, samples_per_problem >= 1");
  const int64_t total = problems * samples_per_problem;
  int rc = check_solve_args(ctx, s, c, n_pieces, total);
  if (rc) return rc;
  if (total == 0) return ANET_OK;
  if (!head || !tail || !T || !cost || (n_pieces > 1 && !wps) || ld < total || ldp < problems)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_sample_costs_dev: NULL pointer, ld < problems * samples_per_problem or ldp < problems");
  anet::SampleArgs a{head, tail, wps, T, cost, total, ld, ldp, samples_per_problem, n_pieces, c, rho};
  hipStream_t st = (hipStream_t)stream;
  switch (s) {
    case 2: return launch_sample<2>(ctx, a, st);
    case 3: return launch_sample<3>(ctx, a, st);
    default: return launch_sample<4>(ctx, a, st);
  }
}

int anet_minco_solve_wide_spread_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                     const double *head, const double *tail, const double *wps, const double *T,
                                     double min_spread, double *coeffs, double *energy, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps) || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve_wide_spread_dev: NULL input or ld < batch");
  if (!coeffs && !energy) return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve_wide_spread_dev: no output requested");
  anet::DenseSolveArgs a{head, tail, wps, T, coeffs, energy, batch, ld, n_pieces, c, min_spread};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)batch), block(64);
  auto launch = [&](auto kernel, size_t lds) -> int {
    if (lds > 64 * 1024) ANET_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  };
  if (s == 2) return launch(anet::k_minco_solve_dense<2>, anet::minco_dense_lds_bytes<2>(n_pieces));
  if (s == 3) return launch(anet::k_minco_solve_dense<3>, anet::minco_dense_lds_bytes<3>(n_pieces));
  return launch(anet::k_minco_solve_dense<4>, anet::minco_dense_lds_bytes<4>(n_pieces));
}

int anet_minco_solve(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, const double *wps, const double *T, double *coeffs,
                     double *energy);

int anet_traj_eval_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, int nq, const double *tq, int deriv,
                       double *out, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (deriv < 0 || deriv > 3 || nq < 0) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval: deriv in [0,3], nq >= 0");
  if (batch == 0 || nq == 0) return ANET_OK;
  if (!coeffs || !T || !tq || !out || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval_dev: NULL pointer or ld < batch");
  anet::EvalArgs a{coeffs, T, tq, out, batch, ld, n_pieces, nq, deriv};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_eval<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_eval<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_eval<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_traj_cost_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, double m34, double *cost, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !cost || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_dev: NULL pointer or ld < batch");
  anet::CostArgs a{coeffs, T, cost, nullptr, batch, ld, n_pieces, m34};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_cost<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_cost<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_cost<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

// Host (trajectory-major) wrappers: stage -> batch-minor -> kernel -> back.
namespace {
struct Stager {
  anet_ctx *ctx;
  int64_t batch, ld;
  double *stage;   // batch * max_fields doubles
  double *cursor;  // next free batch-minor region
  int upload(const double *host, int64_t nf, double **dev) {
    *dev = cursor;
    cursor += nf * ld;
    if (nf == 0) return ANET_OK;
    if (batch == 1) {
      // one trajectory: both layouts coincide (ld = 1), no transpose kernel; the inputs are packed into
      // pinned memory and go out with a single copy when the first output region is reserved
      if (!pack_base) pack_base = *dev;
      const size_t off = (size_t)(*dev - pack_base);
      if (off + (size_t)nf > ctx->h_pack_doubles) {
        const size_t want = (off + (size_t)nf) * 2 + 1024;
        double *np_ = nullptr;
        hipError_t e1 = hipHostMalloc((void **)&np_, sizeof(double) * want, hipHostMallocDefault);
        if (e1 != hipSuccess) return hip_fail(ctx, e1, "hipHostMalloc(pack)");
        if (ctx->h_pack) {
          memcpy(np_, ctx->h_pack, sizeof(double) * off);
          (void)hipHostFree(ctx->h_pack);
        }
        ctx->h_pack = np_;
        ctx->h_pack_doubles = want;
      }
      memcpy(ctx->h_pack + off, host, sizeof(double) * nf);
      pack_doubles = off + (size_t)nf;
      return ANET_OK;
    }
    hipError_t e = hipMemcpyAsync(stage, host, sizeof(double) * batch * nf, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(H2D)");
    return anet_to_batch_minor_dev(ctx, batch, nf, ld, stage, *dev, ctx->stream);
  }
  double *pack_base = nullptr;
  size_t pack_doubles = 0;
  int flush() {  // send the packed single-trajectory inputs (no-op otherwise)
    if (pack_base && pack_doubles) {
      hipError_t e = hipMemcpyAsync(pack_base, ctx->h_pack, sizeof(double) * pack_doubles, hipMemcpyHostToDevice, ctx->stream);
      pack_doubles = 0;
      if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(H2D packed)");
    }
    return ANET_OK;
  }
  double *reserve(int64_t nf) {  // called after all uploads and before the kernels in every entry point
    (void)flush();
    double *p = cursor;
    cursor += nf * ld;
    return p;
  }
  int download(const double *dev, int64_t nf, double *host) {
    if (batch == 1) {
      hipError_t e1 = hipMemcpyAsync(host, dev, sizeof(double) * nf, hipMemcpyDeviceToHost, ctx->stream);
      if (e1 != hipSuccess) return hip_fail(ctx, e1, "hipMemcpyAsync(D2H)");
      e1 = hipStreamSynchronize(ctx->stream);
      if (e1 != hipSuccess) return hip_fail(ctx, e1, "hipStreamSynchronize");
      return ANET_OK;
    }
    int rc = anet_to_traj_major_dev(ctx, batch, nf, ld, dev, stage, ctx->stream);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(host, stage, sizeof(double) * batch * nf, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(D2H)");
    // the staging buffer is reused by the next transfer
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize");
    return ANET_OK;
  }
};
int make_stager(anet_ctx *ctx, int64_t batch, int64_t max_field, int64_t total_fields, Stager *st) {
  // (the entry point that stages has made the context's device current: ANET_ON_DEVICE)
  const int64_t ld = batch == 1 ? 1 : anet_recommended_ld(batch);
  int rc = ensure_scratch(ctx, sizeof(double) * (size_t)(batch * max_field + total_fields * ld));
  if (rc) return rc;
  st->ctx = ctx; st->batch = batch; st->ld = ld;
  st->stage = (double *)ctx->scratch;
  st->cursor = st->stage + batch * max_field;
  st->pack_base = nullptr;
  st->pack_doubles = 0;
  return ANET_OK;
}
}  // namespace

int anet_minco_solve(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, const double *wps, const double *T, double *coeffs,
                     double *energy) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps))
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve: NULL input");
  const int N = n_pieces;
  const int64_t n_in = 6 * (int64_t)c + (int64_t)(N - 1) * 3 + N;
  const int64_t n_co = (int64_t)N * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, n_in > n_co ? n_in : n_co, n_in + n_co + 1, &st);
  if (rc) return rc;
  double *d_head, *d_tail, *d_wps, *d_T;
  if ((rc = st.upload(head, 3 * c, &d_head))) return rc;
  if ((rc = st.upload(tail, 3 * c, &d_tail))) return rc;
  if ((rc = st.upload(wps, (int64_t)(N - 1) * 3, &d_wps))) return rc;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  double *d_co = st.reserve(n_co), *d_en = st.reserve(1);
  rc = anet_minco_solve_dev(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, coeffs ? d_co : nullptr, d_en,
                            ctx->stream);
  if (rc) return rc;
  {
    // The durations are in host memory here, so the check is free: trajectories whose durations spread over more
    // than kWideSpread are redone by the pivoted collocation solve (the reduced form of the fast kernel loses the
    // north star's 1e-6 on the coefficients beyond a spread of ~100).  Device callers decide for
    // themselves (anet_minco_solve_wide_spread_dev): the fast path never pays for the check.
    bool wide = false;
    for (int64_t b = 0; b < batch && !wide; ++b) {
      double lo = T[b * N], hi = lo;
      for (int i = 1; i < N; ++i) {
        lo = T[b * N + i] < lo ? T[b * N + i] : lo;
        hi = T[b * N + i] > hi ? T[b * N + i] : hi;
      }
      wide = hi > kWideSpread * lo;
    }
    if (wide) {
      rc = anet_minco_solve_wide_spread_dev(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, kWideSpread,
                                            coeffs ? d_co : nullptr, d_en, ctx->stream);
      if (rc) return rc;
    }
  }
  if (energy) ANET_HIP(ctx, hipMemcpyAsync(energy, d_en, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
  if (coeffs) return st.download(d_co, n_co, coeffs);
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int anet_minco_sample_costs(anet_ctx *ctx, int s, int c, int n_pieces, int64_t samples, const double *head,
                            const double *tail, const double *wps, const double *T, double rho, double *cost) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, samples);
  if (rc) return rc;
  if (samples == 0) return ANET_OK;
  if (!head || !tail || !T || !cost || (n_pieces > 1 && !wps)) return fail(ctx, ANET_ERR_INVALID, "anet_minco_sample_costs: NULL pointer");
  const int N = n_pieces;
  const int64_t npb = 6 * (int64_t)c + 3 * (int64_t)(N - 1);   // the one problem: head, tail, waypoints (ldp = 1)
  Stager st;
  rc = make_stager(ctx, samples, N, N + 1 + (npb + samples - 1) / samples + 1, &st);
  if (rc) return rc;
  double *d_T;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  if ((rc = st.flush())) return rc;
  double *d_cost = st.reserve(1);
  double *d_prob = st.reserve((npb + samples - 1) / samples + 1);
  ANET_HIP(ctx, hipMemcpyAsync(d_prob, head, sizeof(double) * 3 * c, hipMemcpyHostToDevice, ctx->stream));
  ANET_HIP(ctx, hipMemcpyAsync(d_prob + 3 * c, tail, sizeof(double) * 3 * c, hipMemcpyHostToDevice, ctx->stream));
  if (N > 1) ANET_HIP(ctx, hipMemcpyAsync(d_prob + 6 * c, wps, sizeof(double) * 3 * (N - 1), hipMemcpyHostToDevice, ctx->stream));
  rc = anet_minco_sample_costs_dev(ctx, s, c, N, 1, samples, st.ld, 1, d_prob, d_prob + 3 * c, d_prob + 6 * c, d_T, rho, d_cost,
                                   ctx->stream);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * samples, hipMemcpyDeviceToHost, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int anet_traj_eval(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, int nq, const double *tq, int deriv, double *out) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0 || nq <= 0) return nq < 0 ? fail(ctx, ANET_ERR_INVALID, "nq < 0") : ANET_OK;
  if (!coeffs || !T || !tq || !out) return fail(ctx, ANET_ERR_INVALID, "anet_traj_eval: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  const int64_t mx = nco > 3 * (int64_t)nq ? nco : 3 * (int64_t)nq;
  Stager st;
  rc = make_stager(ctx, batch, mx, nco + n_pieces + nq + 3 * (int64_t)nq, &st);
  if (rc) return rc;
  double *d_co, *d_T, *d_tq;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  if ((rc = st.upload(tq, nq, &d_tq))) return rc;
  double *d_out = st.reserve(3 * (int64_t)nq);
  rc = anet_traj_eval_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, nq, d_tq, deriv, d_out, ctx->stream);
  if (rc) return rc;
  return st.download(d_out, 3 * (int64_t)nq, out);
}

int anet_traj_cost(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, double m34, double *cost) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !cost) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, nco, nco + n_pieces + 1, &st);
  if (rc) return rc;
  double *d_co, *d_T;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  double *d_cost = st.reserve(1);
  rc = anet_traj_cost_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, m34, d_cost, ctx->stream);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

int anet_piece_normalized_coeffs_dev(anet_ctx *ctx, int s, int64_t pieces, const double *coeffs, const double *T, int deriv,
                                     double *out, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (s < 2 || s > 4 || pieces < 0 || deriv < 0 || deriv > 2)
    return fail(ctx, ANET_ERR_INVALID, "anet_piece_normalized_coeffs: order in [2, 4], pieces >= 0, deriv in [0, 2]");
  if (pieces == 0) return ANET_OK;
  if (!coeffs || !T || !out) return fail(ctx, ANET_ERR_INVALID, "anet_piece_normalized_coeffs: NULL pointer");
  anet::NormArgs a{coeffs, T, out, pieces, deriv};
  const dim3 grid((unsigned)((pieces + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_piece_normalize<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_piece_normalize<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_piece_normalize<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_piece_normalized_coeffs(anet_ctx *ctx, int s, int64_t pieces, const double *coeffs, const double *T, int deriv,
                                 double *out) {
  ANET_ON_DEVICE(ctx);
  if (s < 2 || s > 4 || pieces < 0 || deriv < 0 || deriv > 2)
    return fail(ctx, ANET_ERR_INVALID, "anet_piece_normalized_coeffs: order in [2, 4], pieces >= 0, deriv in [0, 2]");
  if (pieces == 0) return ANET_OK;
  if (!coeffs || !T || !out) return fail(ctx, ANET_ERR_INVALID, "anet_piece_normalized_coeffs: NULL pointer");
  const size_t nin = (size_t)pieces * 3 * 2 * s, nout = (size_t)pieces * 3 * (2 * s - deriv);
  int rc = ensure_scratch(ctx, sizeof(double) * (nin + (size_t)pieces + nout));
  if (rc) return rc;
  double *d_co = (double *)ctx->scratch, *d_T = d_co + nin, *d_out = d_T + pieces;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_co, coeffs, sizeof(double) * nin, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_T, T, sizeof(double) * pieces, hipMemcpyHostToDevice, st));
  if ((rc = anet_piece_normalized_coeffs_dev(ctx, s, pieces, d_co, d_T, deriv, d_out, st))) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

// ---- cost / gradient entry points ---------------------------------------------------------------
static int check_penalty(anet_ctx *ctx, const anet_penalty *pen) {
  if (!pen) return ANET_OK;
  if (!(pen->smooth_mu > 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_penalty.smooth_mu must be > 0");
  if (pen->res < 1) return fail(ctx, ANET_ERR_INVALID, "anet_penalty.res must be >= 1");
  if (pen->poly_rows < 0 || pen->poly_rows > ANET_MAX_POLY_ROWS)
    return fail(ctx, ANET_ERR_INVALID, "anet_penalty.poly_rows must be in [0, ANET_MAX_POLY_ROWS]");
  return ANET_OK;
}

// The basis table of k_piece_grad for (order, res): built once per context on the stream that first needs it; other streams are
// ordered behind the build by its event.
static int basis_table(anet_ctx *ctx, int s, int res, hipStream_t st, const double **out) {
  int rc = ANET_OK;
  *out = nullptr;
  if (res > 4096) return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_penalty.res too large for the basis table");
  for (auto &t : ctx->tabs)
    if (t.s == s && t.res == res) {
      // built on another stream: order this stream behind the build (no host or device-wide synchronisation)
      if (t.built_on != st) ANET_HIP(ctx, hipStreamWaitEvent(st, t.ready, 0));
      *out = t.d;
    }
  if (!*out) {
    // (never freed before anet_destroy -- a launch on another stream may still read one: a caller that sweeps res over
    //  hundreds of values is told so instead of growing the context without bound, as for the tables of k_qp_ipm)
    if (ctx->tabs.size() >= kMaxTablesPerContext)
      return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_minco_partial_grads: more than 256 distinct (order, res) on one context");
    anet_ctx::BasisTable t{s, res, nullptr, st, nullptr};
    const int need = res * 4 * 2 * s;
    if ((rc = new_table(ctx, sizeof(double) * need, &t.d, &t.ready))) return rc;
    hipLaunchKernelGGL(anet::k_build_basis_table, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, st, t.d, res, 2 * s);
    hipError_t e1 = hipGetLastError();
    if (e1 == hipSuccess) e1 = hipEventRecord(t.ready, st);
    if (e1 != hipSuccess) {
      drop_table(t.d, t.ready);
      return hip_fail(ctx, e1, "k_build_basis_table");
    }
    ctx->tabs.push_back(t);
    *out = t.d;
  }
  return rc;
}

// The large-batch penalty kernel with the basis-table contractions on the FP64 matrix instructions (csrc/piece_grad_mx.h): built for
// res = 20, orders 3 and 4 (131 072 x 8 snap pieces: 295 us against 344; 65 536 x 16 jerk pieces: 287 against 296 -- six coefficients
// fill three quarters of the instructions' k and column tiles --, profiles/r06_piece_grad_mx.txt).  ANET_PG_MX=0: never (A-B runs).
static int anet_piece_grad_mx_res() { return 20; }
static bool piece_grad_mx_enabled() {
  static const int v = [] { const char *e = getenv("ANET_PG_MX"); return e ? atoi(e) : 1; }();
  return v != 0;
}

// The launch shape of the penalty / energy-gradient kernel (launch_piece_grad): 0 a lane per (trajectory, piece); 1 two lanes per
// pair (small batches); 2 two lanes and the samples over a workgroup's four waves (fewest pairs); 3 k_piece_grad_mx
static int piece_grad_shape(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const anet_penalty *pen) {
  // ANET_PIECE_SW_MAX_PAIRS overrides (tuning / A-B runs)
  static const int64_t sw_env = [] { const char *e = getenv("ANET_PIECE_SW_MAX_PAIRS"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
  const int64_t sw_max_pairs = sw_env >= 0 ? sw_env : per_cu(ctx, kPieceSampleSplitMaxPairs);
  if (pen && batch <= axis_variant_max_batch(ctx)) return batch * n_pieces <= sw_max_pairs ? 2 : 1;
  if (pen && pen->res == anet_piece_grad_mx_res() && (s == 3 || s == 4) && piece_grad_mx_enabled()) return 3;
  return 0;
}

int anet_minco_piece_grad_shape(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const anet_penalty *pen) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s < 2 || s > 4 || n_pieces < 1 || n_pieces > ANET_MAX_PIECES || batch < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_piece_grad_shape: bad shape");
  return piece_grad_shape(ctx, s, n_pieces, batch, pen);
}

int anet_minco_partial_grads_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                                 const double *coeffs, const double *T, const double *hpolys,
                                 const anet_penalty *pen, int with_energy, double *gdC, double *gdT,
                                 double *piece_cost, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gdC || !gdT || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_partial_grads_dev: NULL pointer or ld < batch");
  anet::PieceGradArgs a{};
  a.coeffs = coeffs; a.T = T; a.hpolys = (pen && pen->poly_rows > 0) ? hpolys : nullptr;
  a.gdC = gdC; a.gdT = gdT; a.pcost = piece_cost;
  a.B = batch; a.ld = ld; a.N = n_pieces; a.with_energy = with_energy ? 1 : 0; a.with_penalty = pen ? 1 : 0;
  if (pen) a.pp = anet::Penalty{pen->rho, pen->w_corridor, pen->w_vel, pen->w_acc, pen->smooth_mu,
                               pen->max_vel, pen->max_acc, pen->res, pen->poly_rows};
  const dim3 grid((unsigned)((batch + 255) / 256), (unsigned)n_pieces), block(256);
  hipStream_t st = (hipStream_t)stream;
  const double *tab = nullptr;
  if (pen && (rc = basis_table(ctx, s, pen->res, st, &tab))) return rc;
  const int shape = piece_grad_shape(ctx, s, n_pieces, batch, pen);
  if (shape == 2) {
    // fewest waves: two lanes per (trajectory, piece) AND the samples spread over the four waves of a workgroup
    const dim3 g4((unsigned)((2 * batch + 63) / 64), (unsigned)n_pieces);
    anet::launch_piece_grad(s, 2, g4, block, st, a, tab);
  } else if (shape == 1) {  // small batches: two lanes per (trajectory, piece)
    const dim3 g2((unsigned)((2 * batch + 255) / 256), (unsigned)n_pieces);
    anet::launch_piece_grad(s, 1, g2, block, st, a, tab);
  } else {  // 0: a lane per (trajectory, piece); 3: four lanes per pair and the matrix instructions -- 64 pairs per wave either way
    anet::launch_piece_grad(s, shape, grid, block, st, a, tab, ctx->cus);
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}


int anet_minco_propagate_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                  const double *T, const double *coeffs, const double *gdC,
                                  const double *gdT, double *gradP, double *gradT, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!T || !coeffs || !gdC || !gdT || !gradT || (n_pieces > 1 && !gradP) || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_propagate_grad_dev: NULL pointer or ld < batch");
  anet::PropArgs a{T, coeffs, gdC, gdT, gradP, gradT, nullptr, nullptr, nullptr, 0.0, batch, ld, n_pieces, c};
  return do_propagate(ctx, s, a, (hipStream_t)stream);
}

int64_t anet_minco_cost_grad_workspace(int s, int n_pieces, int64_t ld) {
  // coeffs + gdC + gdT + piece cost + energy
  return ((int64_t)n_pieces * 3 * 2 * s * 2 + 2 * (int64_t)n_pieces + 1) * ld;
}

// Does this evaluation run as ONE launch (k_minco_cost_grad_fused) or as solve -> piece gradients -> adjoint?  (c: the boundary
// count, or -1 when the caller does not know it: the thresholds of the vector phase 2 then)
static bool cost_grad_in_one_launch(const anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const anet_penalty *pen) {
  // Small batches in ONE launch (minco_fused_kernel.h): up to THREE rounds of one workgroup per CU for problems of up to eight pieces,
  // two rounds for longer ones (measured with the chains eliminated from both ends, one launch against three: 8192 x 8-seg snap 54.0
  // against 64.9 us, 12000: 79.5 / 83.8, 16384 = four rounds: 105.5 / 102.4 -- not taken; 4096 x 16-seg jerk 51.8 / 61.0, 2500: 48.7 /
  // 59.3) -- beyond that the three streaming kernels have the chip full anyway and are the better shape.
  // Round 6: where phase 2 runs on the matrix instructions and eight waves (the exact shapes at 20 samples per piece: launch_fused_t)
  // a round of groups costs 17 us instead of 27 and the crossover moves out -- 16 384 x 8-seg snap (four rounds) 68.7 us against 102.9,
  // 24 576 (six) 100.5 / 107.4, 32 768 (eight) 133.4 / 125.5: SIX rounds; 16 384 x 16-seg jerk (eight rounds of groups of 8) 131.4 / 150.6:
  // EIGHT.  (ANET_FUSED_MAX_GROUPS overrides; 0 disables.)
  static const int64_t fused_groups_env = [] { const char *e = getenv("ANET_FUSED_MAX_GROUPS"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
  static const int fused_mx_env = [] { const char *e = getenv("ANET_FUSED_MX"); return e ? atoi(e) : 1; }();
  const bool mx = fused_mx_env && pen && pen->res == anet_piece_grad_mx_res() && c == 3 &&
                  ((s == 4 && n_pieces == 8) || (s == 3 && n_pieces == 16));
  const int rounds = mx ? (n_pieces <= 8 ? 6 : 8) : (n_pieces <= 8 ? 3 : 2);
  const int64_t fused_max_groups = fused_groups_env >= 0 ? fused_groups_env : rounds * (int64_t)ctx->cus;
  const int fg = pen ? anet::cost_grad_fused_group(s, n_pieces) : 0;
  return fg > 0 && (batch + fg - 1) / fg <= fused_max_groups && pen->res <= anet::kFusedMaxRes;
}

// tau != nullptr: the durations are T = forward_T(tau) and gradT is returned as dJ/dtau (L-BFGS driver)
static int cost_grad_dev_impl(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                              const double *head, const double *tail, const double *wps,
                              const double *T, const double *hpolys, const anet_penalty *pen,
                              double *work, double *cost, double *gradP, double *gradT,
                              double *coeffs_out, void *stream, const double *tau) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!work || !cost || !gradT || (n_pieces > 1 && !gradP))
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad_dev: NULL output or workspace");
  if (cost_grad_in_one_launch(ctx, s, c, n_pieces, batch, pen)) {
    if (!head || !tail || !T || (n_pieces > 1 && !wps) || ld < batch)
      return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad_dev: NULL input or ld < batch");
    const double *tab = nullptr;
    if ((rc = basis_table(ctx, s, pen->res, (hipStream_t)stream, &tab))) return rc;
    anet::FusedArgs fa{head, tail, wps, T, pen->poly_rows > 0 ? hpolys : nullptr, cost, gradP, gradT, coeffs_out, tau, batch, ld,
                       n_pieces, c, anet::Penalty{pen->rho, pen->w_corridor, pen->w_vel, pen->w_acc, pen->smooth_mu, pen->max_vel,
                                                  pen->max_acc, pen->res, pen->poly_rows}, 0};
#ifdef ANET_FUSED_PROF
    static long long *d_fprof = nullptr;
    if (!d_fprof) ANET_HIP(ctx, hipMalloc((void **)&d_fprof, 16 * sizeof(long long)));
    fa.prof = d_fprof;
#endif
    if (anet::launch_cost_grad_fused(s, fa, tab, (hipStream_t)stream, ctx->cus)) {
      ANET_HIP(ctx, hipGetLastError());
#ifdef ANET_FUSED_PROF
      if (getenv("ANET_FUSED_PROF_PRINT")) {
        long long h[16];
        ANET_HIP(ctx, hipMemcpyAsync(h, d_fprof, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
        ANET_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
        static const char *nm[13] = {"loads1", "factor", "solve", "stash", "barrier1", "lds-in", "penalty", "energy+pairsum+reduce", "nodeform",
                                     "barrier2", "lds-in3", "fwd", "bwd"};
        fprintf(stderr, "fused_prof cycles (workgroup 0, thread 0):");
        for (int k = 0; k < 13; ++k) fprintf(stderr, " %s %lld", nm[k], h[k + 1] - h[k]);
        fprintf(stderr, " | total %lld\n", h[13] - h[0]);
      }
#endif
      return ANET_OK;
    }
  }
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  double *w_co = coeffs_out ? coeffs_out : work;
  double *w_gdC = work + nco * ld;
  double *w_gdT = w_gdC + nco * ld;
  double *w_pc = w_gdT + (int64_t)n_pieces * ld;
  double *w_en = w_pc + (int64_t)n_pieces * ld;
  rc = anet_minco_solve_dev(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, w_co, w_en, stream);
  if (rc) return rc;
  rc = anet_minco_partial_grads_dev(ctx, s, n_pieces, batch, ld, w_co, T, hpolys, pen, 1, w_gdC, w_gdT,
                                    w_pc, stream);
  if (rc) return rc;
  anet::PropArgs a{T, w_co, w_gdC, w_gdT, gradP, gradT, w_en, pen ? w_pc : nullptr, cost,
                   pen ? pen->rho : 0.0, batch, ld, n_pieces, c, tau};
  return do_propagate(ctx, s, a, (hipStream_t)stream);
}

int anet_minco_cost_grad_launches(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const anet_penalty *pen) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  if (s < 2 || s > 4 || c < 1 || c > s || n_pieces < 1 || n_pieces > ANET_MAX_PIECES || batch < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad_launches: bad shape");
  return cost_grad_in_one_launch(ctx, s, c, n_pieces, batch, pen) ? 1 : 3;
}

int anet_minco_cost_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                             const double *head, const double *tail, const double *wps,
                             const double *T, const double *hpolys, const anet_penalty *pen,
                             double *work, double *cost, double *gradP, double *gradT,
                             double *coeffs_out, void *stream) {
  return cost_grad_dev_impl(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, hpolys, pen, work, cost, gradP, gradT,
                            coeffs_out, stream, nullptr);
}

int anet_minco_cost_grad(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                         const double *tail, const double *wps, const double *T, const double *hpolys,
                         const anet_penalty *pen, double *cost, double *gradP, double *gradT,
                         double *coeffs_out) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && (!wps || !gradP)) || !cost || !gradT)
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_cost_grad: NULL pointer");
  const int N = n_pieces;
  const int64_t nco = (int64_t)N * 3 * 2 * s;
  const int64_t M = (pen && hpolys) ? pen->poly_rows : 0;
  const int64_t nhp = (int64_t)N * M * 4;
  const int64_t n_in = 6 * (int64_t)c + (int64_t)(N - 1) * 3 + N + nhp;
  const int64_t n_out = 1 + (int64_t)(N - 1) * 3 + N + nco;
  int64_t mx = nco > nhp ? nco : nhp;
  if (mx < 3 * (int64_t)c) mx = 3 * c;
  Stager st;
  rc = make_stager(ctx, batch, mx, n_in + n_out + anet_minco_cost_grad_workspace(s, N, 1), &st);
  if (rc) return rc;
  double *d_head, *d_tail, *d_wps, *d_T, *d_hp = nullptr;
  if ((rc = st.upload(head, 3 * c, &d_head))) return rc;
  if ((rc = st.upload(tail, 3 * c, &d_tail))) return rc;
  if ((rc = st.upload(wps, (int64_t)(N - 1) * 3, &d_wps))) return rc;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  if (nhp && (rc = st.upload(hpolys, nhp, &d_hp))) return rc;
  double *d_cost = st.reserve(1), *d_gP = st.reserve((int64_t)(N - 1) * 3), *d_gT = st.reserve(N);
  double *d_co = st.reserve(nco);
  double *d_work = st.reserve(anet_minco_cost_grad_workspace(s, N, 1));
  rc = anet_minco_cost_grad_dev(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, d_hp, pen, d_work,
                                d_cost, d_gP, d_gT, d_co, ctx->stream);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
  if (N > 1 && (rc = st.download(d_gP, (int64_t)(N - 1) * 3, gradP))) return rc;
  if ((rc = st.download(d_gT, N, gradT))) return rc;
  if (coeffs_out && (rc = st.download(d_co, nco, coeffs_out))) return rc;
  ANET_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ANET_OK;
}

// ---- L-BFGS entry points -------------------------------------------------------------------------
void anet_lbfgs_default_params(anet_lbfgs_params *p) {
  if (!p) return;
  p->mem_size = 8; p->g_epsilon = 1.0e-5; p->past = 3; p->delta = 1.0e-6; p->max_iterations = 0;
  p->max_linesearch = 64; p->min_step = 1.0e-20; p->max_step = 1.0e+20; p->f_dec_coeff = 1.0e-4;
  p->s_curv_coeff = 0.9; p->cautious_factor = 1.0e-6; p->machine_prec = 1.0e-16;
}

int anet_lbfgs_check_params(int n, const anet_lbfgs_params *p) {
  if (!p) return -1024;
  if (n <= 0) return -1023;
  if (p->mem_size <= 0) return -1022;
  if (p->g_epsilon < 0.0) return -1021;
  if (p->past < 0) return -1020;
  if (p->delta < 0.0) return -1019;
  if (p->min_step < 0.0) return -1018;
  if (p->max_step < p->min_step) return -1017;
  if (!(p->f_dec_coeff > 0.0 && p->f_dec_coeff < 1.0)) return -1016;
  if (!(p->s_curv_coeff < 1.0 && p->s_curv_coeff > p->f_dec_coeff)) return -1015;
  if (!(p->machine_prec > 0.0)) return -1014;
  if (p->max_linesearch <= 0) return -1013;
  return 0;
}

const char *anet_lbfgs_strerror(int code) {
  switch (code) {
    case 0: return "Success: reached convergence (g_epsilon).";
    case 1: return "Success: met stopping criteria (past f decrease less than delta).";
    case 2: return "The iteration has been canceled by the monitor callback.";
    case -1024: return "Unknown error.";
    case -1023: return "Invalid number of variables specified.";
    case -1022: return "Invalid parameter lbfgs_parameter_t::mem_size specified.";
    case -1021: return "Invalid parameter lbfgs_parameter_t::g_epsilon specified.";
    case -1020: return "Invalid parameter lbfgs_parameter_t::past specified.";
    case -1019: return "Invalid parameter lbfgs_parameter_t::delta specified.";
    case -1018: return "Invalid parameter lbfgs_parameter_t::min_step specified.";
    case -1017: return "Invalid parameter lbfgs_parameter_t::max_step specified.";
    case -1016: return "Invalid parameter lbfgs_parameter_t::f_dec_coeff specified.";
    case -1015: return "Invalid parameter lbfgs_parameter_t::s_curv_coeff specified.";
    case -1014: return "Invalid parameter lbfgs_parameter_t::machine_prec specified.";
    case -1013: return "Invalid parameter lbfgs_parameter_t::max_linesearch specified.";
    case -1012: return "The function value became NaN or Inf.";
    case -1011: return "The line-search step became smaller than lbfgs_parameter_t::min_step.";
    case -1010: return "The line-search step became larger than lbfgs_parameter_t::max_step.";
    case -1009: return "Line search reaches the maximum try number, assumptions not satisfied or precision not achievable.";
    case -1008: return "The algorithm routine reaches the maximum number of iterations.";
    case -1007: return "Relative search interval width is at least lbfgs_parameter_t::machine_prec.";
    case -1006: return "A logic error (negative line-search step) occurred.";
    case -1005: return "The current search direction increases the cost function value.";
    case ANET_LBFGS_RUNNING: return "Still running: the evaluation budget (max_evals) was exhausted.";
    default: return "(unknown)";
  }
}

static int check_lbfgs(anet_ctx *ctx, int n, const anet_lbfgs_params *params, int max_evals) {
  const int code = anet_lbfgs_check_params(n, params);
  if (code) return fail(ctx, ANET_ERR_INVALID, std::string("lbfgs parameters rejected: ") + anet_lbfgs_strerror(code));
  if (max_evals <= 0) return fail(ctx, ANET_ERR_INVALID, "max_evals must be > 0");
  return ANET_OK;
}

int anet_lbfgs_mvie(anet_ctx *ctx, int64_t batch, int M, const double *A, double smooth_eps,
                    double penalty_wt, double *x, double *f, const anet_lbfgs_params *params,
                    int max_evals, int32_t *status, int32_t *iters, int32_t *evals) {
  ANET_ON_DEVICE(ctx);
  int rc = check_lbfgs(ctx, 9, params, max_evals);
  if (rc) return rc;
  if (batch < 0 || M < 1 || !(smooth_eps > 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_mvie: bad batch, M or smooth_eps");
  if (batch == 0) return ANET_OK;
  if (!A || !x) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_mvie: NULL pointer");
  const int n = 9, m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  Stager st;
  const int64_t wdoubles = LbfgsLayout::doubles(n, m, npf, 1);
  const int64_t mx = 3 * (int64_t)M > n ? 3 * (int64_t)M : n;
  rc = make_stager(ctx, batch, mx, 3 * (int64_t)M + n + wdoubles + 3, &st);
  if (rc) return rc;
  double *d_A, *d_x0;
  if ((rc = st.upload(A, 3 * (int64_t)M, &d_A))) return rc;
  if ((rc = st.upload(x, n, &d_x0))) return rc;
  LbfgsLayout L{n, m, npf, st.ld};
  L.carve(st.reserve(wdoubles));
  int *d_res = (int *)st.reserve(3);  // status, iters, evals rows (int32, ld each; 3*ld doubles is ample)
  hipStream_t s0 = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(L.x, d_x0, sizeof(double) * n * st.ld, hipMemcpyDeviceToDevice, s0));
  anet::MvieArgs ma{d_A, L.x, L.feval, L.g, L.is, batch, st.ld, M, smooth_eps, penalty_wt};
  const dim3 grid((unsigned)((batch + 63) / 64)), block(64);
  if (batch <= lbfgs_wave_max_batch() && params->mem_size <= 64) {
    // one wave per problem, the whole optimisation in one launch (k_lbfgs_mvie_persistent)
    ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_ * L.ld, s0));
    ANET_HIP(ctx, hipMemsetAsync(L.ds, 0, sizeof(double) * anet::DS_COUNT_ * L.ld, s0));
    anet::LbfgsArgs la{L.n, batch, L.ld, L.x, L.g, L.xp, L.gp, L.d, L.lm_s, L.lm_y, L.lm_ys, L.lm_alpha, L.pf, L.ds,
                       L.feval, L.is, to_kernel_params(*params), nullptr, 1, L.n, nullptr, 0};
    auto launch = [&](auto kernel, int waves) {
      hipLaunchKernelGGL(kernel, dim3((unsigned)((batch + waves - 1) / waves)), dim3(64u * waves), 0, s0, la, ma, max_evals);
    };
    // everything in registers when it fits (<= 128 rows, mem_size <= 20, past <= 64); else the state goes through memory
    const bool resident = M <= 128 && m <= 20 && params->past <= 64 && !getenv("ANET_MVIE_STATE_IN_MEMORY");
    if (resident && m <= 8 && M <= 64) launch(anet::k_lbfgs_mvie_resident<8, 1>, 1);
    else if (resident && m <= 8) launch(anet::k_lbfgs_mvie_resident<8, 2>, 1);
    else if (resident && M <= 64) launch(anet::k_lbfgs_mvie_resident<20, 1>, 1);
    else if (resident) launch(anet::k_lbfgs_mvie_resident<20, 2>, 1);
    else if (m <= 8) launch(anet::k_lbfgs_mvie_persistent<8>, anet::LbfgsWaveShape<8>::kWaves);
    else if (m <= 20) launch(anet::k_lbfgs_mvie_persistent<20>, anet::LbfgsWaveShape<20>::kWaves);
    else launch(anet::k_lbfgs_mvie_persistent<0>, anet::LbfgsWaveShape<0>::kWaves);
    ANET_HIP(ctx, hipGetLastError());
  } else {
    rc = lbfgs_drive(ctx, L, batch, *params, max_evals, s0, [&]() -> int {
      hipLaunchKernelGGL(anet::k_mvie_eval, grid, block, 0, s0, ma);
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    });
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_lbfgs_results, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s0, L.is, L.ds, batch,
                     st.ld, d_res, d_res + st.ld, d_res + 2 * st.ld, L.feval);
  ANET_HIP(ctx, hipGetLastError());
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_res, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_res + st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (evals) ANET_HIP(ctx, hipMemcpyAsync(evals, d_res + 2 * st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (f) ANET_HIP(ctx, hipMemcpyAsync(f, L.feval, sizeof(double) * batch, hipMemcpyDeviceToHost, s0));
  return st.download(L.x, n, x);
}

int64_t anet_lbfgs_workspace(int n, int64_t ld, const anet_lbfgs_params *params) {
  if (!params || n < 1 || ld < 1) return 0;
  return LbfgsLayout::doubles(n, params->mem_size, params->past > 1 ? params->past : 1, ld);
}

int anet_lbfgs_optimize_dev(anet_ctx *ctx, int n, int64_t batch, int64_t ld, double *x, double *f, double *g,
                            anet_lbfgs_evaluate_t proc_evaluate, void *instance, const anet_lbfgs_params *params,
                            int max_evals, int bound_from, double bound_min, double *work, int32_t *status,
                            int32_t *iters, int32_t *evals, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_lbfgs(ctx, n, params, max_evals);
  if (rc) return rc;
  if (batch < 0 || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_optimize_dev: batch < 0 or ld < batch");
  if (batch == 0) return ANET_OK;
  if (!x || !f || !g || !proc_evaluate || !work) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_optimize_dev: NULL pointer");
  const int m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  LbfgsLayout L{n, m, npf, ld};
  L.carve(work);
  L.x = x;      // the caller's buffers: what its callback reads and fills
  L.g = g;
  L.feval = f;
  hipStream_t st = (hipStream_t)stream;
  int cb_rc = 0;
  rc = lbfgs_drive(ctx, L, batch, *params, max_evals, st, [&]() -> int {
    cb_rc = proc_evaluate(instance, L.x, L.feval, L.g, batch, ld, n, stream);
    return cb_rc ? fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_optimize_dev: proc_evaluate returned non-zero") : ANET_OK;
  }, nullptr, bound_from < n ? (bound_from > 0 ? bound_from : 0) : n, true, bound_from < n ? 1 : 0, bound_min, ctx->cancel_flag);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lbfgs_results, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, L.is, L.ds, batch, ld, status,
                     iters, evals, f);
  ANET_HIP(ctx, hipGetLastError());
  // the run synchronises `stream` as it goes (completion polls); so does its end: status / iters / evals / f are complete
  // when this returns, whatever stream the caller reads them on
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_lbfgs_optimize_host(anet_ctx *ctx, int n, double *x, double *f, anet_lbfgs_host_evaluate_t proc_evaluate,
                             anet_lbfgs_host_stepbound_t proc_stepbound, anet_lbfgs_host_progress_t proc_progress,
                             void *instance, const anet_lbfgs_params *params, int32_t *ret, int32_t *iters, int32_t *evals) {
  ANET_ON_DEVICE(ctx);
  if (!x || !f || !proc_evaluate || !ret) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_optimize_host: NULL argument");
  if (iters) *iters = 0;
  if (evals) *evals = 0;
  // lbfgs_optimize's own parameter validation, in its order, is its return value (lbfgs.hpp:449-495)
  const int code = anet_lbfgs_check_params(n, params);
  if (code) {
    *ret = code;
    return ANET_OK;
  }
  const int m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  // device: the optimiser's state (row stride 1: one problem) + f + the cancel word; host: g, xp, d.  An allocation of THIS
  // call, not the context's scratch: the callbacks run while the state is live and may call any host-staged entry point on
  // the same context (anet_minco_cost_grad, anet_traj_*, another anet_lbfgs_optimize_host ...), which re-carve -- or free and
  // re-allocate -- that scratch (lbfgs::lbfgs_optimize<V>, Piece and Trajectory all use Context::thread_default()).
  const int64_t wd = LbfgsLayout::doubles(n, m, npf, 1);
  struct OwnBuffer {
    void *p = nullptr;
    ~OwnBuffer() { if (p) (void)hipFree(p); }
  } own;
  {
    hipError_t e = hipMalloc(&own.p, sizeof(double) * (size_t)(wd + 4));
    if (e != hipSuccess) {
      own.p = nullptr;
      return fail(ctx, ANET_ERR_NOMEM, std::string("anet_lbfgs_optimize_host: hipMalloc: ") + hipGetErrorString(e));
    }
  }
  int rc = ANET_OK;
  LbfgsLayout L{n, m, npf, 1};
  L.carve((double *)own.p);
  int *d_cancel = (int *)((double *)own.p + wd + 2);
  hipStream_t st = ctx->stream;
  std::vector<double> hg((size_t)n), hxp((size_t)n), hd((size_t)n);
  ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_, st));
  ANET_HIP(ctx, hipMemsetAsync(L.ds, 0, sizeof(double) * anet::DS_COUNT_, st));
  ANET_HIP(ctx, hipMemsetAsync(d_cancel, 0, sizeof(int), st));
  ANET_HIP(ctx, hipMemcpyAsync(L.x, x, sizeof(double) * n, hipMemcpyHostToDevice, st));
  anet::LbfgsArgs a{n, 1, 1, L.x, L.g, L.xp, L.gp, L.d, L.lm_s, L.lm_y, L.lm_ys, L.lm_alpha, L.pf, L.ds, L.feval, L.is,
                    to_kernel_params(*params), nullptr, 1, 1, nullptr, 0, 0, 0, 0.0, d_cancel};
  a.host_pg = proc_progress ? 1 : 0;
  a.host_sb = proc_stepbound ? 1 : 0;
  int his[anet::IS_COUNT_];
  double hds[anet::DS_COUNT_];
  // sources of the small host-to-device copies below: they live until the stream is synchronised in tick()
  double fv = 0.0, bound = 0.0;
  int word = 0;
  auto tick = [&]() -> int {  // one launch of the state machine, then its state on the host
    hipLaunchKernelGGL(anet::k_lbfgs_update, dim3(1), dim3(64), 0, st, a);
    ANET_HIP(ctx, hipGetLastError());
    ANET_HIP(ctx, hipMemcpyAsync(his, L.is, sizeof(his), hipMemcpyDeviceToHost, st));
    ANET_HIP(ctx, hipMemcpyAsync(hds, L.ds, sizeof(hds), hipMemcpyDeviceToHost, st));
    ANET_HIP(ctx, hipStreamSynchronize(st));
    return ANET_OK;
  };
  for (;;) {
    // the point to evaluate is in L.x (the host's copy in x: the start point, or what the last tick left)
    fv = proc_evaluate(instance, x, hg.data(), n);
    ANET_HIP(ctx, hipMemcpyAsync(L.g, hg.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    ANET_HIP(ctx, hipMemcpyAsync(L.feval, &fv, sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = tick())) return rc;
    while (!his[anet::IS_DONE] && (his[anet::IS_PHASE] == anet::LB_PHASE_AWAIT_PROGRESS || his[anet::IS_PHASE] == anet::LB_PHASE_AWAIT_STEPBOUND)) {
      if (his[anet::IS_PHASE] == anet::LB_PHASE_AWAIT_PROGRESS) {
        // lbfgs.hpp:580-587: x is the accepted point (the one just evaluated), g its gradient
        const int verdict = proc_progress(instance, x, hg.data(), hds[anet::DS_FX], hds[anet::DS_STEP], his[anet::IS_K], his[anet::IS_COUNT], n);
        word = verdict ? 1 : 0;
        ANET_HIP(ctx, hipMemcpyAsync(d_cancel, &word, sizeof(int), hipMemcpyHostToDevice, st));
      } else {
        // lbfgs.hpp:557-565: xp (= the current point) and the search direction
        ANET_HIP(ctx, hipMemcpyAsync(hxp.data(), L.xp, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        ANET_HIP(ctx, hipMemcpyAsync(hd.data(), L.d, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        ANET_HIP(ctx, hipStreamSynchronize(st));
        bound = proc_stepbound(instance, hxp.data(), hd.data(), n);
        ANET_HIP(ctx, hipMemcpyAsync(L.ds + anet::DS_SMAX, &bound, sizeof(double), hipMemcpyHostToDevice, st));
      }
      if ((rc = tick())) return rc;
    }
    // the next point to evaluate -- or, when the run has ended, the result (a failed line search put xp back)
    ANET_HIP(ctx, hipMemcpyAsync(x, L.x, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    ANET_HIP(ctx, hipStreamSynchronize(st));
    if (his[anet::IS_DONE]) break;
  }
  *ret = his[anet::IS_RET];
  *f = hds[anet::DS_FX];
  if (iters) *iters = his[anet::IS_K];
  if (evals) *evals = his[anet::IS_EVALS];
  return ANET_OK;
}

// ---- batched FIRI -------------------------------------------------------------------------------
void anet_firi_default_params(anet_firi_params *p) {
  if (!p) return;
  p->iterations = 4;       // firi.hpp:273
  p->epsilon = 1.0e-6;     // firi.hpp:274
  p->smooth_eps = 1.0e-2;  // firi.hpp:218
  p->penalty_wt = 1.0e+3;  // firi.hpp:219
  p->mvie_max_evals = 20000;  // a cap, not a tolerance: the reference has none; corridors cut off by it report ok = 2
}

static int firi_check(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const anet_firi_params &P) {
  if (batch < 0 || n_bd < 1 || n_bd > 64 || max_points < 0 || max_rows < 4 || P.iterations < 1 || !(P.epsilon >= 0.0) ||
      !(P.smooth_eps > 0.0) || P.mvie_max_evals < 1)
    return fail(ctx, ANET_ERR_INVALID, "anet_firi: bad argument (1 <= n_bd <= 64, max_rows >= 4, iterations >= 1)");
  if ((size_t)max_rows * 4 * sizeof(double) > 60 * 1024) return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_firi: max_rows too large");
  return ANET_OK;
}

// doubles of device workspace of anet_firi_dev: ellipsoid state, forward points, MVIE rows, L-BFGS state, flags
int64_t anet_firi_workspace(int64_t batch, int max_points, int max_rows) {
  if (batch < 0 || max_points < 0 || max_rows < 4) return -1;
  const int64_t Np = max_points > 0 ? max_points : 1, ld = anet_recommended_ld(batch);
  return batch * anet::kFiriEll + batch * Np * 4 + 3 * (int64_t)max_rows * ld + LbfgsLayout::doubles(9, 18, 3, ld) +
         (batch * (2 + Np)) / 2 + 16;
}

int anet_firi_dev(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                  const double *pc, const int32_t *n_points, const double *a, const double *b,
                  const anet_firi_params *params, double *work, double *hpoly, int32_t *n_rows, int32_t *ok,
                  double *ellipsoid, void *stream) {
  return anet_firi_var_dev(ctx, batch, n_bd, max_points, max_rows, bd, pc, n_points, a, b, nullptr, params, work, hpoly, n_rows,
                           ok, ellipsoid, stream);
}

int anet_firi_var_dev(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                      const double *pc, const int32_t *n_points, const double *a, const double *b,
                      const int32_t *iterations, const anet_firi_params *params, double *work, double *hpoly,
                      int32_t *n_rows, int32_t *ok, double *ellipsoid, void *stream) {
  ANET_ON_DEVICE(ctx);
  anet_firi_params P;
  anet_firi_default_params(&P);
  if (params) P = *params;
  int rc = firi_check(ctx, batch, n_bd, max_points, max_rows, P);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!bd || (max_points > 0 && (!pc || !n_points)) || !a || !b || !work || !hpoly || !n_rows || !ok)
    return fail(ctx, ANET_ERR_INVALID, "anet_firi_dev: NULL pointer");
  const int H = max_rows, Np = max_points > 0 ? max_points : 1;
  // firi.hpp:212-217
  anet_lbfgs_params lp;
  anet_lbfgs_default_params(&lp);
  lp.mem_size = 18; lp.g_epsilon = 0.0; lp.min_step = 1.0e-32; lp.past = 3; lp.delta = 1.0e-7;
  const int n = 9, m = lp.mem_size, npf = lp.past;
  const int64_t ld = anet_recommended_ld(batch);
  const int64_t w_l = LbfgsLayout::doubles(n, m, npf, ld);
  const size_t n_ell = (size_t)batch * anet::kFiriEll, n_fpc = (size_t)batch * Np * 4, n_hp = (size_t)batch * H * 4;
  const size_t n_A = (size_t)3 * H * ld;
  double *d_ell = work, *d_fpc = d_ell + n_ell, *d_A = d_fpc + n_fpc, *d_l = d_A + n_A;
  int *d_flag = (int *)(d_l + w_l), *d_mok = d_flag + (size_t)batch * Np, *d_np0 = d_mok + batch;
  hipStream_t st = (hipStream_t)stream;
  const int *d_np = n_points;
  if (max_points == 0) {  // no obstacle points at all: a zero count per corridor
    ANET_HIP(ctx, hipMemsetAsync(d_np0, 0, sizeof(int) * batch, st));
    d_np = d_np0;
  }
  ANET_HIP(ctx, hipMemsetAsync(hpoly, 0, sizeof(double) * n_hp, st));
  anet::FiriArgs fa{bd, pc, d_np, a, b, d_ell, d_fpc, d_flag, hpoly, n_rows, ok, batch, n_bd, Np, H, P.epsilon};
  const dim3 g64((unsigned)((batch + 63) / 64)), b64(64), gB((unsigned)batch), b256(256);
  hipLaunchKernelGGL(anet::k_firi_init, g64, b64, 0, st, fa);
  ANET_HIP(ctx, hipGetLastError());
  LbfgsLayout L{n, m, npf, ld};
  L.carve(d_l);
  anet::FiriMvieArgs ma{hpoly, n_rows, ok, d_ell, d_A, L.x, L.is + (int64_t)anet::IS_DONE * ld, L.is + (int64_t)anet::IS_RET * ld,
                        d_mok, batch, ld, H};
  anet::MvieArgs ev{d_A, L.x, L.feval, L.g, L.is, batch, ld, H, P.smooth_eps, P.penalty_wt};
  // wave-per-problem layout of the internal vectors (element i of problem b at [i + b*n]), no "still running" counter
  anet::LbfgsArgs la{L.n, batch, L.ld, L.x, L.g, L.xp, L.gp, L.d, L.lm_s, L.lm_y, L.lm_ys, L.lm_alpha, L.pf, L.ds,
                     L.feval, L.is, to_kernel_params(lp), nullptr, 1, L.n, nullptr, 0};
  fa.iters = iterations; ma.iters = iterations;
  for (int loop = 0; loop < P.iterations; ++loop) {
    fa.pass = loop; ma.pass = loop;
    hipLaunchKernelGGL(anet::k_firi_planes, gB, b256, 0, st, fa);
    ANET_HIP(ctx, hipGetLastError());
    if (loop == P.iterations - 1) break;
    ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_ * ld, st));
    ANET_HIP(ctx, hipMemsetAsync(L.ds, 0, sizeof(double) * anet::DS_COUNT_ * ld, st));
    hipLaunchKernelGGL(anet::k_firi_mvie_setup, gB, b256, sizeof(double) * H * 4, st, ma);
    ANET_HIP(ctx, hipGetLastError());
    {  // the whole MVIE optimisation in one launch, one wave per corridor; rows and optimiser state in registers when they fit
      constexpr int kw = anet::LbfgsWaveShape<20>::kWaves;
      const bool in_memory = getenv("ANET_MVIE_STATE_IN_MEMORY") != nullptr;  // A/B switch (tools/README.md)
      if (!in_memory && H <= 64 && lp.mem_size <= 20 && lp.past <= 64)
        hipLaunchKernelGGL((anet::k_lbfgs_mvie_resident<20, 1>), dim3((unsigned)batch), dim3(64), 0, st, la, ev, P.mvie_max_evals);
      else if (!in_memory && H <= 128 && lp.mem_size <= 20 && lp.past <= 64)
        hipLaunchKernelGGL((anet::k_lbfgs_mvie_resident<20, 2>), dim3((unsigned)batch), dim3(64), 0, st, la, ev, P.mvie_max_evals);
      else
        hipLaunchKernelGGL(anet::k_lbfgs_mvie_persistent<20>, dim3((unsigned)((batch + kw - 1) / kw)), dim3(64u * kw), 0, st, la, ev,
                           P.mvie_max_evals);
      ANET_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(anet::k_firi_mvie_finish, g64, b64, 0, st, ma);
    ANET_HIP(ctx, hipGetLastError());
  }
  if (ellipsoid)
    ANET_HIP(ctx, hipMemcpy2DAsync(ellipsoid, sizeof(double) * 15, d_ell, sizeof(double) * anet::kFiriEll, sizeof(double) * 15, batch,
                                   hipMemcpyDeviceToDevice, st));
  return ANET_OK;
}

int anet_firi(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
              const double *pc, const int32_t *n_points, const double *a, const double *b,
              const anet_firi_params *params, double *hpoly, int32_t *n_rows, int32_t *ok, double *ellipsoid) {
  return anet_firi_var(ctx, batch, n_bd, max_points, max_rows, bd, pc, n_points, a, b, nullptr, params, hpoly, n_rows, ok, ellipsoid);
}

int anet_firi_var(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                  const double *pc, const int32_t *n_points, const double *a, const double *b, const int32_t *iterations,
                  const anet_firi_params *params, double *hpoly, int32_t *n_rows, int32_t *ok, double *ellipsoid) {
  if (!ctx) return fail(nullptr, ANET_ERR_INVALID, "ctx is NULL");
  anet_firi_params P;
  anet_firi_default_params(&P);
  if (params) P = *params;
  int rc = firi_check(ctx, batch, n_bd, max_points, max_rows, P);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!bd || (max_points > 0 && (!pc || !n_points)) || !a || !b || !hpoly || !n_rows)
    return fail(ctx, ANET_ERR_INVALID, "anet_firi: NULL pointer");
  ANET_ON_DEVICE(ctx);
  const int H = max_rows, Np = max_points > 0 ? max_points : 1;
  const size_t n_bdv = (size_t)batch * n_bd * 4, n_pc = (size_t)batch * Np * 3, n_ab = (size_t)batch * 3;
  const size_t n_hp = (size_t)batch * H * 4, n_ell = (size_t)batch * 15;
  const size_t n_work = (size_t)anet_firi_workspace(batch, max_points, max_rows);
  rc = ensure_scratch(ctx, sizeof(double) * (n_bdv + n_pc + 2 * n_ab + n_hp + n_ell + n_work + (size_t)(4 * batch) / 2 + 16));
  if (rc) return rc;
  double *d_bd = (double *)ctx->scratch, *d_pc = d_bd + n_bdv, *d_a = d_pc + n_pc, *d_b = d_a + n_ab, *d_hp = d_b + n_ab;
  double *d_el = d_hp + n_hp, *d_work = d_el + n_ell;
  int *d_np = (int *)(d_work + n_work), *d_nh = d_np + batch, *d_ok = d_nh + batch, *d_it = d_ok + batch;
  hipStream_t st = ctx->stream;
  if (iterations) {  // (host array: checked here; the device variant clamps instead, it cannot look without a synchronisation)
    for (int64_t b = 0; b < batch; ++b)
      if (iterations[b] < 1 || iterations[b] > P.iterations)
        return fail(ctx, ANET_ERR_INVALID, "anet_firi_var: iterations[b] must be in [1, params->iterations]");
    ANET_HIP(ctx, hipMemcpyAsync(d_it, iterations, sizeof(int) * batch, hipMemcpyHostToDevice, st));
  }
  ANET_HIP(ctx, hipMemcpyAsync(d_bd, bd, sizeof(double) * n_bdv, hipMemcpyHostToDevice, st));
  if (max_points > 0) {
    ANET_HIP(ctx, hipMemcpyAsync(d_pc, pc, sizeof(double) * n_pc, hipMemcpyHostToDevice, st));
    ANET_HIP(ctx, hipMemcpyAsync(d_np, n_points, sizeof(int) * batch, hipMemcpyHostToDevice, st));
  }
  ANET_HIP(ctx, hipMemcpyAsync(d_a, a, sizeof(double) * n_ab, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_b, b, sizeof(double) * n_ab, hipMemcpyHostToDevice, st));
  rc = anet_firi_var_dev(ctx, batch, n_bd, max_points, max_rows, d_bd, d_pc, d_np, d_a, d_b, iterations ? d_it : nullptr, &P,
                         d_work, d_hp, d_nh, d_ok, ellipsoid ? d_el : nullptr, st);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(hpoly, d_hp, sizeof(double) * n_hp, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipMemcpyAsync(n_rows, d_nh, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
  if (ok) ANET_HIP(ctx, hipMemcpyAsync(ok, d_ok, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
  if (ellipsoid) ANET_HIP(ctx, hipMemcpyAsync(ellipsoid, d_el, sizeof(double) * n_ell, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_polytope_depth_dev(anet_ctx *ctx, int64_t batch, int max_rows, const double *hpoly, int normalise,
                            double *depth, double *point, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (batch < 0 || max_rows < 1) return fail(ctx, ANET_ERR_INVALID, "anet_polytope_depth: bad batch or max_rows");
  // (the vertex enumeration is C(rows, 4): 1.7e8 candidates at 256 rows -- corridor polytopes have a few dozen)
  if (max_rows > 256) return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_polytope_depth: more than 256 rows per polytope");
  if (batch == 0) return ANET_OK;
  if (!hpoly || !depth) return fail(ctx, ANET_ERR_INVALID, "anet_polytope_depth_dev: NULL pointer");
  // active-set ascent, one lane per polytope, certified; what it cannot certify (NaN) goes to the vertex enumeration
  const bool enumerate_all = getenv("ANET_POLYTOPE_DEPTH_ENUMERATE") != nullptr;  // A/B switch
  anet::DepthArgs a{hpoly, depth, point, batch, max_rows, normalise ? 1 : 0, enumerate_all ? 0 : 1};
  if (!enumerate_all) {
    hipLaunchKernelGGL(anet::k_polytope_depth_simplex, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    ANET_HIP(ctx, hipGetLastError());
  }
  if (getenv("ANET_POLYTOPE_DEPTH_NO_FALLBACK")) return ANET_OK;  // diagnostics: NaN marks what the ascent did not certify
  hipLaunchKernelGGL(anet::k_polytope_depth, dim3((unsigned)batch), dim3(256), sizeof(double) * max_rows * 4, (hipStream_t)stream, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_polytope_depth(anet_ctx *ctx, int64_t batch, int max_rows, const double *hpoly, int normalise,
                        double *depth, double *point) {
  ANET_ON_DEVICE(ctx);
  if (batch < 0 || max_rows < 1) return fail(ctx, ANET_ERR_INVALID, "anet_polytope_depth: bad batch or max_rows");
  if (batch == 0) return ANET_OK;
  if (!hpoly || !depth) return fail(ctx, ANET_ERR_INVALID, "anet_polytope_depth: NULL pointer");
  const size_t n_hp = (size_t)batch * max_rows * 4;
  int rc = ensure_scratch(ctx, sizeof(double) * (n_hp + 4 * (size_t)batch));
  if (rc) return rc;
  double *d_hp = (double *)ctx->scratch, *d_depth = d_hp + n_hp, *d_pt = d_depth + batch;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_hp, hpoly, sizeof(double) * n_hp, hipMemcpyHostToDevice, st));
  rc = anet_polytope_depth_dev(ctx, batch, max_rows, d_hp, normalise, d_depth, point ? d_pt : nullptr, st);
  if (rc) return rc;
  ANET_HIP(ctx, hipMemcpyAsync(depth, d_depth, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
  if (point) ANET_HIP(ctx, hipMemcpyAsync(point, d_pt, sizeof(double) * 3 * batch, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

// Thresholds of the two-launch form of the one-launch L-BFGS (lbfgs_minco_dev_impl; the environment overrides are for A/B runs)
static int lbfgs_split_evals() { static const int v = [] { const char *e = getenv("ANET_LBFGS_SPLIT_EVALS"); return e ? atoi(e) : 1000; }(); return v; }
static int64_t lbfgs_split_min_batch(const anet_ctx *ctx) {  // (4096 problems on 256 CUs = twice the resident waves)
  static const int64_t v = [] { const char *e = getenv("ANET_LBFGS_SPLIT_MIN_BATCH"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
  return v >= 0 ? v : per_cu(ctx, 4096);
}
static int lbfgs_split_min_vars() { static const int v = [] { const char *e = getenv("ANET_LBFGS_SPLIT_MIN_VARS"); return e ? atoi(e) : 36; }(); return v; }

int64_t anet_lbfgs_minco_workspace(int s, int n_pieces, int64_t ld, const anet_lbfgs_params *params) {
  if (!params || params->mem_size <= 0) return -1;
  const int n = 3 * (n_pieces - 1) + n_pieces;
  const int npf = params->past > 1 ? params->past : 1;
  // L-BFGS state + cost/grad workspace + gradP + gradT ...
  int64_t w = LbfgsLayout::doubles(n, params->mem_size, npf, ld) + anet_minco_cost_grad_workspace(s, n_pieces, ld) + (int64_t)n * ld;
  // ... then, where the two-launch form of the one-launch shape can run (enough variables; whether a BATCH takes it is the
  // context's decision -- lbfgs_minco_dev_impl, by the device's compute units -- and does not enter the size: a workspace of
  // this size is enough for either form at any batch <= ld), the parked optimisers, their scores and the order of the second
  // launch (int32 each) and the bins of the counting sort
  if (lbfgs_split_evals() > 1 && n >= lbfgs_split_min_vars())
    w += (int64_t)anet::kPersistContDoubles * ld + ld + 2 + kOrderBuckets / 2;
  return w;
}

// Order of the second launch of a two-launch L-BFGS run (lbfgs_minco_persistent.h PersistArgs::park): larger = expected to need more
// evaluations.  What predicts it at the split point (4096 problems of BASELINE configs[3], parked after 1000 evaluations;
// Spearman 0.61 with the evaluations left, and the order it gives simulates to the longest-first makespan): the gradient norm
// relative to the cost and the relative decrease of the cost over the last half of the first part, in decades.  Problems that
// ended in the first part score 0 and come last (their waves leave at once).
__global__ void k_lbfgs_resume_score(const int *is, const double *cont, int64_t B, int64_t ld, int *score) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int sc = 0;
  if (is[(int64_t)anet::IS_DONE * ld + b] == 0) {
    const double *u = cont + b * (int64_t)anet::kPersistContDoubles + 22 * 64;
    const double fx = fabs(u[8]) + 1e-300, dec = fmax((u[23] - u[8]) / fx, 1e-16), gn = sqrt(fmax(u[24], 0.0)) / fx;
    double v = 4000.0 + 150.0 * (log10(fmax(gn, 1e-16)) + log10(dec));
    if (!(v == v)) v = 4000.0;
    sc = (int)fmin(fmax(v, 1.0), 4000.0);
  }
  score[b] = sc;
}

static int launch_order_impl(anet_ctx *ctx, int64_t batch, const int32_t *counts, int32_t *launch_order, int32_t *work,
                             void *stream, int shift) {
  ANET_ON_DEVICE(ctx);
  if (batch < 0 || batch > 0x7fffffff) return fail(ctx, ANET_ERR_INVALID, "anet_launch_order_from_counts: bad batch");
  if (batch == 0) return ANET_OK;
  if (!counts || !launch_order || !work) return fail(ctx, ANET_ERR_INVALID, "anet_launch_order_from_counts_dev: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  ANET_HIP(ctx, hipMemsetAsync(work, 0, sizeof(int) * kOrderBuckets, st));
  hipLaunchKernelGGL(k_order_hist, grid, block, 0, st, counts, batch, work, shift);
  hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(1024), 0, st, work);
  hipLaunchKernelGGL(k_order_scatter, grid, block, 0, st, counts, batch, work, launch_order, shift);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_launch_order_from_counts_dev(anet_ctx *ctx, int64_t batch, const int32_t *counts, int32_t *launch_order,
                                      int32_t *work, void *stream) {
  return launch_order_impl(ctx, batch, counts, launch_order, work, stream, 4);
}

int anet_launch_order_from_steps_dev(anet_ctx *ctx, int64_t batch, const int32_t *steps, int32_t *launch_order,
                                     int32_t *work, void *stream) {
  return launch_order_impl(ctx, batch, steps, launch_order, work, stream, 0);
}

int anet_minco_spread_flags_dev(anet_ctx *ctx, int n_pieces, int64_t batch, int64_t ld, const double *T, double min_spread,
                                int32_t *flags, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (n_pieces < 1 || batch < 0) return fail(ctx, ANET_ERR_INVALID, "anet_minco_spread_flags_dev: n_pieces >= 1, batch >= 0");
  if (batch == 0) return ANET_OK;
  if (!T || !flags || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_minco_spread_flags_dev: NULL pointer or ld < batch");
  hipLaunchKernelGGL(k_spread_flags, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, batch, ld,
                     n_pieces, min_spread > 0.0 ? min_spread : kWideSpread, flags);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

// Coefficients of the RETURNED waypoints / durations: the fast (reduced-system) solve, then the pivoted collocation solve
// for the trajectories whose optimised durations spread over more than kWideSpread -- inside the optimisation loop the
// cost and its gradient keep the reduced system's accuracy envelope, what is handed back does not.
static int final_coeffs(anet_ctx *ctx, int s, int c, int N, int64_t batch, int64_t ld, const double *head, const double *tail,
                        const double *wps, const double *T, double *coeffs_out, hipStream_t st) {
  int rc = anet_minco_solve_dev(ctx, s, c, N, batch, ld, head, tail, wps, T, coeffs_out, nullptr, st);
  if (rc) return rc;
  return anet_minco_solve_wide_spread_dev(ctx, s, c, N, batch, ld, head, tail, wps, T, kWideSpread, coeffs_out, nullptr, st);
}

int anet_set_cancel_flag(anet_ctx *ctx, const int32_t *flag) {
  if (!ctx) return ANET_ERR_INVALID;
  ctx->cancel_flag = flag;
  return ANET_OK;
}

int anet_lbfgs_minco_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head, const double *tail, double *wps, double *T,
                         const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                         int opt_flags, int max_evals, double *work, double *cost, double *coeffs_out,
                         int32_t *status, int32_t *iters, int32_t *evals, void *stream) {
  return anet_lbfgs_minco_ordered_dev(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, hpolys, pen, params, opt_flags,
                                      max_evals, nullptr, work, cost, coeffs_out, status, iters, evals, stream);
}

static int lbfgs_minco_dev_impl(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                               const double *head, const double *tail, double *wps, double *T,
                               const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                               int opt_flags, int max_evals, double min_duration, const int32_t *launch_order, double *work,
                               double *cost, double *coeffs_out, int32_t *status, int32_t *iters, int32_t *evals, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  const int N = n_pieces;
  const int nw = (opt_flags & ANET_OPT_WAYPOINTS) ? 3 * (N - 1) : 0;
  const int nt = (opt_flags & ANET_OPT_TIMES) ? N : 0;
  const int n = nw + nt;
  if (n <= 0) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: nothing to optimise (opt_flags / N)");
  if ((rc = check_lbfgs(ctx, n, params, max_evals))) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (N > 1 && !wps) || !work || ld < batch)
    return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco_dev: NULL pointer or ld < batch");
  const int m = params->mem_size, npf = params->past > 1 ? params->past : 1;
  LbfgsLayout L{n, m, npf, ld};
  L.carve(work);
  double *w_cg = work + LbfgsLayout::doubles(n, m, npf, ld);
  double *w_gP = w_cg + anet_minco_cost_grad_workspace(s, N, ld);
  double *w_gT = w_gP + (int64_t)3 * (N - 1) * ld;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g256((unsigned)((batch + 255) / 256)), b256(256);
  anet::MapArgs mp{L.x, wps, T, batch, ld, nw, nt, 0};
  const dim3 gmap(g256.x, (unsigned)n);
  hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
  ANET_HIP(ctx, hipGetLastError());
  // No per-evaluation mapping launches: the optimised waypoints ARE the first nw rows of x (same layout),
  // their gradient goes straight into g, the update kernel writes T = forward_T(tau) next to x, and the
  // propagate kernel applies dT/dtau to the duration gradient.
  const double *wps_eval = nw ? L.x : wps;
  double *gP_out = nw ? L.g : w_gP, *gT_out = nt ? L.g + (int64_t)nw * ld : w_gT;
  const double *tau = nt ? L.x + (int64_t)nw * ld : nullptr;
  // One launch, one wave per problem (lbfgs_minco_persistent.h) whenever the problem fits a wave: every problem runs
  // until ITS OWN stop instead of the batch advancing in lockstep, four launches per evaluation.  The launch-per-
  // evaluation kernels (all 64 lanes busy in the chains) have up to twice the throughput per evaluation STEP at batches
  // of 10^5, but a run to convergence is as long as its slowest member times the whole batch there: 131072 x 8-segment
  // snap 2.3 s in one launch against 4.1 s in lockstep, 131072 x 16-segment jerk 3.5 s against 10.0 s.  So one launch
  // at any batch (ANET_LBFGS_PERSISTENT_MAX_BATCH caps it for A/B runs); callers with a small fixed evaluation budget
  // at a huge batch ask for the lockstep shape (ANET_OPT_LOCKSTEP).
  static const int64_t persist_max_batch = [] {
    const char *e = getenv("ANET_LBFGS_PERSISTENT_MAX_BATCH");
    return e ? (int64_t)atoll(e) : (int64_t)0x7fffffff;
  }();
  const int Mrows = (pen && hpolys) ? pen->poly_rows : 0;
  const size_t row_bytes = sizeof(double) * anet::persist_lds_row_doubles(N, Mrows);
  if (!(opt_flags & ANET_OPT_LOCKSTEP) && batch <= persist_max_batch && (s == 3 || s == 4) && n <= 64 &&
      params->mem_size <= 8 && params->past <= 64) {
    anet::PersistArgs pa{};
    pa.head = head; pa.tail = tail; pa.wps = wps; pa.T = T; pa.hpolys = Mrows ? hpolys : nullptr;
    pa.x = L.x; pa.is = L.is; pa.ds = L.ds; pa.order = launch_order; pa.B = batch; pa.ld = ld;
    pa.N = N; pa.c = c; pa.nw = nw; pa.nt = nt; pa.M = Mrows; pa.max_evals = max_evals; pa.with_penalty = pen ? 1 : 0;
    if (pen) pa.pp = anet::Penalty{pen->rho, pen->w_corridor, pen->w_vel, pen->w_acc, pen->smooth_mu, pen->max_vel,
                                   pen->max_acc, pen->res, Mrows};
    else pa.pp = anet::Penalty{0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1, 0};
    pa.inv_mu = 1.0 / pa.pp.mu; pa.inv_res = 1.0 / (double)pa.pp.res;
    pa.p = to_kernel_params(*params);
    pa.step_bound = (min_duration > 0.0 && nt > 0) ? 1 : 0;  // (gcopter's backwardT, minco_core.h backward_T)
    pa.cancel = (const int *)ctx->cancel_flag;
    pa.tau_min = min_duration > 1.0 ? sqrt(2.0 * min_duration - 1.0) - 1.0 : (min_duration > 0.0 ? 1.0 - sqrt(2.0 / min_duration - 1.0) : 0.0);
#ifdef ANET_PERSIST_PROF
    static long long *d_prof = nullptr;
    if (!d_prof) ANET_HIP(ctx, hipMalloc((void **)&d_prof, 16 * sizeof(long long)));
    ANET_HIP(ctx, hipMemsetAsync(d_prof, 0, 16 * sizeof(long long), st));
    pa.prof = d_prof;
#endif
    // a caller-supplied launch order is not checked: a problem it skips (out-of-range or repeated entries) must report
    // ANET_LBFGS_RUNNING with zero counters, not whatever the workspace held
    if (launch_order) ANET_HIP(ctx, hipMemsetAsync(L.is, 0, sizeof(int) * anet::IS_COUNT_ * ld, st));
    // Batches well beyond the 2048 resident waves in TWO launches: the first takes every problem through the same number of
    // evaluations (equally long waves: no late starters), the second resumes the unfinished ones longest-expected first.  The
    // batch of BASELINE configs[3] (4096 problems, 300..7400 evaluations) otherwise ends with whichever long problem happened to
    // start in the second round: 0.160 s against 0.117 s with the problems longest first by their true counts.
    const int split_evals = lbfgs_split_evals();
    // (a SECOND park / re-sort, measured in round 5 and left off: profiles/r05_lbfgs_second_park.txt -- every stage boundary is a
    //  barrier for the whole batch, and what the better order of the third stage gains is less than what the second stage's own
    //  tail loses: configs[3] 136 -> 147 / 157 / 165 ms with the second boundary at 2000 / 2500 / 3000 evaluations)
    static const int split_evals2 = [] { const char *e = getenv("ANET_LBFGS_SPLIT_EVALS2"); return e ? atoi(e) : 0; }();
    const int64_t split_min_batch = lbfgs_split_min_batch(ctx);
    // ... where it was measured to pay (tools/time_lbfgs_batch.py, 4096 problems unless noted, one launch -> two): 16 jerk pieces
    // 165 -> 140 ms (bench: 0.169 -> 0.133 s), 16 snap pieces 424 -> 389, 12 jerk pieces 107 -> 98, 10 jerk pieces 79 -> 77, 16 jerk
    // pieces x 8192 235 -> 216, x 16384 415 -> 403, x 3072 no change; 8 snap pieces 102 -> 100..109, 5 jerk pieces 23.5 -> 25, 5
    // snap pieces 37 -> 39 (their runs are a few hundred evaluations long: the split point lies behind most of them, and at 250..700
    // evaluations the parked state does not tell the long problems yet).  Hence: problems of at least 36 variables (ten pieces).
    const int split_min_vars = lbfgs_split_min_vars();
    const bool two_launches = split_evals > 1 && batch >= split_min_batch && n >= split_min_vars && !launch_order && max_evals > split_evals;
    double *cont = w_gP + (int64_t)n * ld;
    int32_t *score = (int32_t *)(cont + (int64_t)anet::kPersistContDoubles * ld);
    int32_t *order2 = score + ld + (ld & 1);
    int32_t *bins = order2 + ld + (ld & 1);
    bool order_failed = false;
    auto launch = [&](auto kernel, size_t fixed_bytes) {
      if (!two_launches) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(64), fixed_bytes + row_bytes, st, pa);
        return;
      }
      // stages: [0, split) for everybody, then the problems still running, each stage's workgroups handed their problems
      // longest-expected first by what the stage before parked ([split, split2), [split2, ...) with ANET_LBFGS_SPLIT_EVALS2 set)
      const int stops[3] = {split_evals, (split_evals2 > split_evals && max_evals > split_evals2) ? split_evals2 : 0, max_evals};
      pa.cont = cont;
      int prev = 0;
      for (int stage = 0; stage < 3; ++stage) {
        if (stops[stage] <= 0) continue;
        const bool last = stage == 2;
        pa.park = last ? 0 : 1;
        pa.resume = prev > 0 ? 1 : 0;
        pa.half_mark = (prev + stops[stage]) / 2;
        pa.max_evals = stops[stage];
        hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(64), fixed_bytes + row_bytes, st, pa);
        if (last) break;
        hipLaunchKernelGGL(k_lbfgs_resume_score, g256, b256, 0, st, L.is, cont, batch, ld, score);
        if (launch_order_impl(ctx, batch, score, order2, bins, st, 0) != ANET_OK) {  // (cannot fail with these arguments; if it ever does:
          order_failed = true;                                                       //  the error is the caller's return code)
          return;
        }
        pa.order = order2;
        prev = stops[stage];
      }
    };
    const size_t lds_cap = 64 * 1024;
    bool launched = true;
    if (s == 3 && N <= 8 && anet::persist_lds_fixed_bytes<3, 8>() + row_bytes <= lds_cap)
      launch(anet::k_lbfgs_minco_persistent<3, 8, 8>, anet::persist_lds_fixed_bytes<3, 8>());
    else if (s == 3 && anet::persist_lds_fixed_bytes<3, 16>() + row_bytes <= lds_cap)
      launch(anet::k_lbfgs_minco_persistent<3, 16, 8>, anet::persist_lds_fixed_bytes<3, 16>());
    else if (s == 4 && N <= 8 && anet::persist_lds_fixed_bytes<4, 8>() + row_bytes <= lds_cap)
      launch(anet::k_lbfgs_minco_persistent<4, 8, 8>, anet::persist_lds_fixed_bytes<4, 8>());
    else if (s == 4 && anet::persist_lds_fixed_bytes<4, 16>() + row_bytes <= lds_cap)
      launch(anet::k_lbfgs_minco_persistent<4, 16, 8>, anet::persist_lds_fixed_bytes<4, 16>());
    else
      launched = false;
    if (order_failed) return ANET_ERR_INVALID;
    if (launched) {
      ANET_HIP(ctx, hipGetLastError());
#ifdef ANET_PERSIST_PROF
      {
        long long h[16];
        ANET_HIP(ctx, hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, st));
        ANET_HIP(ctx, hipStreamSynchronize(st));
        fprintf(stderr, "persist_prof cycles (problem 0): E1 %lld E2 %lld E3 %lld E4 %lld E5 %lld E6 %lld E7 %lld E8 %lld update %lld\n",
                h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
      }
#endif
      mp.mode = 1;
      hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
      ANET_HIP(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_lbfgs_results, g256, b256, 0, st, L.is, L.ds, batch, ld, status, iters, evals, cost);
      ANET_HIP(ctx, hipGetLastError());
      return coeffs_out ? final_coeffs(ctx, s, c, N, batch, ld, head, tail, wps, T, coeffs_out, st) : ANET_OK;
    }
  }
  // the lockstep shape: the same minimum-duration bound (one maximum over the duration variables per iteration) and the
  // same cancel word, looked at after every successful line search (lbfgs.hpp:557-565, 580-587)
  const double tau_min = min_duration > 1.0 ? sqrt(2.0 * min_duration - 1.0) - 1.0
                                            : (min_duration > 0.0 ? 1.0 - sqrt(2.0 / min_duration - 1.0) : 0.0);
  rc = lbfgs_drive(ctx, L, batch, *params, max_evals, st, [&]() -> int {
    return cost_grad_dev_impl(ctx, s, c, N, batch, ld, head, tail, wps_eval, T, hpolys, pen, w_cg, L.feval, gP_out,
                              gT_out, nullptr, st, tau);
  }, nt ? T : nullptr, nw, true, (min_duration > 0.0 && nt > 0) ? 1 : 0, tau_min, ctx->cancel_flag);
  if (rc) return rc;
  // final parameters (x may have been reverted by a failed line search) and outputs
  mp.mode = 1;
  hipLaunchKernelGGL(anet::k_minco_map, gmap, b256, 0, st, mp);
  ANET_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_lbfgs_results, g256, b256, 0, st, L.is, L.ds, batch, ld, status, iters, evals, cost);
  ANET_HIP(ctx, hipGetLastError());
  return coeffs_out ? final_coeffs(ctx, s, c, N, batch, ld, head, tail, wps, T, coeffs_out, st) : ANET_OK;
}

int anet_lbfgs_minco_ordered_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                 const double *head, const double *tail, double *wps, double *T,
                                 const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                                 int opt_flags, int max_evals, const int32_t *launch_order, double *work, double *cost,
                                 double *coeffs_out, int32_t *status, int32_t *iters, int32_t *evals, void *stream) {
  return lbfgs_minco_dev_impl(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, hpolys, pen, params, opt_flags, max_evals, 0.0,
                              launch_order, work, cost, coeffs_out, status, iters, evals, stream);
}

int anet_lbfgs_minco_bounded_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                 const double *head, const double *tail, double *wps, double *T,
                                 const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                                 int opt_flags, int max_evals, double min_duration, const int32_t *launch_order, double *work,
                                 double *cost, double *coeffs_out, int32_t *status, int32_t *iters, int32_t *evals, void *stream) {
  if (ctx && !(min_duration >= 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco_bounded_dev: min_duration must be >= 0");
  return lbfgs_minco_dev_impl(ctx, s, c, n_pieces, batch, ld, head, tail, wps, T, hpolys, pen, params, opt_flags, max_evals,
                              min_duration, launch_order, work, cost, coeffs_out, status, iters, evals, stream);
}

static int lbfgs_minco_host_impl(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                                 const double *tail, double *wps, double *T, const double *hpolys,
                                 const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                                 int max_evals, double min_duration, double *cost, double *coeffs_out, int32_t *status,
                                 int32_t *iters, int32_t *evals) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if ((rc = check_penalty(ctx, pen))) return rc;
  if (!params) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: params is NULL");
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: NULL pointer");
  const int N = n_pieces;
  const int64_t nco = (int64_t)N * 3 * 2 * s;
  const int64_t M = (pen && hpolys) ? pen->poly_rows : 0;
  const int64_t nhp = (int64_t)N * M * 4;
  // (the workspace has terms that do not scale with ld: asked for with the stager's own row stride, reserved in rows of it)
  const int64_t ld_h = batch == 1 ? 1 : anet_recommended_ld(batch);
  const int64_t wtotal = anet_lbfgs_minco_workspace(s, N, ld_h, params);
  if (wtotal < 0) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco: bad lbfgs parameters");
  const int64_t wdoubles = (wtotal + ld_h - 1) / ld_h;
  int64_t mx = nco > nhp ? nco : nhp;
  if (mx < 3 * (int64_t)c) mx = 3 * c;
  Stager st;
  rc = make_stager(ctx, batch, mx, 6 * (int64_t)c + 3 * (int64_t)(N - 1) + N + nhp + nco + wdoubles + 4, &st);
  if (rc) return rc;
  double *d_head, *d_tail, *d_wps, *d_T, *d_hp = nullptr;
  if ((rc = st.upload(head, 3 * c, &d_head))) return rc;
  if ((rc = st.upload(tail, 3 * c, &d_tail))) return rc;
  if ((rc = st.upload(wps, (int64_t)(N - 1) * 3, &d_wps))) return rc;
  if ((rc = st.upload(T, N, &d_T))) return rc;
  if (nhp && (rc = st.upload(hpolys, nhp, &d_hp))) return rc;
  double *d_co = st.reserve(nco), *d_work = st.reserve(wdoubles), *d_cost = st.reserve(1);
  int *d_res = (int *)st.reserve(3);
  rc = lbfgs_minco_dev_impl(ctx, s, c, N, batch, st.ld, d_head, d_tail, d_wps, d_T, d_hp, pen, params, opt_flags,
                            max_evals, min_duration, nullptr, d_work, d_cost, coeffs_out ? d_co : nullptr, d_res, d_res + st.ld,
                            d_res + 2 * st.ld, ctx->stream);
  if (rc) return rc;
  hipStream_t s0 = ctx->stream;
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_res, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_res + st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (evals) ANET_HIP(ctx, hipMemcpyAsync(evals, d_res + 2 * st.ld, sizeof(int) * batch, hipMemcpyDeviceToHost, s0));
  if (cost) ANET_HIP(ctx, hipMemcpyAsync(cost, d_cost, sizeof(double) * batch, hipMemcpyDeviceToHost, s0));
  if (N > 1 && (rc = st.download(d_wps, (int64_t)(N - 1) * 3, wps))) return rc;
  if ((rc = st.download(d_T, N, T))) return rc;
  if (coeffs_out && (rc = st.download(d_co, nco, coeffs_out))) return rc;
  ANET_HIP(ctx, hipStreamSynchronize(s0));
  return ANET_OK;
}

int anet_lbfgs_minco(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, double *wps, double *T, const double *hpolys,
                     const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                     int max_evals, double *cost, double *coeffs_out, int32_t *status, int32_t *iters,
                     int32_t *evals) {
  return lbfgs_minco_host_impl(ctx, s, c, n_pieces, batch, head, tail, wps, T, hpolys, pen, params, opt_flags, max_evals, 0.0,
                               cost, coeffs_out, status, iters, evals);
}

int anet_lbfgs_minco_bounded(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                             const double *tail, double *wps, double *T, const double *hpolys,
                             const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                             int max_evals, double min_duration, double *cost, double *coeffs_out, int32_t *status,
                             int32_t *iters, int32_t *evals) {
  if (ctx && !(min_duration >= 0.0)) return fail(ctx, ANET_ERR_INVALID, "anet_lbfgs_minco_bounded: min_duration must be >= 0");
  return lbfgs_minco_host_impl(ctx, s, c, n_pieces, batch, head, tail, wps, T, hpolys, pen, params, opt_flags, max_evals,
                               min_duration, cost, coeffs_out, status, iters, evals);
}

// ---- QP assembly entry points --------------------------------------------------------------------
int anet_qp_dims_of(int s, int n_pieces, int res, const int32_t *rows, anet_qp_dims *out) {
  if (!out || !rows || (s != 3 && s != 4) || n_pieces < 1 || res < 1) return ANET_ERR_INVALID;
  int64_t tot = 0;
  for (int i = 0; i < n_pieces; ++i) {
    if (rows[i] < 0) return ANET_ERR_INVALID;
    tot += rows[i];
  }
  out->n = (int64_t)3 * 2 * s * n_pieces;
  out->m_e = 3 * (6 + (int64_t)s * (n_pieces - 1));
  out->m_g = (int64_t)res * (tot + 12 * (int64_t)n_pieces);
  return ANET_OK;
}

int anet_qp_assemble_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M,
                         double max_vel, double max_acc, double m34, int float_time, int row_order,
                         const double *state, const double *T, const double *hpolys, const int32_t *rows,
                         double *Q, double *A, double *b, double *G, double *h, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (s != 3 && s != 4) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: order must be 3 (jerk) or 4 (snap), qp_solver.hpp:61-83");
  if (n_pieces < 1 || batch < 0 || res < 1 || M < 0 || (row_order != 0 && row_order != 1))
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || !rows || (M > 0 && !hpolys) || !Q || !A || !b || !G || !h)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: NULL pointer");
  // one shape for the whole batch: read the first trajectory's row counts (device -> host, tiny)
  std::vector<int32_t> r0((size_t)n_pieces * batch);
  ANET_HIP(ctx, hipMemcpy(r0.data(), rows, sizeof(int32_t) * n_pieces * batch, hipMemcpyDeviceToHost));
  anet_qp_dims dm;
  if (anet_qp_dims_of(s, n_pieces, res, r0.data(), &dm)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad row counts");
  for (int64_t bb = 0; bb < batch; ++bb) {
    int64_t tot = 0;
    for (int i = 0; i < n_pieces; ++i) {
      const int32_t v = r0[(size_t)bb * n_pieces + i];
      if (v < 0 || v > M) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: rows[b][i] must be in [0, M]");
      tot += v;
    }
    if ((int64_t)res * (tot + 12 * (int64_t)n_pieces) != dm.m_g)
      return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: every trajectory of a batch needs the same total polytope row count");
  }
  anet::QpArgs a{state, T, hpolys, rows, Q, A, b, G, h, batch, dm.n, dm.m_e, dm.m_g, n_pieces, res, M,
                 float_time ? 1 : 0, row_order, max_vel, max_acc, m34};
  hipStream_t st = (hipStream_t)stream;
  const int64_t ne = dm.n * dm.n + dm.m_e * dm.n + dm.m_e, ng = dm.m_g * dm.n;
  const dim3 blk(256), g1((unsigned)((ne + 255) / 256), (unsigned)batch), g2((unsigned)((ng + 255) / 256), (unsigned)batch);
  if (s == 4) {
    if (float_time) { hipLaunchKernelGGL((anet::k_qp_eq_obj<4, float>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<4, float>), g2, blk, 0, st, a); }
    else { hipLaunchKernelGGL((anet::k_qp_eq_obj<4, double>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<4, double>), g2, blk, 0, st, a); }
  } else {
    if (float_time) { hipLaunchKernelGGL((anet::k_qp_eq_obj<3, float>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<3, float>), g2, blk, 0, st, a); }
    else { hipLaunchKernelGGL((anet::k_qp_eq_obj<3, double>), g1, blk, 0, st, a); if (ng) hipLaunchKernelGGL((anet::k_qp_ineq<3, double>), g2, blk, 0, st, a); }
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_qp_assemble(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                     double max_acc, double m34, int float_time, int row_order, const double *state,
                     const double *T, const double *hpolys, const int32_t *rows, double *Q, double *A,
                     double *b, double *G, double *h) {
  ANET_ON_DEVICE(ctx);
  if ((s != 3 && s != 4) || n_pieces < 1 || batch < 0 || res < 1 || M < 0 || !rows)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad argument");
  if (batch == 0) return ANET_OK;
  anet_qp_dims dm;
  if (anet_qp_dims_of(s, n_pieces, res, rows, &dm)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_assemble: bad row counts");
  const size_t n_state = 18 * (size_t)batch, n_T = (size_t)n_pieces * batch, n_hp = (size_t)batch * n_pieces * M * 4;
  const size_t n_rows = ((size_t)n_pieces * batch + 1) / 2;  // int32 pairs in doubles
  const size_t nQ = (size_t)(dm.n * dm.n) * batch, nA = (size_t)(dm.m_e * dm.n) * batch, nb = (size_t)dm.m_e * batch;
  const size_t nG = (size_t)(dm.m_g * dm.n) * batch, nh = (size_t)dm.m_g * batch;
  int rc = ensure_scratch(ctx, sizeof(double) * (n_state + n_T + n_hp + n_rows + nQ + nA + nb + nG + nh + 8));
  if (rc) return rc;
  double *d_state = (double *)ctx->scratch, *d_T = d_state + n_state, *d_hp = d_T + n_T;
  int32_t *d_rows = (int32_t *)(d_hp + n_hp);
  double *d_Q = d_hp + n_hp + n_rows, *d_A = d_Q + nQ, *d_b = d_A + nA, *d_G = d_b + nb, *d_h = d_G + nG;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_state, state, sizeof(double) * n_state, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_T, T, sizeof(double) * n_T, hipMemcpyHostToDevice, st));
  if (n_hp) ANET_HIP(ctx, hipMemcpyAsync(d_hp, hpolys, sizeof(double) * n_hp, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_rows, rows, sizeof(int32_t) * n_pieces * batch, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  rc = anet_qp_assemble_dev(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, float_time, row_order, d_state,
                            d_T, d_hp, d_rows, d_Q, d_A, d_b, d_G, d_h, st);
  if (rc) return rc;
  if (Q) ANET_HIP(ctx, hipMemcpyAsync(Q, d_Q, sizeof(double) * nQ, hipMemcpyDeviceToHost, st));
  if (A) ANET_HIP(ctx, hipMemcpyAsync(A, d_A, sizeof(double) * nA, hipMemcpyDeviceToHost, st));
  if (b) ANET_HIP(ctx, hipMemcpyAsync(b, d_b, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
  if (G && nG) ANET_HIP(ctx, hipMemcpyAsync(G, d_G, sizeof(double) * nG, hipMemcpyDeviceToHost, st));
  if (h && nh) ANET_HIP(ctx, hipMemcpyAsync(h, d_h, sizeof(double) * nh, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

// ---- QP solve (ADMM) entry points ------------------------------------------------------------------
void anet_qp_default_settings(anet_qp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->max_iter = 4000; s->check_termination = 25; s->adaptive_rho_interval = 100; s->scaled_termination = 0;
  s->method = ANET_QP_METHOD_INTERIOR_POINT;
}

int64_t anet_qp_solve_workspace(int s, int n_pieces, int64_t batch, int res, int M) {
  const int64_t m = 3 * (6 + (int64_t)s * (n_pieces - 1)) + (int64_t)n_pieces * res * (M + 12);
  // z, y, residuals; then (interior point, two-launch form) the parked state of every problem, its score and the launch order of
  // the second part (int32 each) and the 4096 bins of the counting sort
  const int64_t cont = (int64_t)3 * s * (n_pieces + 1) + anet::kIpmContScalars;
  return 2 * m * batch + 2 * batch + cont * batch + batch + 4096 / 2 + 8;
}

// Second part of a two-launch interior-point solve: what order its workgroups take the problems in.  Score of a parked problem,
// larger = expected to take longer (tools/qp_split_features.py: the parked scalars of 3 x 4096 problems against the steps they still
// needed).  The problems a batch waits for -- the infeasible ones, told at step 30..50, and the hard feasible ones -- stand out
// after four steps already: their PRIMAL residual is still above 2e-4 (1e-3..4e-2 against <= 5e-5 for the rest) and their steps
// are short; they go first, by residual.  Behind them the rest by the decades their complementarity stands above the tolerance
// (Spearman 0.75..0.94 with the steps left: an interior point gains a fixed number of digits per step at the end).  Problems
// decided in the first part score 0 and come last (their workgroups leave at once).
__global__ void k_qp_resume_score(const int *status, const double *cont, int64_t B, int ny, double tol, int *score) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int sc = 0;
  if (status[b] == 0) {
    const double *c = cont + b * (int64_t)(ny + anet::kIpmContScalars) + ny;
    const double pres = c[8], gap = c[10];
    double v = 200.0 + 100.0 * log10(fmax(gap / tol, 1.0));
    if (!(pres <= 2e-4)) v = 2000.0 + 100.0 * log10(fmax(pres / 2e-4, 1.0));
    if (!(v == v)) v = 4000.0;
    sc = (int)fmin(fmax(v, 1.0), 4000.0);
  }
  score[b] = sc;
}

// A caller's launch order is not checked (device memory): whatever it skips -- out-of-range or repeated entries -- must say so
__global__ void k_qp_mark_not_run(int64_t B, int *status, int *iters, double *obj) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  status[b] = ANET_QP_UNSOLVED;
  iters[b] = 0;
  obj[b] = __builtin_nan("");
}

static int qp_solve_dev_impl(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                             double max_acc, double m34, const double *state, const double *T,
                             const double *hpolys, const anet_qp_settings *settings, double *work, double *coeffs,
                             double *obj, int32_t *status, int32_t *iters, double *residuals, double *grad_T,
                             void *stream, const double *grad_z = nullptr, double *vjp_T = nullptr,
                             const int32_t *launch_order = nullptr) {
  ANET_ON_DEVICE(ctx);
  if (s != 3 && s != 4) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: order must be 3 (jerk) or 4 (snap)");
  if (n_pieces < 1 || batch < 0 || res < 1 || M < 0) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || (M > 0 && !hpolys) || !work || !coeffs || !obj || !status || !iters)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: NULL pointer");
  anet_qp_settings st_;
  anet_qp_default_settings(&st_);
  if (settings) st_ = *settings;
  if (!(st_.rho > 0) || !(st_.sigma > 0) || !(st_.alpha > 0 && st_.alpha < 2) || st_.max_iter < 1 ||
      st_.check_termination < 1 || st_.eps_abs < 0 || st_.eps_rel < 0 || st_.adaptive_rho_interval < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad settings");
  if (st_.method != ANET_QP_METHOD_ADMM && st_.method != ANET_QP_METHOD_INTERIOR_POINT)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: unknown method");
  if (st_.method == ANET_QP_METHOD_INTERIOR_POINT) {
    const size_t ldsb = (s == 4) ? anet::qp_ipm_lds_bytes<4>(n_pieces, res, M) : anet::qp_ipm_lds_bytes<3>(n_pieces, res, M);
    if (ldsb > 160 * 1024)
      return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_qp_solve: problem too large for the 160 KB LDS (interior-point method)");
    const int64_t mi = (int64_t)n_pieces * res * (M + 12);
    double tol = st_.eps_rel < st_.eps_abs ? st_.eps_rel : st_.eps_abs;
    if (!(tol > 0.0) || tol > 1e-6) tol = 1e-6;   // Newton's method: the last digits cost one or two steps
    if (tol < 1e-10) tol = 1e-10;                 // (below that the slacks of the touched rows underflow the factorisation)
    // the backward pass differentiates the central path at the barrier parameter the solve stopped at: its error is
    // of that order, so it asks for three more digits (one or two Newton steps)
    const double tol_plain = tol;
    if (grad_z && tol > 1e-9) tol = 1e-9;
    anet::IpmArgs ia{state, T, hpolys, work, work + mi * batch, coeffs, obj, status, iters,
                     residuals ? residuals : work + 2 * mi * batch, grad_T, grad_z, vjp_T, batch, n_pieces, res, M, max_vel,
                     max_acc, m34, tol, st_.max_iter < 200 ? st_.max_iter : 200, tol_plain > tol ? tol_plain : 0.0, 0.1 * tol, 0, launch_order, 0, 0, nullptr, nullptr};
    static const int ipm_twist_min_pieces = [] {
      const char *e = getenv("ANET_IPM_TWIST_MIN_PIECES");
      return e ? atoi(e) : 2;
    }();
    ia.twist_min_pieces = ipm_twist_min_pieces;
    hipStream_t sti = (hipStream_t)stream;
    if (launch_order) {
      hipLaunchKernelGGL(k_qp_mark_not_run, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, sti, batch, status, iters, obj);
      ANET_HIP(ctx, hipGetLastError());
    }
    {  // the tables of (order, res, m34): built once, on the stream that first needs them
      const double *tab = nullptr;
      for (auto &tb : ctx->ipm_tabs)
        if (tb.s == s && tb.res == res && tb.m34 == m34) {
          if (tb.built_on != sti) ANET_HIP(ctx, hipStreamWaitEvent(sti, tb.ready, 0));
          tab = tb.d;
        }
      if (!tab) {
        // (never freed before anet_destroy -- a launch on another stream may still read one: a caller that sweeps m34 or res over
        //  hundreds of values is told so instead of growing the context without bound)
        if (ctx->ipm_tabs.size() >= kMaxTablesPerContext)
          return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_qp_solve: more than 256 distinct (order, res, m34) on one context");
        anet_ctx::IpmTable tb{s, res, m34, nullptr, sti, nullptr};
        const size_t need = (size_t)2 * (2 * s) * (2 * s) + (size_t)res * anet::ipm_ht_stride(2 * s);
        int rc_t = new_table(ctx, sizeof(double) * need, &tb.d, &tb.ready);
        if (rc_t) return rc_t;
        hipError_t e1 = hipMemsetAsync(tb.d, 0, sizeof(double) * need, sti);
        if (e1 == hipSuccess) {
          if (s == 4) hipLaunchKernelGGL(anet::k_qp_ipm_tables<4>, dim3(1), dim3(256), 0, sti, tb.d, res, m34);
          else hipLaunchKernelGGL(anet::k_qp_ipm_tables<3>, dim3(1), dim3(256), 0, sti, tb.d, res, m34);
          e1 = hipGetLastError();
        }
        if (e1 == hipSuccess) e1 = hipEventRecord(tb.ready, sti);
        if (e1 != hipSuccess) {  // nothing half-built stays behind
          drop_table(tb.d, tb.ready);
          return hip_fail(ctx, e1, "k_qp_ipm_tables");
        }
        ctx->ipm_tabs.push_back(tb);
        tab = tb.d;
      }
      ia.tab = tab;
    }
#ifdef ANET_IPM_PROF
    static long long *d_iprof = nullptr;
    if (!d_iprof) ANET_HIP(ctx, hipMalloc((void **)&d_iprof, 16 * sizeof(long long)));
    ANET_HIP(ctx, hipMemsetAsync(d_iprof, 0, 16 * sizeof(long long), sti));
    ia.prof = d_iprof;
    struct ProfDump {
      anet_ctx *c; long long *d; hipStream_t st;
      ~ProfDump() {
        long long h[16];
        if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
        fprintf(stderr, "ipm_prof cycles (problem 0): setup %lld | passA %lld resid %lld assemble %lld rhs %lld factor %lld solve1 %lld passB %lld passC %lld rhs2 %lld solve2 %lld passD %lld passE %lld | before the loop (tables, first iterate) %lld\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13]);
      }
    } prof_dump{ctx, d_iprof, sti};
#endif
    // two workgroups per CU (registers bounded to 256) from this batch on, when two fit the LDS
    static const int64_t two_per_cu_env = [] {
      const char *e = getenv("ANET_IPM_TWO_PER_CU_MIN_BATCH");
      return e ? (int64_t)atoll(e) : (int64_t)-1;
    }();
    // more than two rounds of one workgroup per CU (measured on 256 CUs): below that a batch lasts as long as its slowest problem
    // and a problem alone on its CU is faster (512 problems are a draw -- 2.86 / 1.88 / 2.95 ms against 2.64 / 1.63 /
    // 3.42 ms for 8 snap / 5 jerk / 5 snap pieces --, 768 problems gain 15-20 % from two per CU, 320 lose 15 %)
    const int64_t ipm_two_per_cu_min_batch = two_per_cu_env >= 0 ? two_per_cu_env : 2 * (int64_t)ctx->cus + 1;
    const bool two_per_cu = batch >= ipm_two_per_cu_min_batch && 2 * ldsb <= 160 * 1024;
    // ... and THREE for jerk problems whose LDS allows it, from a batch on that fills them several times over (registers bounded
    // to 168: 464 B of scratch).  Measured (round 5, same box, 5 jerk pieces): 4096 problems 3.96-4.00 -> 3.77-3.85 ms; 3000:
    // 3.02-3.07 -> 3.21-3.25; 2048: 2.09-2.12 -> 2.42-2.43 (1024: 1.60 -> 1.91 in round 4) -- selected by batch like every other
    // shape here (ANET_IPM_THREE_PER_CU_MIN_BATCH overrides; 0 disables)
    static const int64_t three_per_cu_env = [] {
      const char *e = getenv("ANET_IPM_THREE_PER_CU_MIN_BATCH");
      return e ? (int64_t)atoll(e) : (int64_t)-1;
    }();
    const int64_t ipm_three_per_cu_min_batch = three_per_cu_env >= 0 ? three_per_cu_env : per_cu(ctx, 4096);
    const bool three_per_cu = s == 3 && two_per_cu && ipm_three_per_cu_min_batch > 0 && batch >= ipm_three_per_cu_min_batch &&
                              3 * ldsb <= 160 * 1024;
    auto launch_ipm = [&](auto kern) -> int {
      ANET_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(256), ldsb, sti, ia);
      return ANET_OK;
    };
    auto launch_throughput = [&]() -> int {  // the shape of large batches: four row passes, registers bounded for 2 or 3 per CU
      if (s == 4) return launch_ipm(anet::k_qp_ipm<4, 2, false>);
      return three_per_cu ? launch_ipm(anet::k_qp_ipm<3, 3, false>) : launch_ipm(anet::k_qp_ipm<3, 1, false>);
    };
    int rc_l;
    // Large batches in TWO launches (qp_ipm.h, IpmArgs::it_stop): the first takes every problem through the same number of Newton
    // steps -- no tail: all workgroups are equally long --, the second resumes the unfinished ones longest-expected first.  A batch
    // of 4096 in one launch ends 27 % above its balanced figure because its 30..50-step problems start whenever their turn comes.
    static const int ipm_split_steps = [] { const char *e = getenv("ANET_IPM_SPLIT_STEPS"); return e ? atoi(e) : 4; }();
    static const int64_t ipm_split_env = [] { const char *e = getenv("ANET_IPM_SPLIT_MIN_BATCH"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
    const int64_t ipm_split_min_batch = ipm_split_env >= 0 ? ipm_split_env : per_cu(ctx, 576);  // (256 CUs: 520..560 problems lose 7-10 %, 600..1280 gain 10-19 %)
    if (two_per_cu && ipm_split_steps > 0 && batch >= ipm_split_min_batch && !launch_order && ia.max_iter > ipm_split_steps) {
      const int ny = 3 * s * (n_pieces + 1);
      const int64_t m_adm = 3 * (6 + (int64_t)s * (n_pieces - 1)) + mi;
      double *cont = work + 2 * m_adm * batch + 2 * batch;
      int32_t *score = (int32_t *)(cont + (int64_t)(ny + anet::kIpmContScalars) * batch);
      int32_t *order2 = score + batch + (batch & 1);
      int32_t *bins = order2 + batch + (batch & 1);
      ia.cont = cont;
      ia.it_stop = ipm_split_steps;
      rc_l = launch_throughput();
      if (rc_l != ANET_OK) return rc_l;
      hipLaunchKernelGGL(k_qp_resume_score, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, sti, status, cont, batch, ny, tol, score);
      rc_l = launch_order_impl(ctx, batch, score, order2, bins, sti, 0);
      if (rc_l != ANET_OK) return rc_l;
      ia.it_stop = 0;
      ia.resume = 1;
      ia.order = order2;
      rc_l = launch_throughput();
      if (rc_l != ANET_OK) return rc_l;
      ANET_HIP(ctx, hipGetLastError());
      return ANET_OK;
    }
    // Shapes that put two workgroups on a CU visit the rows once more per step instead of carrying the next step's sums
    // through the updating pass (registers: qp_ipm.h FUSE); a lone problem, a small batch or a problem whose LDS fills the
    // CU takes the fused form.  (jerk: the unbounded instantiation needs <= 256 registers as it is -- two workgroups per CU
    // -- and the compiler schedules it for latency; bounded to 256 it is 18 % slower per problem at the same occupancy)
    if (!two_per_cu) {  // (qp_ipm_fuse_unit.hip: the FUSE instantiations, scheduled for ILP)
      ANET_HIP(ctx, (hipError_t)anet::launch_qp_ipm_fuse(s, batch, ldsb, sti, ia));
      rc_l = ANET_OK;
    } else rc_l = launch_throughput();
    if (rc_l != ANET_OK) return rc_l;
    ANET_HIP(ctx, hipGetLastError());
    return ANET_OK;
  }
  if (grad_z) return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_qp_solve_vjp: the backward pass needs the interior-point method");
  size_t lds = (s == 4) ? anet::qp_admm_lds_bytes<4>(n_pieces, res, M, true) : anet::qp_admm_lds_bytes<3>(n_pieces, res, M, true);
  int zy_in_lds = 1;
  if (lds > 160 * 1024) {
    zy_in_lds = 0;
    lds = (s == 4) ? anet::qp_admm_lds_bytes<4>(n_pieces, res, M, false) : anet::qp_admm_lds_bytes<3>(n_pieces, res, M, false);
  }
  if (lds > 160 * 1024)
    return fail(ctx, ANET_ERR_UNSUPPORTED, "anet_qp_solve: the block factor of this many pieces does not fit the 160 KB LDS");
  const int64_t m = 3 * (6 + (int64_t)s * (n_pieces - 1)) + (int64_t)n_pieces * res * (M + 12);
  int adapt = st_.adaptive_rho_interval;
  if (adapt > 0) adapt = (adapt + st_.check_termination - 1) / st_.check_termination * st_.check_termination;
  anet::AdmmArgs a{state, T, hpolys, work, work + m * batch, coeffs, obj, status, iters,
                   residuals ? residuals : work + 2 * m * batch, batch, n_pieces, res, M, max_vel, max_acc, m34,
                   anet::AdmmParams{st_.rho, st_.sigma, st_.alpha, st_.eps_abs, st_.eps_rel, st_.max_iter,
                                    st_.check_termination, adapt, st_.scaled_termination ? 1 : 0},
                   zy_in_lds, grad_T};
  hipStream_t st = (hipStream_t)stream;
  if (s == 4) {
    ANET_HIP(ctx, hipFuncSetAttribute((const void *)anet::k_qp_admm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((anet::k_qp_admm<4>), dim3((unsigned)batch), dim3(256), lds, st, a);
  } else {
    ANET_HIP(ctx, hipFuncSetAttribute((const void *)anet::k_qp_admm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((anet::k_qp_admm<3>), dim3((unsigned)batch), dim3(256), lds, st, a);
  }
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_qp_solve_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                      double max_acc, double m34, const double *state, const double *T,
                      const double *hpolys, const anet_qp_settings *settings, double *work, double *coeffs,
                      double *obj, int32_t *status, int32_t *iters, double *residuals, void *stream) {
  return qp_solve_dev_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, work,
                           coeffs, obj, status, iters, residuals, nullptr, stream);
}

int anet_qp_solve_ordered_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                              double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                              const anet_qp_settings *settings, const int32_t *launch_order, double *work, double *coeffs,
                              double *obj, int32_t *status, int32_t *iters, double *residuals, void *stream) {
  return qp_solve_dev_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, work,
                           coeffs, obj, status, iters, residuals, nullptr, stream, nullptr, nullptr, launch_order);
}

int anet_qp_solve_time_grad_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                                double max_acc, double m34, const double *state, const double *T,
                                const double *hpolys, const anet_qp_settings *settings, double *work,
                                double *coeffs, double *obj, int32_t *status, int32_t *iters, double *residuals,
                                double *grad_T, void *stream) {
  if (ctx && batch > 0 && !grad_T) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve_time_grad: grad_T is NULL");
  return qp_solve_dev_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, work,
                           coeffs, obj, status, iters, residuals, grad_T, stream);
}

int anet_qp_solve_vjp_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                          double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                          const anet_qp_settings *settings, const double *grad_z, double *work, double *coeffs,
                          double *obj, int32_t *status, int32_t *iters, double *residuals, double *grad_T,
                          void *stream) {
  if (ctx && batch > 0 && (!grad_z || !grad_T)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve_vjp: grad_z / grad_T is NULL");
  return qp_solve_dev_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, work,
                           coeffs, obj, status, iters, residuals, nullptr, stream, grad_z, grad_T);
}

static int qp_solve_host_impl(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                              double max_acc, double m34, const double *state, const double *T,
                              const double *hpolys, const anet_qp_settings *settings, double *coeffs, double *obj,
                              int32_t *status, int32_t *iters, double *residuals, double *grad_T,
                              const double *grad_z = nullptr) {
  ANET_ON_DEVICE(ctx);
  if ((s != 3 && s != 4) || n_pieces < 1 || batch < 0 || res < 1 || M < 0)
    return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: bad argument");
  if (batch == 0) return ANET_OK;
  if (!state || !T || (M > 0 && !hpolys) || !coeffs) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve: NULL pointer");
  const size_t n = (size_t)3 * 2 * s * n_pieces;
  const size_t n_state = 18 * (size_t)batch, n_T = (size_t)n_pieces * batch, n_hp = (size_t)batch * n_pieces * M * 4;
  const size_t n_work = (size_t)anet_qp_solve_workspace(s, n_pieces, batch, res, M);
  const size_t n_int = (size_t)batch;  // 2 int32 arrays fit in `batch` doubles
  int rc = ensure_scratch(ctx, sizeof(double) * (n_state + n_T + n_hp + n_work + 2 * n * batch + 3 * batch + n_int + n_T + 8));
  if (rc) return rc;
  double *d_state = (double *)ctx->scratch, *d_T = d_state + n_state, *d_hp = d_T + n_T, *d_work = d_hp + n_hp;
  double *d_co = d_work + n_work, *d_obj = d_co + n * batch, *d_res = d_obj + batch;
  int32_t *d_status = (int32_t *)(d_res + 2 * batch), *d_iters = d_status + batch;
  double *d_gT = d_res + 2 * batch + n_int;
  double *d_gz = d_gT + n_T;
  hipStream_t st = ctx->stream;
  ANET_HIP(ctx, hipMemcpyAsync(d_state, state, sizeof(double) * n_state, hipMemcpyHostToDevice, st));
  ANET_HIP(ctx, hipMemcpyAsync(d_T, T, sizeof(double) * n_T, hipMemcpyHostToDevice, st));
  if (n_hp) ANET_HIP(ctx, hipMemcpyAsync(d_hp, hpolys, sizeof(double) * n_hp, hipMemcpyHostToDevice, st));
  if (grad_z) ANET_HIP(ctx, hipMemcpyAsync(d_gz, grad_z, sizeof(double) * n * batch, hipMemcpyHostToDevice, st));
  rc = qp_solve_dev_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, d_state, d_T, d_hp, settings, d_work,
                         d_co, d_obj, d_status, d_iters, d_res, (grad_T && !grad_z) ? d_gT : nullptr, st,
                         grad_z ? d_gz : nullptr, grad_z ? d_gT : nullptr);
  if (rc) return rc;
  if (grad_T) ANET_HIP(ctx, hipMemcpyAsync(grad_T, d_gT, sizeof(double) * n_T, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipMemcpyAsync(coeffs, d_co, sizeof(double) * n * batch, hipMemcpyDeviceToHost, st));
  if (obj) ANET_HIP(ctx, hipMemcpyAsync(obj, d_obj, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
  if (status) ANET_HIP(ctx, hipMemcpyAsync(status, d_status, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, st));
  if (iters) ANET_HIP(ctx, hipMemcpyAsync(iters, d_iters, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, st));
  if (residuals) ANET_HIP(ctx, hipMemcpyAsync(residuals, d_res, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_qp_solve(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                  double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                  const anet_qp_settings *settings, double *coeffs, double *obj, int32_t *status,
                  int32_t *iters, double *residuals) {
  return qp_solve_host_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, coeffs,
                            obj, status, iters, residuals, nullptr);
}

int anet_qp_solve_time_grad(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                            double max_acc, double m34, const double *state, const double *T,
                            const double *hpolys, const anet_qp_settings *settings, double *coeffs, double *obj,
                            int32_t *status, int32_t *iters, double *residuals, double *grad_T) {
  if (ctx && batch > 0 && !grad_T) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve_time_grad: grad_T is NULL");
  return qp_solve_host_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, coeffs,
                            obj, status, iters, residuals, grad_T);
}

int anet_qp_solve_vjp(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                      double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                      const anet_qp_settings *settings, const double *grad_z, double *coeffs, double *obj,
                      int32_t *status, int32_t *iters, double *residuals, double *grad_T) {
  if (ctx && batch > 0 && (!grad_z || !grad_T)) return fail(ctx, ANET_ERR_INVALID, "anet_qp_solve_vjp: grad_z / grad_T is NULL");
  return qp_solve_host_impl(ctx, s, n_pieces, batch, res, M, max_vel, max_acc, m34, state, T, hpolys, settings, coeffs,
                            obj, status, iters, residuals, grad_T, grad_z);
}

int anet_traj_cost_grad_T_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                              const double *coeffs, const double *T, double m34, double *gradT, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gradT || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_grad_T_dev: NULL pointer or ld < batch");
  anet::CostArgs a{coeffs, T, nullptr, gradT, batch, ld, n_pieces, m34};
  const dim3 grid((unsigned)((batch + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_traj_cost<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_traj_cost<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_traj_cost<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_traj_cost_grad_T(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                          const double *T, double m34, double *gradT) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !gradT) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_grad_T: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, nco, nco + 2 * (int64_t)n_pieces, &st);
  if (rc) return rc;
  double *d_co, *d_T;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  double *d_g = st.reserve(n_pieces);
  rc = anet_traj_cost_grad_T_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, m34, d_g, ctx->stream);
  if (rc) return rc;
  return st.download(d_g, n_pieces, gradT);
}

// ---- RCCL (loaded at run time) -------------------------------------------------------------------
namespace {
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
int load_rccl(anet_ctx *ctx) {
  if (g_rccl.handle) return ANET_OK;
  const char *env = getenv("ANET_RCCL_PATH");
  const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *c : cands) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(ctx, ANET_ERR_UNSUPPORTED, std::string("cannot load librccl.so: ") + dlerror());
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy)
    return fail(ctx, ANET_ERR_UNSUPPORTED, "librccl.so lacks the expected nccl* symbols");
  g_rccl = a;
  return ANET_OK;
}
int rccl_fail(anet_ctx *ctx, ncclResult_t r, const char *what) {
  return fail(ctx, ANET_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
}
}  // namespace

int anet_comm_unique_id(anet_ctx *ctx, unsigned char id[ANET_COMM_ID_BYTES]) {
  if (!ctx || !id) return fail(ctx, ANET_ERR_INVALID, "anet_comm_unique_id: NULL argument");
  int rc = load_rccl(ctx);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == ANET_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return rccl_fail(ctx, r, "ncclGetUniqueId");
  memcpy(id, &u, ANET_COMM_ID_BYTES);
  return ANET_OK;
}

int anet_comm_init(anet_ctx *ctx, int nranks, int rank, const unsigned char id[ANET_COMM_ID_BYTES]) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, ANET_ERR_INVALID, "anet_comm_init: bad argument");
  if (ctx->comm) return fail(ctx, ANET_ERR_INVALID, "anet_comm_init: communicator already initialised");
  int rc = load_rccl(ctx);
  if (rc) return rc;
  ANET_ON_DEVICE(ctx);
  ncclUniqueId u;
  memcpy(&u, id, ANET_COMM_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    ctx->comm = nullptr;
    return rccl_fail(ctx, r, "ncclCommInitRank");
  }
  ctx->comm_ranks = nranks;
  return ANET_OK;
}

int anet_comm_allgather_costs_dev(anet_ctx *ctx, const double *send, double *recv, int64_t count, void *stream) {
  ANET_ON_DEVICE(ctx);
  if (!ctx || !ctx->comm) return fail(ctx, ANET_ERR_INVALID, "anet_comm_allgather_costs_dev: call anet_comm_init first");
  if (!send || !recv || count < 0) return fail(ctx, ANET_ERR_INVALID, "anet_comm_allgather_costs_dev: bad argument");
  if (count == 0) return ANET_OK;
  ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)count, ncclFloat64, ctx->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail(ctx, r, "ncclAllGather");
  return ANET_OK;
}

int anet_comm_destroy(anet_ctx *ctx) {
  if (!ctx) return ANET_ERR_INVALID;
  if (ctx->comm && g_rccl.CommDestroy) {
    (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
  }
  return ANET_OK;
}

int anet_traj_max_rate_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                           const double *coeffs, const double *T, int which, double *rate, void *stream) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (which != 1 && which != 2) return fail(ctx, ANET_ERR_INVALID, "anet_traj_max_rate: which must be 1 (velocity) or 2 (acceleration)");
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !rate || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_max_rate_dev: NULL pointer or ld < batch");
  anet::RateArgs a{coeffs, T, rate, batch, ld, n_pieces, which};
  const dim3 grid((unsigned)((batch + 63) / 64), (unsigned)n_pieces), block(64);
  hipStream_t st = (hipStream_t)stream;
  if (s == 2) hipLaunchKernelGGL(anet::k_piece_max_rate<2>, grid, block, 0, st, a);
  else if (s == 3) hipLaunchKernelGGL(anet::k_piece_max_rate<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(anet::k_piece_max_rate<4>, grid, block, 0, st, a);
  ANET_HIP(ctx, hipGetLastError());
  return ANET_OK;
}

int anet_traj_max_rate(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                       const double *T, int which, double *rate) {
  ANET_ON_DEVICE(ctx);
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !rate) return fail(ctx, ANET_ERR_INVALID, "anet_traj_max_rate: NULL pointer");
  const int64_t nco = (int64_t)n_pieces * 3 * 2 * s;
  Stager st;
  rc = make_stager(ctx, batch, nco, nco + 2 * (int64_t)n_pieces, &st);
  if (rc) return rc;
  double *d_co, *d_T;
  if ((rc = st.upload(coeffs, nco, &d_co))) return rc;
  if ((rc = st.upload(T, n_pieces, &d_T))) return rc;
  double *d_r = st.reserve(n_pieces);
  rc = anet_traj_max_rate_dev(ctx, s, n_pieces, batch, st.ld, d_co, d_T, which, d_r, ctx->stream);
  if (rc) return rc;
  return st.download(d_r, n_pieces, rate);
}

}  // extern "C"
