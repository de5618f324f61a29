// allocnet_amd: gfx950 kernels + C ABI (include/allocnet_amd.h).  Built by allocnet_amd/build.py:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch, no Eigen.  There is no CPU fallback in this library: without a device every entry
// point fails with ANET_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <new>

#include "../../include/allocnet_amd.h"
#include "minco_core.h"

namespace anet {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
struct SolveArgs {
  const double *head, *tail, *wps, *T;
  double *coeffs, *energy;
  int64_t B, ld;
  int N, c;
};

constexpr int kSolveBlock = 64;

// One lane = one trajectory.  The block-tridiagonal factor is shared by the three axes and stays
// in registers; the axes are swept one after the other so only one axis' right-hand side is live.
// NEXACT: the piece count is exactly NB (compile time); NPC >= 0: c-1 is NPC (compile time).
// Both let every end-node / pinned-derivative mask fold away; the generic instantiation
// (NEXACT = false, NPC = -1) keeps them as wave-uniform selects.
template <int S, int NB, bool NEXACT = false, int NPC = -1>
__global__ void __launch_bounds__(kSolveBlock) k_minco_solve(SolveArgs a) {
  constexpr int m = S - 1, D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * kSolveBlock + threadIdx.x;
  if (b >= a.B) return;
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int c = np + 1;
  const int64_t ld = a.ld;

  Factor<S, NB> F;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < N) F.r[i] = fast_rcp(a.T[i * ld + b]);
  F.factorize(N, np);

  double etot = 0.0;
#pragma unroll 1
  for (int ax = 0; ax < 3; ++ax) {
    double P[NB + 1], hv[m], tv[m], X[NB + 1][m];
    const double *hp = a.head + (int64_t)(ax * c) * ld + b;
    const double *tp = a.tail + (int64_t)(ax * c) * ld + b;
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      if (k == 0)
        P[k] = hp[0];
      else if (k < N)
        P[k] = a.wps[(int64_t)((k - 1) * 3 + ax) * ld + b];
      else if (k == N)
        P[k] = tp[0];
      else
        P[k] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < m; ++j) {
      hv[j] = (j < np) ? hp[(int64_t)(1 + j) * ld] : 0.0;
      tv[j] = (j < np) ? tp[(int64_t)(1 + j) * ld] : 0.0;
    }
    double *cp = a.coeffs ? a.coeffs + (int64_t)(ax * D) * ld + b : nullptr;
    etot += solve_axis<S, NB>(F, N, np, P, hv, tv, X, [&](int piece, int col, double v) {
      if (cp) cp[(int64_t)(piece * 3 * D + col) * ld] = v;
    });
  }
  if (a.energy) a.energy[b] = etot;
}

// Trajectory<D>::getPos/Vel/Acc/Jer: one lane per trajectory, nq queries each.  The accumulation
// order is the reference's (ascending powers, tn *= t), trajectory.hpp:75-133.
struct EvalArgs {
  const double *coeffs, *T, *tq;
  double *out;
  int64_t B, ld;
  int N, nq, deriv;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_eval(EvalArgs a) {
  constexpr int D = 2 * S, DEG = D - 1;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  const int N = a.N, d = a.deriv;
  for (int q = 0; q < a.nq; ++q) {
    double t = a.tq[(int64_t)q * ld + b];
    // locatePieceIdx (trajectory.hpp:496-514)
    int idx = 0;
    double dur = 0.0;
    for (; idx < N; ++idx) {
      dur = a.T[(int64_t)idx * ld + b];
      if (!(t > dur)) break;
      t -= dur;
    }
    if (idx == N) {
      --idx;
      t += a.T[(int64_t)idx * ld + b];
    }
    const double *cm = a.coeffs + (int64_t)(idx * 3 * D) * ld + b;
    double acc[3] = {0.0, 0.0, 0.0};
    double tn = 1.0;
    for (int i = DEG - d; i >= 0; --i) {
      const int k = DEG - i;  // power of column i
      double f = 1.0;
      for (int e = 0; e < d; ++e) f *= (double)(k - e);
      const double w = f * tn;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) acc[ax] += w * cm[(int64_t)(ax * D + i) * ld];
      tn *= t;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) a.out[(int64_t)(q * 3 + ax) * ld + b] = acc[ax];
  }
}

// Trajectory<D>::getTrajCost (trajectory.hpp:354-427).
struct CostArgs {
  const double *coeffs, *T;
  double *cost;
  int64_t B, ld;
  int N;
  double m34;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_cost(CostArgs a) {
  constexpr int D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  double energy = 0.0;
  for (int i = 0; i < a.N; ++i) {
    const double t = a.T[(int64_t)i * ld + b];
    const double t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3;
    double Q[S][S];
    if constexpr (S == 4) {
      const double t6 = t3 * t3, t7 = t4 * t3;
      Q[0][0] = 100800 * t7; Q[0][1] = 50400 * t6; Q[0][2] = 20160 * t5; Q[0][3] = 5040 * t4;
      Q[1][1] = 25920 * t5;  Q[1][2] = 10800 * t4; Q[1][3] = 2880 * t3;
      Q[2][2] = 4800 * t3;   Q[2][3] = a.m34 * t2;
      Q[3][3] = 576 * t;
    } else if constexpr (S == 3) {
      Q[0][0] = 720 * t5; Q[0][1] = 360 * t4; Q[0][2] = 120 * t3;
      Q[1][1] = 192 * t3; Q[1][2] = 72 * t2;
      Q[2][2] = 36 * t;
    } else {
      Q[0][0] = 12 * t3; Q[0][1] = 6 * t2;
      Q[1][1] = 4 * t;
    }
#pragma unroll
    for (int j = 1; j < S; ++j)
#pragma unroll
      for (int k = 0; k < j; ++k) Q[j][k] = Q[k][j];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      double z[S];
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] = a.coeffs[(int64_t)((i * 3 + ax) * D + j) * ld + b];
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) r += Q[j][k] * z[k];
        acc += z[j] * r;
      }
      energy += 0.5 * acc;
    }
  }
  a.cost[b] = energy;
}

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

// ------------------------------------------------------------------------------------------
// context + error plumbing
// ------------------------------------------------------------------------------------------
struct anet_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  // grow-only device scratch for the host (trajectory-major) entry points
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
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

template <int S>
int launch_solve(anet_ctx *ctx, const anet::SolveArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.B + anet::kSolveBlock - 1) / anet::kSolveBlock));
  const dim3 block(anet::kSolveBlock);
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
  }
  if constexpr (S == 3) {
    if (a.N == 16 && a.c == 3) {
      hipLaunchKernelGGL((anet::k_minco_solve<3, 16, true, 2>), grid, block, 0, st, a);
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

}  // namespace

extern "C" {

int anet_abi_version(void) { return ANET_ABI_VERSION; }

int anet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

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

int anet_to_batch_minor_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                            const double *src, double *dst, void *stream) {
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

int anet_minco_solve(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, const double *wps, const double *T, double *coeffs,
                     double *energy) {
  int rc = check_solve_args(ctx, s, c, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!head || !tail || !T || (n_pieces > 1 && !wps))
    return fail(ctx, ANET_ERR_INVALID, "anet_minco_solve: NULL input");
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const int N = n_pieces, D = 2 * s;
  const int64_t ld = round_up(batch, 64);
  const int64_t n_in = 3 * c * 2 + (int64_t)(N - 1) * 3 + N;  // fields in per trajectory
  const int64_t n_co = (int64_t)N * 3 * D;
  const int64_t n_stage = (n_in > n_co ? n_in : n_co);
  // scratch: [stage: batch*n_stage][soa_in: n_in*ld][soa_co: n_co*ld][energy: ld]
  const size_t bytes = sizeof(double) * (size_t)(batch * n_stage + (n_in + n_co + 1) * ld);
  rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  double *stage = (double *)ctx->scratch;
  double *s_head = stage + batch * n_stage;
  double *s_tail = s_head + 3 * c * ld;
  double *s_wps = s_tail + 3 * c * ld;
  double *s_T = s_wps + (int64_t)(N - 1) * 3 * ld;
  double *s_co = s_T + (int64_t)N * ld;
  double *s_en = s_co + n_co * ld;
  hipStream_t st = ctx->stream;
  struct In { const double *h; double *d; int64_t nf; } ins[4] = {
      {head, s_head, 3 * c}, {tail, s_tail, 3 * c}, {wps, s_wps, (int64_t)(N - 1) * 3}, {T, s_T, N}};
  for (auto &in : ins) {
    if (in.nf == 0) continue;
    ANET_HIP(ctx, hipMemcpyAsync(stage, in.h, sizeof(double) * batch * in.nf, hipMemcpyHostToDevice, st));
    rc = anet_to_batch_minor_dev(ctx, batch, in.nf, ld, stage, in.d, st);
    if (rc) return rc;
  }
  rc = anet_minco_solve_dev(ctx, s, c, N, batch, ld, s_head, s_tail, s_wps, s_T, coeffs ? s_co : nullptr,
                            s_en, st);
  if (rc) return rc;
  if (coeffs) {
    rc = anet_to_traj_major_dev(ctx, batch, n_co, ld, s_co, stage, st);
    if (rc) return rc;
    ANET_HIP(ctx, hipMemcpyAsync(coeffs, stage, sizeof(double) * batch * n_co, hipMemcpyDeviceToHost, st));
  }
  if (energy)
    ANET_HIP(ctx, hipMemcpyAsync(energy, s_en, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
  ANET_HIP(ctx, hipStreamSynchronize(st));
  return ANET_OK;
}

int anet_traj_eval_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, int nq, const double *tq, int deriv,
                       double *out, void *stream) {
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
  int rc = check_solve_args(ctx, s, 1, n_pieces, batch);
  if (rc) return rc;
  if (batch == 0) return ANET_OK;
  if (!coeffs || !T || !cost || ld < batch) return fail(ctx, ANET_ERR_INVALID, "anet_traj_cost_dev: NULL pointer or ld < batch");
  anet::CostArgs a{coeffs, T, cost, batch, ld, n_pieces, m34};
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
    hipError_t e = hipMemcpyAsync(stage, host, sizeof(double) * batch * nf, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(H2D)");
    return anet_to_batch_minor_dev(ctx, batch, nf, ld, stage, *dev, ctx->stream);
  }
  double *reserve(int64_t nf) {
    double *p = cursor;
    cursor += nf * ld;
    return p;
  }
  int download(const double *dev, int64_t nf, double *host) {
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
  ANET_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t ld = round_up(batch, 64);
  int rc = ensure_scratch(ctx, sizeof(double) * (size_t)(batch * max_field + total_fields * ld));
  if (rc) return rc;
  st->ctx = ctx; st->batch = batch; st->ld = ld;
  st->stage = (double *)ctx->scratch;
  st->cursor = st->stage + batch * max_field;
  return ANET_OK;
}
}  // namespace

int anet_traj_eval(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, int nq, const double *tq, int deriv, double *out) {
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

}  // extern "C"
