// Batched L-BFGS state machine (lane-per-problem and wave-per-problem), the costMVIE objective and the
// MINCO variable maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "minco_kernels.h"
#include "wave_ops.h"

namespace anet {

// ------------------------------------------------------------------------------------------
// batched L-BFGS (lbfgs.hpp:276-384, 434-717) as a per-trajectory state machine
// ------------------------------------------------------------------------------------------
struct LbfgsP {
  int mem_size;
  double g_epsilon;
  int past;
  double delta;
  int max_iterations, max_linesearch;
  double min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
};
enum { DS_FX = 0, DS_STEP, DS_FINIT, DS_DGTEST, DS_DSTEST, DS_MU, DS_NU, DS_SMAX, DS_COUNT_ };  // DS_SMAX: stpmax of the running line search
enum { IS_DONE = 0, IS_RET, IS_K, IS_END, IS_BOUND, IS_COUNT, IS_BRACKT, IS_TOUCHED, IS_EVALS, IS_PHASE, IS_COUNT_ };
enum {  // lbfgs.hpp:135-184
  LB_CONVERGENCE = 0, LB_STOP = 1, LB_CANCELED = 2,
  LBERR_INVALID_FUNCVAL = -1012, LBERR_MINIMUMSTEP = -1011, LBERR_MAXIMUMSTEP = -1010,
  LBERR_MAXIMUMLINESEARCH = -1009, LBERR_MAXIMUMITERATION = -1008, LBERR_WIDTHTOOSMALL = -1007,
  LBERR_INVALIDPARAMETERS = -1006, LBERR_INCREASEGRADIENT = -1005
};

struct LbfgsArgs {
  int n;
  int64_t B, ld;
  double *x, *g, *xp, *gp, *d, *lm_s, *lm_y, *lm_ys, *lm_alpha, *pf, *ds;
  const double *feval;
  int *is;
  LbfgsP p;
  int *n_active;  // optional: set to 1 by every problem that is still running after this tick
  int64_t vs, ps;  // internal vectors (xp, gp, d, lm_s, lm_y): element i of problem b at [i*vs + b*ps]
  // MINCO objective: variables [map_nw, n) are tau of the durations; wherever x is written the mapped
  // duration T = forward_T(tau) is written too (saves a launch per evaluation).  nullptr: no mapping.
  double *map_T = nullptr;
  int map_nw = 0;
  // lbfgs_optimize's proc_stepbound (lbfgs.hpp:221-224, applied at :557-565) as the built-in bound of the MINCO objective:
  // variables [sb_lo, n) (tau of the durations) may not fall below sb_xmin within one line search; 0: no bound
  int sb_on = 0, sb_lo = 0;
  double sb_xmin = 0.0;
  // proc_progress's one effect (lbfgs.hpp:580-587): a device-visible word, non-zero cancels after the running iteration
  const int *cancel = nullptr;
  // HOST callbacks (anet_lbfgs_optimize_host; lane kernel only): the state machine PARKS where lbfgs_optimize would call
  // them and the host resumes it -- host_pg: at an accepted step (IS_PHASE = 2: the host calls proc_progress with x, g, fx,
  // step, k and ls = IS_COUNT, leaves its verdict in the cancel word, launches again); host_sb: at the entry of a line search
  // (IS_PHASE = 3: xp, gp, d in place; the host calls proc_stepbound(xp, d), leaves its value in DS_SMAX, launches again).
  // A resuming launch consumes no evaluation.
  int host_pg = 0, host_sb = 0;
};
enum { LB_PHASE_FIRST = 0, LB_PHASE_SEARCH = 1, LB_PHASE_AWAIT_PROGRESS = 2, LB_PHASE_AWAIT_STEPBOUND = 3 };
__device__ __forceinline__ int read_cancel_word(const int *w) {  // system scope: written while the kernels run
  return w ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
}
// x = xp + step * d as the reference computes it (lbfgs.hpp:308): the reference is built with -O3 and no -march
// (src/planner/CMakeLists.txt:4-6), i.e. for baseline x86-64, where this is a multiply and an add -- two roundings.  The
// trial point decides every later comparison of a line search, so the kernels round it the same way instead of fusing.
__device__ __forceinline__ double trial_point(double step, double d, double xp) {
#pragma clang fp contract(off)
  const double p = step * d;
  return xp + p;
}

__device__ __forceinline__ void store_x(const LbfgsArgs &a, int64_t b, int i, double v) {
  a.x[(int64_t)i * a.ld + b] = v;
  if (a.map_T && i >= a.map_nw) a.map_T[(int64_t)(i - a.map_nw) * a.ld + b] = forward_T(v);
}

// One lane per problem.  Every launch consumes ONE objective evaluation (f = feval[b], gradient in
// g, both taken at the point currently in x) and leaves in x the next point to evaluate.  The
// control flow per problem is lbfgs_optimize's: phase 0 = the initial evaluation, phase 1 = inside
// line_search_lewisoverton.  Finished problems are untouched (x, g hold the result).
__device__ __forceinline__ void lbfgs_update_lane(const LbfgsArgs &a, const int64_t b) {
  const int64_t ld = a.ld;
  int *is = a.is + b;
  if (is[IS_DONE * ld]) return;
  double *ds = a.ds + b;
  const int n = a.n, m = a.p.mem_size;
  const LbfgsP &P = a.p;
  double *x = a.x + b, *g = a.g + b, *xp = a.xp + b, *gp = a.gp + b, *d = a.d + b;
  const int phase_in = is[IS_PHASE * ld];
  const bool resume_pg = phase_in == LB_PHASE_AWAIT_PROGRESS, resume_sb = phase_in == LB_PHASE_AWAIT_STEPBOUND;
  double fx = ds[DS_FX * ld];
  const double f = (resume_pg || resume_sb) ? fx : a.feval[b];  // (a resumed step: the value it parked with)
  if (!(resume_pg || resume_sb)) is[IS_EVALS * ld] += 1;
  double step = ds[DS_STEP * ld];
  int k = is[IS_K * ld];
  bool start_ls = false;
  int finish = 0x7fffffff;  // sentinel: keep running

  auto conv_test = [&]() {
    double gn = 0.0, xn = 0.0;
    for (int i = 0; i < n; ++i) {
      gn = fmax(gn, fabs(g[i * ld]));
      xn = fmax(xn, fabs(x[i * ld]));
    }
    return gn / fmax(1.0, xn) < P.g_epsilon;
  };

  if (phase_in == LB_PHASE_FIRST) {
    fx = f;
    a.pf[b] = fx;
    double dd = 0.0;
    for (int i = 0; i < n; ++i) {
      const double gi = g[i * ld];
      d[i * ld] = -gi;
      dd = __builtin_fma(gi, gi, dd);
    }
    if (conv_test()) {
      finish = LB_CONVERGENCE;
    } else {
      step = 1.0 / sqrt(dd);
      k = 1;
      is[IS_END * ld] = 0;
      is[IS_BOUND * ld] = 0;
      is[IS_PHASE * ld] = LB_PHASE_SEARCH;
      start_ls = true;
    }
  } else if (resume_sb) {
    start_ls = true;  // the direction, xp, gp, step and k are in place; the host's bound is in DS_SMAX
  } else {
    // ---- one trial of line_search_lewisoverton (lbfgs.hpp:307-383)
    const double finit = ds[DS_FINIT * ld], dgtest = ds[DS_DGTEST * ld], dstest = ds[DS_DSTEST * ld];
    double mu = ds[DS_MU * ld], nu = ds[DS_NU * ld];
    const double smax = ds[DS_SMAX * ld];
    int count = is[IS_COUNT * ld] + 1, brackt = is[IS_BRACKT * ld], touched = is[IS_TOUCHED * ld];
    bool success = resume_pg;  // (parked behind a successful trial: straight to the accepted step)
    int err = 0;
    if (resume_pg) {
      count -= 1;
    } else if (isinf(f) || isnan(f)) {
      err = LBERR_INVALID_FUNCVAL;
    } else {
      if (f > finit + step * dgtest) {
        nu = step;
        brackt = 1;
      } else {
        double dg = 0.0;
        for (int i = 0; i < n; ++i) dg = __builtin_fma(g[i * ld], d[i * ld], dg);
        if (dg < dstest)
          mu = step;
        else
          success = true;
      }
      if (!success) {
        if (P.max_linesearch <= count) {
          err = LBERR_MAXIMUMLINESEARCH;
        } else if (brackt && (nu - mu) < P.machine_prec * nu) {
          err = LBERR_WIDTHTOOSMALL;
        } else {
          step = brackt ? 0.5 * (mu + nu) : step * 2.0;
          if (step < P.min_step) {
            err = LBERR_MINIMUMSTEP;
          } else if (step > smax) {
            if (touched) {
              err = LBERR_MAXIMUMSTEP;
            } else {
              touched = 1;
              step = smax;
            }
          }
        }
      }
    }
    if (err) {
      // revert to the previous point; the reported f stays the last trial's (lbfgs.hpp:570-577,713)
      for (int i = 0; i < n; ++i) {
        store_x(a, b, i, xp[i * ld]);
        g[i * ld] = gp[i * ld];
      }
      fx = f;
      finish = err;
    } else if (!success) {
      for (int i = 0; i < n; ++i) store_x(a, b, i, trial_point(step, d[i * ld], xp[i * ld]));
      ds[DS_MU * ld] = mu;
      ds[DS_NU * ld] = nu;
      is[IS_COUNT * ld] = count;
      is[IS_BRACKT * ld] = brackt;
      is[IS_TOUCHED * ld] = touched;
    } else if (a.host_pg && !resume_pg) {
      // ---- accepted step, host progress callback: park (lbfgs.hpp:580-587 is the host's to run)
      fx = f;
      is[IS_COUNT * ld] = count;
      is[IS_PHASE * ld] = LB_PHASE_AWAIT_PROGRESS;
    } else {
      // ---- accepted step (lbfgs.hpp:579-709); the progress report (:580-587) comes first: non-zero cancels
      fx = f;
      if (resume_pg) is[IS_PHASE * ld] = LB_PHASE_SEARCH;
      if (read_cancel_word(a.cancel)) {
        finish = LB_CANCELED;
      } else if (conv_test()) {
        finish = LB_CONVERGENCE;
      } else {
        if (0 < P.past) {
          if (P.past <= k) {
            const double rate = fabs(a.pf[(int64_t)(k % P.past) * ld + b] - fx) / fmax(1.0, fabs(fx));
            if (rate < P.delta) finish = LB_STOP;
          }
          if (finish == 0x7fffffff) a.pf[(int64_t)(k % P.past) * ld + b] = fx;
        }
        if (finish == 0x7fffffff && P.max_iterations != 0 && P.max_iterations <= k) finish = LBERR_MAXIMUMITERATION;
        if (finish == 0x7fffffff) {
          ++k;
          int end = is[IS_END * ld], bound = is[IS_BOUND * ld];
          double *se = a.lm_s + (int64_t)end * n * ld + b, *ye = a.lm_y + (int64_t)end * n * ld + b;
          double ys = 0.0, yy = 0.0, ss = 0.0, gpgp = 0.0;
          for (int i = 0; i < n; ++i) {
            const double si = x[i * ld] - xp[i * ld], yi = g[i * ld] - gp[i * ld], gpi = gp[i * ld];
            se[i * ld] = si;
            ye[i * ld] = yi;
            ys = __builtin_fma(yi, si, ys);
            yy = __builtin_fma(yi, yi, yy);
            ss = __builtin_fma(si, si, ss);
            gpgp = __builtin_fma(gpi, gpi, gpgp);
            d[i * ld] = -g[i * ld];
          }
          a.lm_ys[(int64_t)end * ld + b] = ys;
          const double cau = ss * sqrt(gpgp) * P.cautious_factor;
          if (ys > cau) {
            ++bound;
            bound = m < bound ? m : bound;
            end = (end + 1) % m;
            int j = end;
            for (int it = 0; it < bound; ++it) {
              j = (j + m - 1) % m;
              const double *sj = a.lm_s + (int64_t)j * n * ld + b, *yj = a.lm_y + (int64_t)j * n * ld + b;
              double sd = 0.0;
              for (int i = 0; i < n; ++i) sd = __builtin_fma(sj[i * ld], d[i * ld], sd);
              const double al = sd / a.lm_ys[(int64_t)j * ld + b];
              a.lm_alpha[(int64_t)j * ld + b] = al;
              for (int i = 0; i < n; ++i) d[i * ld] = __builtin_fma(-al, yj[i * ld], d[i * ld]);
            }
            const double sc = ys / yy;
            for (int i = 0; i < n; ++i) d[i * ld] *= sc;
            for (int it = 0; it < bound; ++it) {
              const double *sj = a.lm_s + (int64_t)j * n * ld + b, *yj = a.lm_y + (int64_t)j * n * ld + b;
              double yd = 0.0;
              for (int i = 0; i < n; ++i) yd = __builtin_fma(yj[i * ld], d[i * ld], yd);
              const double beta = yd / a.lm_ys[(int64_t)j * ld + b];
              const double cf = a.lm_alpha[(int64_t)j * ld + b] - beta;
              for (int i = 0; i < n; ++i) d[i * ld] = __builtin_fma(cf, sj[i * ld], d[i * ld]);
              j = (j + 1) % m;
            }
          }
          is[IS_END * ld] = end;
          is[IS_BOUND * ld] = bound;
          step = 1.0;
          start_ls = true;
        }
      }
    }
  }
  if (start_ls) {
    // ---- entry of line_search_lewisoverton (lbfgs.hpp:287-305) for the new direction
    double dginit = 0.0;
    for (int i = 0; i < n; ++i) {
      const double xi = x[i * ld], gi = g[i * ld];
      xp[i * ld] = xi;
      gp[i * ld] = gi;
      dginit = __builtin_fma(gi, d[i * ld], dginit);
    }
    double smax = P.max_step;
    bool parked = false;
    if (a.host_sb && !resume_sb) {  // the host's proc_stepbound(xp, d) comes first: park with xp, gp, d in place
      is[IS_PHASE * ld] = LB_PHASE_AWAIT_STEPBOUND;
      parked = true;
    } else if (a.host_sb) {        // lbfgs.hpp:557-565 with the host's value
      const double bnd = ds[DS_SMAX * ld];
      smax = bnd < P.max_step ? bnd : P.max_step;
      step = step < smax ? step : 0.5 * smax;
      is[IS_PHASE * ld] = LB_PHASE_SEARCH;
    } else if (a.sb_on) {  // lbfgs.hpp:557-565: step_max = min(proc_stepbound(xp, d), max_step); step = step < step_max ? step : step_max / 2
      double worst = 0.0;
      for (int i = a.sb_lo; i < n; ++i) {
        const double di = d[i * ld], room = x[i * ld] - a.sb_xmin;
        if (di < 0.0) worst = fmax(worst, -di / (room > 1e-300 ? room : 1e-300));
      }
      const double bnd = worst > 0.0 ? 1.0 / worst : INFINITY;
      smax = bnd < P.max_step ? bnd : P.max_step;
      step = step < smax ? step : 0.5 * smax;
    }
    if (!parked) ds[DS_SMAX * ld] = smax;
    if (parked) {
      // (nothing more until the host has answered)
    } else if (!(step > 0.0)) {
      finish = LBERR_INVALIDPARAMETERS;
    } else if (0.0 < dginit) {
      finish = LBERR_INCREASEGRADIENT;
    } else {
      ds[DS_FINIT * ld] = fx;
      ds[DS_DGTEST * ld] = P.f_dec_coeff * dginit;
      ds[DS_DSTEST * ld] = P.s_curv_coeff * dginit;
      ds[DS_MU * ld] = 0.0;
      ds[DS_NU * ld] = smax;
      is[IS_COUNT * ld] = 0;
      is[IS_BRACKT * ld] = 0;
      is[IS_TOUCHED * ld] = 0;
      for (int i = 0; i < n; ++i) store_x(a, b, i, trial_point(step, d[i * ld], xp[i * ld]));
    }
  }
  ds[DS_FX * ld] = fx;
  ds[DS_STEP * ld] = step;
  is[IS_K * ld] = k;
  if (finish != 0x7fffffff) {
    is[IS_DONE * ld] = 1;
    is[IS_RET * ld] = finish;
  } else if (a.n_active) {
    *a.n_active = 1;  // a flag, not a count: 10^5 atomics on one address cost more than the rest of the tick
  }
}


__global__ void __launch_bounds__(64) k_lbfgs_update(LbfgsArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b < a.B) lbfgs_update_lane(a, b);
}

// a value every lane holds identically (loaded from a wave-uniform address) -> SGPR pair
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// Two problems per wave (32 lanes each): the same scans, chained over the two rows of a half only; each half then
// reads its total from its own last lane.  Needs the lanes of a HALF to be active together, not the whole wave.
__device__ __forceinline__ double half_pick(double v) {
  const double lo = last_lane<31>(v), hi = last_lane<63>(v);
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ double half_sum(double v) {
  v += dpp_f64<0x111>(v);
  v += dpp_f64<0x112>(v);
  v += dpp_f64<0x114>(v);
  v += dpp_f64<0x118>(v);
  v += dpp_f64<0x142, 0xa>(v);
  return half_pick(v);
}
__device__ __forceinline__ double half_max_nonneg(double v) {
  v = fmax(v, dpp_f64<0x111>(v));
  v = fmax(v, dpp_f64<0x112>(v));
  v = fmax(v, dpp_f64<0x114>(v));
  v = fmax(v, dpp_f64<0x118>(v));
  v = fmax(v, dpp_f64<0x142, 0xa>(v));
  return half_pick(v);
}

// Same state machine, ONE WAVE per problem: the n <= 128 variables are spread over the 64 lanes (two
// per lane, held in registers for the whole tick), every dot product / norm is a wave reduction,
// scalars are wave-uniform.  Used for small batches, where one lane per problem leaves the chip idle
// and serialises ~16 n-long dependent loops per accepted step.
// Memory round trips per tick: (1) state + the lane's x, g, d, xp, gp; (2) history + past-f ring
// (speculatively, the accepted-step branch is the common one); then only stores.
// LBFGS_WAVE_MREG: history slots kept in registers -- instantiated for 8 (the default mem_size;
// 4 waves/SIMD), 20 (FIRI's 18; 2 waves/SIMD) and 0 (any mem_size, history re-read per slot).
// A workgroup is 4 to 16 waves (by register budget) = ADJACENT problems: x, g and the state
// are batch-minor, so one 128-byte line holds the same variable of 16 neighbouring problems -- on one
// CU they share it in L1; as separate workgroups they were dealt round-robin to the 8 XCDs and every
// line was fetched 16 times.
template <int LBFGS_WAVE_MREG>
struct LbfgsWaveShape {
  static constexpr int kWaves = (LBFGS_WAVE_MREG > 8) ? 4 : (LBFGS_WAVE_MREG > 0) ? 8 : 16;  // register budget
};
// The register copy of the (s, y) history: slot `it` = the it-th pair behind the one an accepted step is about to
// add (slot 0 = that new pair).  The per-launch kernel fills it from memory every tick; the one-launch kernel
// (CARRY) keeps it across its iterations and only shifts it by one slot when a pair is stored.
template <int MR, int NV>
struct WaveHistory {
  double s[MR][NV], y[MR][NV], ys[MR];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int it = 0; it < MR; ++it) {
#pragma unroll
      for (int q = 0; q < NV; ++q) s[it][q] = y[it][q] = 0.0;
      ys[it] = 1.0;
    }
  }
};
// NV: variables per lane (1 for n <= 64, 2 for n <= 128)
// RL: highest lane that can hold a variable (63 in general; 15 when the caller knows n <= 16)
// HALF: the wave carries TWO problems of at most 32 variables, `lane` is the lane within the half (0..31) and `b`
// differs between the halves: nothing is wave-uniform any more (state in vector registers, branches by exec mask),
// every reduction runs in both halves at once.  Half the instruction issue per problem -- what bounds the kernel.
template <int LBFGS_WAVE_MREG, int NV = 2, bool CARRY = false, int RL = 63, bool HALF = false>
__device__ __forceinline__ void lbfgs_update_wave_body(const LbfgsArgs &a, const int64_t b, const int lane,
                                                       WaveHistory<(LBFGS_WAVE_MREG > 0 ? LBFGS_WAVE_MREG : 1), NV> &H) {
  constexpr int MR = LBFGS_WAVE_MREG > 0 ? LBFGS_WAVE_MREG : 1;
  double (&hs)[MR][NV] = H.s, (&hy)[MR][NV] = H.y, (&hys)[MR] = H.ys;
  const int64_t ld = a.ld;
  int *is = a.is + b;
  double *ds = a.ds + b;
  const int n = a.n, m = a.p.mem_size;
  const LbfgsP &P = a.p;
  const int64_t vs = a.vs, ps = a.ps;
  double *x = a.x + b, *g = a.g + b;                       // batch-minor (shared with the objective)
  double *xp = a.xp + b * ps, *gp = a.gp + b * ps, *d = a.d + b * ps;
  bool h[NV];
  int64_t iv[NV];  // this lane's variable indices
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    iv[q] = lane + 64 * q;
    h[q] = iv[q] < n;
  }

  // ---- round trip 1
  const int done = is[IS_DONE * ld];
  const double f = a.feval[b];
  double fx = ds[DS_FX * ld];
  double step = ds[DS_STEP * ld];
  int k = is[IS_K * ld];
  int evals = is[IS_EVALS * ld] + 1;
  int end = is[IS_END * ld], bound = is[IS_BOUND * ld], phase = is[IS_PHASE * ld];
  int count = is[IS_COUNT * ld], brackt = is[IS_BRACKT * ld], touched = is[IS_TOUCHED * ld];
  double finit = ds[DS_FINIT * ld], dgtest = ds[DS_DGTEST * ld], dstest = ds[DS_DSTEST * ld];
  double mu = ds[DS_MU * ld], nu = ds[DS_NU * ld];
  double smax = ds[DS_SMAX * ld];
  const int cancel = read_cancel_word(a.cancel);
  double xr[NV], gr[NV], dr[NV], xpr[NV], gpr[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    xr[q] = h[q] ? x[iv[q] * ld] : 0.0;
    gr[q] = h[q] ? g[iv[q] * ld] : 0.0;
    dr[q] = h[q] ? d[iv[q] * vs] : 0.0;
    xpr[q] = h[q] ? xp[iv[q] * vs] : 0.0;
    gpr[q] = h[q] ? gp[iv[q] * vs] : 0.0;
  }
  if (done) return;
  if constexpr (!HALF) {
    // the scalars are the same in every lane: make that visible (scalar branches, SGPR operands)
    k = __builtin_amdgcn_readfirstlane(k);
    end = __builtin_amdgcn_readfirstlane(end);
    bound = __builtin_amdgcn_readfirstlane(bound);
    phase = __builtin_amdgcn_readfirstlane(phase);
  }

  // ---- round trip 2 (speculative): history slots other than the one this tick writes, past-f entry
  double *lms = a.lm_s + b * ps * m, *lmy = a.lm_y + b * ps * m;  // [j][i] at (j*js + i*vs)
  const int64_t js = (vs == 1) ? ps : (int64_t)n * vs;             // stride between history slots
  const bool hist_in_regs = LBFGS_WAVE_MREG > 0 && m <= LBFGS_WAVE_MREG;
  double pf_old = 0.0;
  if (phase != 0) {
    if (0 < P.past && P.past <= k) pf_old = a.pf[(int64_t)(k % P.past) * ld + b];
    if (hist_in_regs && !CARRY) {
      const int nb = (bound + 1 < m) ? bound + 1 : m;  // bound after an accepted step
      int jj = end;                                     // walks backwards through the ring (no division per slot)
#pragma unroll
      for (int it = 1; it < MR; ++it) {
#pragma unroll
        for (int q = 0; q < NV; ++q) hs[it][q] = hy[it][q] = 0.0;
        hys[it] = 1.0;
        jj = (jj == 0 ? m : jj) - 1;                    // it-th slot behind the new one
        if (it < nb) {
          const double *sj = lms + (int64_t)jj * js, *yj = lmy + (int64_t)jj * js;
#pragma unroll
          for (int q = 0; q < NV; ++q)
            if (h[q]) {
              hs[it][q] = sj[iv[q] * vs];
              hy[it][q] = yj[iv[q] * vs];
            }
          hys[it] = a.lm_ys[(int64_t)jj * ld + b];
        }
      }
      // (made wave-uniform only after every slot's loads are out: a readfirstlane right behind its load would put
      //  one full wait per slot into the loop above)
      if constexpr (!HALF) {
#pragma unroll
        for (int it = 1; it < MR; ++it) hys[it] = uniform_f64(hys[it]);
      }
    }
  }

  bool start_ls = false;
  int finish = 0x7fffffff;
  auto dot = [&](const double (&u)[NV], const double (&v)[NV]) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NV; ++q) acc = __builtin_fma(u[q], v[q], acc);
    if constexpr (HALF) return half_sum(acc);
    else return wave_sum<RL>(acc);
  };
  auto conv_test = [&]() {
    double gm = 0.0, xm = 0.0;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      gm = fmax(gm, fabs(gr[q]));
      xm = fmax(xm, fabs(xr[q]));
    }
    if constexpr (HALF) return half_max_nonneg(gm) / fmax(1.0, half_max_nonneg(xm)) < P.g_epsilon;
    else return wave_max_nonneg<RL>(gm) / fmax(1.0, wave_max_nonneg<RL>(xm)) < P.g_epsilon;
  };

  if (phase == 0) {
    fx = f;
    if (lane == 0) a.pf[b] = fx;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      dr[q] = -gr[q];
      if (h[q]) d[iv[q] * vs] = dr[q];
    }
    const double dd = dot(gr, gr);
    if (conv_test()) {
      finish = LB_CONVERGENCE;
    } else {
      step = 1.0 / sqrt(dd);
      k = 1;
      end = 0;
      bound = 0;
      phase = 1;
      start_ls = true;
    }
  } else {
    ++count;
    bool success = false;
    int err = 0;
    if (isinf(f) || isnan(f)) {
      err = LBERR_INVALID_FUNCVAL;
    } else {
      if (f > finit + step * dgtest) {
        nu = step;
        brackt = 1;
      } else {
        const double dg = dot(gr, dr);
        if (dg < dstest)
          mu = step;
        else
          success = true;
      }
      if (!success) {
        if (P.max_linesearch <= count) {
          err = LBERR_MAXIMUMLINESEARCH;
        } else if (brackt && (nu - mu) < P.machine_prec * nu) {
          err = LBERR_WIDTHTOOSMALL;
        } else {
          step = brackt ? 0.5 * (mu + nu) : step * 2.0;
          if (step < P.min_step) {
            err = LBERR_MINIMUMSTEP;
          } else if (step > smax) {
            if (touched) {
              err = LBERR_MAXIMUMSTEP;
            } else {
              touched = 1;
              step = smax;
            }
          }
        }
      }
    }
    if (err) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
        if (h[q]) {
          store_x(a, b, (int)iv[q], xpr[q]);
          g[iv[q] * ld] = gpr[q];
        }
      fx = f;
      finish = err;
    } else if (!success) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
        if (h[q]) store_x(a, b, (int)iv[q], trial_point(step, dr[q], xpr[q]));
    } else {
      fx = f;
      if (cancel) {  // lbfgs.hpp:580-587: the progress report comes first after a line search; non-zero cancels
        finish = LB_CANCELED;
      } else if (conv_test()) {
        finish = LB_CONVERGENCE;
      } else {
        if (0 < P.past) {
          if (P.past <= k) {
            const double rate = fabs(pf_old - fx) / fmax(1.0, fabs(fx));
            if (rate < P.delta) finish = LB_STOP;
          }
          if (finish == 0x7fffffff && lane == 0) a.pf[(int64_t)(k % P.past) * ld + b] = fx;
        }
        if (finish == 0x7fffffff && P.max_iterations != 0 && P.max_iterations <= k) finish = LBERR_MAXIMUMITERATION;
        if (finish == 0x7fffffff) {
          ++k;
          double *se = lms + (int64_t)end * js, *ye = lmy + (int64_t)end * js;
          double sreg[NV], yreg[NV], dv[NV];
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            sreg[q] = xr[q] - xpr[q];
            yreg[q] = gr[q] - gpr[q];
            dv[q] = -gr[q];
            if (h[q]) {
              se[iv[q] * vs] = sreg[q];
              ye[iv[q] * vs] = yreg[q];
            }
          }
          const double ys = dot(yreg, sreg), yy = dot(yreg, yreg), ss = dot(sreg, sreg), gpgp = dot(gpr, gpr);
          if (lane == 0) a.lm_ys[(int64_t)end * ld + b] = ys;
          const double cau = ss * sqrt(gpgp) * P.cautious_factor;
          if (ys > cau) {
            ++bound;
            bound = m < bound ? m : bound;
            const int newest = end;
            end = (end + 1) % m;
            if (hist_in_regs) {
#pragma unroll
              for (int q = 0; q < NV; ++q) {
                hs[0][q] = sreg[q];
                hy[0][q] = yreg[q];
              }
              hys[0] = ys;
              double alpha[MR];
#pragma unroll
              for (int it = 0; it < MR; ++it) {
                alpha[it] = 0.0;
                if (it < bound) {
                  alpha[it] = dot(hs[it], dv) / hys[it];
#pragma unroll
                  for (int q = 0; q < NV; ++q) dv[q] = __builtin_fma(-alpha[it], hy[it][q], dv[q]);
                }
              }
              const double sc = ys / yy;
#pragma unroll
              for (int q = 0; q < NV; ++q) dv[q] *= sc;
#pragma unroll
              for (int it = MR - 1; it >= 0; --it) {
                if (it < bound) {
                  const double cf = alpha[it] - dot(hy[it], dv) / hys[it];
#pragma unroll
                  for (int q = 0; q < NV; ++q) dv[q] = __builtin_fma(cf, hs[it][q], dv[q]);
                }
              }
              {  // the pair was stored: it is one slot behind the next new pair (dead code unless H is carried)
#pragma unroll
                for (int it = MR - 1; it > 0; --it) {
#pragma unroll
                  for (int q = 0; q < NV; ++q) {
                    hs[it][q] = hs[it - 1][q];
                    hy[it][q] = hy[it - 1][q];
                  }
                  hys[it] = hys[it - 1];
                }
              }
            } else {
              // (history re-read from memory: wave-wide lane indexing below, so never with two problems per wave --
              //  the host only dispatches HALF for mem_size <= LBFGS_WAVE_MREG)
              if constexpr (HALF) __builtin_trap();
              int j = end;
              double alpha = 0.0;  // lane `it` keeps alpha of the it-th visited slot (mem_size <= 64, host-checked)
              for (int it = 0; it < bound; ++it) {
                j = (j + m - 1) % m;
                const double *sj = lms + (int64_t)j * js, *yj = lmy + (int64_t)j * js;
                double sv[NV], yv[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) sv[q] = yv[q] = 0.0;
#pragma unroll
                for (int q = 0; q < NV; ++q)
                  if (h[q]) {
                    sv[q] = sj[iv[q] * vs];
                    yv[q] = yj[iv[q] * vs];
                  }
                const double ysj = (j == newest) ? ys : a.lm_ys[(int64_t)j * ld + b];
                const double al = dot(sv, dv) / ysj;
                alpha = (lane == it) ? al : alpha;
#pragma unroll
                for (int q = 0; q < NV; ++q) dv[q] = __builtin_fma(-al, yv[q], dv[q]);
              }
              const double sc = ys / yy;
#pragma unroll
              for (int q = 0; q < NV; ++q) dv[q] *= sc;
              for (int it = 0; it < bound; ++it) {
                const double *sj = lms + (int64_t)j * js, *yj = lmy + (int64_t)j * js;
                double sv[NV], yv[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) sv[q] = yv[q] = 0.0;
#pragma unroll
                for (int q = 0; q < NV; ++q)
                  if (h[q]) {
                    sv[q] = sj[iv[q] * vs];
                    yv[q] = yj[iv[q] * vs];
                  }
                const double ysj = (j == newest) ? ys : a.lm_ys[(int64_t)j * ld + b];
                const double cf = __shfl(alpha, bound - 1 - it) - dot(yv, dv) / ysj;
#pragma unroll
                for (int q = 0; q < NV; ++q) dv[q] = __builtin_fma(cf, sv[q], dv[q]);
                j = (j + 1) % m;
              }
            }
          }
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            dr[q] = dv[q];
            if (h[q]) d[iv[q] * vs] = dv[q];
          }
          step = 1.0;
          start_ls = true;
        }
      }
    }
  }
  if (start_ls) {
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (h[q]) {
        xp[iv[q] * vs] = xr[q];
        gp[iv[q] * vs] = gr[q];
      }
    smax = P.max_step;
    if (a.sb_on) {  // lbfgs.hpp:557-565 (see lbfgs_update_lane)
      double q = 0.0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const double room = xr[v] - a.sb_xmin;
        if (h[v] && iv[v] >= a.sb_lo && dr[v] < 0.0) q = fmax(q, -dr[v] / (room > 1e-300 ? room : 1e-300));
      }
      double worst;
      if constexpr (HALF) worst = half_max_nonneg(q);
      else worst = wave_max_nonneg<RL>(q);
      const double bnd = worst > 0.0 ? 1.0 / worst : INFINITY;
      smax = bnd < P.max_step ? bnd : P.max_step;
      step = step < smax ? step : 0.5 * smax;
    }
    const double dginit = dot(gr, dr);
    if (!(step > 0.0)) {
      finish = LBERR_INVALIDPARAMETERS;
    } else if (0.0 < dginit) {
      finish = LBERR_INCREASEGRADIENT;
    } else {
      finit = fx;
      dgtest = P.f_dec_coeff * dginit;
      dstest = P.s_curv_coeff * dginit;
      mu = 0.0;
      nu = smax;
      count = 0;
      brackt = 0;
      touched = 0;
#pragma unroll
      for (int q = 0; q < NV; ++q)
        if (h[q]) store_x(a, b, (int)iv[q], trial_point(step, dr[q], xr[q]));
    }
  }
  if (lane == 0) {
    ds[DS_FX * ld] = fx; ds[DS_STEP * ld] = step; ds[DS_FINIT * ld] = finit; ds[DS_DGTEST * ld] = dgtest;
    ds[DS_DSTEST * ld] = dstest; ds[DS_MU * ld] = mu; ds[DS_NU * ld] = nu; ds[DS_SMAX * ld] = smax;
    is[IS_K * ld] = k; is[IS_END * ld] = end; is[IS_BOUND * ld] = bound; is[IS_PHASE * ld] = phase;
    is[IS_COUNT * ld] = count; is[IS_BRACKT * ld] = brackt; is[IS_TOUCHED * ld] = touched;
    is[IS_EVALS * ld] = evals;
    if (finish != 0x7fffffff) {
      is[IS_DONE * ld] = 1;
      is[IS_RET * ld] = finish;
    } else if (a.n_active) {
      *a.n_active = 1;  // (flag: see k_lbfgs_update)
    }
  }
}

template <int LBFGS_WAVE_MREG, int NV, bool HALF = false>
__global__ void __launch_bounds__(64 * LbfgsWaveShape<LBFGS_WAVE_MREG>::kWaves, (LBFGS_WAVE_MREG == 8 && NV == 1 && !HALF) ? 4 : 1)
k_lbfgs_update_wave(LbfgsArgs a) {
  static_assert(!HALF || (NV == 1 && LBFGS_WAVE_MREG > 0), "two problems per wave: one variable per lane, history in registers");
  const int64_t w = (int64_t)blockIdx.x * LbfgsWaveShape<LBFGS_WAVE_MREG>::kWaves + (threadIdx.x >> 6);
  const int64_t b = HALF ? 2 * w + ((threadIdx.x >> 5) & 1) : w;
  if (b >= a.B) return;  // whole waves (halves) only: the reductions need all their lanes
  WaveHistory<(LBFGS_WAVE_MREG > 0 ? LBFGS_WAVE_MREG : 1), NV> H;
  lbfgs_update_wave_body<LBFGS_WAVE_MREG, NV, false, 63, HALF>(a, b, HALF ? (threadIdx.x & 31) : (threadIdx.x & 63), H);
}

// firi::costMVIE (gcopter/firi.hpp:86-157): x = [p, rtd, cde], A is M x 3 column-major per problem
// (field k*M + r), the reference's optData packing (firi.hpp:186-200).
struct MvieArgs {
  const double *A, *x;
  double *f, *g;
  const int *done;
  int64_t B, ld;
  int M;
  double eps, wt;
};
__device__ __forceinline__ void mvie_eval_lane(const MvieArgs &a, const int64_t b) {
  if (a.done && a.done[b]) return;
  const int64_t ld = a.ld;
  const double *x = a.x + b;
  double p[3], rtd[3], cde[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    p[q] = x[q * ld];
    rtd[q] = x[(3 + q) * ld];
    cde[q] = x[(6 + q) * ld];
  }
  const double L00 = rtd[0] * rtd[0] + 2.220446049250313e-16, L11 = rtd[1] * rtd[1] + 2.220446049250313e-16,
               L22 = rtd[2] * rtd[2] + 2.220446049250313e-16;
  const double L10 = cde[0], L21 = cde[1], L20 = cde[2];
  double cost = 0.0, gdp[3] = {0, 0, 0}, gdr[3] = {0, 0, 0}, gdc[3] = {0, 0, 0};
  const double inv_mu = 1.0 / a.eps;
  for (int r = 0; r < a.M; ++r) {
    const double a0 = a.A[(int64_t)r * ld + b], a1 = a.A[(int64_t)(a.M + r) * ld + b],
                 a2 = a.A[(int64_t)(2 * a.M + r) * ld + b];
    const double al0 = a0 * L00 + a1 * L10 + a2 * L20, al1 = a1 * L11 + a2 * L21, al2 = a2 * L22;
    const double nrm = sqrt(al0 * al0 + al1 * al1 + al2 * al2);
    const double viol = nrm + (a0 * p[0] + a1 * p[1] + a2 * p[2]) - 1.0;
    if (viol >= 0.0) {
      double c, dc;
      smoothed_l1(a.eps, inv_mu, viol, c, dc);
      const double inv = 1.0 / nrm;
      const double adj0 = al0 * inv, adj1 = al1 * inv, adj2 = al2 * inv;
      const double v0 = dc * a0, v1 = dc * a1, v2 = dc * a2;
      cost += c;
      gdp[0] += v0; gdp[1] += v1; gdp[2] += v2;
      gdr[0] += adj0 * v0; gdr[1] += adj1 * v1; gdr[2] += adj2 * v2;
      gdc[0] += adj0 * v1;
      gdc[1] += adj1 * v2;
      gdc[2] += adj0 * v2;
    }
  }
  cost *= a.wt;
  cost -= log(L00) + log(L11) + log(L22);
  const double Ld[3] = {L00, L11, L22};
  double *g = a.g + b;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    g[q * ld] = gdp[q] * a.wt;
    g[(3 + q) * ld] = (gdr[q] * a.wt - 1.0 / Ld[q]) * 2.0 * rtd[q];
    g[(6 + q) * ld] = gdc[q] * a.wt;
  }
  a.f[b] = cost;
}
__global__ void __launch_bounds__(64) k_mvie_eval(MvieArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b < a.B) mvie_eval_lane(a, b);
}

// costMVIE with one WAVE per problem: the rows of A are spread over the lanes, the ten sums (cost, nine gradient
// parts) are wave reductions.  Same quantities as mvie_eval_lane, summed in a different order.
__device__ __forceinline__ void mvie_eval_wave(const MvieArgs &a, const int64_t b, const int lane) {
  const int64_t ld = a.ld;
  const double *x = a.x + b;
  double p[3], rtd[3], cde[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    p[q] = x[q * ld];
    rtd[q] = x[(3 + q) * ld];
    cde[q] = x[(6 + q) * ld];
  }
  const double L00 = rtd[0] * rtd[0] + 2.220446049250313e-16, L11 = rtd[1] * rtd[1] + 2.220446049250313e-16,
               L22 = rtd[2] * rtd[2] + 2.220446049250313e-16;
  const double L10 = cde[0], L21 = cde[1], L20 = cde[2];
  double cost = 0.0, gdp[3] = {0, 0, 0}, gdr[3] = {0, 0, 0}, gdc[3] = {0, 0, 0};
  const double inv_mu = 1.0 / a.eps;
  for (int r = lane; r < a.M; r += 64) {
    const double a0 = a.A[(int64_t)r * ld + b], a1 = a.A[(int64_t)(a.M + r) * ld + b],
                 a2 = a.A[(int64_t)(2 * a.M + r) * ld + b];
    const double al0 = a0 * L00 + a1 * L10 + a2 * L20, al1 = a1 * L11 + a2 * L21, al2 = a2 * L22;
    const double nrm = sqrt(al0 * al0 + al1 * al1 + al2 * al2);
    const double viol = nrm + (a0 * p[0] + a1 * p[1] + a2 * p[2]) - 1.0;
    if (viol >= 0.0) {
      double c, dc;
      smoothed_l1(a.eps, inv_mu, viol, c, dc);
      const double inv = 1.0 / nrm;
      const double adj0 = al0 * inv, adj1 = al1 * inv, adj2 = al2 * inv;
      const double v0 = dc * a0, v1 = dc * a1, v2 = dc * a2;
      cost += c;
      gdp[0] += v0; gdp[1] += v1; gdp[2] += v2;
      gdr[0] += adj0 * v0; gdr[1] += adj1 * v1; gdr[2] += adj2 * v2;
      gdc[0] += adj0 * v1;
      gdc[1] += adj1 * v2;
      gdc[2] += adj0 * v2;
    }
  }
  cost = wave_sum(cost);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    gdp[q] = wave_sum(gdp[q]);
    gdr[q] = wave_sum(gdr[q]);
    gdc[q] = wave_sum(gdc[q]);
  }
  cost *= a.wt;
  cost -= log(L00) + log(L11) + log(L22);
  const double Ld[3] = {L00, L11, L22};
  if (lane == 0) {
    double *g = a.g + b;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      g[q * ld] = gdp[q] * a.wt;
      g[(3 + q) * ld] = (gdr[q] * a.wt - 1.0 / Ld[q]) * 2.0 * rtd[q];
      g[(6 + q) * ld] = gdc[q] * a.wt;
    }
    a.f[b] = cost;
  }
}

// A whole MVIE optimisation in ONE launch: one wave per problem loops evaluation + L-BFGS update (the same
// update body as k_lbfgs_update_wave, state in the same arrays).  The launch-per-evaluation driver spends a
// corridor search of a handful of segments almost entirely on launch latency (hundreds of evaluations of a
// 9-variable problem); here an evaluation costs a few memory round trips.  The fences order the cross-lane
// traffic through global memory inside the wave (workgroup scope: the lanes share one L1).
template <int LBFGS_WAVE_MREG>
__global__ void __launch_bounds__(64 * LbfgsWaveShape<LBFGS_WAVE_MREG>::kWaves)
k_lbfgs_mvie_persistent(LbfgsArgs la, MvieArgs ma, int max_evals) {
  const int64_t b = (int64_t)blockIdx.x * LbfgsWaveShape<LBFGS_WAVE_MREG>::kWaves + (threadIdx.x >> 6);
  if (b >= la.B) return;
  const int lane = threadIdx.x & 63;
  const int *done = la.is + (int64_t)IS_DONE * la.ld + b;
  // When mem_size fits, the history stays in registers across the iterations: the first one fills it from memory
  // as the per-launch kernel does (nothing to read in a fresh run), the later ones carry it.
  WaveHistory<(LBFGS_WAVE_MREG > 0 ? LBFGS_WAVE_MREG : 1), 1> H;
  H.clear();
  const bool carry = LBFGS_WAVE_MREG > 0 && la.p.mem_size <= LBFGS_WAVE_MREG;
  for (int e = 0; e < max_evals; ++e) {
    if (__builtin_amdgcn_readfirstlane(*(volatile const int *)done)) break;
    mvie_eval_wave(ma, b, lane);
    __threadfence_block();
    // nine variables: one per lane
    if (carry && e > 0) lbfgs_update_wave_body<LBFGS_WAVE_MREG, 1, true, 15>(la, b, lane, H);
    else lbfgs_update_wave_body<LBFGS_WAVE_MREG, 1, false, 15>(la, b, lane, H);
    __threadfence_block();
  }
}

struct MapArgs {
  double *x;                 // optimisation variables [n][ld]
  double *wps, *T;           // trajectory parameters
  int64_t B, ld;
  int nw, nt;                // optimised waypoint coordinates (0 or 3(N-1)), optimised durations (0 or N)
  int mode;                  // 0: params -> x (before the first evaluation), 1: x -> params (results)
};
// Only at the two ends of a run: during it the waypoint rows of x are read in place, the update kernel
// writes T = forward_T(tau) (store_x) and the propagate kernel applies dT/dtau (PropArgs::tau).
__global__ void __launch_bounds__(256) k_minco_map(MapArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  const int v = blockIdx.y;  // variable index: [0, nw) waypoint coordinates, [nw, nw+nt) durations
  if (v < a.nw) {
    const int64_t i = (int64_t)v * ld + b;
    if (a.mode == 0) a.x[i] = a.wps[i];
    else a.wps[i] = a.x[i];
  } else {
    const int64_t xi = (int64_t)v * ld + b, ti = (int64_t)(v - a.nw) * ld + b;
    if (a.mode == 0) a.x[xi] = backward_T(a.T[ti]);
    else a.T[ti] = forward_T(a.x[xi]);
  }
}

}  // namespace anet
