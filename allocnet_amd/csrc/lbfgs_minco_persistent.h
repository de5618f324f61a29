// L-BFGS over the MINCO cost as ONE launch: one wave per problem runs evaluation + update until ITS problem stops
// (lbfgs.hpp:551-709 is one loop per problem; the launch-per-evaluation driver makes the whole batch wait for its
// slowest member and pays four kernel boundaries per evaluation).
//
// The evaluation is the one of k_minco_solve / k_piece_grad / k_minco_propagate, re-mapped onto the 64 lanes of a
// wave with every intermediate in LDS (10-13 KB per wave, plus the corridor rows):
//   * of the block-tridiagonal solve only the factor's Schur-complement chain is walked node by node (one lane); the
//     forward / backward sweeps, primal and adjoint, are parallel scans over (node, axis) rows of 16 lanes (chain_solve);
//     with the durations fixed the factor is computed once per problem;
//   * everything per node or per piece (right-hand sides, coefficients, adjoint contributions, gradient terms) runs
//     on lanes = (node | piece, axis);
//   * the penalty functional runs on lanes = (piece, sample group): each lane takes every G-th sample of its piece
//     (G = 64 / NB), the basis rows are built from tau_j on the fly, the corridor rows come from LDS (staged once per
//     problem) and the partial gradients of a piece are summed across its G lanes with DPP swaps.
// The L-BFGS state (x, g, d, xp, gp and the (s, y) history) never leaves the registers: lane i owns variable i.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lbfgs_kernels.h"

namespace anet {

struct PersistArgs {
  const double *head, *tail, *wps, *T, *hpolys;  // batch-minor problem data (wps / T: initial values of unoptimised blocks)
  double *x;                                     // [n][ld] start point in, last point out
  int *is;                                       // IS_* rows (results)
  double *ds;                                    // DS_* rows (results)
  const int32_t *order;                          // workgroup w solves problem order[w] (nullptr: problem w)
  int64_t B, ld;
  int N, c, nw, nt, M, max_evals, with_penalty;
  Penalty pp;
  double inv_mu, inv_res;  // 1 / pp.mu, 1 / pp.res
  LbfgsP p;
  int step_bound;  // 1: every line search is bounded so that the durations stay >= the minimum behind tau_min
  double tau_min;  // backward_T(minimum duration)
  const int *cancel;  // optional device-visible word, polled once per evaluation: non-zero ends every problem with LBFGS_CANCELED
                      // after its next completed iteration (lbfgs.hpp:580-587: what proc_progress returning non-zero does)
  // A run in TWO launches (batches well beyond the 2048 resident waves: lbfgs_minco_dev_impl).  The first launch takes every
  // problem through max_evals (= the split point) evaluations and, park = 1, parks the optimiser of those still running in
  // `cont` ([B][kPersistContDoubles]: 22 per-lane values x 64 lanes, then 64 wave-uniform ones); the second, resume = 1, loads
  // it and goes on -- longest-expected first (`order` from the parked gradient norm and the decrease of the cost over the last
  // half of the first part).  half_mark: the evaluation count at which the cost is noted for that decrease.
  int park, resume, half_mark;
  double *cont;
#ifdef ANET_PERSIST_PROF
  long long *prof;  // [16] cycle counters of problem 0 (tools/persist_prof.py)
#endif
};
#ifdef ANET_PERSIST_PROF
#define PERSIST_TICK(slot)                                              \
  do {                                                                  \
    if (a.prof && blockIdx.x == 0) {                                    \
      const long long now_ = __builtin_readcyclecounter();              \
      if (threadIdx.x == 0) a.prof[slot] += now_ - prof_t_;             \
      prof_t_ = now_;                                                   \
    }                                                                   \
  } while (0)
#define PERSIST_TICK_DECL long long prof_t_ = __builtin_readcyclecounter()
#else
#define PERSIST_TICK(slot) do {} while (0)
#define PERSIST_TICK_DECL do {} while (0)
#endif

constexpr int kPersistContLaneFields = 22, kPersistContDoubles = (kPersistContLaneFields + 1) * 64;

template <int S, int NB>
struct PersistLds {
  static constexpr int m = S - 1, D = 2 * S, nl = m * (m - 1) / 2;
  double P[3][NB + 2];       // node positions per axis
  double T[NB], r[NB];       // durations, 1/T
  double hv[3][m], tv[3][m];  // pinned end derivatives
  double Ad[NB + 1][m][m];   // diagonal block of node k before elimination (lower triangle, end nodes pinned)
  double Kp[NB][m][m];       // coupling block Ko_k of piece k
  double Si[NB + 1][m][m];   // inverse Schur complements S_k^-1
  double H[NB][m][m];        // H_k = S_k^-1 Ko_k: y_k+1 = rhs_k+1 - H_k' y_k ;  x_k = z_k - H_k x_k+1
  // X[axis][component][node], rows of XW doubles: the (node, axis) lanes and the 16-lane scan rows of chain_solve read one
  // component of many nodes at a time -- node-minor keeps those on consecutive banks (node-major with 2 components per node put
  // 51 lanes on 16 same-parity double-banks: four-way); XW makes the axes start 16 double-banks apart where a small pad can
  static constexpr int xw_pad() {
    for (int p = 0; p <= 8; ++p)
      if ((m * (NB + 1 + p)) % 32 == 16) return p;
    return 0;
  }
  static constexpr int XW = NB + 1 + xw_pad();
  double X[3][m][XW];        // primal: right-hand side -> node derivatives; then adjoint: right-hand side -> multipliers
  // (rows of D + 1 doubles: lanes = (piece, axis) read and write these at a fixed column, and at 2 S doubles per row -- 48 or
  //  64 bytes -- consecutive lanes fall on 16 resp. 4 bank positions)
  double co[NB][3][D + 1];   // coefficients, highest power first; then the node-state adjoint contributions of a piece
                             // to its start node (first S entries) and its end node (last S), written by their reader
  double gc[NB][3][D + 1];   // penalty part of dJ/dc
  double gdT[NB], pc[NB];    // penalty part of dJ/dT, penalty cost per piece
  double wl[NB + 1][3], gTp[NB][3], ep[NB][3];
  double mid[3][2][m];       // twisted sweeps: what the last node of either chain contributes to the middle node's right-hand side
};
// The block-tridiagonal system is eliminated from BOTH ends towards the middle node (1) or from node 0 to node N (0: the
// one-ended walk, kept for A/B runs: tools/ab_build.sh "-DANET_PERSIST_TWISTED=0").
#ifndef ANET_PERSIST_TWISTED
#define ANET_PERSIST_TWISTED 1
#endif

template <int S, int NB>
constexpr size_t persist_lds_fixed_bytes() { return (sizeof(PersistLds<S, NB>) + 15) / 16 * 16; }
// corridor rows of piece i start at i * (4 M4 + 4) doubles, M4 = M rounded up to a multiple of four with zero rows
// (a zero row is never violated); the pad of four doubles spreads the pieces over the LDS banks
inline size_t persist_lds_row_doubles(int N, int M) { return (size_t)N * (4 * (size_t)((M + 3) & ~3) + 4); }

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) { return v + dpp_f64<CTRL>(v); }
// sum over the G adjacent lanes of a group (G = 4, 8, 16; every lane of the group ends with the total)
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  v = dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                          // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if constexpr (G >= 16) v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

template <int S>
struct BlkOps {
  static constexpr int m = S - 1, nl = m * (m - 1) / 2, NLA = nl > 0 ? nl : 1;
  __device__ __forceinline__ static int li(int i, int j) { return i * (i - 1) / 2 + j; }
  __device__ __forceinline__ static void solve_L(const double (&L)[NLA], double (&v)[m]) {
#pragma unroll
    for (int i = 1; i < m; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) v[i] = __builtin_fma(-L[li(i, j)], v[j], v[i]);
  }
  __device__ __forceinline__ static void solve_LT(const double (&L)[NLA], double (&v)[m]) {
#pragma unroll
    for (int i = m - 2; i >= 0; --i)
#pragma unroll
      for (int j = i + 1; j < m; ++j) v[i] = __builtin_fma(-L[li(j, i)], v[j], v[i]);
  }
};

// rhs_primal_node (minco_core.h) for a node index only known at run time: the neighbours arrive as scalars
template <int S>
__device__ __forceinline__ void rhs_primal_node_rt(int k, int N, int np, double rk, double rkm, double Pm, double P0,
                                                   double Pp, const double (&hv)[S - 1], const double (&tv)[S - 1],
                                                   double (&y)[S - 1]) {
  constexpr int m = S - 1;
#pragma unroll
  for (int l = 0; l < m; ++l) y[l] = 0.0;
  if (k < N) {
    Pw<S> p(rk);
    const double dl = Pp - P0;
#pragma unroll
    for (int l = 0; l < m; ++l) y[l] = Tab<S>::M[1 + l][0] * p[2 * S - 2 - l] * dl;
    if (k == N - 1) {
#pragma unroll
      for (int l = 0; l < m; ++l)
#pragma unroll
        for (int j = 0; j < m; ++j)
          if (j < np) y[l] = __builtin_fma(-Tab<S>::M[1 + l][S + 1 + j] * p[2 * S - 3 - l - j], tv[j], y[l]);
    }
  }
  if (k > 0) {
    Pw<S> p(rkm);
    const double dl = P0 - Pm;
#pragma unroll
    for (int l = 0; l < m; ++l) y[l] = __builtin_fma(Tab<S>::M[S + 1 + l][0] * p[2 * S - 2 - l], dl, y[l]);
    if (k == 1) {
#pragma unroll
      for (int l = 0; l < m; ++l)
#pragma unroll
        for (int j = 0; j < m; ++j)
          if (j < np) y[l] = __builtin_fma(-Tab<S>::M[1 + j][S + 1 + l] * p[2 * S - 3 - l - j], hv[j], y[l]);
    }
  }
  if (k == 0) {
    Pw<S> p(rk);
#pragma unroll
    for (int l = 0; l < m; ++l)
#pragma unroll
      for (int j = 0; j < m; ++j)
        if (j < np && l >= np) y[l] = __builtin_fma(-Tab<S>::M[1 + l][1 + j] * p[2 * S - 3 - l - j], hv[j], y[l]);
  }
  if (k == N && k > 0) {
    Pw<S> p(rkm);
#pragma unroll
    for (int l = 0; l < m; ++l)
#pragma unroll
      for (int j = 0; j < m; ++j)
        if (j < np && l >= np)
          y[l] = __builtin_fma(-Tab<S>::M[S + 1 + l][S + 1 + j] * p[2 * S - 3 - l - j], tv[j], y[l]);
  }
  if (k == 0 || k == N) {
#pragma unroll
    for (int l = 0; l < m; ++l)
      if (l < np) y[l] = (k == 0) ? hv[l] : tv[l];
  }
}

// The register-resident L-BFGS of one problem: lbfgs_update_wave_body (one variable per lane, history carried)
// without its loads and stores.  pf: lane j holds pf[j] of the past-f ring.
// LAST: the highest lane that can hold a non-zero component (63: any n <= 64; 15: n <= 16, reductions stop after one row).
// lbfgs_optimize's proc_stepbound (lbfgs.hpp:221-224, 557-565) as a built-in: the largest step along d that keeps the variables
// of lanes [lo, hi) at or above xmin -- for the MINCO objective the duration variables tau and xmin = backward_T(minimum
// duration), so that no line search ever leaves T >= T_min.  on = 0: no bound (step_max = max_step, as with a NULL callback).
struct StepBound {
  int on, lo, hi;
  double xmin;
};

template <int MR, int LAST = 63>
struct LbfgsResident {
  double x, g, d, xp, gp;
  double smax;  // stpmax of the current line search: min(step bound, max_step)
  double hs[MR], hy[MR], hys[MR];  // hys: 1 / (y.s) of the slot
  double fx, step, finit, dgtest, dstest, mu, nu, pf;
  int k, bound, count, brackt, touched, evals, phase;

  __device__ __forceinline__ void init(double x0) {
    x = x0;
    g = d = xp = gp = 0.0;
#pragma unroll
    for (int it = 0; it < MR; ++it) {
      hs[it] = hy[it] = 0.0;
      hys[it] = 1.0;
    }
    fx = step = finit = dgtest = dstest = mu = nu = pf = smax = 0.0;
    k = bound = count = brackt = touched = evals = phase = 0;
  }
  // the whole state to / from global memory ([field][lane], then 64 wave-uniform values): what a second launch needs to go on
  // exactly where this one stopped (PersistArgs::park / resume)
  __device__ __forceinline__ void park(double *c, const int lane, const double f_half) const {
    static_assert(MR <= 8, "layout of the parked state");
    double *pl = c + lane;
    pl[0 * 64] = x; pl[1 * 64] = g; pl[2 * 64] = d; pl[3 * 64] = xp; pl[4 * 64] = gp; pl[5 * 64] = pf;
#pragma unroll
    for (int it = 0; it < MR; ++it) {
      pl[(6 + it) * 64] = hs[it];
      pl[(14 + it) * 64] = hy[it];
    }
    const double gn2 = dot(gp, gp);
    if (lane == 0) {
      double *u = c + 22 * 64;
#pragma unroll
      for (int it = 0; it < MR; ++it) u[it] = hys[it];
      u[8] = fx; u[9] = step; u[10] = finit; u[11] = dgtest; u[12] = dstest; u[13] = mu; u[14] = nu; u[15] = smax;
      u[16] = (double)k; u[17] = (double)bound; u[18] = (double)count; u[19] = (double)brackt; u[20] = (double)touched;
      u[21] = (double)evals; u[22] = (double)phase;
      u[23] = f_half;  // (for the order of the second launch only)
      u[24] = gn2;
    }
  }
  __device__ __forceinline__ void unpark(const double *c, const int lane) {
    const double *pl = c + lane;
    x = pl[0 * 64]; g = pl[1 * 64]; d = pl[2 * 64]; xp = pl[3 * 64]; gp = pl[4 * 64]; pf = pl[5 * 64];
#pragma unroll
    for (int it = 0; it < MR; ++it) {
      hs[it] = pl[(6 + it) * 64];
      hy[it] = pl[(14 + it) * 64];
    }
    const double *u = c + 22 * 64;
#pragma unroll
    for (int it = 0; it < MR; ++it) hys[it] = u[it];
    fx = u[8]; step = u[9]; finit = u[10]; dgtest = u[11]; dstest = u[12]; mu = u[13]; nu = u[14]; smax = u[15];
    k = (int)u[16]; bound = (int)u[17]; count = (int)u[18]; brackt = (int)u[19]; touched = (int)u[20];
    evals = (int)u[21]; phase = (int)u[22];
  }
  __device__ __forceinline__ static double dot(double u, double v) { return wave_sum<LAST>(u * v); }
  // |g|_inf / max(1, |x|_inf) < g_epsilon (lbfgs.hpp:520-524, 592-596), the quotient cleared
  // (g_epsilon = 0, the setting of the reference's only call site: a norm is never below zero, the two reductions are skipped)
  __device__ __forceinline__ bool conv_test(const LbfgsP &P) const {
    if (!(P.g_epsilon > 0.0)) return false;
    return wave_max_nonneg<LAST>(fabs(g)) < P.g_epsilon * fmax(1.0, wave_max_nonneg<LAST>(fabs(x)));
  }
  // consumes f = objective at x (gradient already in g); leaves the next point in x.  Returns the lbfgs.hpp
  // return code when the problem stops, 0x7fffffff while it runs.
  __device__ __forceinline__ int update(const LbfgsP &P, const int lane, const double f, const StepBound sb = StepBound{0, 0, 0, 0.0},
                                        const int cancel = 0) {
    const int m = P.mem_size;
    ++evals;
    bool start_ls = false;
    int finish = 0x7fffffff;
    if (phase == 0) {
      fx = f;
      pf = (lane == 0) ? fx : pf;
      d = -g;
      const double dd = dot(g, g);
      if (conv_test(P)) {
        finish = LB_CONVERGENCE;
      } else {
        step = 1.0 / sqrt(dd);
        k = 1;
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
          const double dg = dot(g, d);
          if (dg < dstest) mu = step;
          else success = true;
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
      if (err) {  // revert; the reported f stays the last trial's (lbfgs.hpp:570-577,713)
        x = xp;
        g = gp;
        fx = f;
        finish = err;
      } else if (!success) {
        x = trial_point(step, d, xp);
      } else {
        fx = f;
        if (cancel) {  // lbfgs.hpp:580-587: the progress report comes first after a line search; non-zero cancels
          finish = LB_CANCELED;
        } else if (conv_test(P)) {
          finish = LB_CONVERGENCE;
        } else {
          if (0 < P.past) {
            const int slot = k % P.past;
            if (P.past <= k) {
              const double pf_old = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pf), slot),
                                                     __builtin_amdgcn_readlane(__double2loint(pf), slot));
              const double rate = fabs(pf_old - fx) / fmax(1.0, fabs(fx));
              if (rate < P.delta) finish = LB_STOP;
            }
            if (finish == 0x7fffffff) pf = (lane == slot) ? fx : pf;
          }
          if (finish == 0x7fffffff && P.max_iterations != 0 && P.max_iterations <= k) finish = LBERR_MAXIMUMITERATION;
          if (finish == 0x7fffffff) {
            ++k;
            const double sreg = x - xp, yreg = g - gp;
            double dv = -g;
            const double ys = dot(yreg, sreg), yy = dot(yreg, yreg), ss = dot(sreg, sreg), gpgp = dot(gp, gp);
            const double cau = ss * sqrt(gpgp) * P.cautious_factor;
            if (ys > cau) {
              ++bound;
              bound = m < bound ? m : bound;
              hs[0] = sreg;
              hy[0] = yreg;
              hys[0] = 1.0 / ys;  // one division per stored pair instead of two per slot and iteration
              double alpha[MR];
#pragma unroll
              for (int it = 0; it < MR; ++it) {
                alpha[it] = 0.0;
                if (it < bound) {
                  alpha[it] = dot(hs[it], dv) * hys[it];
                  dv = __builtin_fma(-alpha[it], hy[it], dv);
                }
              }
              dv *= ys / yy;
#pragma unroll
              for (int it = MR - 1; it >= 0; --it) {
                if (it < bound) {
                  const double cf = alpha[it] - dot(hy[it], dv) * hys[it];
                  dv = __builtin_fma(cf, hs[it], dv);
                }
              }
#pragma unroll
              for (int it = MR - 1; it > 0; --it) {  // the stored pair is one slot behind the next new pair
                hs[it] = hs[it - 1];
                hy[it] = hy[it - 1];
                hys[it] = hys[it - 1];
              }
            }
            d = dv;
            step = 1.0;
            start_ls = true;
          }
        }
      }
    }
    if (start_ls) {  // lbfgs.hpp:553-565, then the entry of line_search_lewisoverton (lbfgs.hpp:287-305)
      xp = x;
      gp = g;
      smax = P.max_step;
      if (sb.on) {  // step_max = proc_stepbound(xp, d); step_max = min(step_max, max_step); step = step < step_max ? step : step_max / 2
        const bool mine = lane >= sb.lo && lane < sb.hi && d < 0.0;
        const double room = x - sb.xmin;
        const double q = mine ? -d / (room > 1e-300 ? room : 1e-300) : 0.0;
        const double worst = wave_max_nonneg<LAST>(q);
        const double bnd = worst > 0.0 ? 1.0 / worst : INFINITY;
        smax = bnd < P.max_step ? bnd : P.max_step;
        step = step < smax ? step : 0.5 * smax;
      }
      const double dginit = dot(g, d);
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
        x = trial_point(step, d, x);
      }
    }
    return finish;
  }
};

// firi::maxVolInsEllipsoid's optimisation (firi.hpp:207-227) in one launch with NOTHING in memory between the iterations:
// one wave per polytope, the rows of A in registers (lane = row, RG groups of 64), the nine variables and the L-BFGS
// state in LbfgsResident (variable = lane), costMVIE (firi.hpp:86-157) as ten wave sums over the rows.  The variant with
// the state in memory (k_lbfgs_mvie_persistent: x, g, f and the optimiser state through L1 / L2 every iteration) spent
// about half of an iteration on those round trips.  Same arithmetic as mvie_eval_wave + lbfgs_update_wave_body.
template <int MR, int RG>
__global__ void __launch_bounds__(64) k_lbfgs_mvie_resident(LbfgsArgs la, MvieArgs ma, int max_evals) {
  const int64_t b = blockIdx.x, ld = la.ld;
  const int lane = threadIdx.x;
  if (la.is[(int64_t)IS_DONE * ld + b]) return;  // (corridors FIRI's set-up found empty)
  double a0[RG], a1[RG], a2[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int r = lane + 64 * g;
    const bool v = r < ma.M;
    a0[g] = v ? ma.A[(int64_t)r * ld + b] : 0.0;
    a1[g] = v ? ma.A[(int64_t)(ma.M + r) * ld + b] : 0.0;
    a2[g] = v ? ma.A[(int64_t)(2 * ma.M + r) * ld + b] : 0.0;
  }
  LbfgsResident<MR, 15> st;
  st.init(lane < 9 ? la.x[(int64_t)lane * ld + b] : 0.0);
  const double inv_mu = 1.0 / ma.eps;
  int finish = 0x7fffffff;
#pragma unroll 1
  for (int e = 0; e < max_evals; ++e) {
    double xv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
      xv[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(st.x), q), __builtin_amdgcn_readlane(__double2loint(st.x), q));
    const double L00 = xv[3] * xv[3] + 2.220446049250313e-16, L11 = xv[4] * xv[4] + 2.220446049250313e-16,
                 L22 = xv[5] * xv[5] + 2.220446049250313e-16;
    const double L10 = xv[6], L21 = xv[7], L20 = xv[8];
    double cost = 0.0, gdp[3] = {0, 0, 0}, gdr[3] = {0, 0, 0}, gdc[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const double al0 = a0[g] * L00 + a1[g] * L10 + a2[g] * L20, al1 = a1[g] * L11 + a2[g] * L21, al2 = a2[g] * L22;
      const double nrm = sqrt(al0 * al0 + al1 * al1 + al2 * al2);
      const double viol = nrm + (a0[g] * xv[0] + a1[g] * xv[1] + a2[g] * xv[2]) - 1.0;
      if (viol >= 0.0) {
        double c, dc;
        smoothed_l1(ma.eps, inv_mu, viol, c, dc);
        const double inv = fast_rcp(nrm);  // (v_rcp_f64 + two Newton steps: 5 instructions where the IEEE division takes ~30)
        const double adj0 = al0 * inv, adj1 = al1 * inv, adj2 = al2 * inv;
        const double v0 = dc * a0[g], v1 = dc * a1[g], v2 = dc * a2[g];
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
    cost *= ma.wt;
    cost -= log(L00 * L11 * L22);  // (one logarithm instead of three: ~70 wave instructions each)
    const double Ld[3] = {L00, L11, L22};
    double g = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      g = (lane == q) ? gdp[q] * ma.wt : g;
      g = (lane == 3 + q) ? (gdr[q] * ma.wt - fast_rcp(Ld[q])) * 2.0 * xv[3 + q] : g;
      g = (lane == 6 + q) ? gdc[q] * ma.wt : g;
    }
    st.g = g;
    finish = __builtin_amdgcn_readfirstlane(st.update(la.p, lane, cost));
    if (finish != 0x7fffffff) break;
  }
  if (lane < 9) la.x[(int64_t)lane * ld + b] = st.x;
  if (lane == 0) {
    la.is[(int64_t)IS_DONE * ld + b] = finish != 0x7fffffff;
    la.is[(int64_t)IS_RET * ld + b] = finish;
    la.is[(int64_t)IS_K * ld + b] = st.k;
    la.is[(int64_t)IS_EVALS * ld + b] = st.evals;
    la.ds[(int64_t)DS_FX * ld + b] = st.fx;
  }
}

// K v = rhs for the three axes at once, in place, given S_k^-1 and H_k (E2).
// The two sweeps of the block LDL^T solve, y_k = rhs_k - H_{k-1}' y_{k-1} and x_k = z_k - H_k x_{k+1} (z_k = S_k^-1 y_k), are
// affine recurrences with m x m matrices: instead of walking the nodes on three lanes they run as PARALLEL SCANS, one DPP row
// of 16 lanes per axis (row 3 shadows axis 2), lane j = node j+1 forwards and node j backwards (the ends are folded into
// their neighbours, so 17 nodes fit 16 lanes).  A round combines (M, v) -- "value = M * value(d nodes away) + v" -- with
// the pair d lanes away: log2(N) rounds of row_shr / row_shl moves and m^3 + m^2 FMAs replace N dependent node steps
// with their LDS round trips; rounding differs from the walk by a factor below two (tests/prototypes/scan_sweeps.py).
template <int CTRL, int m>
__device__ __forceinline__ void scan_round(double (&M)[m][m], double (&v)[m], const bool more) {
  double Mp[m][m], vp[m];
#pragma unroll
  for (int a = 0; a < m; ++a) {
    vp[a] = dpp_f64<CTRL>(v[a]);
#pragma unroll
    for (int b = 0; b < m; ++b) Mp[a][b] = dpp_f64<CTRL>(M[a][b]);
  }
#pragma unroll
  for (int a = 0; a < m; ++a)
#pragma unroll
    for (int b = 0; b < m; ++b) v[a] = __builtin_fma(M[a][b], vp[b], v[a]);
  if (more) {  // (wave-uniform: the last round only needs the values)
    double Mn[m][m];
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < m; ++q) acc = __builtin_fma(M[a][q], Mp[q][b], acc);
        Mn[a][b] = acc;
      }
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) M[a][b] = Mn[a][b];
  }
}

template <int S, int NB>
__device__ __forceinline__ void chain_solve(PersistLds<S, NB> &Lm, double (&V)[3][S - 1][PersistLds<S, NB>::XW], const int N,
                                            const int lane, const int, const int) {
  constexpr int m = S - 1;
  const int row = lane >> 4, j = lane & 15, ax = row < 3 ? row : 2;
  const bool valid = j < N;                  // forward node j + 1 <= N, backward node j <= N - 1
  const int jc = valid ? j : 0;
  double Hj[m][m], M[m][m], v[m], y0[m], z0[m], zf[m];
#pragma unroll
  for (int a = 0; a < m; ++a) {
    y0[a] = V[ax][a][0];
    v[a] = valid ? V[ax][a][jc + 1] : 0.0;
#pragma unroll
    for (int b = 0; b < m; ++b) Hj[a][b] = valid ? Lm.H[jc][a][b] : 0.0;
  }
  // ---- forwards: y_{j+1} = rhs_{j+1} - H_j' y_j; lane 0 takes y_0 = rhs_0 in and starts the chain
#pragma unroll
  for (int a = 0; a < m; ++a)
#pragma unroll
    for (int b = 0; b < m; ++b) M[a][b] = -Hj[b][a];
  if (j == 0) {
#pragma unroll
    for (int a = 0; a < m; ++a) {
#pragma unroll
      for (int b = 0; b < m; ++b) {
        v[a] = __builtin_fma(M[a][b], y0[b], v[a]);
      }
    }
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) M[a][b] = 0.0;
  }
  if (N > 1) scan_round<0x111, m>(M, v, N > 2);
  if (N > 2) scan_round<0x112, m>(M, v, N > 4);
  if (N > 4) scan_round<0x114, m>(M, v, N > 8);
  if constexpr (NB > 8) {
    if (N > 8) scan_round<0x118, m>(M, v, false);
  }
  // ---- z = S^-1 y on the lane of its node; z_j arrives from the left neighbour, z_0 is computed by lane 0
#pragma unroll
  for (int a = 0; a < m; ++a) {
    double acc = 0.0, acc0 = 0.0;
#pragma unroll
    for (int b = 0; b < m; ++b) {
      acc = __builtin_fma(Lm.Si[jc + 1][a][b], v[b], acc);
      acc0 = __builtin_fma(Lm.Si[0][a][b], y0[b], acc0);
    }
    zf[a] = valid ? acc : 0.0;
    z0[a] = acc0;
  }
#pragma unroll
  for (int a = 0; a < m; ++a) {
    const double zl = dpp_f64<0x111>(zf[a]);
    v[a] = (j == 0) ? z0[a] : zl;
    if (!valid) v[a] = 0.0;
  }
  // ---- backwards: x_j = z_j - H_j x_{j+1}; lane N-1 takes x_N = z_N in and starts the chain
#pragma unroll
  for (int a = 0; a < m; ++a)
#pragma unroll
    for (int b = 0; b < m; ++b) M[a][b] = -Hj[a][b];
  if (j == N - 1) {
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) v[a] = __builtin_fma(M[a][b], zf[b], v[a]);
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) M[a][b] = 0.0;
  }
  if (N > 1) scan_round<0x101, m>(M, v, N > 2);
  if (N > 2) scan_round<0x102, m>(M, v, N > 4);
  if (N > 4) scan_round<0x104, m>(M, v, N > 8);
  if constexpr (NB > 8) {
    if (N > 8) scan_round<0x108, m>(M, v, false);
  }
  if (valid && row < 3) {
#pragma unroll
    for (int a = 0; a < m; ++a) V[ax][a][j] = v[a];
    if (j == N - 1) {
#pragma unroll
      for (int a = 0; a < m; ++a) V[ax][a][N] = zf[a];
    }
  }
  __syncthreads();
}

// The same solve with the TWISTED factor of E2: nodes 0 .. pm-1 were eliminated upwards (S_k), nodes N .. pm+1 downwards (R_k),
// the middle node pm = N / 2 last.  Lm.Si[k] holds S_k^-1 (k < pm), R_k^-1 (k > pm) or the inverse of the middle block;
// Lm.H[i] (piece i, between nodes i and i+1) holds S_i^-1 Ko_i for i < pm and R_{i+1}^-1 Ko_i' for i >= pm -- in either chain
// "the inverse of the node FARTHER from the middle times the coupling towards the nearer one".  A row of 16 lanes per axis
// carries BOTH chains: lanes 0..7 the lower one (lane jj = node jj), lanes 8..15 the upper one reversed (lane jj = node N - jj);
// in both, towards the middle is towards higher lanes, so one row_shr scan runs both forward sweeps
//   y_node = rhs_node - H[piece behind]' y_prev
// and one row_shl scan both backward sweeps x_node = z_node - H[piece ahead] x_next (z = inverse block times y); a zero matrix
// at the head of a chain stops anything from crossing between lanes 7 and 8.  Chains of at most 8 nodes: three scan rounds
// where the one-ended sweep over 16 pieces took four, and the factor's chain (E2) is walked by two lanes at once.  The middle
// node's right-hand side takes one term from the last node of either chain (through Lm.mid: the source lanes depend on N).
template <int S, int NB>
__device__ __forceinline__ void chain_solve_twisted(PersistLds<S, NB> &Lm, double (&V)[3][S - 1][PersistLds<S, NB>::XW], const int N,
                                                    const int lane) {
  constexpr int m = S - 1;
  const int row = lane >> 4, j = lane & 15, ax = row < 3 ? row : 2;
  const int half = j >> 3, jj = j & 7;
  const int pm = N >> 1;
  const int cnt = half == 0 ? pm : N - pm;  // nodes of this lane's chain
  const int L = N - pm;                     // the longer chain (wave-uniform): how many scan rounds
  const bool valid = jj < cnt;
  const bool last = valid && jj == cnt - 1;
  const int k = valid ? (half == 0 ? jj : N - jj) : 0;                  // this lane's node
  const int pb = valid ? (half == 0 ? jj : N - jj - 1) : 0;             // piece between it and the next node towards the middle
  const int pf = (valid && jj > 0) ? (half == 0 ? jj - 1 : N - jj) : 0;  // piece between it and the previous node of its chain
  double Hf[m][m], Hb[m][m], M[m][m], v[m], Sk[m][m];
#pragma unroll
  for (int a = 0; a < m; ++a) {
    v[a] = valid ? V[ax][a][k] : 0.0;
#pragma unroll
    for (int b = 0; b < m; ++b) {
      Hf[a][b] = (valid && jj > 0) ? Lm.H[pf][a][b] : 0.0;
      Hb[a][b] = valid ? Lm.H[pb][a][b] : 0.0;
      Sk[a][b] = Lm.Si[k][a][b];
    }
  }
  if (row < 3 && jj == 0) {  // (an empty lower chain -- N = 1 -- contributes nothing)
#pragma unroll
    for (int a = 0; a < m; ++a) Lm.mid[ax][half][a] = 0.0;
  }
  // ---- forwards, both chains: y = rhs - Hf' y_prev
#pragma unroll
  for (int a = 0; a < m; ++a)
#pragma unroll
    for (int b = 0; b < m; ++b) M[a][b] = -Hf[b][a];
  if (L > 1) scan_round<0x111, m>(M, v, L > 2);
  if (L > 2) scan_round<0x112, m>(M, v, L > 4);
  if (L > 4) scan_round<0x114, m>(M, v, false);
  // ---- the middle node: rhs - Hb' y of the last node of either chain, then its inverse block
  if (last && row < 3) {
#pragma unroll
    for (int a = 0; a < m; ++a) {
      double t = 0.0;
#pragma unroll
      for (int b = 0; b < m; ++b) t = __builtin_fma(Hb[b][a], v[b], t);
      Lm.mid[ax][half][a] = t;
    }
  }
  double ym[m], xm[m], z[m];
#pragma unroll
  for (int a = 0; a < m; ++a) ym[a] = V[ax][a][pm] - Lm.mid[ax][0][a] - Lm.mid[ax][1][a];
#pragma unroll
  for (int a = 0; a < m; ++a) {
    double acc = 0.0, accz = 0.0;
#pragma unroll
    for (int b = 0; b < m; ++b) {
      acc = __builtin_fma(Lm.Si[pm][a][b], ym[b], acc);
      accz = __builtin_fma(Sk[a][b], v[b], accz);
    }
    xm[a] = acc;
    z[a] = valid ? accz : 0.0;
  }
  // ---- backwards, both chains: x = z - Hb x_next; the last node of a chain takes the middle node's value in
#pragma unroll
  for (int a = 0; a < m; ++a) {
    v[a] = z[a];
#pragma unroll
    for (int b = 0; b < m; ++b) M[a][b] = -Hb[a][b];
  }
  if (last) {
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) v[a] = __builtin_fma(M[a][b], xm[b], v[a]);
  }
  if (last || !valid) {
#pragma unroll
    for (int a = 0; a < m; ++a)
#pragma unroll
      for (int b = 0; b < m; ++b) M[a][b] = 0.0;
  }
  if (L > 1) scan_round<0x101, m>(M, v, L > 2);
  if (L > 2) scan_round<0x102, m>(M, v, L > 4);
  if (L > 4) scan_round<0x104, m>(M, v, false);
  if (row < 3) {
    if (valid) {
#pragma unroll
      for (int a = 0; a < m; ++a) V[ax][a][k] = v[a];
    }
    if (j == 0) {
#pragma unroll
      for (int a = 0; a < m; ++a) V[ax][a][pm] = xm[a];
    }
  }
  __syncthreads();
}

// ---- one objective evaluation of one problem by one wave ---------------------------------------------------------
// in: Lm.P (node positions), Lm.T (durations), Lm.hv / tv, rows; out: f (wave-uniform) and this lane's gradient
// component (lane < nw: waypoint coordinate lane = 3 (k-1) + axis; lane in [nw, nw+nt): dJ/dT of piece lane - nw,
// NOT yet multiplied by dT/dtau).
template <int S, int NB>
__device__ __forceinline__ void persist_eval(PersistLds<S, NB> &Lm, const double *rows, const PersistArgs &a, const int lane_in,
                                             const bool refactor, double &f_out, double &g_out) {
  // (an opaque copy: what the phases derive from the lane index -- node, axis, piece, sample group, LDS addresses -- is the
  //  same in every evaluation, and hoisted out of the evaluation loop it is carried through the optimiser's update in
  //  registers that has none to spare)
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  constexpr int m = S - 1, D = 2 * S, NLA = BlkOps<S>::NLA;
  constexpr int G = 64 / NB;
  using F = Factor<S, NB>;
  const int N = a.N, np = a.c - 1;
  int na = lane / 3, ax = lane - 3 * na;  // (node | piece, axis) mapping of the per-node / per-piece phases
  PERSIST_TICK_DECL;
  // (every phase starts from an opaque lane index of its own: its addresses die with it instead of being computed once,
  //  ahead of the first phase, and carried through all of them)
#define PERSIST_PHASE()                  \
  do {                                   \
    asm volatile("" : "+v"(lane));       \
    na = lane / 3;                       \
    ax = lane - 3 * na;                  \
  } while (0)

  // ---- E1: 1/T and the primal right-hand sides, lanes = (node, axis)
  if (na <= N) {
    const int k = na;
    const double rk = (k < N) ? fast_rcp(Lm.T[k < N ? k : 0]) : 0.0;
    const double rkm = (k > 0) ? fast_rcp(Lm.T[k > 0 ? k - 1 : 0]) : 0.0;
    if (ax == 0 && k < N) Lm.r[k] = rk;
    double hv[m], tv[m], y[m];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      hv[j] = Lm.hv[ax][j];
      tv[j] = Lm.tv[ax][j];
    }
    const double P0 = Lm.P[ax][k], Pm = Lm.P[ax][k > 0 ? k - 1 : 0], Pp = Lm.P[ax][k < N ? k + 1 : k];
    rhs_primal_node_rt<S>(k, N, np, rk, rkm, Pm, P0, Pp, hv, tv, y);
#pragma unroll
    for (int l = 0; l < m; ++l) Lm.X[ax][l][k] = y[l];
    if (ax == 0 && refactor) {  // the blocks of the system (minco_core.h Factor::factorize, assembly part)
      double Ak[m][m];
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l < m; ++l) Ak[j][l] = 0.0;
      if (k < N) {
        Pw<S> p(rk);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l) Ak[j][l] = __builtin_fma(Tab<S>::M[1 + j][1 + l], p[2 * S - 3 - j - l], Ak[j][l]);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l < m; ++l) Lm.Kp[k][j][l] = F::ko_const(k, N, np, j, l) * p[2 * S - 3 - j - l];
      }
      if (k > 0) {
        Pw<S> p(rkm);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l)
            Ak[j][l] = __builtin_fma(Tab<S>::M[S + 1 + j][S + 1 + l], p[2 * S - 3 - j - l], Ak[j][l]);
      }
      if (k == 0 || k == N) {
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l)
            if (j < np || l < np) Ak[j][l] = (j == l) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l) Lm.Ad[k][j][l] = Ak[j][l];
    }
  }
  __syncthreads();

  PERSIST_TICK(1);
  PERSIST_PHASE();
  // ---- E2: block LDL^T factorisation of the Schur complements S_k = A_k - Ko_{k-1}' S_{k-1}^-1 Ko_{k-1}, one lane.
  //      This is the only part that is sequential in earnest: per node two (three) reciprocals in a row.  Its inputs
  //      come from LDS a few nodes at a time into alternating register buffers (a load-then-use per node is an LDS
  //      round trip of dead time per node); what the sweeps need of it leaves as S_k^-1 and H_k = S_k^-1 Ko_k, so
  //      the sweeps are bare m x m recurrences (chain_solve) and the products with S_k^-1 run on lanes = (node, axis).
  // The system depends on the durations only: with the durations fixed (waypoints-only optimisation) it is factorised
  // by the first evaluation and S_k^-1, H_k stay in LDS for the rest of the run.
#if ANET_PERSIST_TWISTED
  // TWISTED: lane 0 walks nodes 0 .. pm-1 upwards, lane 1 walks nodes N .. pm+1 downwards -- the same instructions on
  // two lanes, so half the chain costs nothing extra -- and lane 0 finishes with the middle node pm = N / 2, whose block takes a
  // Schur term from either side.  The downward chain is the upward one with every coupling block transposed.  (Measured by
  // walking half the chain a second time: 3.7 k of the 9.6 k cycles of this phase per half chain at 16 jerk pieces.)
  if (lane < 2 && refactor) {
    const int pm = N >> 1;
    const int cnt = lane == 0 ? pm : N - pm;
    const int steps = N - pm;  // (the longer of the two chains: wave-uniform)
    double Dk[m][m];
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l < m; ++l) Dk[j][l] = 0.0;
    // operands of step t of this lane's chain: diagonal block of node kk, coupling block of piece kp (transposed downwards)
    auto load = [&](int t, double (&BA)[m][m], double (&BK)[m][m]) {
      const int tt = (t < cnt) ? t : (cnt > 0 ? cnt - 1 : 0);
      const int kk = lane == 0 ? tt : N - tt;
      const int kp = lane == 0 ? (tt < N ? tt : N - 1) : N - tt - 1;
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l < m; ++l) {
          BA[j][l] = (l <= j) ? Lm.Ad[kk][j][l] : 0.0;
          BK[j][l] = lane == 0 ? Lm.Kp[kp][j][l] : Lm.Kp[kp][l][j];
        }
    };
    // one node: S = Dk + A; inverse (2 x 2) or LDL^T factor (3 x 3) stored for the sweeps; W = S^-1 K stored as H of the piece
    // towards the middle; the Schur seed -K' W for the next node stays in Dk.  `on`: this lane's chain still has a node here.
    auto step = [&](const bool on, const int kk, const int kp, const double (&Ak)[m][m], const double (&Kk)[m][m]) {
      double Sd[m][m];
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l) Sd[j][l] = Dk[j][l] + Ak[j][l];
      if constexpr (m == 2) {
        const double r = fast_rcp(__builtin_fma(Sd[0][0], Sd[1][1], -Sd[1][0] * Sd[1][0]));
        const double s00 = Sd[1][1] * r, s11 = Sd[0][0] * r, s10 = -Sd[1][0] * r;
        double W[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          W[0][b] = __builtin_fma(s00, Kk[0][b], s10 * Kk[1][b]);
          W[1][b] = __builtin_fma(s10, Kk[0][b], s11 * Kk[1][b]);
        }
        if (on) {
          Lm.Si[kk][0][0] = s00; Lm.Si[kk][0][1] = s10; Lm.Si[kk][1][0] = s10; Lm.Si[kk][1][1] = s11;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            Lm.H[kp][0][b] = W[0][b];
            Lm.H[kp][1][b] = W[1][b];
          }
#pragma unroll
          for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb <= aa; ++bb) Dk[aa][bb] = -__builtin_fma(Kk[0][aa], W[0][bb], Kk[1][aa] * W[1][bb]);
        }
      } else {
        double Lk[NLA] = {}, dd[m], dik[m];
#pragma unroll
        for (int j = 0; j < m; ++j) {
          double dj = Sd[j][j];
          double ld_[m];
#pragma unroll
          for (int q = 0; q < j; ++q) {
            ld_[q] = Lk[BlkOps<S>::li(j, q)] * dd[q];
            dj = __builtin_fma(-ld_[q], Lk[BlkOps<S>::li(j, q)], dj);
          }
          dd[j] = dj;
          dik[j] = fast_rcp(dj);
#pragma unroll
          for (int i = j + 1; i < m; ++i) {
            double v = Sd[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) v = __builtin_fma(-Lk[BlkOps<S>::li(i, q)], ld_[q], v);
            Lk[BlkOps<S>::li(i, j)] = v * dik[j];
          }
        }
        double Y[m][m], Z[m][m];
#pragma unroll
        for (int l = 0; l < m; ++l) {
          double col[m];
#pragma unroll
          for (int j = 0; j < m; ++j) col[j] = Kk[j][l];
          BlkOps<S>::solve_L(Lk, col);
#pragma unroll
          for (int j = 0; j < m; ++j) {
            Y[j][l] = col[j];
            Z[j][l] = col[j] * dik[j];
          }
        }
        if (on) {
#pragma unroll
          for (int aa = 0; aa < m; ++aa)
#pragma unroll
            for (int bb = 0; bb <= aa; ++bb) {
              double acc = 0.0;
#pragma unroll
              for (int j = 0; j < m; ++j) acc = __builtin_fma(-Y[j][aa], Z[j][bb], acc);
              Dk[aa][bb] = acc;
            }
          // the factor of the node, for the lanes that turn it into the inverse block and H below: [1/d | L] in the slot
          double *slot = &Lm.Si[kk][0][0];
#pragma unroll
          for (int j = 0; j < m; ++j) slot[j] = dik[j];
#pragma unroll
          for (int q = 0; q < BlkOps<S>::nl; ++q) slot[m + q] = Lk[q];
        }
      }
    };
    auto node_of = [&](int t) { return lane == 0 ? t : N - t; };
    auto piece_of = [&](int t) { return lane == 0 ? t : N - t - 1; };
    double A1[m][m], K1[m][m], A2[m][m], K2[m][m];
    load(0, A1, K1);
#pragma unroll 1
    for (int t = 0; t < steps; t += 2) {
      load(t + 1, A2, K2);
      step(t < cnt, node_of(t), piece_of(t), A1, K1);
      load(t + 2, A1, K1);
      if (t + 1 < steps) step(t + 1 < cnt, node_of(t + 1), piece_of(t + 1), A2, K2);
    }
    // ---- the middle node: Schur terms of both chains (lane 1's arrive by a DPP swap), then its inverse block
    double Sd[m][m];
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l <= j; ++l) Sd[j][l] = Dk[j][l] + dpp_f64<0xB1>(Dk[j][l]) + Lm.Ad[pm][j][l];
    if (lane == 0) {
      if constexpr (m == 2) {
        const double r = fast_rcp(__builtin_fma(Sd[0][0], Sd[1][1], -Sd[1][0] * Sd[1][0]));
        const double s10 = -Sd[1][0] * r;
        Lm.Si[pm][0][0] = Sd[1][1] * r; Lm.Si[pm][0][1] = s10; Lm.Si[pm][1][0] = s10; Lm.Si[pm][1][1] = Sd[0][0] * r;
      } else {
        double Lk[NLA] = {}, dd[m], dik[m];
#pragma unroll
        for (int j = 0; j < m; ++j) {
          double dj = Sd[j][j];
          double ld_[m];
#pragma unroll
          for (int q = 0; q < j; ++q) {
            ld_[q] = Lk[BlkOps<S>::li(j, q)] * dd[q];
            dj = __builtin_fma(-ld_[q], Lk[BlkOps<S>::li(j, q)], dj);
          }
          dd[j] = dj;
          dik[j] = fast_rcp(dj);
#pragma unroll
          for (int i = j + 1; i < m; ++i) {
            double v = Sd[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) v = __builtin_fma(-Lk[BlkOps<S>::li(i, q)], ld_[q], v);
            Lk[BlkOps<S>::li(i, j)] = v * dik[j];
          }
        }
        double *slot = &Lm.Si[pm][0][0];
#pragma unroll
        for (int j = 0; j < m; ++j) slot[j] = dik[j];
#pragma unroll
        for (int q = 0; q < BlkOps<S>::nl; ++q) slot[m + q] = Lk[q];
      }
    }
  }
  __syncthreads();
  // inverse blocks and H = (inverse block) (coupling towards the middle) are off the chain: every node on its own lane (3 x 3)
  if (m > 2 && refactor && lane <= N) {
    const int k = lane, pm = N >> 1;
    double Lk[NLA] = {}, dik[m];
    const double *slot = &Lm.Si[k][0][0];
#pragma unroll
    for (int j = 0; j < m; ++j) dik[j] = slot[j];
#pragma unroll
    for (int q = 0; q < BlkOps<S>::nl; ++q) Lk[q] = slot[m + q];
    if (k != pm) {
      const int kp = k < pm ? k : k - 1;  // the piece towards the middle
#pragma unroll
      for (int l = 0; l < m; ++l) {
        double col[m];
#pragma unroll
        for (int j = 0; j < m; ++j) col[j] = k < pm ? Lm.Kp[kp][j][l] : Lm.Kp[kp][l][j];
        BlkOps<S>::solve_L(Lk, col);
#pragma unroll
        for (int j = 0; j < m; ++j) col[j] *= dik[j];
        BlkOps<S>::solve_LT(Lk, col);
#pragma unroll
        for (int j = 0; j < m; ++j) Lm.H[kp][j][l] = col[j];
      }
    }
    double Sk[m][m];
#pragma unroll
    for (int cc = 0; cc < m; ++cc) {
      double col[m];
#pragma unroll
      for (int j = 0; j < m; ++j) col[j] = (j == cc) ? 1.0 : 0.0;
      BlkOps<S>::solve_L(Lk, col);
#pragma unroll
      for (int j = 0; j < m; ++j) col[j] *= dik[j];
      BlkOps<S>::solve_LT(Lk, col);
#pragma unroll
      for (int j = 0; j < m; ++j) Sk[j][cc] = col[j];
    }
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int cc = 0; cc < m; ++cc) Lm.Si[k][j][cc] = Sk[j][cc];
  }
  if constexpr (m > 2) __syncthreads();
  chain_solve_twisted<S, NB>(Lm, Lm.X, N, lane);
#else
  if (lane == 0 && refactor) {
    constexpr int CH = (m <= 2) ? 4 : 2;
    double Dk[m][m];
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l < m; ++l) Dk[j][l] = 0.0;
    auto load = [&](int k0, double (&BA)[CH][m][m], double (&BK)[CH][m][m]) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int kk = (k0 + u <= N) ? k0 + u : N, kp = (kk < N) ? kk : (N > 0 ? N - 1 : 0);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l < m; ++l) {
            if (l <= j) BA[u][j][l] = Lm.Ad[kk][j][l];
            BK[u][j][l] = Lm.Kp[kp][j][l];
          }
      }
    };
    auto step = [&](const int k, const double (&Ak)[m][m], const double (&Kk)[m][m]) {
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l) Dk[j][l] += Ak[j][l];
      if constexpr (m == 2) {
        // 2 x 2: the inverse by its adjugate (same rounding as the LDL^T route -- tests/prototypes/scan_sweeps.py -- one
        // reciprocal instead of two on the chain), S_k^-1 and H_k = S_k^-1 Ko_k stored right away
        const double r = fast_rcp(__builtin_fma(Dk[0][0], Dk[1][1], -Dk[1][0] * Dk[1][0]));
        const double s00 = Dk[1][1] * r, s11 = Dk[0][0] * r, s10 = -Dk[1][0] * r;
        Lm.Si[k][0][0] = s00; Lm.Si[k][0][1] = s10; Lm.Si[k][1][0] = s10; Lm.Si[k][1][1] = s11;
        if (k < N) {
          double W[2][2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            W[0][b] = __builtin_fma(s00, Kk[0][b], s10 * Kk[1][b]);
            W[1][b] = __builtin_fma(s10, Kk[0][b], s11 * Kk[1][b]);
            Lm.H[k][0][b] = W[0][b];
            Lm.H[k][1][b] = W[1][b];
          }
#pragma unroll
          for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb <= aa; ++bb) Dk[aa][bb] = -__builtin_fma(Kk[0][aa], W[0][bb], Kk[1][aa] * W[1][bb]);
        }
        return;
      }
      double Lk[NLA] = {}, dd[m], dik[m];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        double dj = Dk[j][j];
        double ld_[m];
#pragma unroll
        for (int q = 0; q < j; ++q) {
          ld_[q] = Lk[BlkOps<S>::li(j, q)] * dd[q];
          dj = __builtin_fma(-ld_[q], Lk[BlkOps<S>::li(j, q)], dj);
        }
        dd[j] = dj;
        dik[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          double v = Dk[i][j];
#pragma unroll
          for (int q = 0; q < j; ++q) v = __builtin_fma(-Lk[BlkOps<S>::li(i, q)], ld_[q], v);
          Lk[BlkOps<S>::li(i, j)] = v * dik[j];
        }
      }
      if (k < N) {  // Schur complement seed for node k+1: the chain continues with it
        double Y[m][m], Z[m][m];
#pragma unroll
        for (int l = 0; l < m; ++l) {
          double col[m];
#pragma unroll
          for (int j = 0; j < m; ++j) col[j] = Kk[j][l];
          BlkOps<S>::solve_L(Lk, col);
#pragma unroll
          for (int j = 0; j < m; ++j) {
            Y[j][l] = col[j];
            Z[j][l] = col[j] * dik[j];
          }
        }
#pragma unroll
        for (int aa = 0; aa < m; ++aa)
#pragma unroll
          for (int bb = 0; bb <= aa; ++bb) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) acc = __builtin_fma(-Y[j][aa], Z[j][bb], acc);
            Dk[aa][bb] = acc;
          }
      }
      // the factor of the node, for the lanes that turn it into S_k^-1 and H_k below: [1/d | L] in the slot of S_k^-1
      double *slot = &Lm.Si[k][0][0];
#pragma unroll
      for (int j = 0; j < m; ++j) slot[j] = dik[j];
#pragma unroll
      for (int q = 0; q < BlkOps<S>::nl; ++q) slot[m + q] = Lk[q];
    };
    double A1[CH][m][m], K1[CH][m][m], A2[CH][m][m], K2[CH][m][m];
    load(0, A1, K1);
#pragma unroll 1
    for (int k0 = 0; k0 <= N; k0 += 2 * CH) {
      load(k0 + CH, A2, K2);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (k0 + u <= N) step(k0 + u, A1[u], K1[u]);
      load(k0 + 2 * CH, A1, K1);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (k0 + CH + u <= N) step(k0 + CH + u, A2[u], K2[u]);
    }
  }
  __syncthreads();
  // S_k^-1 = L^-T D^-1 L^-1 and H_k = S_k^-1 Ko_k are off the chain: every node on its own lane (3 x 3 blocks)
  if (m > 2 && refactor && lane <= N) {
    const int k = lane;
    double Lk[NLA] = {}, dik[m];
    const double *slot = &Lm.Si[k][0][0];
#pragma unroll
    for (int j = 0; j < m; ++j) dik[j] = slot[j];
#pragma unroll
    for (int q = 0; q < BlkOps<S>::nl; ++q) Lk[q] = slot[m + q];
    if (k < N) {
#pragma unroll
      for (int l = 0; l < m; ++l) {
        double col[m];
#pragma unroll
        for (int j = 0; j < m; ++j) col[j] = Lm.Kp[k][j][l];
        BlkOps<S>::solve_L(Lk, col);
#pragma unroll
        for (int j = 0; j < m; ++j) col[j] *= dik[j];
        BlkOps<S>::solve_LT(Lk, col);
#pragma unroll
        for (int j = 0; j < m; ++j) Lm.H[k][j][l] = col[j];
      }
    }
    double Sk[m][m];
#pragma unroll
    for (int cc = 0; cc < m; ++cc) {
      double col[m];
#pragma unroll
      for (int j = 0; j < m; ++j) col[j] = (j == cc) ? 1.0 : 0.0;
      BlkOps<S>::solve_L(Lk, col);
#pragma unroll
      for (int j = 0; j < m; ++j) col[j] *= dik[j];
      BlkOps<S>::solve_LT(Lk, col);
#pragma unroll
      for (int j = 0; j < m; ++j) Sk[j][cc] = col[j];
    }
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int cc = 0; cc < m; ++cc) Lm.Si[k][j][cc] = Sk[j][cc];
  }
  if constexpr (m > 2) __syncthreads();
  chain_solve<S, NB>(Lm, Lm.X, N, lane, na, ax);
#endif

  PERSIST_TICK(2);
  PERSIST_PHASE();
  // ---- E3: coefficients and energy share of every (piece, axis)
  if (na < N) {
    const int i = na;
    Pw<S> p(Lm.r[i]);
    double x0[m], x1[m];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      x0[l] = Lm.X[ax][l][i];
      x1[l] = Lm.X[ax][l][i + 1];
    }
    const double e = emit_piece<S>(i, p, Lm.P[ax][i], Lm.P[ax][i + 1], x0, x1,
                                   [&](int piece, int col, double v) { Lm.co[piece][ax][col] = v; });
    Lm.ep[i][ax] = e;
  }
  __syncthreads();

  PERSIST_TICK(3);
  PERSIST_PHASE();
  // ---- E4: penalty functional, lanes = (piece, sample group).  A lane holds NS samples of its piece at a time: their
  //      positions first, then the corridor rows are walked ONCE for all of them (four rows per LDS round trip), then the
  //      velocity / acceleration limits and the gradient per sample.  Same arithmetic as k_piece_grad (minco_kernels.h):
  //      * everything in units of mu: the rows are staged divided by mu, u = a.p - b, the smoothed L1 is mu F(u) with
  //        F(u) = uc^3 (1 - uc/2) + max(u - 1, 0), F' = uc^2 (3 - 2 uc), uc = clamp(u, 0, 1); a limit is
  //        u = |a1| kv - cv straight from the normalised-time sum a1 = sum c~ tb' (kv = 1 / (T mu), cv = vmax / mu);
  //      * gradient w.r.t. c~_k = c_k T^k accumulated per sample, scaled to d/dc once at the end;
  //      * no jerk and no per-sample d/dt: tau tb^(d+1)[col] = (k - d) tb^(d)[col], so the sample-time part of dJ/dT is
  //        (1/T) (sum_col c~[col] k gN[col] - sum_j (s1 a1 + 2 s2 a2)), gN being accumulated anyway;
  //      (One wave-uniform test per (row, sample slot) instead of one per row was measured: a row is violated by 0.5 % of
  //      the (lane, sample) pairs, so the per-row test is true four times out of five and the finer one once in four -- but
  //      a lone wave pays ~25 cycles per branch: 12.3 k cycles against 11.0 k for this phase.)
  {
    const int i = lane / G, grp = lane - G * i;
    constexpr int NS = (G == 4) ? 5 : 3;  // G * NS >= 20 samples (planner.yaml:21) in one pass
    double gC[3][D], gT = 0.0, pc = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int col = 0; col < D; ++col) gC[q][col] = 0.0;
    if (a.with_penalty) {
      if (i < N) {
        const Penalty pp = a.pp;
        const double Ti = Lm.T[i];
        const double inv_mu = a.inv_mu, inv_res = a.inv_res;  // (host-side reciprocals: an IEEE division is ~30 wave instructions)
        const double step = Ti * inv_res;
        const double rT = Lm.r[i], rT2 = rT * rT;  // (1 / T_i of E1)
        const double wcm = pp.wc * pp.mu, wvm = pp.wv * pp.mu, wam = pp.wa * pp.mu;
        const double kv = rT * inv_mu, ka = rT2 * inv_mu, cv = pp.vmax * inv_mu, ca = pp.amax * inv_mu;
        const double K0 = step * wcm, K1 = step * rT * pp.wv, K2 = step * rT2 * pp.wa;
        double ct[3][D];
        {
          double tk = 1.0;
#pragma unroll
          for (int col = D - 1; col >= 0; --col) {
#pragma unroll
            for (int q = 0; q < 3; ++q) ct[q][col] = Lm.co[i][q][col] * tk;
            tk *= Ti;
          }
        }
        const int M4 = a.hpolys ? (a.M + 3) & ~3 : 0;  // rows per piece in LDS (divided by mu), zero rows up to a multiple of four
        const double *rp = rows + (size_t)i * (4 * (size_t)M4 + 4);
        const bool all_ok = pp.res % (G * NS) == 0;  // every lane has a full set of samples in every pass
        double csum = 0.0, Rs1 = 0.0, Rs2 = 0.0;
        for (int base = 0; base < pp.res; base += G * NS) {
          double tau[NS], pos[NS][3], gp[NS][3], Fs[NS];
          bool ok[NS];
#pragma unroll
          for (int sI = 0; sI < NS; ++sI) {
            const int j = base + grp + sI * G;
            ok[sI] = j < pp.res;
            tau[sI] = (double)j * inv_res;
            Fs[sI] = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              double acc = ct[q][0];
#pragma unroll
              for (int col = 1; col < D; ++col) acc = __builtin_fma(acc, tau[sI], ct[q][col]);
              pos[sI][q] = acc;
              gp[sI][q] = 0.0;
            }
          }
          for (int r0 = 0; r0 < M4; r0 += 4) {
            double h[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int q = 0; q < 4; ++q) h[u][q] = rp[4 * (r0 + u) + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              double viol[NS], worst = 0.0;
#pragma unroll
              for (int sI = 0; sI < NS; ++sI) {
                viol[sI] = __builtin_fma(h[u][0], pos[sI][0], __builtin_fma(h[u][1], pos[sI][1], __builtin_fma(h[u][2], pos[sI][2], -h[u][3])));
                if (!all_ok) viol[sI] = ok[sI] ? viol[sI] : -1.0;
                worst = fmax(worst, viol[sI]);
              }
              if (__any(worst > 0.0)) {  // (one test per row: a lone wave pays ~25 cycles per branch, more than the skipped work)
#pragma unroll
                for (int sI = 0; sI < NS; ++sI) {
                  double f, df;
                  smoothed_l1_unit(viol[sI], f, df);
                  Fs[sI] += f;
                  gp[sI][0] = __builtin_fma(df, h[u][0], gp[sI][0]);
                  gp[sI][1] = __builtin_fma(df, h[u][1], gp[sI][1]);
                  gp[sI][2] = __builtin_fma(df, h[u][2], gp[sI][2]);
                }
              }
            }
          }
#pragma unroll
          for (int sI = 0; sI < NS; ++sI) {
            // basis rows of the sample: tb^(d)[col] = k!/(k-d)! tau^(k-d), k = D-1-col (tb^(0)[col] = pw[k])
            double pw[D];
            pw[0] = 1.0;
#pragma unroll
            for (int e = 1; e < D; ++e) pw[e] = pw[e - 1] * tau[sI];
            double tb1[D], tb2[D];  // (the entries with k < d are not touched: D - 1 and D - 2 columns take part)
#pragma unroll
            for (int col = 0; col < D - 1; ++col) tb1[col] = (double)(D - 1 - col) * pw[D - 2 - col];
#pragma unroll
            for (int col = 0; col < D - 2; ++col) tb2[col] = (double)((D - 1 - col) * (D - 2 - col)) * pw[D - 3 - col];
            double a1[3], a2[3], worst = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              double x1 = 0.0, x2 = 0.0;
#pragma unroll
              for (int col = 0; col < D - 1; ++col) x1 = __builtin_fma(ct[q][col], tb1[col], x1);
#pragma unroll
              for (int col = 0; col < D - 2; ++col) x2 = __builtin_fma(ct[q][col], tb2[col], x2);
              a1[q] = x1;
              a2[q] = x2;
              worst = fmax(worst, fmax(__builtin_fma(fabs(x1), kv, -cv), __builtin_fma(fabs(x2), ka, -ca)));
            }
            if (!all_ok) worst = ok[sI] ? worst : 0.0;
            double cost = wcm * Fs[sI];
            if (__any(worst > 0.0)) {  // only one of +v, -v (+a, -a) can be violated: the slope has the sign of a1 (a2)
              const double live = (all_ok || ok[sI]) ? 1.0 : 0.0;
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                double f, df;
                smoothed_l1_unit(__builtin_fma(fabs(a1[q]), kv, -cv), f, df);
                if (!all_ok) { f *= live; df *= live; }
                cost = __builtin_fma(wvm, f, cost);
                const double s1 = K1 * copysign(df, a1[q]);
                Rs1 = __builtin_fma(s1, a1[q], Rs1);
                smoothed_l1_unit(__builtin_fma(fabs(a2[q]), ka, -ca), f, df);
                if (!all_ok) { f *= live; df *= live; }
                cost = __builtin_fma(wam, f, cost);
                const double s2 = K2 * copysign(df, a2[q]);
                Rs2 = __builtin_fma(s2, a2[q], Rs2);
#pragma unroll
                for (int col = 0; col < D - 1; ++col) gC[q][col] = __builtin_fma(s1, tb1[col], gC[q][col]);
#pragma unroll
                for (int col = 0; col < D - 2; ++col) gC[q][col] = __builtin_fma(s2, tb2[col], gC[q][col]);
              }
            }
            if (__any(cost > 0.0)) {
              csum += cost;
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                const double s0 = K0 * gp[sI][q];
#pragma unroll
                for (int col = 0; col < D; ++col) gC[q][col] = __builtin_fma(s0, pw[D - 1 - col], gC[q][col]);
              }
            }
          }
        }
        pc = step * csum;
        {  // d/dT at fixed c: the quadrature weight T/res and the sample times tau_j T (gC still holds d/dc~)
          double acc = 0.0;
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int col = 0; col < D - 1; ++col) acc = __builtin_fma(ct[q][col] * (double)(D - 1 - col), gC[q][col], acc);
          gT = csum * inv_res + rT * (acc - __builtin_fma(2.0, Rs2, Rs1));
        }
        {  // d/dc = T^k d/dc~
          double tk = 1.0;
#pragma unroll
          for (int col = D - 1; col >= 0; --col) {
#pragma unroll
            for (int q = 0; q < 3; ++q) gC[q][col] *= tk;
            tk *= Ti;
          }
        }
      }
      // sum over the sample groups of the piece (all lanes take part)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int col = 0; col < D; ++col) gC[q][col] = group_sum<G>(gC[q][col]);
      gT = group_sum<G>(gT);
      pc = group_sum<G>(pc);
    }
    if (i < N && grp == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int col = 0; col < D; ++col) Lm.gc[i][q][col] = gC[q][col];
      Lm.gdT[i] = gT;
      Lm.pc[i] = pc;
    }
  }
  __syncthreads();

  PERSIST_TICK(4);
  PERSIST_PHASE();
  // ---- E5: energy part of dJ/dc, node-state adjoint contributions and the direct dPhi/dT term; lanes = (piece, axis)
  double x0s[S], x1s[S];  // node states of this lane's piece (position, derivatives), reused by E7
#pragma unroll
  for (int j = 0; j < S; ++j) x0s[j] = x1s[j] = 0.0;
  if (na < N) {
    const int i = na;
    const double Ti = Lm.T[i];
    Pw<S> p(Lm.r[i]);
    double c[D], gc[D];
#pragma unroll
    for (int col = 0; col < D; ++col) {
      c[col] = Lm.co[i][ax][col];
      gc[col] = Lm.gc[i][ax][col];
    }
    double gTl = 0.0;
    {  // d/dc and d/dT of int (p^(S))^2 (k_piece_grad, energy part)
      double tp[D];
      tp[0] = 1.0;
#pragma unroll
      for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Ti;
      double ps = 0.0;
#pragma unroll
      for (int j = S; j < D; ++j) {
        double fj = 1.0;
#pragma unroll
        for (int e = 0; e < S; ++e) fj *= (double)(j - e);
        ps = __builtin_fma(fj * tp[j - S], c[D - 1 - j], ps);
        double acc = 0.0;
#pragma unroll
        for (int k = S; k < D; ++k) {
          double fk = 1.0;
#pragma unroll
          for (int e = 0; e < S; ++e) fk *= (double)(k - e);
          acc = __builtin_fma(2.0 * fj * fk / (double)(j + k - 2 * S + 1) * tp[j + k - 2 * S + 1], c[D - 1 - k], acc);
        }
        gc[D - 1 - j] += acc;
      }
      gTl = ps * ps;
    }
    x0s[0] = Lm.P[ax][i];
    x1s[0] = Lm.P[ax][i + 1];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      x0s[l + 1] = Lm.X[ax][l][i];
      x1s[l + 1] = Lm.X[ax][l][i + 1];
    }
    double cA[S], cB[S];
    double fact = 1.0;
#pragma unroll
    for (int k = 0; k < S; ++k) {
      if (k > 0) fact *= (double)k;
      cA[k] = gc[D - 1 - k] * (1.0 / fact);
      cB[k] = 0.0;
    }
    double h[S];
#pragma unroll
    for (int q = 0; q < S; ++q) h[q] = gc[S - 1 - q] * p[q];
    double dsum = 0.0;
#pragma unroll
    for (int bb = 0; bb < 2 * S; ++bb) {
      const int dg = bb % S;
      double u = 0.0, qd = 0.0;
#pragma unroll
      for (int q = 0; q < S; ++q) {
        u = __builtin_fma(Tab<S>::BHI[q][bb], h[q], u);
        qd = __builtin_fma((double)(S + q - dg) * Tab<S>::BHI[q][bb], h[q], qd);
      }
      const double sc = p[S - dg];
      const double xb = (bb < S) ? x0s[dg] : x1s[dg];
      if (bb < S) cA[dg] += u * sc;
      else cB[dg] += u * sc;
      dsum = __builtin_fma(xb * sc, qd, dsum);
    }
    gTl = __builtin_fma(-p[1], dsum, gTl);
#pragma unroll
    for (int k = 0; k < S; ++k) {
      Lm.co[i][ax][k] = cA[k];
      Lm.co[i][ax][S + k] = cB[k];
    }
    Lm.gTp[i][ax] = gTl;
  }
  __syncthreads();

  PERSIST_TICK(5);
  PERSIST_PHASE();
  // ---- E6: adjoint solve K lam = g_x|free (pinned rows 0): right-hand sides on lanes = (node, axis), then the sweeps
  if (na <= N) {
    const int k = na;
#pragma unroll
    for (int l = 0; l < m; ++l) {
      double v = 0.0;
      if (k > 0) v += Lm.co[k - 1][ax][S + l + 1];
      if (k < N) v += Lm.co[k < N ? k : 0][ax][l + 1];
      Lm.X[ax][l][k] = ((k == 0 || k == N) && l < np) ? 0.0 : v;
    }
  }
  __syncthreads();
#if ANET_PERSIST_TWISTED
  chain_solve_twisted<S, NB>(Lm, Lm.X, N, lane);
#else
  chain_solve<S, NB>(Lm, Lm.X, N, lane, na, ax);
#endif

  PERSIST_TICK(6);
  PERSIST_PHASE();
  // ---- E7: position-row term and -lam' (dW/dT) x of every (piece, axis)
  if (na < N) {
    const int k = na;
    Pw<S> p(Lm.r[k]);
    double la[m], lb[m];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      la[l] = Lm.X[ax][l][k];
      lb[l] = Lm.X[ax][l][k + 1];
    }
    double wl = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) {
      wl = __builtin_fma(Tab<S>::M[0][1 + l] * p[2 * S - 2 - l], la[l], wl);
      wl = __builtin_fma(Tab<S>::M[0][S + 1 + l] * p[2 * S - 2 - l], lb[l], wl);
    }
    double xs[2 * S];
#pragma unroll
    for (int bb = 0; bb < 2 * S; ++bb) xs[bb] = ((bb < S) ? x0s[bb % S] : x1s[bb % S]) * p[S - bb % S];
    double acc = 0.0;
#pragma unroll
    for (int aa = 0; aa < 2 * S; ++aa) {
      const int da = aa % S;
      if (da == 0) continue;
      double row = 0.0;
#pragma unroll
      for (int bb = 0; bb < 2 * S; ++bb)
        row = __builtin_fma(Tab<S>::M[aa][bb] * (double)(2 * S - 1 - da - bb % S), xs[bb], row);
      const double ls = ((aa < S) ? la[da - 1] : lb[da - 1]) * p[S - da];
      acc = __builtin_fma(ls, row, acc);
    }
    Lm.wl[k][ax] = wl;
    Lm.gTp[k][ax] += acc;
  }
  __syncthreads();

  PERSIST_TICK(7);
  PERSIST_PHASE();
  // ---- E8: total gradient component of this lane and the cost
  double g = 0.0;
  if (lane < a.nw) {
    const int k = na + 1;  // waypoint node 1 .. N-1
    g = Lm.co[k][ax][0] + Lm.co[k - 1][ax][S] - Lm.wl[k][ax] + Lm.wl[k - 1][ax];
  } else if (lane < a.nw + a.nt) {
    const int i = lane - a.nw;
    g = Lm.gdT[i] + ((Lm.gTp[i][0] + Lm.gTp[i][1]) + Lm.gTp[i][2]) + a.pp.rho;
  }
  double fp = 0.0;
  if (lane < N) fp = ((Lm.ep[lane][0] + Lm.ep[lane][1]) + Lm.ep[lane][2]) + a.pp.rho * Lm.T[lane] + Lm.pc[lane];
  f_out = wave_sum<63>(fp);
  g_out = g;
  __syncthreads();  // (the next evaluation overwrites P / T)
  PERSIST_TICK(8);
}
#undef PERSIST_PHASE

// MR: history slots in registers (mem_size <= MR).
template <int S, int NB, int MR>
__global__ void __launch_bounds__(64, 2) k_lbfgs_minco_persistent(PersistArgs a) {
  extern __shared__ double smem[];
  PersistLds<S, NB> &Lm = *reinterpret_cast<PersistLds<S, NB> *>(smem);
  double *rows = smem + persist_lds_fixed_bytes<S, NB>() / sizeof(double);
  constexpr int m = S - 1;
  const int lane = threadIdx.x;
  const int64_t b = a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x, ld = a.ld;
  if ((uint64_t)b >= (uint64_t)a.B) return;  // (an out-of-range entry of a caller's order: leave it, touch nothing)
  if (a.resume && a.is[(int64_t)IS_DONE * ld + b] != 0) return;  // decided in the first launch
  const int N = a.N, np = a.c - 1, n = a.nw + a.nt;

  // ---- problem data -> LDS (batch-minor global: strided, once per problem)
  for (int e = lane; e < 3 * (N + 1); e += 64) {
    const int k = e / 3, q = e - 3 * k;
    double v;
    if (k == 0) v = a.head[(int64_t)(q * a.c) * ld + b];
    else if (k == N) v = a.tail[(int64_t)(q * a.c) * ld + b];
    else v = a.wps[(int64_t)((k - 1) * 3 + q) * ld + b];
    Lm.P[q][k] = v;
  }
  if (lane < N) Lm.T[lane] = a.T[(int64_t)lane * ld + b];
  if (lane < 3 * m) {
    const int q = lane / m, j = lane - m * q;
    Lm.hv[q][j] = (j < np) ? a.head[(int64_t)(q * a.c + 1 + j) * ld + b] : 0.0;
    Lm.tv[q][j] = (j < np) ? a.tail[(int64_t)(q * a.c + 1 + j) * ld + b] : 0.0;
  }
  if (a.with_penalty && a.hpolys) {
    const int per = 4 * a.M, per4 = 4 * ((a.M + 3) & ~3);
    for (int e = lane; e < N * per4; e += 64) {
      const int i = e / per4, w = e - per4 * i;
      rows[(size_t)i * (per4 + 4) + w] = (w < per) ? a.hpolys[(int64_t)(i * per + w) * ld + b] * a.inv_mu : 0.0;  // (in units of mu: E4)
    }
  }
  LbfgsResident<MR> st;
  st.init(lane < n ? a.x[(int64_t)lane * ld + b] : 0.0);
  double *cont = a.cont ? a.cont + b * (int64_t)kPersistContDoubles : nullptr;
  int e_first = 0;
  double f_half = 0.0;
  if (a.resume) {
    st.unpark(cont, lane);
    e_first = __builtin_amdgcn_readfirstlane(st.evals);
  }
  const int na = lane / 3, ax = lane - 3 * na;
  int finish = 0x7fffffff;
  __syncthreads();
#pragma unroll 1
  for (int e = e_first; e < a.max_evals; ++e) {
    // (the cancel word is fetched now and looked at after the evaluation: its round trip costs nothing)
    int cancel = 0;
    // system-scope load (sc0 sc1): the word is written while the kernel runs -- by another stream or by the host through
    // mapped pinned memory -- and a plain load could be served from this CU's vector L1 for the life of the kernel
    if (a.cancel) cancel = __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // publish the point to evaluate
    if (lane < a.nw) Lm.P[ax][na + 1] = st.x;
    else if (lane < n) Lm.T[lane - a.nw] = forward_T(st.x);
    __syncthreads();
    double f, g;
    persist_eval<S, NB>(Lm, rows, a, lane, e == e_first || a.nt > 0, f, g);
    PERSIST_TICK_DECL;
    if (lane >= a.nw && lane < n) g *= dforward_T(st.x);
    st.g = (lane < n) ? g : 0.0;
    finish = st.update(a.p, lane, f, StepBound{a.step_bound, a.nw, n, a.tau_min}, __builtin_amdgcn_readfirstlane(cancel));
    finish = __builtin_amdgcn_readfirstlane(finish);
    PERSIST_TICK(9);
    if (finish != 0x7fffffff) break;
    if (a.park && e + 1 == a.half_mark) f_half = st.fx;
  }
  if (a.park && finish == 0x7fffffff) st.park(cont, lane, f_half);
  if (lane < n) a.x[(int64_t)lane * ld + b] = st.x;
  if (lane == 0) {
    a.is[(int64_t)IS_DONE * ld + b] = finish != 0x7fffffff;
    a.is[(int64_t)IS_RET * ld + b] = finish;
    a.is[(int64_t)IS_K * ld + b] = st.k;
    a.is[(int64_t)IS_EVALS * ld + b] = st.evals;
    a.ds[(int64_t)DS_FX * ld + b] = st.fx;
  }
}

}  // namespace anet
