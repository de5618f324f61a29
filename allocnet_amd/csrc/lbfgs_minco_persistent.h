// L-BFGS over the MINCO cost as ONE launch: one wave per problem runs evaluation + update until ITS problem stops
// (lbfgs.hpp:551-709 is one loop per problem; the launch-per-evaluation driver makes the whole batch wait for its
// slowest member and pays four kernel boundaries per evaluation).
//
// The evaluation is the one of k_minco_solve / k_piece_grad / k_minco_propagate, re-mapped onto the 64 lanes of a
// wave with every intermediate in LDS (about 11 KB per wave at 16 pieces, plus the corridor rows):
//   * only the bare block-tridiagonal recurrences (factor, forward, backward; primal and adjoint) run on three
//     lanes (one per axis) -- same elimination order and operand order as minco_core.h;
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
  int64_t B, ld;
  int N, c, nw, nt, M, max_evals, with_penalty;
  Penalty pp;
  LbfgsP p;
};

template <int S, int NB>
struct PersistLds {
  static constexpr int m = S - 1, D = 2 * S, nl = m * (m - 1) / 2;
  double P[3][NB + 2];       // node positions per axis
  double T[NB], r[NB];       // durations, 1/T
  double hv[3][m], tv[3][m];  // pinned end derivatives
  double FL[NB + 1][nl > 0 ? nl : 1], Fd[NB + 1][m];  // block LDL^T factor
  double X[3][NB + 1][m];    // primal: right-hand side -> node derivatives
  double A[3][NB + 1][m];    // adjoint: right-hand side -> multipliers
  double co[NB][3][D];       // coefficients, highest power first
  double gc[NB][3][D];       // penalty part of dJ/dc
  double gdT[NB], pc[NB];    // penalty part of dJ/dT, penalty cost per piece
  double cA[NB][3][S], cB[NB][3][S];  // node-state adjoint contributions of a piece to its start / end node
  double wl[NB + 1][3], gTp[NB][3], ep[NB][3];
};

template <int S, int NB>
constexpr size_t persist_lds_fixed_bytes() { return (sizeof(PersistLds<S, NB>) + 15) / 16 * 16; }
// corridor rows of piece i start at i * (4 M + 4) doubles: the pad spreads the pieces over the LDS banks
inline size_t persist_lds_row_doubles(int N, int M) { return (size_t)N * (4 * (size_t)M + 4); }

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
template <int MR>
struct LbfgsResident {
  double x, g, d, xp, gp;
  double hs[MR], hy[MR], hys[MR];
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
    fx = step = finit = dgtest = dstest = mu = nu = pf = 0.0;
    k = bound = count = brackt = touched = evals = phase = 0;
  }
  __device__ __forceinline__ static double dot(double u, double v) { return wave_sum<63>(u * v); }
  __device__ __forceinline__ bool conv_test(const LbfgsP &P) const {
    return wave_max_nonneg<63>(fabs(g)) / fmax(1.0, wave_max_nonneg<63>(fabs(x))) < P.g_epsilon;
  }
  // consumes f = objective at x (gradient already in g); leaves the next point in x.  Returns the lbfgs.hpp
  // return code when the problem stops, 0x7fffffff while it runs.
  __device__ __forceinline__ int update(const LbfgsP &P, const int lane, const double f) {
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
            } else if (step > P.max_step) {
              if (touched) {
                err = LBERR_MAXIMUMSTEP;
              } else {
                touched = 1;
                step = P.max_step;
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
        x = __builtin_fma(step, d, xp);
      } else {
        fx = f;
        if (conv_test(P)) {
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
              hys[0] = ys;
              double alpha[MR];
#pragma unroll
              for (int it = 0; it < MR; ++it) {
                alpha[it] = 0.0;
                if (it < bound) {
                  alpha[it] = dot(hs[it], dv) / hys[it];
                  dv = __builtin_fma(-alpha[it], hy[it], dv);
                }
              }
              dv *= ys / yy;
#pragma unroll
              for (int it = MR - 1; it >= 0; --it) {
                if (it < bound) {
                  const double cf = alpha[it] - dot(hy[it], dv) / hys[it];
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
    if (start_ls) {  // entry of line_search_lewisoverton (lbfgs.hpp:287-305)
      xp = x;
      gp = g;
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
        nu = P.max_step;
        count = 0;
        brackt = 0;
        touched = 0;
        x = __builtin_fma(step, d, x);
      }
    }
    return finish;
  }
};

// ---- one objective evaluation of one problem by one wave ---------------------------------------------------------
// in: Lm.P (node positions), Lm.T (durations), Lm.hv / tv, rows; out: f (wave-uniform) and this lane's gradient
// component (lane < nw: waypoint coordinate lane = 3 (k-1) + axis; lane in [nw, nw+nt): dJ/dT of piece lane - nw,
// NOT yet multiplied by dT/dtau).
template <int S, int NB>
__device__ __forceinline__ void persist_eval(PersistLds<S, NB> &Lm, const double *rows, const PersistArgs &a, const int lane,
                                             double &f_out, double &g_out) {
  constexpr int m = S - 1, D = 2 * S, NLA = BlkOps<S>::NLA;
  constexpr int G = 64 / NB;
  using F = Factor<S, NB>;
  const int N = a.N, np = a.c - 1;
  const int na = lane / 3, ax = lane - 3 * na;  // (node | piece, axis) mapping of the per-node / per-piece phases

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
    for (int l = 0; l < m; ++l) Lm.X[ax][k][l] = y[l];
  }
  __syncthreads();

  // ---- E2: block LDL^T factor fused with the primal forward sweep, then the backward sweep; lanes 0..2 = axes
  if (lane < 3) {
    double Dk[m][m], Lp[NLA] = {}, dip[m] = {}, wp[m] = {};
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int l = 0; l < m; ++l) Dk[j][l] = 0.0;
#pragma unroll 1
    for (int k = 0; k <= N; ++k) {
      const double rk = Lm.r[k < N ? k : 0], rkm = Lm.r[k > 0 ? k - 1 : 0];
      if (k < N) {
        Pw<S> p(rk);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l) Dk[j][l] = __builtin_fma(Tab<S>::M[1 + j][1 + l], p[2 * S - 3 - j - l], Dk[j][l]);
      }
      if (k > 0) {
        Pw<S> p(rkm);
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l)
            Dk[j][l] = __builtin_fma(Tab<S>::M[S + 1 + j][S + 1 + l], p[2 * S - 3 - j - l], Dk[j][l]);
      }
      if (k == 0 || k == N) {
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l)
            if (j < np || l < np) Dk[j][l] = (j == l) ? 1.0 : 0.0;
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
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < BlkOps<S>::nl; ++q) Lm.FL[k][q] = Lk[q];
#pragma unroll
        for (int j = 0; j < m; ++j) Lm.Fd[k][j] = dik[j];
      }
      // forward step of this axis
      double y[m];
#pragma unroll
      for (int l = 0; l < m; ++l) y[l] = Lm.X[lane][k][l];
      if (k > 0) {
        Pw<S> p(rkm);
        double v[m];
#pragma unroll
        for (int j = 0; j < m; ++j) v[j] = wp[j] * dip[j];
        BlkOps<S>::solve_LT(Lp, v);
        F::sub_KoT(k - 1, N, np, p, v, y);
      }
      BlkOps<S>::solve_L(Lk, y);
#pragma unroll
      for (int l = 0; l < m; ++l) Lm.X[lane][k][l] = y[l];
      // Schur complement seed for node k+1
      if (k < N) {
        Pw<S> p(rk);
        double Y[m][m], Z[m][m];
#pragma unroll
        for (int l = 0; l < m; ++l) {
          double col[m];
#pragma unroll
          for (int j = 0; j < m; ++j) col[j] = F::ko_const(k, N, np, j, l) * p[2 * S - 3 - j - l];
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
#pragma unroll
      for (int q = 0; q < NLA; ++q) Lp[q] = Lk[q];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        dip[j] = dik[j];
        wp[j] = y[j];
      }
    }
    // backward sweep (the factor of the last node is still in registers; the others come back from LDS)
    double xn[m] = {};
#pragma unroll 1
    for (int k = N; k >= 0; --k) {
      double Lk[NLA] = {}, dik[m], x[m];
#pragma unroll
      for (int q = 0; q < BlkOps<S>::nl; ++q) Lk[q] = Lm.FL[k][q];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        dik[j] = Lm.Fd[k][j];
        x[j] = Lm.X[lane][k][j];
      }
      if (k < N) {
        Pw<S> p(Lm.r[k]);
        double t[m];
        F::mul_Ko(k, N, np, p, xn, t);
        BlkOps<S>::solve_L(Lk, t);
#pragma unroll
        for (int l = 0; l < m; ++l) x[l] -= t[l];
      }
#pragma unroll
      for (int l = 0; l < m; ++l) x[l] *= dik[l];
      BlkOps<S>::solve_LT(Lk, x);
#pragma unroll
      for (int l = 0; l < m; ++l) {
        Lm.X[lane][k][l] = x[l];
        xn[l] = x[l];
      }
    }
  }
  __syncthreads();

  // ---- E3: coefficients and energy share of every (piece, axis)
  if (na < N) {
    const int i = na;
    Pw<S> p(Lm.r[i]);
    double x0[m], x1[m];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      x0[l] = Lm.X[ax][i][l];
      x1[l] = Lm.X[ax][i + 1][l];
    }
    const double e = emit_piece<S>(i, p, Lm.P[ax][i], Lm.P[ax][i + 1], x0, x1,
                                   [&](int piece, int col, double v) { Lm.co[piece][ax][col] = v; });
    Lm.ep[i][ax] = e;
  }
  __syncthreads();

  // ---- E4: penalty functional, lanes = (piece, sample group)
  {
    const int i = lane / G, grp = lane - G * i;
    double gC[3][D], gT = 0.0, pc = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int col = 0; col < D; ++col) gC[q][col] = 0.0;
    if (a.with_penalty) {
      if (i < N) {
        const Penalty pp = a.pp;
        const double Ti = Lm.T[i];
        const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
        const double step = Ti * inv_res;
        const double rT = 1.0 / Ti, rT2 = rT * rT, rT3 = rT2 * rT;
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
        const double *rp = rows + (size_t)i * (4 * (size_t)a.M + 4);
        const int M = a.hpolys ? a.M : 0;
        for (int j = grp; j < pp.res; j += G) {
          const double tau = (double)j * inv_res;
          // basis rows of the sample: tb[d][col] = k!/(k-d)! tau^(k-d), k = D-1-col
          double pw[D];
          pw[0] = 1.0;
#pragma unroll
          for (int e = 1; e < D; ++e) pw[e] = pw[e - 1] * tau;
          double tb[4][D];
#pragma unroll
          for (int col = 0; col < D; ++col) {
            const int k = D - 1 - col;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              double fct = 1.0;
#pragma unroll
              for (int q = 0; q < d; ++q) fct *= (double)(k - q);
              tb[d][col] = (k >= d) ? fct * pw[k >= d ? k - d : 0] : 0.0;
            }
          }
          double st[4][3];
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              double acc = 0.0;
#pragma unroll
              for (int col = 0; col < D; ++col) acc = __builtin_fma(ct[q][col], tb[d][col], acc);
              st[d][q] = acc * (d == 0 ? 1.0 : d == 1 ? rT : d == 2 ? rT2 : rT3);
            }
          double cost = 0.0, g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
          bool active = false;
          for (int r = 0; r < M; ++r) {
            const double h0 = rp[4 * r], h1 = rp[4 * r + 1], h2 = rp[4 * r + 2], h3 = rp[4 * r + 3];
            const double viol = __builtin_fma(h0, st[0][0], __builtin_fma(h1, st[0][1], h2 * st[0][2])) - h3;
            if (__any(viol > 0.0)) {
              double f, df;
              smoothed_l1_clamped(pp.mu, inv_mu, viol, f, df);
              cost = __builtin_fma(pp.wc, f, cost);
              df *= pp.wc;
              g[0][0] = __builtin_fma(df, h0, g[0][0]);
              g[0][1] = __builtin_fma(df, h1, g[0][1]);
              g[0][2] = __builtin_fma(df, h2, g[0][2]);
              active = true;
            }
          }
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const double av = fabs(st[1][q]) - pp.vmax, aa_ = fabs(st[2][q]) - pp.amax;
            if (__any(av > 0.0)) {
              double f, df;
              smoothed_l1_clamped(pp.mu, inv_mu, av, f, df);
              cost = __builtin_fma(pp.wv, f, cost);
              g[1][q] = __builtin_fma(pp.wv * (st[1][q] < 0.0 ? -1.0 : 1.0), df, g[1][q]);
              active = true;
            }
            if (__any(aa_ > 0.0)) {
              double f, df;
              smoothed_l1_clamped(pp.mu, inv_mu, aa_, f, df);
              cost = __builtin_fma(pp.wa, f, cost);
              g[2][q] = __builtin_fma(pp.wa * (st[2][q] < 0.0 ? -1.0 : 1.0), df, g[2][q]);
              active = true;
            }
          }
          if (__any(active)) {
            pc = __builtin_fma(step, cost, pc);
            double dt = 0.0;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int q = 0; q < 3; ++q) dt = __builtin_fma(g[d][q], st[d + 1][q], dt);
            gT += cost * inv_res + step * dt * ((double)j * inv_res);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const double g0 = step * g[0][q], g1 = step * g[1][q] * rT, g2 = step * g[2][q] * rT2;
#pragma unroll
              for (int col = 0; col < D; ++col) {
                double acc = g0 * tb[0][col];
                acc = __builtin_fma(g1, tb[1][col], acc);
                acc = __builtin_fma(g2, tb[2][col], acc);
                gC[q][col] += acc;
              }
            }
          }
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
      x0s[l + 1] = Lm.X[ax][i][l];
      x1s[l + 1] = Lm.X[ax][i + 1][l];
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
      Lm.cA[i][ax][k] = cA[k];
      Lm.cB[i][ax][k] = cB[k];
    }
    Lm.gTp[i][ax] = gTl;
  }
  __syncthreads();

  // ---- E6: adjoint solve K lam = g_x|free with the factor of E2; lanes 0..2 = axes
  if (lane < 3) {
    double Lp[NLA] = {}, dip[m] = {}, wp[m] = {};
#pragma unroll 1
    for (int k = 0; k <= N; ++k) {
      double Lk[NLA] = {}, dik[m], y[m];
#pragma unroll
      for (int q = 0; q < BlkOps<S>::nl; ++q) Lk[q] = Lm.FL[k][q];
#pragma unroll
      for (int j = 0; j < m; ++j) dik[j] = Lm.Fd[k][j];
#pragma unroll
      for (int l = 0; l < m; ++l) {
        double v = 0.0;
        if (k > 0) v += Lm.cB[k - 1][lane][l + 1];
        if (k < N) v += Lm.cA[k < N ? k : 0][lane][l + 1];
        y[l] = ((k == 0 || k == N) && l < np) ? 0.0 : v;
      }
      if (k > 0) {
        Pw<S> p(Lm.r[k - 1]);
        double v[m];
#pragma unroll
        for (int j = 0; j < m; ++j) v[j] = wp[j] * dip[j];
        BlkOps<S>::solve_LT(Lp, v);
        F::sub_KoT(k - 1, N, np, p, v, y);
      }
      BlkOps<S>::solve_L(Lk, y);
#pragma unroll
      for (int l = 0; l < m; ++l) Lm.A[lane][k][l] = y[l];
#pragma unroll
      for (int q = 0; q < NLA; ++q) Lp[q] = Lk[q];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        dip[j] = dik[j];
        wp[j] = y[j];
      }
    }
    double xn[m] = {};
#pragma unroll 1
    for (int k = N; k >= 0; --k) {
      double Lk[NLA] = {}, dik[m], x[m];
#pragma unroll
      for (int q = 0; q < BlkOps<S>::nl; ++q) Lk[q] = Lm.FL[k][q];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        dik[j] = Lm.Fd[k][j];
        x[j] = Lm.A[lane][k][j];
      }
      if (k < N) {
        Pw<S> p(Lm.r[k]);
        double t[m];
        F::mul_Ko(k, N, np, p, xn, t);
        BlkOps<S>::solve_L(Lk, t);
#pragma unroll
        for (int l = 0; l < m; ++l) x[l] -= t[l];
      }
#pragma unroll
      for (int l = 0; l < m; ++l) x[l] *= dik[l];
      BlkOps<S>::solve_LT(Lk, x);
#pragma unroll
      for (int l = 0; l < m; ++l) {
        Lm.A[lane][k][l] = x[l];
        xn[l] = x[l];
      }
    }
  }
  __syncthreads();

  // ---- E7: position-row term and -lam' (dW/dT) x of every (piece, axis)
  if (na < N) {
    const int k = na;
    Pw<S> p(Lm.r[k]);
    double la[m], lb[m];
#pragma unroll
    for (int l = 0; l < m; ++l) {
      la[l] = Lm.A[ax][k][l];
      lb[l] = Lm.A[ax][k + 1][l];
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

  // ---- E8: total gradient component of this lane and the cost
  double g = 0.0;
  if (lane < a.nw) {
    const int k = na + 1;  // waypoint node 1 .. N-1
    g = Lm.cA[k][ax][0] + Lm.cB[k - 1][ax][0] - Lm.wl[k][ax] + Lm.wl[k - 1][ax];
  } else if (lane < a.nw + a.nt) {
    const int i = lane - a.nw;
    g = Lm.gdT[i] + ((Lm.gTp[i][0] + Lm.gTp[i][1]) + Lm.gTp[i][2]) + a.pp.rho;
  }
  double fp = 0.0;
  if (lane < N) fp = ((Lm.ep[lane][0] + Lm.ep[lane][1]) + Lm.ep[lane][2]) + a.pp.rho * Lm.T[lane] + Lm.pc[lane];
  f_out = wave_sum<63>(fp);
  g_out = g;
  __syncthreads();  // (the next evaluation overwrites P / T)
}

// MR: history slots in registers (mem_size <= MR).
template <int S, int NB, int MR>
__global__ void __launch_bounds__(64, 2) k_lbfgs_minco_persistent(PersistArgs a) {
  extern __shared__ double smem[];
  PersistLds<S, NB> &Lm = *reinterpret_cast<PersistLds<S, NB> *>(smem);
  double *rows = smem + persist_lds_fixed_bytes<S, NB>() / sizeof(double);
  constexpr int m = S - 1;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x, ld = a.ld;
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
    const int per = 4 * a.M;
    for (int e = lane; e < N * per; e += 64) {
      const int i = e / per, w = e - per * i;
      rows[(size_t)i * (per + 4) + w] = a.hpolys[(int64_t)e * ld + b];
    }
  }
  LbfgsResident<MR> st;
  st.init(lane < n ? a.x[(int64_t)lane * ld + b] : 0.0);
  const int na = lane / 3, ax = lane - 3 * na;
  int finish = 0x7fffffff;
  __syncthreads();
#pragma unroll 1
  for (int e = 0; e < a.max_evals; ++e) {
    // publish the point to evaluate
    if (lane < a.nw) Lm.P[ax][na + 1] = st.x;
    else if (lane < n) Lm.T[lane - a.nw] = forward_T(st.x);
    __syncthreads();
    double f, g;
    persist_eval<S, NB>(Lm, rows, a, lane, f, g);
    if (lane >= a.nw && lane < n) g *= dforward_T(st.x);
    st.g = (lane < n) ? g : 0.0;
    finish = st.update(a.p, lane, f);
    finish = __builtin_amdgcn_readfirstlane(finish);
    if (finish != 0x7fffffff) break;
  }
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
