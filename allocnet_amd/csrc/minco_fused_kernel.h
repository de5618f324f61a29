// One cost + gradient evaluation of a SMALL batch in ONE launch: solve -> penalty / energy partial gradients -> adjoint.
//
// Why (profiles/r05_cost_grad_small_batch.txt): at BASELINE configs[2]'s literal batch (4096 x 8 pieces) the three launches of
// anet_minco_cost_grad_dev take 8.6 + 26.5 + 13.2 us of kernel time back to back, and ~9 us of each are the same whatever the
// arithmetic (k_piece_grad with ONE sample per piece instead of 20: 10.1 us): a kernel boundary makes every wave of the next
// kernel start with cold loads from beyond the L2 (what the previous kernel wrote is written back at its end), the adjoint
// re-factorises the system the solve factorised and re-reads the node states it had in registers.  Here a workgroup owns G
// trajectories from their waypoints to their gradients:
//   phase 1  wave 0, lane = (trajectory, axis): k_minco_solve_axis' chain WITHOUT its per-piece output step -- factor and the two
//            sweeps --; factor and node states are parked in LDS for phases 2 and 3;
//   phase 2  all 256 lanes, at least two per (trajectory, piece): the piece's coefficients (and energy share) from its two node
//            states (emit_piece, the same function the solve kernels call: same bits; to coeffs_out if asked for), then
//            k_piece_grad's penalty / energy code on them; the pair's partial gradient w.r.t. the piece's coefficients is turned into its adjoint
//            w.r.t. the piece's two node states right there (what the adjoint kernel's first loop does piece after piece on
//            one lane) and handed over through LDS;
//   phase 3  wave 0 again: the two sweeps of the adjoint with the factor and the node states of phase 1.
// No global round trip between the phases, one launch, no second factorisation.
// Every shape of two pieces and more eliminates the chain FROM BOTH ENDS in phases 1 and 3: two lanes per (trajectory, axis), the
// second one working on the trajectory reversed in time with the same code, the halves meeting at the middle node -- through a
// DPP swap in the exact shapes (8-piece snap, 16-piece jerk, c = 3: equal halves), through LDS otherwise (TW below; 4096 x 8 snap
// 29.8 -> 28.6 us, 2048 x 16 jerk 31.4 -> 25.6, 512 x 8 snap 20.3 -> 17.1, 700 x 6 snap c = 4 27.5 -> 19.8).
// The group size G is a run-time value (a power of two, at most FusedShape<NB>::G): a batch too small to give every CU a
// workgroup of G trajectories takes a smaller G, and the lanes that frees split the SAMPLES of a piece further: Q = 128 / (G NB)
// lane pairs per (trajectory, piece), 2 Q lanes taking every 2 Q-th sample each (the basis table from a copy in LDS, since the
// sample index differs from lane to lane); their partial gradients are added in a fixed order through LDS.  For batches that fill the chip the three
// streaming kernels remain the better shape (the host picks: allocnet_amd.hip cost_grad_dev_impl).
#pragma once
#include "minco_kernels.h"
#include "piece_grad_mx.h"

namespace anet {

struct FusedArgs {
  const double *head, *tail, *wps, *T, *hpolys;
  double *cost, *gradP, *gradT, *coeffs_out;
  const double *tau;  // optional: durations parametrised as T = forward_T(tau), gradT returned as dJ/dtau (L-BFGS driver)
  int64_t B, ld;
  int N, c;
  Penalty pp;
  int G;   // trajectories per workgroup: a power of two <= FusedShape<NB>::G
#ifdef ANET_FUSED_PROF
  long long *prof;  // [16] cycle stamps of thread 0 of workgroup 0 (tools: ANET_BUILD_FLAGS=-DANET_FUSED_PROF)
#endif
};
#ifdef ANET_FUSED_PROF
// (sched_barrier: nothing is scheduled across a stamp, or max-ILP scheduling moves the arithmetic of a phase past its stamp)
#define ANET_FP(k) do { __builtin_amdgcn_sched_barrier(0); if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[k] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ANET_FP(k) do { } while (0)
#endif

constexpr int kFusedMaxRes = 64;  // samples per piece the LDS copy of the basis table holds (more: the three-launch path)

template <int NB>
struct FusedShape {
  static constexpr int G = (128 / NB) < 16 ? (128 / NB) : 16;  // trajectories per workgroup: at most 128 (trajectory, piece) pairs
  static constexpr int PST = 130;                              // row stride of the LDS arrays (pairs, padded)
};

// MX (round 6; the exact shapes at 20 samples per piece, groups of at least 16 (trajectory, piece) pairs): phase 2 in k_piece_grad_mx's
// mapping on a wave per column set of 16 pairs -- for full groups (G NB = 128) EIGHT waves (512 lanes: two waves per SIMD in phase 2 -- a lone wave issues an instruction every 6 - 7 cycles, two
// fill each other's bubbles; the chain phases still run on waves 0 and 1, and at two waves per SIMD the kernel fits 256 registers
// with 40 - 68 B of scratch outside the hot loops) -- a wave owns one column set of 16 (trajectory, piece) pairs, four lanes per pair with five samples each, the
// contractions with the basis table on the FP64 matrix instructions (piece_grad_mx.h: mx_column_set); the piece's coefficients are
// formed by three of its four lanes (one axis each) and change hands through the LDS rows the adjoint's hand-over uses later.
template <int S, int NB, bool NEXACT = false, int NPC = -1, bool MX = false>
__global__ void __launch_bounds__(MX ? 512 : 256, 1) k_minco_cost_grad_fused(FusedArgs a, const double *__restrict__ tab) {
  constexpr int m = S - 1, D = 2 * S, GM = FusedShape<NB>::G, PST = FusedShape<NB>::PST;
  constexpr int ROW_T = 0, ROW_GX = ROW_T + 1, ROW_GTD = ROW_GX + 3 * D, ROW_GDT = ROW_GTD + 3, ROW_PC = ROW_GDT + 1, ROW_EN = ROW_PC + 1;
  constexpr int NRED = 3 * D + 2;  // values a lane pair hands to the pair that adds up a piece: gC, gT, pc
  __shared__ double lds[(ROW_EN + 1) * PST];
  __shared__ double lred[MX ? 1 : NRED * 128];
  __shared__ double ltab[MX ? 1 : kFusedMaxRes * 3 * D];
  constexpr int MXRB = 8, MXW = 8;                    // (MX) rows per parked block, waves of the workgroup
  constexpr int MXTST = MXRB * 4 + 2;                 // (MX) doubles per pair of a parked row block (+2: 16 pairs' reads on 16 bank groups)
  __shared__ double mx_lag[MX ? kMxNU * 64 * 2 : 1];  // (MX) gradient A operands [u][lane][ct]
  __shared__ double mx_laf[MX ? 4 * 64 * 2 : 1];      // (MX) forward A operands [tile][lane][ks]
  __shared__ double mx_row[MX ? MXW * 16 * MXTST : 1];  // (MX) per wave: the corridor rows of its column set, a block of MXRB at a time
  // What phase 3 needs of phase 1 (factor, node states, durations, energy: ~100 doubles per chain lane) waits in LDS, not in
  // registers across phase 2: with them the sample loop (whose table rows are per-lane values) goes into scratch
  constexpr int nl = Factor<S, NB>::nl > 0 ? Factor<S, NB>::nl : 1;
  // TW: the chain of an exact shape with an even number of pieces is eliminated from BOTH ends (see phase 1): two lanes per
  // (trajectory, axis), each with a chain of NC = NB / 2 pieces
  // (shapes with a run-time piece count, 2 <= N <= NB: halves of (N + 1) / 2 and N / 2 pieces -- the lanes of a pair reach the middle
  //  node at different steps of the unrolled loops when N is odd, so their exchange goes through LDS, in program order, instead of
  //  a DPP swap: the second half hands over its share of the middle node, the first half finishes the node and hands back its solution)
  constexpr bool TW = NB % 2 == 0;
  constexpr int NC = TW ? NB / 2 : NB, CL = TW ? 6 : 3;  // pieces of a chain, chain lanes per trajectory
  constexpr int MROWS = m * (m + 1) / 2 + m;             // what the halves exchange through LDS: a block's lower triangle, or a vector
  __shared__ double lmeet[(TW && !NEXACT) ? MROWS * 3 * GM : 1];
  constexpr int NST = (NC + 1) * (nl + 2 * m + 1) + 2 * NC, SST = CL * GM;
  __shared__ double lst[NST * SST];
  // (TW) the node states in the trajectory's own direction, for phase 2: value (node, j) of (trajectory, axis) at [(node (m + 1) + j) XST + 3 t + axis]
  constexpr int XST = 3 * GM;
  __shared__ double lxs[TW ? (NB + 1) * (m + 1) * XST : 1];
  const int G = a.G;                              // trajectories of this workgroup
  const int LPQ = 2 * G * NB;                     // lanes of one sample subset (a power of two <= 256)
  const int Q = 256 / LPQ;                        // lane pairs per (trajectory, piece)
  const int N = NEXACT ? NB : a.N;
  const int np = NPC >= 0 ? NPC : a.c - 1;
  const int c = np + 1;
  const int64_t ld = a.ld;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * G;
  ANET_FP(0);
  {
    // the rows of position, velocity and acceleration of every sample (read by other threads behind the barrier); by the waves
    // that have no chain to start: a load-to-store round trip in front of wave 0's chain would be in front of everything
    constexpr int CW = TW ? 2 : 1;  // waves with chain lanes
    if constexpr (MX) {
      static_assert(CW == 2, "two idle waves build the operand tables");
      const int lane = tid & 63, r = lane >> 4, col = lane & 15;
      const double inv_mu = 1.0 / a.pp.mu;
      if (wave == 2) {
#pragma unroll
        for (int u = 0; u < kMxNU; ++u)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const int ic = lane & 3, j = r + 4 * (u % kMxNSL), d = u / kMxNSL, cc = 4 * ct + ic;
            mx_lag[(u * 64 + lane) * 2 + ct] = tab[(size_t)(j * 4 + d) * D + (cc < D ? cc : 0)] * (cc < D ? 1.0 : 0.0);
          }
      } else if (wave == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int row = 16 * t + col, u = row >> 2, rr = row & 3, d = u / kMxNSL, j = rr + 4 * (u % kMxNSL), ck = 4 * ks + r;
            const bool in = u < kMxNU && ck < D;
            const double v = tab[(size_t)(j * 4 + (in ? d : 0)) * D + (in ? ck : 0)];
            mx_laf[(t * 64 + lane) * 2 + ks] = in ? (d == 0 ? v * inv_mu : v) : 0.0;
          }
      }
    } else {
      if (wave >= CW)
        for (int e = tid - 64 * CW; e < a.pp.res * 3 * D; e += 256 - 64 * CW) ltab[e] = tab[(size_t)(e / (3 * D)) * 4 * D + e % (3 * D)];
    }
  }

  // (MX) the corridor rows travel global -> registers -> LDS a block ahead of their use; the FIRST block of a wave's first column set
  // is requested during phase 1 -- by the idle waves at once, by the chain waves behind their chain -- and parked at the start of
  // phase 2.
  const int mx_M = (MX && a.hpolys) ? a.pp.M : 0, mx_nrb = (mx_M + MXRB - 1) / MXRB;
  constexpr int MXNM = MXRB / 4;  // rows a lane fetches per block
  double hn[MXNM][4];
  int hok = 0;
  auto fetch_rows = [&](const int rb) {  // lane (r, col): the rows r, r + 4 of its pair's block
    const int r = (tid & 63) >> 4, col = tid & 15, M = mx_M;
    const Penalty &pp = a.pp;
    hok = 0;
    if (16 * wave >= G * NB) return;  // (a smaller group: fewer column sets than waves -- wave-uniform)
    const int pr = 16 * wave + col, pc_ = pr / G, tt2 = pr % G;
    const int64_t bb = b0 + tt2 < a.B ? b0 + tt2 : a.B - 1;
#pragma unroll
    for (int mm = 0; mm < MXNM; ++mm) {
      const int rr = rb * MXRB + r + 4 * mm;
      const bool ok = rr < M;
      const double *src = a.hpolys + (int64_t)((pc_ * pp.M + (ok ? rr : 0)) * 4) * ld + bb;
#pragma unroll
      for (int e = 0; e < 4; ++e) hn[mm][e] = M > 0 ? src[(int64_t)e * ld] : 0.0;
      hok |= ok ? (1 << mm) : 0;
    }
  };

  if constexpr (MX) {
    if (wave >= 2 && mx_nrb > 0) fetch_rows(0);
  }

  // phase 2's lane mapping
  const int q2 = tid / LPQ;                      // which lane pair of its (trajectory, piece)
  const int pair = (tid % LPQ) >> 1, half = tid & 1;
  const int piece = pair / G, t2 = pair % G;
  // (pairs beyond N * G idle -- the two lanes of a pair always agree; GM * NB < 128: the lanes beyond the last subset too)
  const bool pair_ok = piece < N && q2 < Q;
  const int64_t bb2 = (pair_ok && b0 + t2 < a.B) ? b0 + t2 : (a.B - 1);

  // ---- phase 1 (wave 0): the coefficient solve, one lane per (trajectory, axis) ---------------------------------------------
  Factor<S, NC> F;
  double P[NC + 1], X[NC + 1][m], tt[NC];
  // (TW: lane = 2 (3 t + axis) + role; role 0 walks the first half of the chain, role 1 the second half BACKWARDS IN TIME)
  // (ten trajectories per wave, four lanes of a wave idle: the lanes of a trajectory add up across the axes by wave shuffles)
  const int lw6 = (tid & 63) % 6;
  const int role = TW ? (lw6 & 1) : 0;
  const int t1 = TW ? 10 * wave + (tid & 63) / 6 : tid / 3, ax1 = TW ? (lw6 >> 1) : tid % 3;
  const bool chain_lane = TW ? (tid < 128 && (tid & 63) < 60 && t1 < G) : tid < 3 * G;
  const int ci = TW ? 6 * t1 + lw6 : tid;  // the lane's column of the parking area
  const int Nh = (!TW || NEXACT) ? NC : (role ? N / 2 : (N + 1) / 2);  // pieces of this lane's chain
  const bool live1 = chain_lane && b0 + t1 < a.B;
  const int64_t bb1 = live1 ? b0 + t1 : (a.B - 1);
  double hv[m], tv[m];
  // Seen backwards in time a trajectory is a trajectory: node k' = N - k, piece i' = N - 1 - i, every derivative of odd order
  // changes its sign -- and the minimum-effort problem is the same problem.  So the second half of the chain is eliminated by
  // the SAME code on the reversed data (factorize_chain / sweep_*_chain<TF = true>: the last node of a half is the interior node
  // where the halves meet, nothing pinned there); where the halves meet, the two lanes exchange what their half contributes to
  // the middle node's block and right-hand side (a DPP swap, the partner's values with the signs of the reversal) and both
  // finish the middle node for themselves.  Half the sequential depth of the factorisation and of all four sweeps.
  auto pair_swap = [](double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  double *const mt = lmeet + ((TW && !NEXACT && chain_lane) ? 3 * t1 + ax1 : 0);  // the pair's column of the exchange area
  auto meet_block = [&](double (&Dk)[m][m]) {
    if constexpr (NEXACT) {
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l) {
          const double o = pair_swap(Dk[j][l]);
          Dk[j][l] += ((j + l) & 1) ? -o : o;
        }
    } else {
      if (role && chain_lane) {
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l) mt[(size_t)(j * (j + 1) / 2 + l) * 3 * GM] = Dk[j][l];
      }
      // (the two branches must stay two branches IN THIS ORDER: mutually exclusive per lane, the compiler may otherwise fold them
      //  into one if / else and run the reading side first -- seen with an even piece count, where both lanes of a pair are here in
      //  the same step; LDS instructions of a wave execute in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (!role && chain_lane) {
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
          for (int l = 0; l <= j; ++l) {
            const double o = mt[(size_t)(j * (j + 1) / 2 + l) * 3 * GM];
            Dk[j][l] += ((j + l) & 1) ? -o : o;
          }
      }
    }
  };
  auto meet_vector = [&](double (&y)[m]) {  // (component l is the derivative of order l + 1)
    if constexpr (NEXACT) {
#pragma unroll
      for (int l = 0; l < m; ++l) {
        const double o = pair_swap(y[l]);
        y[l] += ((l + 1) & 1) ? -o : o;
      }
    } else {
      if (role && chain_lane) {
#pragma unroll
        for (int l = 0; l < m; ++l) mt[(size_t)l * 3 * GM] = y[l];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (!role && chain_lane) {
#pragma unroll
        for (int l = 0; l < m; ++l) {
          const double o = mt[(size_t)l * 3 * GM];
          y[l] += ((l + 1) & 1) ? -o : o;
        }
      }
    }
  };
  // (the solution of the middle node: both halves have it themselves when they met by the swap; through LDS the first half has)
  auto meet_solution = [&](double (&x)[m]) {
    if constexpr (!NEXACT) {
      if (!role && chain_lane) {
#pragma unroll
        for (int l = 0; l < m; ++l) mt[(size_t)(m + l) * 3 * GM] = x[l];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (role && chain_lane) {
#pragma unroll
        for (int l = 0; l < m; ++l) {
          const double o = mt[(size_t)(m + l) * 3 * GM];
          x[l] = ((l + 1) & 1) ? -o : o;
        }
      }
    }
  };
  if constexpr (TW) {
    if (tid < 128) {  // waves 0 and 1
      const double *hp = a.head + (int64_t)(ax1 * c) * ld + bb1;
      const double *tp = a.tail + (int64_t)(ax1 * c) * ld + bb1;
#pragma unroll
      for (int i = 0; i < NC; ++i) tt[i] = a.T[(int64_t)(i < Nh ? (role ? N - 1 - i : i) : 0) * ld + bb1];
#pragma unroll
      for (int k = 0; k <= NC; ++k) {
        const int kr = k <= Nh ? (role ? N - k : k) : 0;
        const double *src = (kr == 0) ? hp : (kr < N) ? a.wps + (int64_t)((kr - 1) * 3 + ax1) * ld + bb1 : tp;
        P[k] = *src;
      }
#pragma unroll
      for (int j = 0; j < m; ++j) {
        const int64_t row = (j < np) ? 1 + j : 0;
        const double h = (role ? tp : hp)[row * ld];
        hv[j] = (j < np) ? ((role && ((j + 1) & 1)) ? -h : h) : 0.0;
        tv[j] = 0.0;
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) F.r[i] = fast_rcp(tt[i]);
      ANET_FP(1);
      F.template factorize_chain<true>(Nh, np, meet_block);
      ANET_FP(2);
      {
        double rr[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) rr[i] = launder(F.r[i]);
        sweep_forward_chain<true, S, NC>(F, Nh, np, rr, X,
                                         [&](int k, double (&y)[m]) { rhs_primal_node<S, NC, true>(k, Nh, np, rr, P, hv, tv, y); }, meet_vector);
#pragma unroll
        for (int i = 0; i < NC; ++i) rr[i] = launder(rr[i]);
        sweep_backward_chain<true, S, NC>(F, Nh, np, rr, X, [&](int, const Pw<S> &) {}, meet_solution);
      }
      ANET_FP(3);
      if constexpr (MX) {
        // (the chain waves' first row block: behind the chain -- in front of it the chain's own loads would wait behind these,
        //  loads return in order, and a forced wait for the chain's inputs there cost 1.8 k cycles of phase 1 -- ; it flies while
        //  the factor is parked and the barrier is reached)
        if (mx_nrb > 0) fetch_rows(0);
      }
      if (chain_lane) {
        if (ax1 == 0) {
#pragma unroll
          for (int i = 0; i < NC; ++i)
            if (i < Nh) lds[ROW_T * PST + (role ? N - 1 - i : i) * G + t1] = tt[i];
        }
        // the node states as phase 2 reads them: role 0 has nodes 0 .. its last, role 1 the others, backwards, with the signs of the reversal
#pragma unroll
        for (int k = 0; k <= NC; ++k) {
          if (k > Nh || (role && k == Nh)) continue;
          const int kr = role ? N - k : k;
          double *dst = lxs + (size_t)(kr * (m + 1)) * XST + 3 * t1 + ax1;
#pragma unroll
          for (int j = 0; j < m; ++j) dst[(size_t)j * XST] = (role && ((j + 1) & 1)) ? -X[k][j] : X[k][j];
          dst[(size_t)m * XST] = P[k];
        }
        int v = 0;
        auto put = [&](double x) { lst[(v++) * SST + ci] = x; };
#pragma unroll
        for (int k = 0; k <= NC; ++k) {
#pragma unroll
          for (int j = 0; j < nl; ++j) put(F.L[k][j]);
#pragma unroll
          for (int j = 0; j < m; ++j) put(F.dinv[k][j]);
#pragma unroll
          for (int j = 0; j < m; ++j) put(X[k][j]);
          put(P[k]);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          put(F.r[i]);
          put(tt[i]);
        }
      }
    }
  } else
  if (wave == 0) {
    const double *hp = a.head + (int64_t)(ax1 * c) * ld + bb1;
    const double *tp = a.tail + (int64_t)(ax1 * c) * ld + bb1;
#pragma unroll
    for (int i = 0; i < NB; ++i) tt[i] = a.T[(int64_t)(i < N ? i : 0) * ld + bb1];
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      const double *src = (k == 0) ? hp : (k < N) ? a.wps + (int64_t)((k - 1) * 3 + ax1) * ld + bb1 : tp;
      const double v = *src;
      P[k] = (k <= N) ? v : 0.0;
    }
#pragma unroll
    for (int j = 0; j < m; ++j) {
      const int64_t row = (j < np) ? 1 + j : 0;
      const double h = hp[row * ld], t = tp[row * ld];
      hv[j] = (j < np) ? h : 0.0;
      tv[j] = (j < np) ? t : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < N) F.r[i] = fast_rcp(tt[i]);
    ANET_FP(1);
    F.factorize(N, np);
    ANET_FP(2);
    // the two sweeps ONLY: the coefficients of a piece (emit_piece: a quarter of the chain's instructions when it runs here)
    // are formed from the node states by the lanes of phase 2, in parallel
    {
      double rr[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(F.r[i]) : 0.0;
      sweep_forward<S, NB>(F, N, np, rr, X, [&](int k, double (&y)[m]) { rhs_primal_node<S, NB>(k, N, np, rr, P, hv, tv, y); });
#pragma unroll
      for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(rr[i]) : 0.0;
      sweep_backward<S, NB>(F, N, np, rr, X, [&](int, const Pw<S> &) {});
    }
    ANET_FP(3);
    if (chain_lane && ax1 == 0) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i < N) lds[ROW_T * PST + i * G + t1] = tt[i];
    }
    {
      if (chain_lane) {
        int v = 0;
        auto put = [&](double x) { lst[(v++) * SST + ci] = x; };
#pragma unroll
        for (int k = 0; k <= NB; ++k) {
#pragma unroll
          for (int j = 0; j < nl; ++j) put(F.L[k][j]);
#pragma unroll
          for (int j = 0; j < m; ++j) put(F.dinv[k][j]);
#pragma unroll
          for (int j = 0; j < m; ++j) put(X[k][j]);
          put(P[k]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          put(F.r[i]);
          put(tt[i]);
        }
      }
    }
  }
  __syncthreads();
  ANET_FP(4);

  // The adjoint's first step for one (trajectory, piece, axis): g_x = Phi' gC for the piece's two node states and the direct
  // dPhi/dT term (k_minco_propagate's loop over the pieces, minco_kernels.h propagate_axis), from the piece's own coefficients:
  // x0[j] = j! c_j, x1 = the piece's derivatives at its end; handed to phase 3 through LDS.
  auto adjoint_first = [&](const int ax, const double (&cfa)[D], const double (&gca)[D], const double Ti, const int pair) {
    const Pw<S> p(fast_rcp(Ti));
    double tp[D];
    tp[0] = 1.0;
#pragma unroll
    for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Ti;
    double x0[S], x1[S], gs[S], ge[S], h[S];
    double fact = 1.0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (j > 0) fact *= (double)j;
      x0[j] = fact * cfa[D - 1 - j];
      double acc = 0.0;
#pragma unroll
      for (int q = j; q < D; ++q) {
        double f = 1.0;
#pragma unroll
        for (int e = 0; e < j; ++e) f *= (double)(q - e);
        acc = __builtin_fma(f * tp[q - j], cfa[D - 1 - q], acc);
      }
      x1[j] = acc;
      gs[j] = gca[D - 1 - j] * (1.0 / fact);
      ge[j] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < S; ++q) h[q] = gca[S - 1 - q] * p[q];
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
      const double xb = (bb < S) ? x0[dg] : x1[dg];
      if (bb < S) gs[dg] = __builtin_fma(u, sc, gs[dg]);
      else ge[dg] = u * sc;
      dsum = __builtin_fma(xb * sc, qd, dsum);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      lds[(ROW_GX + ax * D + j) * PST + pair] = gs[j];
      lds[(ROW_GX + ax * D + S + j) * PST + pair] = ge[j];
    }
    lds[(ROW_GTD + ax) * PST + pair] = -p[1] * dsum;
  };

  if constexpr (MX) {
    // ---- phase 2, MX: four lanes per (trajectory, piece), one column set of 16 pairs per wave (eight waves for a full group) ----------
    static_assert(TW && NEXACT && FusedShape<NB>::G * NB == 128, "an exact shape whose groups fill the workgroup");
    const int lane = tid & 63, r = lane >> 4, col = lane & 15;
    const Penalty pp = a.pp;
    const double inv_mu = 1.0 / pp.mu, inv_res = 1.0 / (double)pp.res;
    const int M = mx_M, nrb = mx_nrb;
    double *const lr = mx_row + wave * 16 * MXTST;
    double AE, AP;  // energy part: the A operands of k_piece_grad_mx (a row of the energy Hessian's integers, the derivative factors)
    {
      const int ie = lane & 3, ke = lane >> 4;
      double fi = 1.0, fk = 1.0;
      for (int e = 0; e < S; ++e) {
        fi *= (double)(D - 1 - ie - e);
        fk *= (double)(D - 1 - ke - e);
      }
      AE = (ie < S && ke < S) ? 2.0 * fi * fk / (double)(2 * S - 1 - ie - ke) : 0.0;
      AP = ke < S ? fk : 0.0;
    }
    const bool has1 = 4 + r < D;
    auto park_rows = [&]() {
#pragma unroll
      for (int mm = 0; mm < MXNM; ++mm) {
        double *dst = lr + col * MXTST + (r + 4 * mm) * 4;
        const bool ok = (hok >> mm) & 1;
        dst[0] = ok ? hn[mm][0] : 0.0;
        dst[1] = ok ? hn[mm][1] : 0.0;
        dst[2] = ok ? hn[mm][2] : 0.0;
        dst[3] = ok ? hn[mm][3] * inv_mu : 0.0;
      }
    };
    auto landed = [](double &v) { asm volatile("" : "+v"(v)); };
    auto wave_sync = []() {  // what one lane of the wave wrote to LDS is read by another behind this (LDS instructions execute in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // (a group of G < 16 trajectories has G NB / 16 column sets: the workgroup is launched with that many waves, four at least --
    //  two chain waves, two for the operand tables --, and a wave without a column set has nothing to do here)
    if (16 * wave < G * NB) {
      if (nrb > 0) {  // (requested during phase 1)
#pragma unroll
        for (int mm = 0; mm < MXNM; ++mm)
#pragma unroll
          for (int e = 0; e < 4; ++e) landed(hn[mm][e]);
        park_rows();
      }
      const int pair = 16 * wave + col, piece = pair / G, t2m = pair % G;
      const bool live = b0 + t2m < a.B;
      const int64_t bbm = live ? b0 + t2m : a.B - 1;
      const double Ti = lds[ROW_T * PST + pair];
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      double tp[D];
      tp[0] = 1.0;
#pragma unroll
      for (int e = 1; e < D; ++e) tp[e] = tp[e - 1] * Ti;
      // the piece's coefficients of axis r (lanes r < 3) from its two node states, as solve_axis emits them; c~ to the hand-over rows
      double cf[D], e_share = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) cf[k] = 0.0;
      if (r < 3) {
        const Pw<S> pw(fast_rcp(Ti));
        double x0[m], x1[m];
        const double *ns = lxs + 3 * t2m + r;
#pragma unroll
        for (int j = 0; j < m; ++j) {
          x0[j] = ns[(size_t)(piece * (m + 1) + j) * XST];
          x1[j] = ns[(size_t)((piece + 1) * (m + 1) + j) * XST];
        }
        const double P0 = ns[(size_t)(piece * (m + 1) + m) * XST], P1 = ns[(size_t)((piece + 1) * (m + 1) + m) * XST];
        double *cp = (a.coeffs_out && live) ? a.coeffs_out + (int64_t)(piece * 3 * D + r * D) * ld + bbm : nullptr;
        e_share = emit_piece<S>(piece, pw, P0, P1, x0, x1, [&](int, int k, double v) {
          cf[k] = v;
          if (cp) cp[(int64_t)k * ld] = v;
        });
#pragma unroll
        for (int k = 0; k < D; ++k) lds[(ROW_GX + r * D + k) * PST + pair] = cf[k] * tp[D - 1 - k];
      }
      wave_sync();
      double cb[3][2];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        cb[ax][0] = lds[(ROW_GX + ax * D + r) * PST + pair];
        cb[ax][1] = has1 ? lds[(ROW_GX + ax * D + (has1 ? 4 + r : r)) * PST + pair] : 0.0;
      }
      {  // the piece's energy share: the three axes' lanes added by a product with ones (every lane receives the sum)
        const double e_piece = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, e_share, 0.0, 0, 0, 0);
        if (r == 0) lds[ROW_EN * PST + pair] = e_piece;
      }
      ANET_FP(5);
      double gN[3][2], csum, Rs1, Rs2, rT, step;
      mx_column_set<S, MXRB>(pp, inv_mu, inv_res, lane_o, mx_lag, mx_laf, lr + col * MXTST, M, nrb, Ti, cb,
                             [&](const int rb) {  // the pair's next row block on its way while this one is walked
                               if (rb + 1 < nrb) fetch_rows(rb + 1);
                             },
                             [&](const int rb) {
                               if (rb + 1 < nrb) {
#pragma unroll
                                 for (int mm = 0; mm < MXNM; ++mm)
#pragma unroll
                                   for (int e = 0; e < 4; ++e) landed(hn[mm][e]);
                                 park_rows();
                               }
                             },
                             gN, csum, Rs1, Rs2, rT, step);
      ANET_FP(6);
      double acc = 0.0;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        acc = __builtin_fma(cb[ax][0] * (double)(D - 1 - r), gN[ax][0], acc);
        acc = __builtin_fma(cb[ax][1] * (double)(D - 5 - r), gN[ax][1], acc);
      }
      double gT = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, csum * inv_res + rT * (acc - __builtin_fma(2.0, Rs2, Rs1)), 0.0, 0, 0, 0);
      const double pc = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, step * csum, 0.0, 0, 0, 0);
      const double tsel0 = mx_sel4(r, tp[D - 1], tp[D - 2], tp[D - 3], tp[D - 4]);
      const double tsel1 = D == 8 ? mx_sel4(r, tp[3], tp[2], tp[1], tp[0]) : mx_sel4(r, tp[1], tp[0], 0.0, 0.0);
      // (cb already carries the powers: the columns' d/dc = T^k d/dc~ need them once more)
      const double rTS = S == 4 ? (rT * rT) * (rT * rT) : rT * (rT * rT);
      const double TA = tsel0 * (S == 4 ? rT * (rT * rT) : rT * rT);
      wave_sync();  // (every lane of the pair has read c~ from the hand-over rows: they take d/dc now)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const double ye = cb[ax][0] * rTS;
        const double e = __builtin_amdgcn_mfma_f64_4x4x4f64(AE, ye, 0.0, 0, 0, 0);
        const double ps = __builtin_amdgcn_mfma_f64_4x4x4f64(AP, ye, 0.0, 0, 0, 0);
        gT = __builtin_fma(ps, ps, gT);
        lds[(ROW_GX + ax * D + r) * PST + pair] = __builtin_fma(e, TA, gN[ax][0] * tsel0);
        if (has1) lds[(ROW_GX + ax * D + 4 + r) * PST + pair] = gN[ax][1] * tsel1;
      }
      wave_sync();
      ANET_FP(7);
      if (r < 3) {
        double gca[D];
#pragma unroll
        for (int k = 0; k < D; ++k) gca[k] = lds[(ROW_GX + r * D + k) * PST + pair];
        adjoint_first(r, cf, gca, Ti, pair);
      }
      if (r == 0) {
        lds[ROW_GDT * PST + pair] = gT;
        lds[ROW_PC * PST + pair] = pc;
      }
    }
  } else
  // ---- phase 2 (all waves): two lanes per (trajectory, piece) ------------------------------------------------------------------
  {
    const int q = q2;
    double gC[3][D], cf[3][D];
    double gT = 0.0, pc = 0.0, Ti = 1.0;
    if (pair_ok) {
      Ti = lds[ROW_T * PST + pair];
      // the piece's coefficients from its two node states (phase 1 parked them: lst[value][3 t + axis]), as solve_axis emits them
      {
        constexpr int PER = nl + 2 * m + 1;  // values parked per node: L, dinv, X, P
        const Pw<S> pw(fast_rcp(Ti));
        double e_piece = 0.0;
        double *cp = (a.coeffs_out && q2 == 0 && half == 0 && b0 + t2 < a.B) ? a.coeffs_out + (int64_t)(piece * 3 * D) * ld + bb2 : nullptr;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          double x0[m], x1[m], P0, P1;
          if constexpr (TW) {
            const double *ns = lxs + 3 * t2 + ax;
#pragma unroll
            for (int j = 0; j < m; ++j) {
              x0[j] = ns[(size_t)(piece * (m + 1) + j) * XST];
              x1[j] = ns[(size_t)((piece + 1) * (m + 1) + j) * XST];
            }
            P0 = ns[(size_t)(piece * (m + 1) + m) * XST];
            P1 = ns[(size_t)((piece + 1) * (m + 1) + m) * XST];
          } else {
            const double *ns = lst + 3 * t2 + ax;
#pragma unroll
            for (int j = 0; j < m; ++j) {
              x0[j] = ns[(size_t)(piece * PER + nl + m + j) * SST];
              x1[j] = ns[(size_t)((piece + 1) * PER + nl + m + j) * SST];
            }
            P0 = ns[(size_t)(piece * PER + nl + 2 * m) * SST];
            P1 = ns[(size_t)((piece + 1) * PER + nl + 2 * m) * SST];
          }
          e_piece += emit_piece<S>(piece, pw, P0, P1, x0, x1, [&](int, int col, double v) {
            cf[ax][col] = v;
            if (cp) cp[(int64_t)(ax * D + col) * ld] = v;
          });
        }
        if (q2 == 0 && half == 0) lds[ROW_EN * PST + pair] = e_piece;
      }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax)
#pragma unroll
        for (int col = 0; col < D; ++col) gC[ax][col] = 0.0;
      ANET_FP(5);
      // The 2 Q lanes of a (trajectory, piece) split its SAMPLES (lane h of pair q takes j = 2q + h, 2q + h + 2Q, ...) and each
      // visits every corridor row: the position is evaluated once per sample and pass, the limit rows once per sample.  (With
      // the rows split between the two lanes of a pair, as k_piece_grad<S, true> has it, both lanes evaluate every position and
      // the limit block runs for the whole wave on behalf of half its lanes: 18.3 us of sample loop against 14.3 at 4096 x 8.)
      piece_penalty_part<S, false, 1, true>(a.pp, a.hpolys, ld, bb2, piece, 0, 2 * q + half, Ti, cf, ltab, gC, gT, pc, 2 * Q);
      ANET_FP(6);
      if (half == 0 && q == 0) {
        double ch[3][S];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
          for (int col = 0; col < S; ++col) ch[ax][col] = cf[ax][col];
        piece_energy_compute<S>(ch, Ti, gC, gT);
      }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax)
#pragma unroll
        for (int col = 0; col < D; ++col) gC[ax][col] = pair_sum(gC[ax][col]);
      gT = pair_sum(gT);
      pc = pair_sum(pc);
    }
    {  // the lane pairs of a piece, added by pair 0 in the order of q (deterministic)
      const int PPQ = G * NB;  // pairs per subset; (Q - 1) PPQ < 128 slots
      if (pair_ok && q > 0 && half == 0) {
        const int slot = (q - 1) * PPQ + pair;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
          for (int col = 0; col < D; ++col) lred[(ax * D + col) * 128 + slot] = gC[ax][col];
        lred[(3 * D) * 128 + slot] = gT;
        lred[(3 * D + 1) * 128 + slot] = pc;
      }
      __syncthreads();
      if (pair_ok && q == 0 && half == 0) {
        for (int qq = 1; qq < Q; ++qq) {
          const int slot = (qq - 1) * PPQ + pair;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int col = 0; col < D; ++col) gC[ax][col] += lred[(ax * D + col) * 128 + slot];
          gT += lred[(3 * D) * 128 + slot];
          pc += lred[(3 * D + 1) * 128 + slot];
        }
      }
    }
    ANET_FP(7);
    if (pair_ok) {
      if (half == 0 && q == 0) {
        // The adjoint's first step, per piece and in parallel: g_x = Phi' gC for the piece's two node states and the direct
        // dPhi/dT term (k_minco_propagate's loop over the pieces, minco_kernels.h propagate_axis), from the piece's own
        // coefficients: x0[j] = j! c_j, x1 = the piece's derivatives at its end.
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) adjoint_first(ax, cf[ax], gC[ax], Ti, pair);
        lds[ROW_GDT * PST + pair] = gT;
        lds[ROW_PC * PST + pair] = pc;
      }
    }
  }
  ANET_FP(8);
  __syncthreads();
  ANET_FP(9);

  // ---- phase 3 (wave 0): the adjoint sweeps with the factor and the node states of phase 1 -----------------------------------------
  if (wave >= (TW ? 2 : 1)) return;
  {
    const int lane = chain_lane ? ci : 0;
    int v = 0;
    auto get = [&]() { return lst[(v++) * SST + lane]; };
#pragma unroll
    for (int k = 0; k <= NC; ++k) {
#pragma unroll
      for (int j = 0; j < nl; ++j) F.L[k][j] = get();
#pragma unroll
      for (int j = 0; j < m; ++j) F.dinv[k][j] = get();
#pragma unroll
      for (int j = 0; j < m; ++j) X[k][j] = get();
      P[k] = get();
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      F.r[i] = get();
      tt[i] = get();
    }
  }
  // (W_k lam^)[position row of node k] (the row of node k+1 is its negative) and - lam^' (dW/dT) x^ = sum_ab lam_a M_ab e_ab
  // r^(e_ab+1) x_b,  e_ab = 2S-1-deg a-deg b, of piece k of a chain; x^ from phase 1's registers
  auto piece_terms = [&](int k, const Pw<S> &p, const double (&la)[m], const double (&lb)[m], double &wl, double &acc) {
    wl = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) {
      wl = __builtin_fma(Tab<S>::M[0][1 + l] * p[2 * S - 2 - l], la[l], wl);
      wl = __builtin_fma(Tab<S>::M[0][S + 1 + l] * p[2 * S - 2 - l], lb[l], wl);
    }
    double xs[2 * S];
#pragma unroll
    for (int bb = 0; bb < 2 * S; ++bb) {
      const int dg = bb % S;
      const double xv = dg == 0 ? P[bb < S ? k : k + 1] : X[bb < S ? k : k + 1][dg - 1];
      xs[bb] = xv * p[S - dg];
    }
    acc = 0.0;
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
  };
  if constexpr (TW) {
    double GP[NC + 1], XA[NC + 1][m], gTl[NC], rr[NC];
    // the partial gradients of this half's pieces, in the half's own direction: its piece i is the trajectory's piece ir, and for
    // role 1 start and end of a piece change places and the odd derivatives their sign
#pragma unroll
    for (int k = 0; k <= NC; ++k) {
      GP[k] = 0.0;
#pragma unroll
      for (int l = 0; l < m; ++l) XA[k][l] = 0.0;
      if (chain_lane && k <= Nh) {
        if (k < Nh) {  // the half's piece k starts at its node k
          const int ir = role ? N - 1 - k : k, off = role ? S : 0;
          const double *src = lds + (size_t)(ROW_GX + ax1 * D + off) * PST + ir * G + t1;
          GP[k] = src[0];
#pragma unroll
          for (int l = 0; l < m; ++l) {
            const double v = src[(size_t)(1 + l) * PST];
            XA[k][l] = (role && ((l + 1) & 1)) ? -v : v;
          }
        }
        if (k > 0) {  // the half's piece k - 1 ends there
          const int ir = role ? N - k : k - 1, off = role ? 0 : S;
          const double *src = lds + (size_t)(ROW_GX + ax1 * D + off) * PST + ir * G + t1;
          GP[k] += src[0];
#pragma unroll
          for (int l = 0; l < m; ++l) {
            const double v = src[(size_t)(1 + l) * PST];
            XA[k][l] += (role && ((l + 1) & 1)) ? -v : v;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      gTl[i] = (chain_lane && i < Nh) ? lds[(ROW_GTD + ax1) * PST + (role ? N - 1 - i : i) * G + t1] : 0.0;
      rr[i] = launder(F.r[i]);
    }
    ANET_FP(10);
    sweep_forward_chain<true, S, NC>(F, Nh, np, rr, XA,
                                     [&](int k, double (&y)[m]) {
#pragma unroll
                                       for (int l = 0; l < m; ++l) y[l] = (k == 0 && l < np) ? 0.0 : XA[k][l];
                                     },
                                     meet_vector);
    ANET_FP(11);
#pragma unroll
    for (int i = 0; i < NC; ++i) rr[i] = launder(rr[i]);
    sweep_backward_chain<true, S, NC>(F, Nh, np, rr, XA, [&](int k, const Pw<S> &p) {
      double wl, acc;
      piece_terms(k, p, XA[k], XA[k + 1], wl, acc);
      GP[k] -= wl;
      GP[k + 1] += wl;
      gTl[k] += acc;
    }, meet_solution);
    ANET_FP(12);
    {
      double gown = GP[NC];  // this half's share of the middle node (the last node of its chain)
      if constexpr (!NEXACT) {
#pragma unroll
        for (int k = 1; k < NC; ++k) gown = (k == Nh) ? GP[k] : gown;
      }
      const double gmid = pair_sum(gown);  // both halves' shares
      if (live1 && a.gradP) {
        double *gp = a.gradP + (int64_t)ax1 * ld + bb1;
#pragma unroll
        for (int k = 1; k < NC; ++k)
          if (k < Nh) gp[(int64_t)(((role ? N - k : k) - 1) * 3) * ld] = GP[k];
        if (!role) gp[(int64_t)((Nh - 1) * 3) * ld] = gmid;
      }
    }
    double e_tot = 0.0, csum = 0.0, tsum = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const double tot = gTl[i] + __shfl_down(gTl[i], 2) + __shfl_down(gTl[i], 4);  // the three axes: lanes two apart
      if (live1 && ax1 == 0 && i < Nh) {
        const int ir = role ? N - 1 - i : i;
        const double gt = lds[ROW_GDT * PST + ir * G + t1] + tot + a.pp.rho;
        a.gradT[(int64_t)ir * ld + bb1] = a.tau ? gt * dforward_T(a.tau[(int64_t)ir * ld + bb1]) : gt;
      }
    }
    if (a.cost && live1 && ax1 == 0 && !role) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i < N) {
          csum += lds[ROW_PC * PST + i * G + t1];
          e_tot += lds[ROW_EN * PST + i * G + t1];
          tsum += lds[ROW_T * PST + i * G + t1];
        }
      a.cost[bb1] = e_tot + a.pp.rho * tsum + csum;
    }
    ANET_FP(13);
  } else
  {
    double GP[NB + 1], XA[NB + 1][m], gTl[NB], rr[NB];
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      GP[k] = 0.0;
#pragma unroll
      for (int l = 0; l < m; ++l) XA[k][l] = 0.0;
      if (k <= N && chain_lane) {
        if (k < N) {
          GP[k] = lds[(ROW_GX + ax1 * D) * PST + k * G + t1];
#pragma unroll
          for (int l = 0; l < m; ++l) XA[k][l] = lds[(ROW_GX + ax1 * D + 1 + l) * PST + k * G + t1];
        }
        if (k > 0) {
          GP[k] += lds[(ROW_GX + ax1 * D + S) * PST + (k - 1) * G + t1];
#pragma unroll
          for (int l = 0; l < m; ++l) XA[k][l] += lds[(ROW_GX + ax1 * D + S + 1 + l) * PST + (k - 1) * G + t1];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      gTl[i] = (i < N && chain_lane) ? lds[(ROW_GTD + ax1) * PST + i * G + t1] : 0.0;
      rr[i] = (i < N) ? launder(F.r[i]) : 0.0;
    }
    ANET_FP(10);
    // adjoint solve K lam = g_x|free (pinned rows 0)
    sweep_forward<S, NB>(F, N, np, rr, XA, [&](int k, double (&y)[m]) {
#pragma unroll
      for (int l = 0; l < m; ++l) y[l] = ((k == 0 || k == N) && l < np) ? 0.0 : XA[k][l];
    });
    ANET_FP(11);
#pragma unroll
    for (int i = 0; i < NB; ++i) rr[i] = (i < N) ? launder(rr[i]) : 0.0;
    sweep_backward<S, NB>(F, N, np, rr, XA, [&](int k, const Pw<S> &p) {
      double wl, acc;
      piece_terms(k, p, XA[k], XA[k + 1], wl, acc);
      GP[k] -= wl;
      GP[k + 1] += wl;
      gTl[k] += acc;
    });
    ANET_FP(12);
    if (live1 && a.gradP) {
      double *gp = a.gradP + (int64_t)ax1 * ld + bb1;
#pragma unroll
      for (int k = 1; k < NB; ++k)
        if (k < N) gp[(int64_t)((k - 1) * 3) * ld] = GP[k];
    }
    // per trajectory: the three axes' shares (adjacent lanes), the partial dJ/dT of phase 2, rho, the chain rule of tau
    double e_tot = 0.0, csum = 0.0, tsum = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < N) {
        const double tot = gTl[i] + __shfl_down(gTl[i], 1) + __shfl_down(gTl[i], 2);
        if (live1 && ax1 == 0) {
          const double gt = lds[ROW_GDT * PST + i * G + t1] + tot + a.pp.rho;
          a.gradT[(int64_t)i * ld + bb1] = a.tau ? gt * dforward_T(a.tau[(int64_t)i * ld + bb1]) : gt;
          csum += lds[ROW_PC * PST + i * G + t1];
          e_tot += lds[ROW_EN * PST + i * G + t1];
          tsum += tt[i];
        }
      }
    if (a.cost && live1 && ax1 == 0) a.cost[bb1] = e_tot + a.pp.rho * tsum + csum;
    ANET_FP(13);
  }
}

// false: no instantiation for this shape (the caller takes the three-launch path)
bool launch_cost_grad_fused(int s, const FusedArgs &a, const double *tab, hipStream_t st, int cus);
int cost_grad_fused_group(int s, int n_pieces);  // trajectories per workgroup of the instantiation that would run (0: none)

}  // namespace anet
