#!/usr/bin/env python3
"""Benchmark of the MINCO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_PER_GPU] [--pieces 8] [--order 4]

`--gpus N` with N > 1 starts its own N ranks (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1);
launched BY torch.distributed.run (WORLD_SIZE set) the process is one rank of the job, as the driver's multi-GPU form has it.

One "step" = one pass of the hot path over one batch of synthetic input: for every trajectory of
the batch the banded minimum-control-effort coefficient solve + energy (BASELINE.json configs[1]:
8-segment min-snap, random waypoints, energy-only), inputs already resident in HBM in the library's
batch-minor layout, followed (N > 1) by the all-gather of the per-trajectory costs over RCCL/xGMI
that the north star names.  Batches shard across ranks (weak scaling, fixed per-GPU batch).

Prints ONE JSON line (rank 0).  `value` is whole-job trajectories/s.  Extra objects:
  roofline      dominant kernel (k_minco_solve) against HBM: algorithmic bytes / mean kernel time
                measured with HIP events on the launch stream
  cpu_baseline  the C oracle (classic banded-LU MINCO, oracle/minco_oracle.c) on the host cores
  config1_b1024 the literal configs[1] batch (B=1024), which is launch-latency bound
  host_api      PCIe-inclusive rate through the host-pointer entry point (never `value`)
  config5       BASELINE.json configs[4] (SURVEY 8(d) "config 5"): 32768 x 8-segment min-snap with corridor and
                dynamic-limit penalties, sharded over the ranks, one cost + gradient evaluation per step, costs
                all-gathered; every rank takes part, so an N-GPU run measures the configuration BASELINE names for 8 GPUs

  config3       BASELINE.json configs[2]: 4096 x 8-segment min-snap, corridor penalties + time-allocation gradients (seed 1):
                one cost + gradient evaluation per step at the literal batch and at a saturating one, FP64 roofline,
                CPU figure (oracle/minco_costgrad.c, classic banded LU + adjoint, all host cores) beside it
  config4       BASELINE.json configs[3]: 4096 x 16-segment min-jerk, L-BFGS (lbfgs_parameter_t defaults) until every
                problem stops on its own (seed 2): seconds, trajectories/s, evaluation statistics, status histogram, FP64
                roofline, CPU figure (oracle_lbfgs_optimize on the C cost + gradient, all host cores, a sample of the batch)

`--workload config5` makes that evaluation the timed main workload instead (same JSON contract; strong scaling:
the 32768 trajectories are a fixed total).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy


def algorithmic_bytes(s, c, N):
    """compulsory FP64 read+write once per trajectory, energy-only solve."""
    return 8 * (2 * 3 * c + N + 3 * (N - 1) + 3 * 2 * s * N + 1)


def pmc_traffic_bytes(B, N, s):
    """HBM bytes per launch of k_minco_solve from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json, written by tools/summarize_prof.py): WRITE_SIZE + 2 x FETCH_SIZE in KiB
    (gfx950 FETCH_SIZE counts half of a coalesced read stream, MI355X_MICROARCH.md, HBM section).
    None when no profile of this launch shape is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for e in d.get("k_minco_solve", []):
            if e.get("grid") == B and e.get("pieces") == N and e.get("order") == s:
                best = (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0
    return best


def pmc_leg_traffic(leg, picks, tag=None):
    """HBM bytes per step of one leg from the committed rocprofv3 PMC passes of THAT leg (tools/profile_leg.sh runs
    `bench.py --workload <leg> --main-only` under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs and
    tools/summarize_leg.py writes profiles/<round>_<leg>_pmc.json: per (kernel, grid) the mean KiB per launch): the sum over
    `picks` = [(kernel-name substring, grid or None, launches per step)] of launches x (WRITE_SIZE + 2 x FETCH_SIZE) x 1024 (gfx950's
    FETCH_SIZE counts half of a coalesced read stream, MI355X_MICROARCH.md, HBM section).  The newest file wins; None when a
    pick has no entry (no profile of this leg committed, or the launch shape changed since)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{leg}_pmc.json")))
    if tag:
        files = [f for f in files if os.path.basename(f).startswith(tag + "_")]
    for f in reversed(files):
        try:
            kernels = json.load(open(f))["kernels"]
        except Exception:
            continue
        total = 0.0
        for sub, grid, per_step in picks:
            hits = [k for k in kernels if sub in k["name"] and (grid is None or k["grid"] == grid)
                    and k.get("fetch_kib") is not None and k.get("write_kib") is not None]
            if len(hits) != 1:
                total = None
                break
            total += per_step * (2.0 * hits[0]["fetch_kib"] + hits[0]["write_kib"]) * 1024.0
        if total is not None:
            return total
    return None


def config5_bytes(s, c, N, M):
    """compulsory bytes of one cost + gradient evaluation per trajectory: the energy-only figure plus
    the corridor rows 8*4*sum(M_i) and the gradient outputs 8*(N + 3(N-1)) (6248 B at N = 8, s = 4, M = 16)."""
    return algorithmic_bytes(s, c, N) + 8 * 4 * N * M + 8 * (N + 3 * (N - 1))


FP64_PEAK_TFLOPS = 74.5        # measured FMA peak, tools/micro/fp64_peak.hip (profiles/r02_fp64_peak.txt); nominal 78.6


def host_cores():
    """Host cores this process may really use: the scheduler affinity AND the cgroup CPU quota (the GPU boxes report 256
    logical CPUs but run the container with a 16-CPU quota; 256 threads there only oversubscribe 16 cores)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def classic_solve_flops(s, N):
    """SURVEY.md 8(d) "Algorithmic FLOPs": banded LU of the n = 2sN collocation system with p = q = 2s
    (2 n p q) and the forward / backward substitution of the three right-hand sides (3 * 2 n (p + q))."""
    n, p = 2 * s * N, 2 * s
    return 2 * n * p * p + 3 * 2 * n * 2 * p


def cost_grad_flops(s, N, M, res):
    """Analytic, data-independent FP64 operation count of ONE cost + gradient evaluation of one trajectory (FMA = 2), of the
    reference formulation -- what any implementation has to do whatever the data:
      per piece and sample: position, velocity and acceleration of 3 axes (3 x 3 x 2s FMA), the residual of every corridor
      row (3 FMA + 1 per row) and of the 12 box rows (1 each);
      the coefficient solve and the adjoint solve through the same factors (classic banded LU: SURVEY 8(d));
      the energy and its partial gradients (3 axes x s x s FMA, twice).
    The data-DEPENDENT part (smoothed L1 and gradient accumulation of the rows that are violated) is not counted, so the
    fraction of the FP64 peak derived from this is a lower bound on useful work."""
    D = 2 * s
    per_sample = 3 * 3 * D * 2 + M * 7 + 12
    return N * res * per_sample + classic_solve_flops(s, N) + 3 * 2 * (2 * s * N) * 2 * D + N * 3 * s * s * 2 * 2


def lbfgs_update_flops(n, m):
    """two-loop recursion with a full history (4 m dot products / axpys of length n), the pair update and the line-search
    vector operations of one accepted step"""
    return 4 * m * n * 2 + 10 * n * 2


def fp64_roofline(flops_per_launch, seconds, hbm_bytes_per_launch, kernel, traffic=None):
    ach = flops_per_launch / seconds / 1e12
    hb = hbm_bytes_per_launch / seconds / 1e9
    # (FLOPs: analytic, data-independent -- cost_grad_flops; "hbm": SURVEY 8(d)'s compulsory bytes over the same time;
    #  "traffic": HBM bytes per step from the committed PMC passes of this leg, pmc_leg_traffic -- a constant of the tree)
    # (sub-leg rooflines: TFLOP/s against FP64_PEAK_TFLOPS, stated once per line as "fp64_peak_tflops"; a leg that is the
    #  line's main workload gets "peak" and "unit" back: with_peak)
    return {"bound": "fp64", "achieved": ach, "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic, "kernel": kernel,
            # (hbm: GB/s of the compulsory bytes and their fraction of the 8 TB/s peak)
            "hbm": {"achieved": hb, "frac": hb / HBM_PEAK_GBS,
                    "traffic_over_algorithmic": (traffic / hbm_bytes_per_launch) if traffic else None}}


def with_peak(roof):
    """the contract's keys for a roofline object that is a line's MAIN one"""
    return dict(roof, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s")


COST_GRAD_KERNELS = {1: "k_minco_cost_grad_fused", 3: "k_piece_grad (+ solve, propagate)"}   # (+ k_minco_solve, k_minco_propagate)
PIECE_GRAD_KERNELS = {0: "k_piece_grad", 1: "k_piece_grad", 2: "k_piece_grad", 3: "k_piece_grad_mx"}


def cost_grad_kernel_label(aa, ctx, s, N, B, pen):
    """the kernels of one cost + gradient evaluation of this shape, dominant first: what the library's own predicates say runs
    (anet_minco_cost_grad_launches, anet_minco_piece_grad_shape)"""
    if aa.minco_cost_grad_launches(s, N, B, penalty=pen, ctx=ctx) == 1:
        return COST_GRAD_KERNELS[1]
    return PIECE_GRAD_KERNELS[aa.minco_piece_grad_shape(s, N, B, penalty=pen, ctx=ctx)] + " (+ solve, propagate)"


def cost_grad_picks(launches):
    """the kernels of one cost + gradient evaluation as pmc_leg_traffic picks"""
    if launches == 1:
        return [("k_minco_cost_grad_fused", None, 1)]
    return [("k_minco_solve<4, 8, true, 2>", None, 1), ("k_piece_grad", None, 1), ("k_minco_propagate<", None, 1)]


PEN = dict(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20)
PEN_ORACLE = dict(rho=50.0, res=20, vmax=4.0, amax=6.0, wc=1e4, wv=1e3, wa=1e3, mu=1e-2)


def _to_bm(torch, a, B, ld, device):
    import numpy as np
    f = np.ascontiguousarray(a.reshape(B, -1).T)
    t = torch.zeros(f.shape[0], ld, device=device, dtype=torch.float64)
    t[:, :B] = torch.from_numpy(f).to(device)
    return t


def timed_reps(torch, fn, K, reps=3, warm_ms=10.0):
    """[(host ms per call, stream ms per call)] of `reps` repetitions of K back-to-back calls of fn, each repetition bracketed
    by one HIP-event pair on the current stream and a host clock, after at least `warm_ms` of GPU time of the same calls
    (clocks, code objects, first-touch of the outputs: none of it in the timed repetitions)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if K >= 100:
        # The HIP runtime stalls a stream ONCE per process, for 20-60 ms, when the host first gets ~10^3 launches ahead of
        # the GPU (tools/stall_probe.py, profiles/r04_launch_backlog_stall.txt: always in the launches 900-1050 of a
        # back-to-back loop of small kernels, never again).  A loop of 200 three-launch evaluations can be where that
        # happens -- round 3's driver run: 11 ms of kernels + 28 ms = "0.196 ms per step" -- so it is made to happen here.
        for _ in range(450):
            fn()
        torch.cuda.synchronize()
    warmed, rounds = 0.0, 0
    while warmed < warm_ms and rounds < 1000:
        e0.record()
        for _ in range(max(5, K // 8)):
            fn()
        e1.record()
        torch.cuda.synchronize()
        warmed += e0.elapsed_time(e1)
        rounds += 1
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(((time.perf_counter() - t0) / K * 1e3, e0.elapsed_time(e1) / K))
    return out


def cost_grad_kernel_split(torch, aa, ctx, s, c, N, B, ld, th, tt, tw, tT, thp, pen, work, gP, gT, K=50):
    """Device time of the three launches of one evaluation, each between its own HIP events (microseconds, mean of K; an
    interval holds the launch gap in front of its kernel): anet_minco_solve_dev -> anet_minco_partial_grads_dev ->
    anet_minco_propagate_grad_dev on the buffers the fused entry point uses."""
    import ctypes
    nco = N * 3 * 2 * s
    w_co, w_gdC = work[:nco * ld], work[nco * ld:2 * nco * ld]
    w_gdT = work[2 * nco * ld:(2 * nco + N) * ld]
    w_pc = work[(2 * nco + N) * ld:(2 * nco + 2 * N) * ld]
    w_en = work[(2 * nco + 2 * N) * ld:(2 * nco + 2 * N + 1) * ld]
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pp = ctypes.cast(ctypes.pointer(pen), ctypes.c_void_p)
    lib, h = ctx.lib, ctx.handle

    def three(ev):
        ev[0].record()
        ctx.check(lib.anet_minco_solve_dev(h, s, c, N, B, ld, q(th), q(tt), q(tw), q(tT), q(w_co), q(w_en), st))
        ev[1].record()
        ctx.check(lib.anet_minco_partial_grads_dev(h, s, N, B, ld, q(w_co), q(tT), q(thp), pp, 1, q(w_gdC), q(w_gdT), q(w_pc), st))
        ev[2].record()
        ctx.check(lib.anet_minco_propagate_grad_dev(h, s, c, N, B, ld, q(tT), q(w_co), q(w_gdC), q(w_gdT), q(gP), q(gT), st))
        ev[3].record()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    for ev in evs[:5]:
        three(ev)
    torch.cuda.synchronize()
    for ev in evs:
        three(ev)
    torch.cuda.synchronize()
    names = ("k_minco_solve", PIECE_GRAD_KERNELS[aa.minco_piece_grad_shape(s, N, B, penalty=pen, ctx=ctx)], "k_minco_propagate")
    return {n: 1e3 * sum(ev[i].elapsed_time(ev[i + 1]) for ev in evs) / K for i, n in enumerate(names)}


def run_config3(torch, aa, ctx, device, cpu_baseline, cpu_seconds, split=True):
    """BASELINE configs[2] (SURVEY 8(d) "config 3"): B = 4096 x 8-segment min-snap, corridor (M = 16) + limit penalties,
    gradients w.r.t. waypoints and durations, seed 1.  One step = one cost + gradient evaluation of the whole batch -- ONE
    launch (k_minco_cost_grad_fused) at the literal batch, k_minco_solve -> k_piece_grad -> k_minco_propagate at the saturating
    one; the line names the kernel that ran (anet_minco_cost_grad_launches) -- timed with events on the launch stream.
    `split`: also time the three streaming launches one by one on the same buffers (never under a profiler: `--main-only`)."""
    import numpy as np
    from allocnet_amd.synth import corridor_problem
    s, c, N, M = 4, 3, 8, 16
    out = {"pieces": N, "order": s, "poly_rows": M, "res": PEN["res"], "seed": 1,
           "unit": "trajectory cost+gradient evaluations/s"}
    flops = cost_grad_flops(s, N, M, PEN["res"])
    ab = config5_bytes(s, c, N, M)
    pen = aa.make_penalty(poly_rows=M, **PEN)
    host = None
    for key, B, K in (("b4096", 4096, 200), ("saturating", 1 << 17, 30)):
        head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, c, M)
        if key == "b4096":
            host = (head, tail, wps, T, hp)
        ld = aa.recommended_ld(B)
        th, tt, tw, tT, thp = (_to_bm(torch, x, B, ld, device) for x in (head, tail, wps, T, hp))
        cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)

        def evaluate():
            aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP,
                                   gradT=gT, ctx=ctx)
        runs = timed_reps(torch, evaluate, K, reps=5, warm_ms=10.0)
        host_ms = sorted(r[0] for r in runs)
        st_ms = sorted(r[1] for r in runs)
        dt, kms = host_ms[len(host_ms) // 2] * 1e-3, st_ms[len(st_ms) // 2]
        # results of the FUSED entry point, taken before the per-kernel split below runs: the split's third launch is the
        # plain anet_minco_propagate_grad_dev (no rho * sum T term, allocnet_amd.hip) and has output buffers of its own
        if key == "b4096":
            torch.cuda.synchronize()
            snap = (cost[:B].cpu().numpy(), gT[:, :B].cpu().numpy().T.copy(), gP[:, :B].cpu().numpy().T.copy())
        # (timing: 5 repetitions of K back-to-back evaluations after >= 10 ms of warm-up and, for the small batch, a burst of 450
        #  evaluations that takes the runtime's one-off launch-backlog stall; median repetition: DESIGN.md section 7)
        launches = aa.minco_cost_grad_launches(s, N, B, penalty=pen, ctx=ctx)
        out[key] = {"batch": B, "ms_per_step": dt * 1e3, "stream_ms_per_step": kms, "value": B / dt,
                    "stream_ms_min_max": [st_ms[0], st_ms[-1]], "launches_per_step": launches,
                    "roofline": fp64_roofline(B * flops, kms * 1e-3, B * ab, cost_grad_kernel_label(aa, ctx, s, N, B, pen),
                                              traffic=pmc_leg_traffic("config3", cost_grad_picks(launches)))}
        if split:
            # the three streaming launches one by one (at a one-launch batch this is NOT what the step above ran: it is the
            # path the one launch replaces, reported under a name that says so)
            gP2, gT2 = torch.empty_like(gP), torch.empty_like(gT)
            out[key]["kernel_split_us" if launches == 3 else "three_launch_split_us"] = cost_grad_kernel_split(
                torch, aa, ctx, s, c, N, B, ld, th, tt, tw, tT, thp, pen, work, gP2, gT2)
            del gP2, gT2
    out["flops_per_evaluation"] = flops
    out["algorithmic_bytes_per_trajectory"] = ab
    gpu_cost, gpu_gT, gpu_gP = snap
    if cpu_baseline:
        from oracle import cbind
        nthreads = host_cores()
        head, tail, wps, T, hp = host
        cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, nthreads=nthreads, **PEN_ORACLE)      # (threads, pages)
        t0 = time.perf_counter()
        cc, cgP, cgT = cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, nthreads=nthreads, **PEN_ORACLE)
        one = time.perf_counter() - t0
        reps = max(1, min(200, int(cpu_seconds / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(reps):
            cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, nthreads=nthreads, **PEN_ORACLE)
        rate = 4096 * reps / (time.perf_counter() - t0)
        # (classic banded-LU MINCO + adjoint through the same factors + penalty partials in scalar C: oracle/minco_costgrad.c)
        out["cpu_baseline"] = {"value": rate, "unit": out["unit"], "cores": nthreads, "kind": "port",
                               "sample": f"the 4096 trajectories x {reps} passes, oracle/minco_costgrad.c",
                               "gpu_vs_cpu_max_rel_cost_err": float(np.abs(gpu_cost - cc).max() / np.abs(cc).max()),
                               "gpu_vs_cpu_max_rel_gradT_err": float(np.abs(gpu_gT - cgT).max() / np.abs(cgT).max()),
                               "gpu_vs_cpu_max_rel_gradP_err": float(np.abs(gpu_gP - cgP.reshape(gpu_gP.shape)).max()
                                                                     / np.abs(cgP).max())}
    return out


def qp_newton_step_flops(s, N, M, res):
    """Analytic FP64 operation count of ONE Newton step of the interior-point QP in Hermite node coordinates (FMA = 2),
    profiles/r03_qp_ipm_roofline.txt: five row passes at ~30 operations per row, the states of u and du at every sample in
    every pass, the assembly of A'WA from the per-sample weights, the block Cholesky and the two block substitutions."""
    D, BK = 2 * s, 3 * s
    rows = N * res * (M + 12)
    passes = 5 * 30 * rows
    states = 11 * N * res * 9 * D * 2
    assembly = N * D * D * res * 12 * 2
    chol = (N + 1) * (BK ** 3 // 3 + 2 * BK ** 3) + 2 * 2 * (2 * N + 1) * BK * BK
    return passes + states + assembly + chol


def run_qp(torch, aa, ctx, device, cpu_baseline, cpu_seconds, extras=True):
    """The reference's ONLINE solve: the inequality QP QPSolver::solve hands to OSQP
    (planner/qp_solver.hpp:119-358) -- corridor rows and velocity / acceleration boxes at `res` samples per piece -- for
    4096 problems in one launch of k_qp_ipm (interior point; one workgroup per problem), inputs resident, events on the
    launch stream.  Shapes: 8-segment min-snap (BASELINE's problem size) and the planner's own 5 pieces
    (learning_planner.hpp:179), SURVEY 8(d) corridor generator, seed 1, durations x 1.5."""
    import numpy as np
    from allocnet_amd.synth import corridor_problem
    # (a batch lasts as long as its slowest problem; ~1.5 % of the generator's problems are infeasible, status -3)
    out = {"unit": "QP solves/s", "seed": 1, "res": 20, "poly_rows": 16, "max_vel": 4.0, "max_acc": 6.0, "method": "interior point"}
    host = None
    for key, s, N, B in (("snap8", 4, 8, 4096), ("jerk5", 3, 5, 4096)):
        M = 16
        head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
        state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
        T = T * 1.5
        if key == "snap8":
            host = (s, N, M, state, T, hp)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        st, tT, thp = t(state), t(T), t(hp)
        r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx)
        torch.cuda.synchronize()
        K = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps_ms = []
        for _ in range(3):                          # median of three repetitions of K batches
            e0.record()
            for _ in range(K):
                r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx)
            e1.record()
            torch.cuda.synchronize()
            reps_ms.append(e0.elapsed_time(e1) / K)
        ms = sorted(reps_ms)[1]
        iters = r["iters"].double()
        solved = float((r["status"] == 1).double().mean())
        flops = float(iters.sum()) * qp_newton_step_flops(s, N, M, 20)
        ach = flops / (ms * 1e-3) / 1e12
        out[key] = {"order": s, "pieces": N, "batch": B, "ms_per_batch": ms, "value": B / (ms * 1e-3), "solved_frac": solved,
                    "newton_steps_mean": float(iters.mean()), "newton_steps_max": int(iters.max()),
                    # (FLOPs: analytic per Newton step, qp_newton_step_flops, x the steps taken; the kernel is latency-bound --
                    #  block-Cholesky chains and row passes of one 256-thread workgroup per problem: DESIGN.md 8b)
                    "infeasible_frac": float((r["status"] == -3).double().mean()),
                    # (traffic: slacks and multipliers live in global memory -- L2-resident while a problem runs --, everything else in
                    #  LDS and registers; two launches per batch: Newton steps 1-4 of every problem, then the unfinished ones)
                    "roofline": {"bound": "fp64", "achieved": ach, "frac": ach / FP64_PEAK_TFLOPS, "kernel": "k_qp_ipm",
                                 "traffic": pmc_leg_traffic("qp", [(f"k_qp_ipm<{s},", None, 2)])}}
        if not extras:
            continue
        # the same batch with a launch order (anet_qp_solve_ordered_dev: a re-solve of the same / a similar batch): longest first by
        # this batch's own step counts, and by the counts of a PERTURBED copy (durations x U(0.97, 1.03)) -- what a receding-horizon
        # re-solve has.  Beside the as-given number, never instead of it.
        def timed_order(order):
            aa.qp_solve_dev(s, st, tT, thp, ctx=ctx, launch_order=order)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(K):
                aa.qp_solve_dev(s, st, tT, thp, ctx=ctx, launch_order=order)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / K
        own = aa.launch_order_from_counts(r["iters"])
        tTp = tT * torch.from_numpy(np.random.default_rng(9).uniform(0.97, 1.03, size=T.shape)).to(device)
        rp = aa.qp_solve_dev(s, st, tTp, thp, ctx=ctx)
        out[key]["with_launch_order"] = {"own_counts_ms": timed_order(own),
                                         "perturbed_copy_counts_ms": timed_order(aa.launch_order_from_counts(rp["iters"]))}
        if key == "snap8":
            out[key]["gpu_obj"] = r["obj"].cpu().numpy()
            out[key]["gpu_status"] = r["status"].cpu().numpy()
    gpu_obj, gpu_status = out["snap8"].pop("gpu_obj", None), out["snap8"].pop("gpu_status", None)
    if cpu_baseline and extras:
        # dense Mehrotra interior point in numpy / LAPACK (oracle/qp_np.py) on the reference's assembled Q, A, b, G, h
        from oracle import qp_np, minco_np as onp
        s, N, M, state, T, hp = host
        D = 2 * s
        n = 3 * D * N

        def dense(b):
            st9 = np.zeros((9, 2))
            for ax in range(3):
                st9[3 * ax:3 * ax + 3, 0] = state[b, 0, ax]
                st9[3 * ax:3 * ax + 3, 1] = state[b, 1, ax]
            Q, A, bb, G1, h1, G2, h2 = onp.qp_assemble(s, st9, np.transpose(hp[b], (1, 2, 0)), np.full(N, M), T[b], 20, 4.0, 6.0)
            G = np.zeros((G1.shape[0] + G2.shape[0], n))
            r_ = 0
            for i in range(N):
                for _ in range(20):
                    G[r_:r_ + M, i * 3 * D:(i + 1) * 3 * D] = G1[r_:r_ + M]
                    r_ += M
            r2 = 0
            for i in range(N):
                for _ in range(20):
                    for j in range(3):
                        G[r_ + r2:r_ + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2[r2:r2 + 4]
                        r2 += 4
            hh = np.r_[h1, h2]
            keep = (np.abs(G).sum(axis=1) > 0) | (hh != 0)
            return Q, A, bb, G[keep], hh[keep]
        t_asm = t_sol = 0.0
        done, rel = 0, []
        nthreads = host_cores()
        try:                                    # LAPACK's threads = the cores this process may really use
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=nthreads)
        except Exception:
            pass
        t_all = time.perf_counter()
        for b in range(0, 4096, 64):
            t0 = time.perf_counter()
            Q, A, bb, G, h = dense(b)
            t1 = time.perf_counter()
            try:
                z, lam, nu, fo, it = qp_np.qp_ipm(Q, A, bb, G, h, tol=1e-8)
            except (ValueError, FloatingPointError, np.linalg.LinAlgError):   # an infeasible problem that blew up: timed all the same
                fo, it = float("nan"), 199
            t2 = time.perf_counter()
            t_asm += t1 - t0
            t_sol += t2 - t1
            done += 1
            if it < 199 and gpu_status[b] == 1:
                rel.append(abs(gpu_obj[b] - fo) / max(1.0, abs(fo)))
            if time.perf_counter() - t_all > 0.5 * cpu_seconds and done >= 3:
                break
        # (dense Mehrotra interior point in numpy / LAPACK, one problem at a time, on the matrices of qp_solver.hpp:119-296
        #  restated by oracle/minco_np.qp_assemble; solve time only; OSQP itself is not in the image)
        dense_numpy = {"value": done / t_sol, "cores": nthreads, "sample": f"{done} problems (every 64th), oracle/qp_np.py",
                       "gpu_vs_cpu_max_rel_obj_err": float(max(rel)) if rel else None, "compared": len(rel)}
        # like for like: the structured algorithm of k_qp_ipm (block-tridiagonal interior point in Hermite node coordinates,
        # Mehrotra) in scalar C, one problem per task on the host cores (oracle/qp_ipm_port.c)
        from oracle import cbind
        nprobe = 4 * nthreads
        idx = np.linspace(0, 4095, nprobe).astype(int)
        t0 = time.perf_counter()
        cbind.qp_ipm_batch(s, state[idx], T[idx], hp[idx], want_coeffs=False, nthreads=nthreads)
        rate = nprobe / max(time.perf_counter() - t0, 1e-6)
        ns = int(min(4096, max(nprobe, rate * 0.5 * cpu_seconds)))
        idx = np.linspace(0, 4095, ns).astype(int)
        t0 = time.perf_counter()
        po = cbind.qp_ipm_batch(s, state[idx], T[idx], hp[idx], want_coeffs=False, nthreads=nthreads)
        pdt = time.perf_counter() - t0
        both = (po["status"] == 1) & (gpu_status[idx] == 1)
        prel = np.abs(po["obj"] - gpu_obj[idx])[both] / np.maximum(1.0, np.abs(po["obj"][both]))
        out["cpu_baseline"] = {"value": ns / pdt, "unit": "QP solves/s", "cores": nthreads, "kind": "port",
                               "sample": f"{ns} of the 4096 snap8 problems (strided) to 1e-8, oracle/qp_ipm_port.c, {pdt:.1f} s",
                               "newton_steps_mean": float(po["iters"].mean()), "solved_frac": float((po["status"] >= 1).mean()),
                               "solved_to_1e-7_only_frac": float((po["status"] == 2).mean()),
                               "same_verdict_as_gpu_frac": float(((po["status"] >= 1) == (gpu_status[idx] == 1)).mean()),
                               "gpu_vs_cpu_max_rel_obj_err": float(prel.max()) if prel.size else None, "compared": int(both.sum()),
                               "dense_numpy": dense_numpy}
    return out


def run_config4(torch, aa, ctx, device, cpu_baseline, cpu_seconds):
    """BASELINE configs[3] (SURVEY 8(d) "config 4"): B = 4096 x 16-segment min-jerk, L-BFGS with lbfgs_parameter_t
    defaults (lbfgs.hpp:25-128) on waypoints and durations until every problem stops on its own; seed 2.  One step = the
    whole optimisation of the whole batch (one launch of k_lbfgs_minco_persistent), inputs resident in HBM."""
    import numpy as np
    from allocnet_amd.synth import corridor_problem
    B, s, c, N, M = 4096, 3, 3, 16, 16
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(2), B, N, c, M)
    pen = aa.make_penalty(poly_rows=M, **PEN)
    prm = aa.lbfgs_parameter_t()
    ld = aa.recommended_ld(B)
    cap = 40000                                     # a cap, not the stop: every problem must end with its own status
    th, tt, tw, tT, thp = (_to_bm(torch, x, B, ld, device) for x in (head, tail, wps, T, hp))
    # warm-up: one whole run (the same launches as the timed ones, so a profile of this leg holds only one launch shape)
    aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=cap, ctx=ctx)
    secs = []
    for rep in range(2):
        th, tt, tw, tT = (_to_bm(torch, x, B, ld, device) for x in (head, tail, wps, T))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        res = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=cap, ctx=ctx)
        e1.record()
        torch.cuda.synchronize()
        secs.append((time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3))
    dt, kdt = min(secs)
    st = res["status"].cpu().numpy(); it = res["iters"].cpu().numpy(); ev = res["evals"].cpu().numpy()
    cf = res["cost"].cpu().numpy()
    n = 3 * (N - 1) + N
    flops = float(ev.sum()) * cost_grad_flops(s, N, M, PEN["res"]) + float(it.sum()) * lbfgs_update_flops(n, prm.mem_size)
    ab = B * (config5_bytes(s, c, N, M) + 8)        # compulsory: problem data in, optimised waypoints / durations / cost out
    out = {"batch": B, "pieces": N, "order": s, "poly_rows": M, "res": PEN["res"], "seed": 2,
           "seconds": dt, "stream_seconds": kdt, "value": B / dt, "unit": "trajectories optimised to convergence/s",
           # (lbfgs_parameter_t defaults, lbfgs.hpp:25-128: mem 8, g_eps 1e-5, past 3, delta 1e-6)
           "max_evals_cap": cap, "iters_mean": float(it.mean()), "iters_max": int(it.max()),
           "evals_mean": float(ev.mean()), "evals_p50_p90_p99": [float(v) for v in np.percentile(ev, [50, 90, 99])],
           "evals_max": int(ev.max()), "evaluations_per_s": float(ev.sum()) / dt,
           "status_hist": {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
           "cost_final_mean": float(cf.mean()), "flops_per_evaluation": cost_grad_flops(s, N, M, PEN["res"]),
           # (traffic: two launches per run -- 1000 evaluations of every problem, then the unfinished ones: DESIGN.md 5)
           "roofline": fp64_roofline(flops, kdt, ab, "k_lbfgs_minco_persistent",
                                     traffic=pmc_leg_traffic("config4", [("k_lbfgs_minco_persistent<", None, 2)]))}
    # (run time = the LAST problem to stop, so the fraction mixes kernel quality with the spread of the evaluation counts)
    if cpu_baseline:
        from oracle import cbind
        nthreads = host_cores()
        ns = int(min(B, max(64, 64 * nthreads)))
        idx = np.linspace(0, B - 1, ns).astype(int)
        cp = cbind.lbfgs_default_param()
        t0 = time.perf_counter()
        o = cbind.lbfgs_minco_batch(s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], param=cp, nthreads=nthreads,
                                    **PEN_ORACLE)
        cdt = time.perf_counter() - t0
        rel = np.abs(o["cost"] - cf[idx]) / np.abs(o["cost"])
        out["cpu_baseline"] = {"value": ns / cdt, "unit": out["unit"], "cores": nthreads, "kind": "port",
                               # (oracle_lbfgs_optimize = lbfgs.hpp:434-717 restated, on oracle/minco_costgrad.c, one problem per task)
                               "sample": f"{ns} of the 4096 problems (strided), each to its own stop, {cdt:.1f} s",
                               "evals_mean": float(o["evals"].mean()), "evals_max": int(o["evals"].max()),
                               "evaluations_per_s": float(o["evals"].sum()) / cdt,
                               "status_hist": {str(k): int(v) for k, v in zip(*np.unique(o["status"], return_counts=True))},
                               "gpu_vs_cpu_final_cost_rel_median": float(np.median(rel)),
                               "gpu_vs_cpu_final_cost_rel_max": float(rel.max()),
                               "gpu_vs_cpu_same_eval_count_frac": float((o["evals"] == ev[idx]).mean())}
        # "rounding amplified over ~2500 iterations" as a measurement: 256 strided problems at growing iteration budgets
        i256 = np.linspace(0, B - 1, 256).astype(int)
        out["cpu_baseline"]["divergence"] = lbfgs_divergence_profile(aa, cbind, s, head[i256], tail[i256], wps[i256], T[i256],
                                                                     hp[i256], pen, nthreads, ctx=ctx)
    return out


DIVERGENCE_BUDGETS = (25, 50, 100, 200, 400)
DIVERGENCE_CONTROL_EPS = 1e-13      # the relative difference of the two OBJECTIVES (cost 1e-13, gradients 1.5e-13: config3 leg)


def _pair_stats(np, g, r):
    same = (g["status"] == r["status"]) & (g["iters"] == r["iters"]) & (g["evals"] == r["evals"])
    rel = np.abs(g["cost"] - r["cost"]) / np.abs(r["cost"])
    return same, float(same.mean()), (float(rel[same].max()) if same.any() else None), float(np.median(rel))


def lbfgs_divergence_profile(aa, cbind, s, head, tail, wps, T, hp, pen, nthreads, budgets=DIVERGENCE_BUDGETS, ctx=None,
                             eps=DIVERGENCE_CONTROL_EPS):
    """Where do the device run and the C restatement of lbfgs_optimize part ways -- and is that more than rounding?  The SAME
    problems on both sides under lbfgs_parameter_t defaults with max_iterations = each of `budgets` (lbfgs.hpp:690-695: the run
    stops with LBFGSERR_MAXIMUMITERATION unless it stopped on its own earlier).  Per budget: the fraction of problems with
    identical (status, iterations, evaluations), the largest relative cost difference among those, the median over all.
    CONTROL: the C restatement against ITSELF from a start point perturbed by `eps` relative (waypoints and durations times
    1 + eps N(0, 1)) -- how fast this objective amplifies a difference of the size the two objectives have anyway.  A defect
    of the device optimiser that only shows late would make the first profile fall off faster than the second.
    `aa` None: the control alone (no GPU: the CPU suite)."""
    import numpy as np
    B = head.shape[0]
    rng = np.random.default_rng(99)
    wps2 = wps * (1.0 + eps * rng.standard_normal(wps.shape))
    T2 = T * (1.0 + eps * rng.standard_normal(T.shape))
    first = np.zeros(B, dtype=np.int64)                        # 0 = never within the budgets tried
    out = {"problems": int(B), "budgets": list(budgets), "control_eps": eps}
    pairs = {"gpu_vs_cpu": ([], [], []), "cpu_vs_cpu_perturbed": ([], [], [])}
    for mi in budgets:
        r = cbind.lbfgs_minco_batch(s, head, tail, wps, T, hp, param=cbind.lbfgs_default_param(max_iterations=mi),
                                    nthreads=nthreads, **PEN_ORACLE)
        r2 = cbind.lbfgs_minco_batch(s, head, tail, wps2, T2, hp, param=cbind.lbfgs_default_param(max_iterations=mi),
                                     nthreads=nthreads, **PEN_ORACLE)
        todo = [("cpu_vs_cpu_perturbed", r2)]
        if aa is not None:
            g = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(max_iterations=mi),
                               max_evals=40 * mi, want_coeffs=False, ctx=ctx)
            todo.append(("gpu_vs_cpu", g))
        for key, other in todo:
            same, frac, worst, med = _pair_stats(np, other, r)
            for lst, v in zip(pairs[key], (frac, worst, med)):
                lst.append(v)
            if key == "gpu_vs_cpu":
                first[(first == 0) & ~same] = mi
    r3 = lambda v: None if v is None else float(f"{v:.3g}")
    # ("same": fraction with identical (status, iterations, evaluations); "max_rel_same": largest relative cost difference among
    #  those; "median_rel": median relative cost difference over all problems; one entry per budget)
    for key, (frac, worst, med) in pairs.items():
        if frac:
            out[key] = {"same": [r3(v) for v in frac], "max_rel_same": [r3(v) for v in worst], "median_rel": [r3(v) for v in med]}
    if aa is not None:
        div = first[first > 0]
        out["median_first_diverging_budget"] = float(np.median(div)) if div.size else None
    return out


def allgather_probe(torch, dist, og, device, use_dist, reps=10):
    """What the first real multi-GPU run needs to describe itself: how many ranks the RCCL group really has (an all-reduce of
    ones), the collective alone (not overlapped: `reps` blocking all-gathers of the step's payload between one event pair on
    the current stream, which waits for RCCL's stream), its payload, and how many were issued in the timed loop."""
    if not use_dist:
        return {"ranks_seen": 1, "allgather_ms": None, "allgather_bytes_per_rank": 0, "every": og.every, "issued": 0}
    one = torch.ones(1, device=device, dtype=torch.float64)
    dist.all_reduce(one)
    j = 0
    for _ in range(3):
        dist.all_gather_into_tensor(og.recv[j], og.send[j][:og.count])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.all_gather_into_tensor(og.recv[j], og.send[j][:og.count])
    e1.record()
    torch.cuda.synchronize()
    return {"ranks_seen": int(round(float(one.item()))), "allgather_ms": e0.elapsed_time(e1) / reps,
            "allgather_bytes_per_rank": 8 * og.count, "every": og.every, "issued": og.issued}


def run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, steps, warmup, total=32768, every=1):
    """BASELINE configs[4]: `total` x 8-segment min-snap, corridor + velocity / acceleration limit penalties (seed 3),
    contiguous shards over the ranks (remainder to the low ranks), one cost + gradient evaluation of the shard per step
    (k_minco_solve -> k_piece_grad -> k_minco_propagate) and the all-gather of the costs."""
    import numpy as np
    from allocnet_amd.synth import corridor_problem
    from allocnet_amd.distributed import shard_bounds
    s, c, N, M = 4, 3, 8, 16
    lo, hi = shard_bounds(total, world, rank)
    B = hi - lo
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(3), total, N, c, M)
    ld = aa.recommended_ld(max(B, 1))

    def to_bm(a):
        f = np.ascontiguousarray(a[lo:hi].reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=device, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(device)
        return t
    th, tt, tw, tT, thp = (to_bm(x) for x in (head, tail, wps, T, hp))
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=20, poly_rows=M)
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    m = -(-total // world)                      # padded shard length of the gather (ragged shards)
    # The all-gather of a step (8 B per trajectory: latency-bound) runs on RCCL's stream while the next
    # step's evaluation runs on the compute stream: two send / receive slots, a slot is reused only after its
    # collective has completed (allocnet_amd.distributed.OverlappedCostGather).
    from allocnet_amd.distributed import OverlappedCostGather
    og = OverlappedCostGather(m, world, device, every=every, enabled=use_dist)
    count = [0]

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP,
                               gradT=gT, ctx=ctx)
        if ev is not None:
            ev[1].record()
        if use_dist:
            j = og.acquire(count[0])
            og.send[j][:B].copy_(cost[:B])
            og.submit(count[0])
            count[0] += 1

    def sync():
        if use_dist:
            og.drain()
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    sync()
    elapsed, ev, retimed = time_steps(torch, dist, use_dist, device, steps, lambda i, e: step(e), sync)
    if use_dist:
        if og.last is not None and every == 1:
            if not torch.equal(og.recv[og.last][rank * m:rank * m + B], cost[:B]):
                raise SystemExit("config5: all-gather of costs returned wrong data")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    if os.environ.get("ANET_BENCH_DEBUG"):
        print("config5 per-step event ms:", ["%.3f" % a.elapsed_time(b) for a, b in ev], file=sys.stderr)
    ab = config5_bytes(s, c, N, M)
    launches = aa.minco_cost_grad_launches(s, N, B, penalty=pen, ctx=ctx)
    # (traffic: the committed PMC passes are of the one-rank shard, 32768 trajectories; another shard size has no entry)
    roof = fp64_roofline(B * cost_grad_flops(s, N, M, 20), kernel_ms * 1e-3, B * ab, cost_grad_kernel_label(aa, ctx, s, N, B, pen),
                         traffic=pmc_leg_traffic("config5", cost_grad_picks(launches)) if B == total == 32768 else None)
    roof.update(kernel_ms=kernel_ms, algorithmic_bytes_per_trajectory=ab, flops_per_evaluation=cost_grad_flops(s, N, M, 20))
    ms_step = elapsed / steps * 1e3
    ag = allgather_probe(torch, dist, og, device, use_dist)
    return {"value": total * steps / elapsed, "unit": "trajectory cost+gradient evaluations/s", "total_batch": total,
            "batch_this_rank": B, "ms_per_step": ms_step, "kernel_ms": kernel_ms, "steps": steps,
            # shard_ms: this rank's evaluation alone (events around it, mean over the timed steps); exposed_allgather_ms: what a
            # step costs beyond it -- the copy into the send slot, the collective's issue and whatever of it the next
            # evaluation does not hide (the max over ranks of the step time against THIS rank's compute)
            "shard_ms": kernel_ms, "exposed_allgather_ms": max(0.0, ms_step - kernel_ms), "launches_per_step": launches,
            "scaling": "strong", "pieces": N, "order": s, "poly_rows": M, "res": 20, "retimed_after_runtime_stall": retimed,
            "penalty_active_frac": float((cost[:B] > 0).double().mean().item()),
            "allgather": ag, "roofline": roof}


def time_steps(torch, dist, use_dist, device, steps, run_step, sync):
    """Time EXACTLY `steps` calls of run_step(i, (event0, event1)) followed by sync() with the host clock -> (seconds, events of
    the timed pass, the discarded pass's figures or None).  The HIP runtime blocks the host ONCE per process for 20-60 ms at some launch (absorb_runtime_stall
    below makes it happen early, and mostly succeeds); when it still lands in this loop -- the wall time is then more than twice
    what the per-step HIP events say plus 10 ms -- the pass is discarded and the same `steps` steps are timed again, once (the
    stall never comes twice), and the line says so AND carries the discarded pass's own figures (`retimed_after_runtime_stall`:
    null, or {ms_per_step, event_ms_max_step, event_ms_median_step} of the pass that was thrown away).  All ranks decide together."""
    def one_pass():
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            run_step(i, ev[i])
        sync()
        return time.perf_counter() - t0, ev
    elapsed, ev = one_pass()
    per = sorted(a.elapsed_time(b) for a, b in ev)
    stalled = elapsed > 2.0 * steps * per[len(per) // 2] * 1e-3 + 0.010
    if use_dist:
        f = torch.tensor([1.0 if stalled else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        stalled = bool(f.item() > 0.0)
    discarded = None
    if stalled:
        # what is thrown away is reported, so that a systematic slowness (a collective that blocks the host, say) cannot hide
        # behind the re-timing: its wall time per step, the largest single step by the events, the median step by the events
        discarded = {"ms_per_step": elapsed / steps * 1e3, "event_ms_max_step": per[-1], "event_ms_median_step": per[len(per) // 2]}
        elapsed, ev = one_pass()
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, ev, discarded


def absorb_runtime_stall(torch, aa, ctx, device, launches=1400):
    """The HIP runtime stalls a stream ONCE per process for 20-60 ms, some 10^3 launches into it (profiles/
    r04_launch_backlog_stall.txt: between two launches of a loop, whatever the kernels; never a second time).  Which leg of this
    file it lands in depends on how many launches came before -- this round it moved into the 50 timed steps of the configs[4] leg
    (one step of 40.4 ms among 49 of 0.155: "0.85 ms per step").  So it is made to happen HERE, before anything is timed: 1400
    launches of a 32768-trajectory solve (~20 us of GPU each against ~5 us of host: the host gets ~10^3 launches ahead, which is
    the other condition the profile names), not synchronised until the end."""
    # (c = 4, MINCO's boundary count: k_minco_solve<4, 8, true, 3>, an instantiation no timed leg launches, so that a profile of
    #  a leg holds the leg's kernels only)
    s, c, N, B = 4, 4, 8, 32768
    ld = aa.recommended_ld(B)
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, 7, device)
    energy = torch.empty(ld, device=device, dtype=torch.float64)
    call = aa.bind_minco_solve(head, tail, wps, T, s, c, N, B, coeffs=None, energy=energy, ctx=ctx)
    torch.cuda.synchronize()
    for _ in range(launches):
        call()
    torch.cuda.synchronize()


def synth_batch_minor(torch, B, ld, N, c, seed, device):
    """SURVEY.md 8(d) config 2 generator, produced directly on the device in batch-minor layout:
    random walk, step ~U(1,3) m in a random direction, z clamped to [0,5]; T ~U(0.5,2); rest-to-rest."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f64 = torch.float64
    d = torch.randn(N, 3, ld, generator=g, device=device, dtype=f64)
    d = d / d.norm(dim=1, keepdim=True)
    d = d * (1.0 + 2.0 * torch.rand(N, 1, ld, generator=g, device=device, dtype=f64))
    pts = torch.cat([torch.zeros(1, 3, ld, device=device, dtype=f64), torch.cumsum(d, dim=0)], dim=0)
    pts[:, 2] = (pts[:, 2] + 1.0).clamp(0.0, 5.0)
    head = torch.zeros(3, c, ld, device=device, dtype=f64)
    tail = torch.zeros(3, c, ld, device=device, dtype=f64)
    head[:, 0] = pts[0]
    tail[:, 0] = pts[N]
    wps = pts[1:N].contiguous().view((N - 1) * 3, ld)
    T = 0.5 + 1.5 * torch.rand(N, ld, generator=g, device=device, dtype=f64)
    return head.view(3 * c, ld), tail.view(3 * c, ld), wps, T


LINE_BUDGET = 7000      # the driver keeps a ~8 KB tail of stdout: a longer line loses its head (round 4: config3.b4096, config5)
# what goes first if a line is still over budget after rounding (least important first); each entry a key path
DROP_ORDER = (("host_api",), ("config1_b1024", "sampler"), ("qp_solve", "cpu_baseline", "dense_numpy"),
              ("config1_b1024", "streams8"), ("config1_b1024", "graph64x8"), ("config4", "cpu_baseline"),
              ("qp_solve", "jerk5", "with_launch_order"), ("qp_solve", "snap8", "with_launch_order"),
              ("config3", "saturating", "kernel_split_us"), ("qp_solve", "cpu_baseline"), ("config4", "status_hist"))


def _rounded(o):
    """floats to five significant digits (a bench line is read, not recomputed from)"""
    if isinstance(o, float):
        return float(f"{o:.5g}") if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _rounded(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_rounded(v) for v in o]
    return o


def finalize_line(out, budget=LINE_BUDGET):
    """The ONE JSON line: numbers rounded, compact separators, and -- should it still exceed the driver's tail -- the least
    important sub-objects dropped in DROP_ORDER, their names listed under "dropped" (prose lives in DESIGN.md section 7)."""
    out = _rounded(out)
    enc = lambda: json.dumps(out, separators=(",", ":"))
    line = enc()
    for path in DROP_ORDER:
        if len(line) <= budget:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
            if d is None:
                break
        if isinstance(d, dict) and path[-1] in d:
            del d[path[-1]]
            out.setdefault("dropped", []).append(".".join(path))
            line = enc()
    return line


def launch_plan(gpus, env, device_count):
    """How `python bench.py --gpus N` becomes N ranks.  Returns (plan, n, message):
      ("run", world, msg)   this process is a rank already (WORLD_SIZE is set: the driver's torch.distributed.run form) or the
                            job has one rank;
      ("spawn", n, msg)     --gpus N > 1 (or ANET_BENCH_SELF_LAUNCH=1) and no WORLD_SIZE: re-exec under torch.distributed.run
                            with n = min(N, visible GPUs) ranks, one per GPU."""
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        msg = None if world == gpus else f"--gpus {gpus} but WORLD_SIZE={world}: running (and reporting) {world} ranks"
        return "run", world, msg
    n = max(1, int(gpus))
    msg = None
    if n > device_count:
        msg = f"--gpus {gpus} but {device_count} GPU(s) visible: running (and reporting) {max(1, device_count)} rank(s)"
        n = max(1, device_count)
    if n > 1 or env.get("ANET_BENCH_SELF_LAUNCH") == "1":
        return "spawn", n, msg
    return "run", 1, msg


def self_launch_cmd(n, argv, port):
    """The torch.distributed.run command line of n ranks of this script: the caller's flags, `--gpus` replaced by the rank
    count that really runs."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a == "--gpus":
            skip = True
        elif not a.startswith("--gpus="):
            out.append(a)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n)] + out


def self_launch(n, argv):
    """Start n ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free
    port) and return its exit code.  Rank 0 prints the JSON line; the other ranks' stdout goes to /dev/null, so the line
    stays the last line of this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.pop("ANET_BENCH_SELF_LAUNCH", None)
    env.pop("MASTER_PORT", None)                          # torch.distributed.run exports the port it was given
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    sys.stdout.flush()
    return subprocess.run(self_launch_cmd(n, argv, port), env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1 << 20, help="trajectories per GPU per step")
    ap.add_argument("--pieces", type=int, default=8)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--bc", type=int, default=3, help="boundary derivatives fixed per end (3 = reference PVA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed main workload (used under rocprofv3 so kernel stats are not mixed)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--allgather-every", type=int, default=1,
                    help="issue the all-gather of the costs every k-th step only (N > 1; default 1 = every step, the north star)")
    ap.add_argument("--workload", choices=("solve", "config5", "config3", "config4", "qp"), default="solve",
                    help="solve: the headline (configs[1] problem at a saturating batch); config5: BASELINE configs[4]; config3 / "
                         "config4 / qp: that leg of the default line ALONE, as the line's main workload (one GPU; with --main-only: "
                         "no CPU baseline, no per-kernel split -- the form tools/profile_leg.sh runs under rocprofv3)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (allocnet_amd has no CPU fallback)")
    plan, n, msg = launch_plan(args.gpus, os.environ, torch.cuda.device_count())
    if msg:
        print("bench.py: " + msg, file=sys.stderr, flush=True)
    if plan == "spawn":
        raise SystemExit(self_launch(n, sys.argv[1:]))
    import allocnet_amd as aa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        # only rank 0 reports; keep other ranks' library banners (RCCL prints one) off the job's stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # ANET_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("ANET_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    s, c, N, B = args.order, args.bc, args.pieces, args.batch
    D = 2 * s
    ld = aa.recommended_ld(B)      # non-power-of-two row stride (HBM channel/bank spread)
    ctx = aa.Context(local_rank)
    absorb_runtime_stall(torch, aa, ctx, device)
    if args.workload in ("config3", "config4", "qp"):
        # one leg of the default line alone (same generator, same timing code: run_config3 / run_config4 / run_qp), so that a
        # rocprofv3 run holds the kernels of that leg only and profiles/<round>_<leg>_* can be checked against the default line
        if world != 1:
            raise SystemExit(f"--workload {args.workload} is a single-GPU leg")
        cpu = not args.no_cpu_baseline and not args.main_only
        if args.workload == "config3":
            leg = run_config3(torch, aa, ctx, device, cpu, 0.25 * args.cpu_seconds, split=not args.main_only)
            sub = leg["b4096"]
            head_ = {"value": sub["value"], "unit": leg["unit"], "ms_per_step": sub["ms_per_step"], "roofline": sub["roofline"],
                     "workload": "configs[2]: 4096 x 8-segment min-snap, corridor penalties + time-allocation gradients, one cost + "
                                 "gradient evaluation per step (the saturating batch beside it)", "steps": 200}
        elif args.workload == "config4":
            leg = run_config4(torch, aa, ctx, device, cpu, 0.5 * args.cpu_seconds)
            head_ = {"value": leg["value"], "unit": leg["unit"], "ms_per_step": leg["seconds"] * 1e3, "roofline": leg["roofline"],
                     "workload": "configs[3]: 4096 x 16-segment min-jerk, L-BFGS to each problem's own stop, one optimisation of the "
                                 "batch per step", "steps": 1}
        else:
            leg = run_qp(torch, aa, ctx, device, cpu, 0.5 * args.cpu_seconds, extras=not args.main_only)
            sub = leg["snap8"]
            head_ = {"value": sub["value"], "unit": leg["unit"], "ms_per_step": sub["ms_per_batch"], "roofline": sub["roofline"],
                     "workload": "the reference's online solve: 4096 x 8-segment min-snap inequality QPs (interior point), one batch "
                                 "per step (the planner's 5 jerk pieces beside it)", "steps": 5}
        out = {"metric": "MINCO trajectories solved/sec (8-seg min-snap)", "value": head_["value"], "unit": head_["unit"],
               "n_gpus": 1, "steps": head_["steps"], "warmup": args.warmup, "ms_per_step": head_["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": head_["workload"], "leg": args.workload, "main_only": bool(args.main_only)},
               "roofline": with_peak(head_["roofline"]), {"config3": "config3", "config4": "config4", "qp": "qp_solve"}[args.workload]: leg}
        if "cpu_baseline" in leg:
            out["cpu_baseline"] = leg["cpu_baseline"]
        print(finalize_line(out), flush=True)
        return
    if args.workload == "config5":
        c5 = run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, args.steps, args.warmup, every=args.allgather_every)
        if use_dist:
            dist.destroy_process_group()
        if rank != 0:
            return
        out = {"metric": "MINCO trajectories solved/sec (8-seg min-snap)", "value": c5["value"],
               "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": c5["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[4]: 32768 x 8-segment min-snap, corridor + dynamic-limit penalties, one "
                                      "cost + gradient evaluation per trajectory per step, sharded, costs all-gathered",
                          "global_batch": c5["total_batch"], "batch_this_rank": c5["batch_this_rank"],
                          "parallelism": f"dp{world}" + ("+allgather(costs)" if use_dist else "")},
               "roofline": with_peak(c5["roofline"]), "config5": c5}
        out["config"].update(ranks_seen=c5["allgather"]["ranks_seen"], allgather_ms=c5["allgather"]["allgather_ms"],
                             allgather_bytes_per_rank=c5["allgather"]["allgather_bytes_per_rank"],
                             allgather_every=c5["allgather"]["every"])
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(finalize_line(out), flush=True)
        return
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, seed=rank, device=device)
    coeffs = torch.empty(N * 3 * D, ld, device=device, dtype=torch.float64)
    # two cost buffers: the all-gather of step k (RCCL stream) overlaps the solve of step k+1
    from allocnet_amd.distributed import OverlappedCostGather
    og = OverlappedCostGather(B, world, device, alloc=ld, every=args.allgather_every, enabled=use_dist)
    energy = og.send[0]

    def step(i, ev=None):
        j = og.acquire(i)                         # waits for the collective that read this slot two steps ago
        if ev is not None:
            ev[0].record()
        aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=coeffs, energy=og.send[j], ctx=ctx)
        if ev is not None:
            ev[1].record()
        og.submit(i)

    def sync():
        og.drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    elapsed, ev, retimed = time_steps(torch, dist, use_dist, device, args.steps, step, sync)
    if use_dist:
        # the gathered costs of the last gather issued must be every rank's costs in rank order
        if og.last is not None and not torch.equal(og.recv[og.last][rank * B:(rank + 1) * B], og.send[og.last][:B]):
            raise SystemExit("all-gather of costs returned wrong data")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    ag = allgather_probe(torch, dist, og, device, use_dist)

    # every rank takes part in the sharded cost + gradient evaluation (BASELINE configs[4])
    c5 = None
    if not args.main_only:
        del coeffs
        c5 = run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, steps=50, warmup=5, every=args.allgather_every)
        coeffs = torch.empty(N * 3 * D, ld, device=device, dtype=torch.float64)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    total = world * B * args.steps
    value = total / elapsed
    abytes = algorithmic_bytes(s, c, N)
    achieved = B * abytes / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "MINCO trajectories solved/sec (8-seg min-snap)",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[1] problem ({N}-segment order-{s} MINCO, random-walk waypoints, "
                               f"energy-only, PVA boundary c={c}) at saturating batch {B}/GPU",
                   "batch_per_gpu": B, "row_stride_ld": ld, "pieces": N, "order": s, "global_batch": world * B,
                   "parallelism": f"dp{world}" + ("+allgather(costs)" if use_dist else ""),
                   "ranks_seen": ag["ranks_seen"], "allgather_ms": ag["allgather_ms"],
                   "allgather_bytes_per_rank": ag["allgather_bytes_per_rank"], "allgather_every": ag["every"],
                   "allgathers_in_timed_loop_and_warmup": ag["issued"], "retimed_after_runtime_stall": retimed},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_minco_solve", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_trajectory": abytes,
                     "frac_of_measured_copy_6290": achieved / 6290.0},
    }

    out["fp64_peak_tflops"] = FP64_PEAK_TFLOPS        # what the sub-legs' "frac" are fractions of
    out["roofline"]["traffic"] = pmc_traffic_bytes(B, N, s)
    # (the committed rocprofv3 PMC pass of this launch shape, WRITE_SIZE + 2 x FETCH_SIZE: a constant of the tree, not counted
    #  during this run)
    out["roofline"]["traffic_source"] = "profiles/*_pmc.json"
    if c5 is not None:
        # (one rank, no process group: the collective's description says nothing -- ranks_seen etc. are in "config")
        out["config5"] = c5 if use_dist else dict(c5, allgather=None)
    if world == 1 and not args.main_only:
        # BASELINE configs[2] and configs[3] are single-GPU configurations: the north-star loop (cost + gradient, L-BFGS)
        del coeffs
        out["config3"] = run_config3(torch, aa, ctx, device, not args.no_cpu_baseline, 0.25 * args.cpu_seconds)
        out["config4"] = run_config4(torch, aa, ctx, device, not args.no_cpu_baseline, 0.5 * args.cpu_seconds)
        out["qp_solve"] = run_qp(torch, aa, ctx, device, not args.no_cpu_baseline, 0.5 * args.cpu_seconds)
        coeffs = torch.empty(N * 3 * D, ld, device=device, dtype=torch.float64)
        # literal configs[1]: B = 1024 (launch-latency bound; reported, not the headline)
        b2 = 1024
        K2 = 200

        def solve_b1024_wrapper():
            aa.minco_solve_dev(head, tail, wps, T, s, c, N, b2, coeffs=coeffs, energy=energy, ctx=ctx)
        # The launch with its arguments checked and converted once (allocnet_amd.bind_minco_solve: the C entry point behind a ctypes
        # trampoline -- what a C++ caller pays); through minco_solve_dev every launch also pays ~5 us of Python (tensor checks,
        # pointer conversions, stream lookup), which was the bound of this leg: reported beside it as wrapper_ms_per_step.
        solve_b1024 = aa.bind_minco_solve(head, tail, wps, T, s, c, N, b2, coeffs=coeffs, energy=energy, ctx=ctx)
        # (the GPU has idled through seconds of CPU-baseline work: >= 10 ms of warm-up, five repetitions, median -- as the config3 leg)
        runs = timed_reps(torch, solve_b1024, K2, reps=5, warm_ms=10.0)
        h_ms, s_ms = sorted(r[0] for r in runs), sorted(r[1] for r in runs)
        w_ms = sorted(r[0] for r in timed_reps(torch, solve_b1024_wrapper, K2, reps=5, warm_ms=10.0))
        out["config1_b1024"] = {"batch": b2, "value": b2 / (h_ms[2] * 1e-3), "ms_per_step": h_ms[2],
                                "stream_ms_per_step": s_ms[2], "stream_ms_min_median_max": [s_ms[0], s_ms[2], s_ms[-1]],
                                "wrapper_ms_per_step": w_ms[2],
                                "hbm_frac": b2 * abytes / (s_ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # The same 1024-trajectory launches from EIGHT streams (a sampler of time allocations issues many independent
        # batches): a launch is 49 waves on 1024 SIMDs, so independent batches overlap until the chip fills.
        ns = 8
        streams = [torch.cuda.Stream(device=device) for _ in range(ns)]
        ld2 = aa.recommended_ld(b2)
        outs = [(torch.empty(N * 3 * D, ld2, device=device, dtype=torch.float64), torch.empty(ld2, device=device, dtype=torch.float64))
                for _ in range(ns)]
        ins = [x[:, :ld2].contiguous() for x in (head, tail, wps, T)]
        torch.cuda.synchronize()

        def round_robin():
            for j in range(ns):
                aa.minco_solve_dev(ins[0], ins[1], ins[2], ins[3], s, c, N, b2, coeffs=outs[j][0], energy=outs[j][1],
                                   stream=streams[j].cuda_stream, ctx=ctx)
        dt8s = []
        for rep in range(6):                      # first pass = warm-up; then five repetitions, median (host clock: 8 streams)
            t0 = time.perf_counter()
            for k in range(K2 if rep else 25):
                round_robin()
            torch.cuda.synchronize()
            if rep:
                dt8s.append(time.perf_counter() - t0)
        dt8 = sorted(dt8s)[2]
        # ... and as ONE hipGraph of 64 such launches on 8 parallel chains, replayed: what is left of the launch side
        try:
            cap = torch.cuda.Stream(device=device)
            graph = torch.cuda.CUDAGraph()
            NG = 64
            with torch.cuda.graph(graph, stream=cap):
                cur = torch.cuda.current_stream(device)
                for st in streams:
                    st.wait_stream(cur)
                for k in range(NG):
                    j = k % ns
                    aa.minco_solve_dev(ins[0], ins[1], ins[2], ins[3], s, c, N, b2, coeffs=outs[j][0], energy=outs[j][1],
                                       stream=streams[j].cuda_stream, ctx=ctx)
                for st in streams:
                    cur.wait_stream(st)
            reps = 50
            # (64 launches of 1024 trajectories on 8 parallel chains captured in one hipGraph, replayed)
            gr = timed_reps(torch, graph.replay, reps, reps=5, warm_ms=10.0)
            dtg = sorted(r[0] for r in gr)[2] * 1e-3 * reps
            out["config1_b1024"]["graph64x8"] = {"value": b2 * NG * reps / dtg, "ms_per_launch": dtg / (NG * reps) * 1e3,
                                                 "hbm_frac": b2 * abytes / (dtg / (NG * reps)) / 1e9 / HBM_PEAK_GBS}
        except Exception as exc:      # (graph capture is an extra, never the headline)
            out["config1_b1024"]["graph64x8"] = {"error": str(exc)[:200]}
        out["config1_b1024"]["streams8"] = {"ms_per_launch": dt8 / (K2 * ns) * 1e3}    # (the line has a budget: the rate follows)
        # ... and what a sampler of time allocations should call instead of K launches of 1024 replicated problems: ONE
        # launch over K candidate duration vectors of few problems (anet_minco_sample_costs_dev: problem data per problem,
        # durations per sample, only the cost comes back)
        try:
            Ks = 1 << 20
            lds_ = aa.recommended_ld(Ks)
            g2 = torch.Generator(device=device); g2.manual_seed(7)
            Ts = 0.5 + 1.5 * torch.rand(N, lds_, generator=g2, device=device, dtype=torch.float64)
            smp = {}
            for label, P_ in (("one_problem", 1), ("1024_problems", 1024)):
                ph, pt, pw = (x[:, :max(P_, 8)].contiguous() for x in (head, tail, wps))
                cst = aa.minco_sample_costs_dev(ph, pt, pw, Ts, s, c, N, P_, Ks // P_, rho=1.0, ctx=ctx)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    aa.minco_sample_costs_dev(ph, pt, pw, Ts, s, c, N, P_, Ks // P_, rho=1.0, cost=cst, ctx=ctx)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                smp[label] = {"ms_per_launch": ms, "value": Ks / (ms * 1e-3),
                              "fp64_frac": Ks * 4100.0 / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
            # (time-allocation samples/s in one launch of 2^20 samples; 8 (N + 1) bytes per sample: bound by its FP64 work, ~4.1 kFLOP per sample)
            out["config1_b1024"]["sampler"] = smp
        except Exception as exc:
            out["config1_b1024"]["sampler"] = {"error": str(exc)[:200]}
        # PCIe-inclusive host API
        import numpy as np
        from allocnet_amd.synth import random_problem
        rng = np.random.default_rng(0)
        bh = 1 << 16
        h_head, h_tail, h_wps, h_T = random_problem(rng, bh, N, c, rest=True)
        aa.minco_solve(h_head, h_tail, h_wps, h_T, s, ctx=ctx)
        t0 = time.perf_counter()
        for _ in range(3):
            aa.minco_solve(h_head, h_tail, h_wps, h_T, s, ctx=ctx)
        out["host_api"] = {"batch": bh, "value": 3 * bh / (time.perf_counter() - t0)}     # PCIe + layout transposes included

        if not args.no_cpu_baseline:
            from oracle import cbind
            nthreads = host_cores()

            def time_cpu(fn, seconds):
                """rate of `fn`: a sample of at most 2 M trajectories (sized from two probes), solved repeatedly for about
                `seconds` of host time"""
                rate = 0.0
                for probe in (2000, 100000):
                    hp, tp, wp, Tp = random_problem(rng, probe, N, c, rest=True)
                    fn(s, hp, tp, wp, Tp, nthreads=nthreads)                  # (thread start-up, page faults)
                    t0 = time.perf_counter()
                    fn(s, hp, tp, wp, Tp, nthreads=nthreads)
                    rate = probe / max(time.perf_counter() - t0, 1e-6)
                    if rate * seconds < 50000:
                        break
                n_cpu = int(min(max(rate * seconds, 4000), 1_000_000))
                reps = max(1, int(rate * seconds / n_cpu))
                hp, tp, wp, Tp = random_problem(rng, n_cpu, N, c, rest=True)
                outb = (np.zeros((n_cpu, N, 3, 2 * s)), np.zeros(n_cpu))       # outputs allocated (and touched) once
                co_cpu, en_cpu = fn(s, hp, tp, wp, Tp, nthreads=nthreads, out=outb)
                t0 = time.perf_counter()
                for _ in range(reps):
                    co_cpu, en_cpu = fn(s, hp, tp, wp, Tp, nthreads=nthreads, out=outb)
                return n_cpu * reps / (time.perf_counter() - t0), n_cpu * reps, (hp, tp, wp, Tp, co_cpu)
            # (1) the classic banded-LU formulation (the oracle's algorithm)
            rate_lu, n_lu, (hp, tp, wp, Tp, co_cpu) = time_cpu(cbind.minco_solve_batch, 0.5 * args.cpu_seconds)
            # the same sample through the GPU path must agree with the CPU oracle
            co_gpu, en_gpu = aa.minco_solve(hp[:4096], tp[:4096], wp[:4096], Tp[:4096], s, ctx=ctx)
            err = float(np.abs(co_gpu - co_cpu[:4096]).max() / np.abs(co_cpu[:4096]).max())
            # (2) like for like: the kernels' own reduced algorithm compiled for the host cores
            rate_red, n_red, _ = time_cpu(cbind.cpu_reduced_solve_batch, 0.5 * args.cpu_seconds)
            out["cpu_baseline"] = {"value": rate_red, "unit": "trajectories/s", "cores": nthreads, "kind": "port",
                                   # (port = the kernels' own reduced Hermite / block-tridiagonal algorithm compiled for the host)
                                   "sample": f"{n_red} trajectories of the same workload, oracle/minco_cpu_reduced.cpp, scalar FP64",
                                   "classic_banded_lu": {"value": rate_lu, "sample": f"{n_lu} trajectories, oracle/minco_oracle.c"},
                                   "gpu_vs_cpu_max_rel_coeff_err": err}
    if use_dist:
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio, which is only flushed at exit: push it out now so
    # that the JSON line is the LAST line on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    print(finalize_line(out), flush=True)


if __name__ == "__main__":
    main()
