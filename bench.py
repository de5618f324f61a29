#!/usr/bin/env python3
"""Benchmark of the MINCO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_PER_GPU] [--pieces 8] [--order 4]

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


def config5_bytes(s, c, N, M):
    """compulsory bytes of one cost + gradient evaluation per trajectory: the energy-only figure plus
    the corridor rows 8*4*sum(M_i) and the gradient outputs 8*(N + 3(N-1)) (6248 B at N = 8, s = 4, M = 16)."""
    return algorithmic_bytes(s, c, N) + 8 * 4 * N * M + 8 * (N + 3 * (N - 1))


FP64_PEAK_TFLOPS = 74.5        # measured FMA peak, tools/micro/fp64_peak.hip (profiles/r02_fp64_peak.txt); nominal 78.6


def run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, steps, warmup, total=32768):
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
    # step's evaluation runs on the compute stream: two send / receive buffers, a buffer is reused only after its
    # collective has completed (work.wait() orders the compute stream behind it on the device, not the host).
    send = [torch.zeros(m, device=device, dtype=torch.float64) for _ in range(2)]
    gathered = [torch.empty(world * m, device=device, dtype=torch.float64) for _ in range(2)] if use_dist else None
    works = [None, None]
    count = [0]

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP,
                               gradT=gT, ctx=ctx)
        if ev is not None:
            ev[1].record()
        if use_dist:
            b = count[0] & 1
            count[0] += 1
            if works[b] is not None:
                works[b].wait()
            send[b][:B].copy_(cost[:B])
            works[b] = dist.all_gather_into_tensor(gathered[b], send[b], async_op=True)

    def sync():
        if use_dist:
            for w in works:
                if w is not None:
                    w.wait()
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(ev[i])
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        last = gathered[(count[0] - 1) & 1]
        if not torch.equal(last[rank * m:rank * m + B], cost[:B]):
            raise SystemExit("config5: all-gather of costs returned wrong data")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    ab = config5_bytes(s, c, N, M)
    achieved = B * ab / (kernel_ms * 1e-3) / 1e9
    return {"value": total * steps / elapsed, "unit": "trajectory cost+gradient evaluations/s", "total_batch": total,
            "batch_this_rank": B, "ms_per_step": elapsed / steps * 1e3, "kernel_ms": kernel_ms, "steps": steps,
            "scaling": "strong", "pieces": N, "order": s, "poly_rows": M, "res": 20,
            "penalty_active_frac": float((cost[:B] > 0).double().mean().item()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "kernel": "k_piece_grad (+ k_minco_solve, k_minco_propagate)",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_trajectory": ab,
                         "note": "the evaluation is bound by FP64 issue, not by HBM: k_piece_grad runs at 63 % of the "
                                 "measured FP64 FMA peak (profiles/r02_cost_grad_counters.txt)",
                         "fp64_peak_tflops_measured": FP64_PEAK_TFLOPS}}


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
    ap.add_argument("--workload", choices=("solve", "config5"), default="solve",
                    help="solve: the headline (configs[1] problem at a saturating batch); config5: BASELINE configs[4]")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import allocnet_amd as aa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (allocnet_amd has no CPU fallback)")
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
    if args.workload == "config5":
        c5 = run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, args.steps, args.warmup)
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
               "roofline": dict(c5["roofline"], traffic=None), "config5": c5}
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
        return
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, seed=rank, device=device)
    coeffs = torch.empty(N * 3 * D, ld, device=device, dtype=torch.float64)
    # two cost buffers: the all-gather of step k (RCCL stream) overlaps the solve of step k+1
    energies = [torch.empty(ld, device=device, dtype=torch.float64) for _ in range(2)]
    gathered = [torch.empty(world * B, device=device, dtype=torch.float64) for _ in range(2)] if use_dist else None
    works = [None, None]
    energy = energies[0]

    def step(i, ev=None):
        j = i % 2
        if works[j] is not None:
            works[j].wait()                       # the collective that read energies[j] two steps ago
            works[j] = None
        if ev is not None:
            ev[0].record()
        aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=coeffs, energy=energies[j], ctx=ctx)
        if ev is not None:
            ev[1].record()
        if use_dist:
            works[j] = dist.all_gather_into_tensor(gathered[j], energies[j][:B], async_op=True)

    def drain():
        for j in range(2):
            if works[j] is not None:
                works[j].wait()
                works[j] = None

    def sync():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i])
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the gathered costs of the last step must be every rank's costs in rank order
        chk = gathered[(args.steps - 1) % 2][rank * B:(rank + 1) * B]
        if not torch.equal(chk, energies[(args.steps - 1) % 2][:B]):
            raise SystemExit("all-gather of costs returned wrong data")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    # every rank takes part in the sharded cost + gradient evaluation (BASELINE configs[4])
    c5 = None
    if not args.main_only:
        del coeffs
        c5 = run_config5(torch, dist, aa, ctx, device, world, rank, use_dist, steps=50, warmup=5)
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
                   "parallelism": f"dp{world}" + ("+allgather(costs)" if use_dist else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_minco_solve", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_trajectory": abytes,
                     "frac_of_measured_copy_6290": achieved / 6290.0},
    }

    out["roofline"]["traffic"] = pmc_traffic_bytes(B, N, s)
    if c5 is not None:
        out["config5"] = c5
    if world == 1 and not args.main_only:
        # literal configs[1]: B = 1024 (launch-latency bound; reported, not the headline)
        b2 = 1024
        K2 = 200
        for _ in range(20):
            aa.minco_solve_dev(head, tail, wps, T, s, c, N, b2, coeffs=coeffs, energy=energy, ctx=ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K2):
            aa.minco_solve_dev(head, tail, wps, T, s, c, N, b2, coeffs=coeffs, energy=energy, ctx=ctx)
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["config1_b1024"] = {"batch": b2, "value": b2 * K2 / dt, "ms_per_step": dt / K2 * 1e3,
                                "stream_ms_per_step": e0.elapsed_time(e1) / K2,
                                "hbm_frac": b2 * abytes / (e0.elapsed_time(e1) / K2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "launch-latency bound: 2 MB per launch"}
        # The same 1024-trajectory launches from EIGHT streams (a sampler of time allocations issues many independent
        # batches): a launch is 49 waves on 1024 SIMDs, so independent batches overlap until the chip fills.
        ns = 8
        streams = [torch.cuda.Stream(device=device) for _ in range(ns)]
        ld2 = aa.recommended_ld(b2)
        outs = [(torch.empty(N * 3 * D, ld2, device=device, dtype=torch.float64), torch.empty(ld2, device=device, dtype=torch.float64))
                for _ in range(ns)]
        ins = [x[:, :ld2].contiguous() for x in (head, tail, wps, T)]
        torch.cuda.synchronize()
        for rep in range(2):                      # first pass = warm-up
            t0 = time.perf_counter()
            for k in range(K2 * ns if rep else ns * 4):
                j = k % ns
                aa.minco_solve_dev(ins[0], ins[1], ins[2], ins[3], s, c, N, b2, coeffs=outs[j][0], energy=outs[j][1],
                                   stream=streams[j].cuda_stream, ctx=ctx)
            torch.cuda.synchronize()
            dt8 = time.perf_counter() - t0
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
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            torch.cuda.synchronize()
            dtg = time.perf_counter() - t0
            out["config1_b1024"]["graph64x8"] = {"value": b2 * NG * reps / dtg, "ms_per_launch": dtg / (NG * reps) * 1e3,
                                                 "hbm_frac": b2 * abytes / (dtg / (NG * reps)) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "64 launches of 1024 trajectories on 8 parallel chains captured in one hipGraph"}
        except Exception as exc:      # (graph capture is an extra, never the headline)
            out["config1_b1024"]["graph64x8"] = {"error": str(exc)[:200]}
        out["config1_b1024"]["streams8"] = {"value": b2 * K2 * ns / dt8, "ms_per_launch": dt8 / (K2 * ns) * 1e3,
                                            "hbm_frac": b2 * abytes / (dt8 / (K2 * ns)) / 1e9 / HBM_PEAK_GBS,
                                            "note": "1024-trajectory launches round-robin on 8 streams"}
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
        out["host_api"] = {"batch": bh, "value": 3 * bh / (time.perf_counter() - t0),
                           "note": "host pointers in/out, PCIe + layout transposes included"}

        if not args.no_cpu_baseline:
            from oracle import cbind
            nthreads = os.cpu_count() or 1

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
                                   "sample": f"{n_red} trajectories of the same workload, the kernels' own reduced (Hermite / "
                                             f"block-tridiagonal) algorithm compiled for the host (oracle/minco_cpu_reduced.cpp), "
                                             f"scalar FP64, {nthreads} threads",
                                   "classic_banded_lu": {"value": rate_lu, "sample": f"{n_lu} trajectories, oracle/minco_oracle.c, "
                                                                                      f"{nthreads} threads"},
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
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
