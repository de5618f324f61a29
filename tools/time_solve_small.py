#!/usr/bin/env python3
"""Per-launch time of the coefficient solve at small batches, stream launches and ONE hipGraph of 64 launches on one chain:
    gpurun -- 'python tools/time_solve_small.py 4,3,8,1024 3,3,16,1024; ANET_AXIS_TWO_MAX_BATCH=0 python tools/time_solve_small.py ...'
(order, boundary count, pieces, batch).  A/B of the launch shapes: ANET_AXIS_TWO_MAX_BATCH, ANET_AXIS_MAX_BATCH."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from tools.bench_configs import synth, to_bm
dev = torch.device("cuda", 0); ctx = aa.Context(0)
for arg in sys.argv[1:] or ["4,3,8,1024"]:
    s, c, N, B = (int(x) for x in arg.split(","))
    ld = aa.recommended_ld(B)
    head, tail, wps, T, _ = synth(np.random.default_rng(1), B, N, c, 8)
    th, tt, tw, tT = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T))
    coeffs = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64); energy = torch.empty(ld, device=dev, dtype=torch.float64)
    run = lambda stream=None: aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=coeffs, energy=energy, ctx=ctx, **({"stream": stream} if stream else {}))
    for _ in range(200): run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(500): run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 500 * 1e6)
    # the C entry point with its arguments converted once (what a C++ caller pays per launch, plus the ctypes trampoline)
    import ctypes
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    args = (ctx.handle, s, c, N, B, ld, vp(th), vp(tt), vp(tw), vp(tT), vp(coeffs), vp(energy), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    fn = ctx.lib.anet_minco_solve_dev
    tb = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(500): fn(*args)
        torch.cuda.synchronize()
        tb.append((time.perf_counter() - t0) / 500 * 1e6)
    print("   bound arguments: stream us/launch %.2f" % sorted(tb)[2], flush=True)
    cap = torch.cuda.Stream(device=dev); graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        for _ in range(64): run(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(20): graph.replay()
    torch.cuda.synchronize()
    tg = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50): graph.replay()
        torch.cuda.synchronize()
        tg.append((time.perf_counter() - t0) / (50 * 64) * 1e6)
    print("s", s, "c", c, "N", N, "B", B, "stream us/launch %.2f" % sorted(ts)[2], "| graph (one chain) us/launch %.2f" % sorted(tg)[2], flush=True)
