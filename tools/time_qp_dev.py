#!/usr/bin/env python3
"""Device-side time of the batched QP solve (anet_qp_solve_dev, interior point), inputs resident, HIP events:
    gpurun -- 'python tools/time_qp_dev.py'          [s,N,M,B ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
ctx = aa.Context(0); dev = torch.device("cuda", 0)
SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 8, 16, 4096), (3, 5, 16, 4096), (3, 16, 16, 1024), (4, 5, 16, 4096)]  # s,N,M,B
for (s, N, M, B) in SHAPES:
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    state = np.stack([head, tail], axis=1)[..., :3]                      # (B,2,3,3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st, tT, thp = t(state), t(T * 1.5), t(hp)
    r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("s", s, "N", N, "B", B, "ms", ["%.2f" % x for x in ts], "solved %.3f" % float((r["status"] == 1).double().mean()),
          "iters mean %.1f max %d" % (float(r["iters"].double().mean()), int(r["iters"].max())), flush=True)
