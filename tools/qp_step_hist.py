#!/usr/bin/env python3
"""Newton-step histogram of the bench's QP batches (4096 problems, seed 1, durations x 1.5) by verdict, and what the batch
would cost if its steps were spread evenly: sum(steps) / resident workgroups against the measured time (the tail a few
long problems leave).   gpurun -- 'python tools/qp_step_hist.py'"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem

ctx = aa.Context(0)
dev = torch.device("cuda:0")
for key, s, N, B in (("snap8", 4, 8, 4096), ("jerk5", 3, 5, 4096)):
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, 16)
    state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st, tT, thp = t(state), t(T * 1.5), t(hp)
    r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r = aa.qp_solve_dev(s, st, tT, thp, ctx=ctx)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    it, stt = r["iters"].cpu().numpy(), r["status"].cpu().numpy()
    print(f"{key}: {ms:.3f} ms; steps total {it.sum()} mean {it.mean():.2f}")
    for v in np.unique(stt):
        sel = stt == v
        h = np.bincount(it[sel])
        print(f"  status {v}: {sel.sum()} problems, steps mean {it[sel].mean():.1f} max {it[sel].max()}, total {it[sel].sum()} ({100 * it[sel].sum() / it.sum():.1f} % of all steps)")
        print("    histogram (steps: count):", {k: int(c) for k, c in enumerate(h) if c})
    # where in the launch order the long ones sit
    order = np.argsort(-it)[:12]
    print("  longest:", [(int(b), int(it[b]), int(stt[b])) for b in order])
