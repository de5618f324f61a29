#!/usr/bin/env python3
"""Timing of the batched QP solve (OSQP replacement) on config-3-like problems."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import allocnet_amd as aa
from tools.bench_configs import synth
ctx = aa.Context(0)
out = {}
for (s, N, M, B) in [(4, 8, 16, 4096), (3, 5, 16, 4096), (4, 5, 16, 1)]:
    rng = np.random.default_rng(1)
    head, tail, wps, T, hp = synth(rng, B, N, 3, M)
    T = T * 1.5
    ini = head; fin = tail
    aa.qp_solve(s, ini[:8], fin[:8], hp[:8], T[:8], ctx=ctx)
    t0 = time.perf_counter()
    r = aa.qp_solve(s, ini, fin, hp, T, res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
    dt = time.perf_counter() - t0
    st = r["status"]; it = r["iters"]
    out[f"s{s}_N{N}_M{M}_B{B}"] = {"seconds": dt, "solves_per_s": B / dt, "solved_frac": float((st == 1).mean()),
                                   "iters_mean": float(it.mean()), "iters_max": int(it.max()),
                                   "obj_median": float(np.median(r["obj"]))}
print(json.dumps(out, indent=1))
