#!/usr/bin/env python3
"""Timing of the batched QP solve (OSQP replacement) on config-3-like problems, both methods:
the OSQP-faithful ADMM kernel (defaults) and the interior-point kernel (settings.method = 1).
Host-pointer API: PCIe and the device workspace allocation of the first large call are inside the times."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import allocnet_amd as aa  # noqa: E402
from allocnet_amd.synth import corridor_problem  # noqa: E402

ctx = aa.Context(0)
out = {}
for (s, N, M, B) in [(4, 8, 16, 4096), (3, 5, 16, 4096), (3, 16, 16, 1024), (4, 5, 16, 1)]:
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    T = T * 1.5
    for name, st in (("admm", aa.qp_settings(method=aa.qp.QP_METHOD_ADMM)), ("ipm", aa.qp_settings(method=aa.qp.QP_METHOD_INTERIOR_POINT))):
        if name == "admm" and N == 16:
            continue                      # the 16-piece jerk factor of the ADMM kernel needs more LDS than it has at M = 16
        kw = dict(res=20, max_vel=4.0, max_acc=6.0, settings=st, ctx=ctx)
        aa.qp_solve(s, head, tail, hp, T, **kw)            # warm-up: allocates the workspace for this size
        t0 = time.perf_counter()
        r = aa.qp_solve(s, head, tail, hp, T, **kw)
        dt = time.perf_counter() - t0
        out[f"{name}_s{s}_N{N}_M{M}_B{B}"] = {
            "seconds": dt, "solves_per_s": B / dt, "solved_frac": float((r["status"] == 1).mean()),
            "iters_mean": float(r["iters"].mean()), "iters_max": int(r["iters"].max()), "obj_median": float(np.median(r["obj"]))}
print(json.dumps(out, indent=1))
