#!/usr/bin/env python3
"""Per-section cycle counts of one interior-point QP solve (k_qp_ipm), problem 0 of a batch:
    ANET_BUILD_FLAGS=-DANET_IPM_PROF python -m allocnet_amd.build --force
    gpurun -- 'python tools/ipm_prof.py'         (rebuild without the flag afterwards)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
ctx = aa.Context(0)
SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 5, 16, 1), (4, 8, 16, 1), (4, 8, 16, 4096)]  # s,N,M,B
for (s, N, M, B) in SHAPES:
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    sys.stderr.write(f"--- s={s} N={N} M={M} B={B}\n"); sys.stderr.flush()
    r = aa.qp_solve(s, head, tail, hp, T * 1.5, res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
    r = aa.qp_solve(s, head, tail, hp, T * 1.5, res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
    sys.stderr.write(f"iters of problem 0: {r['iters'][0]} status {r['status'][0]}\n")
