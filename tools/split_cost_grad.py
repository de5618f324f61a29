#!/usr/bin/env python3
"""Per-kernel device time of one cost + gradient evaluation (bench.cost_grad_kernel_split) and the fused evaluation's time
(bench.timed_reps) for given shapes:  python tools/split_cost_grad.py 4,3,8,4096 [...]   (order, boundary count, pieces, batch).
Launch shapes are chosen by the library (override: ANET_PIECE_SHAPE, ANET_PIECE_SW_MAX_PAIRS, ANET_AXIS_MAX_BATCH); ANET_RES = samples per piece (default 20)."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import allocnet_amd as aa
import bench
from allocnet_amd.synth import corridor_problem
dev = torch.device("cuda", 0); ctx = aa.Context(0)
for arg in sys.argv[1:] or ["4,3,8,4096"]:
    s, c, N, B = (int(x) for x in arg.split(","))
    M = 16
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, c, M)
    ld = aa.recommended_ld(B)
    pen = aa.make_penalty(poly_rows=M, **dict(bench.PEN, res=int(os.environ.get("ANET_RES", bench.PEN["res"]))))
    th, tt, tw, tT, thp = (bench._to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    def ev():
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)
    runs = bench.timed_reps(torch, ev, 200 if B <= 16384 else 30, reps=5)
    st = sorted(r[1] for r in runs)
    gP2, gT2 = torch.empty_like(gP), torch.empty_like(gT)
    sp = bench.cost_grad_kernel_split(torch, aa, ctx, s, c, N, B, ld, th, tt, tw, tT, thp, pen, work, gP2, gT2)
    print(f"s {s} c {c} N {N} B {B}: eval us min/med/max {st[0]*1e3:.2f} {st[2]*1e3:.2f} {st[-1]*1e3:.2f} | split "
          + " ".join(f"{k.replace('k_minco_','').replace('k_','')} {v:.2f}" for k, v in sp.items()), flush=True)
