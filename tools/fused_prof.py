#!/usr/bin/env python3
"""Phase cycle stamps of the one-launch cost + gradient kernel (library built with ANET_BUILD_FLAGS=-DANET_FUSED_PROF):
    gpurun -- 'bash tools/ab_build.sh "-DANET_FUSED_PROF" python tools/fused_prof.py 4,3,8,4096 4,3,8,512'"""
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
    pen = aa.make_penalty(poly_rows=M, **bench.PEN)
    th, tt, tw, tT, thp = (bench._to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    for _ in range(20):
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)
    torch.cuda.synchronize()
    print(f"s {s} c {c} N {N} B {B}", file=sys.stderr, flush=True)
    os.environ["ANET_FUSED_PROF_PRINT"] = "1"
    for _ in range(2):
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)
    torch.cuda.synchronize()
    del os.environ["ANET_FUSED_PROF_PRINT"]
