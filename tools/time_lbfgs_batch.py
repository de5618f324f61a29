#!/usr/bin/env python3
"""Wall time of the one-launch L-BFGS run to convergence for arbitrary shapes (BASELINE configs[3]'s generator and parameters):
    gpurun -- 'python tools/time_lbfgs_batch.py 3,16,4096 4,8,4096'          # order,pieces,batch
(ANET_LBFGS_SPLIT_EVALS=0 in the environment: one launch instead of two.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
from tools.bench_configs import to_bm
dev = torch.device("cuda", 0); ctx = aa.Context(0)
for arg in sys.argv[1:]:
    s, N, B = (int(v) for v in arg.split(","))
    M = 16; ld = aa.recommended_ld(B)
    data = corridor_problem(np.random.default_rng(2), B, N, 3, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=M)
    ts = []
    for rep in range(3):
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, 3, N, B, hpolys=thp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=40000, opt=3, ctx=ctx)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ev = r["evals"][:B].cpu().numpy()
    print(f"s {s} N {N} B {B}: ms {['%.1f' % t for t in ts]} evals mean {ev.mean():.0f} p99 {np.percentile(ev, 99):.0f} max {ev.max()}", flush=True)
