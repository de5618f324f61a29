#!/usr/bin/env python3
"""Wall time of one cost+gradient evaluation (three launches) for arbitrary shapes:
    gpurun -- 'python tools/time_cost_grad.py 4,3,8,4096 3,3,5,1 ...'     # order,boundary count,pieces,batch
Three repetitions of 200 evaluations each; the first repetition after an allocation sometimes carries a one-off stall."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from tools.bench_configs import synth, to_bm
dev = torch.device("cuda", 0); ctx = aa.Context(0)
def run(s, c, N, B, K=200):
    M = 16
    ld = aa.recommended_ld(B)
    rng = np.random.default_rng(1)
    head, tail, wps, T, hp = synth(rng, B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=int(os.environ.get('ANET_RES', 20)), poly_rows=M)
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    print("s", s, "c", c, "N", N, "B", B, "us/eval", ["%.1f" % t for t in ts], flush=True)
for a in sys.argv[1:]:
    run(*[int(x) for x in a.split(",")])
