#!/usr/bin/env python3
"""Wall time per L-BFGS evaluation step (objective + update) at a given batch, 8 snap pieces, 200 steps:
    gpurun -- 'python tools/time_lbfgs_step.py 131072'"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from tools.bench_configs import synth, to_bm
dev = torch.device("cuda", 0); ctx = aa.Context(0)
B, s, c, N, M = int(sys.argv[1]), 4, 3, 8, 16
ld = aa.recommended_ld(B)
rng = np.random.default_rng(2)
head, tail, wps, T, hp = synth(rng, B, N, c, M)
pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=M)
prm = aa.lbfgs_parameter_t()
for cap in (200, 200):
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=cap, ctx=ctx)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("B", B, "evals", cap, "sec %.4f" % dt, "ms/step %.3f" % (dt / cap * 1e3), flush=True)
