#!/usr/bin/env python3
"""What the parked optimiser of a problem says, after K evaluations, about the evaluations it still needs: runs the two-launch
L-BFGS with the split forced (ANET_LBFGS_SPLIT_EVALS=K, MIN_BATCH / MIN_VARS = 1 from the environment), reads the parked state
back from the workspace and stores it with the final counts.
   gpurun -- 'ANET_LBFGS_SPLIT_EVALS=400 ANET_LBFGS_SPLIT_MIN_BATCH=1 ANET_LBFGS_SPLIT_MIN_VARS=1 python tools/lbfgs_split_features.py 4,8,4096'"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
from tools.bench_configs import to_bm
K = int(os.environ["ANET_LBFGS_SPLIT_EVALS"])
dev = torch.device("cuda", 0); ctx = aa.Context(0)
for arg in sys.argv[1:]:
    s, N, B = (int(v) for v in arg.split(","))
    M = 16; ld = aa.recommended_ld(B)
    data = corridor_problem(np.random.default_rng(2), B, N, 3, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=M)
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in data)
    r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, 3, N, B, hpolys=thp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=40000, opt=3, ctx=ctx, return_work=True)
    torch.cuda.synchronize()
    w = r["_work"]
    tail_len = 1472 * ld + ld + 2 + 2048
    cont = w[w.numel() - tail_len: w.numel() - tail_len + 1472 * ld].reshape(ld, 23, 64)[:B].cpu().numpy()
    n = 3 * (N - 1) + N
    lane = cont[:, :22, :n]   # x g d xp gp pf hs[8] hy[8]
    norms = np.sqrt((lane ** 2).sum(axis=2))                                  # (B, 22): 2-norms of every per-lane vector
    np.savez(f"gpurun_out/lbfgs_split_features_{s}_{N}_{K}.npz", norms=norms, dx=np.sqrt(((lane[:, 0] - lane[:, 3]) ** 2).sum(axis=1)),
             uni=cont[:, 22, :32], evals=r["evals"].cpu().numpy(), iters=r["iters"].cpu().numpy())
    print("saved", s, N, B, K, flush=True)
