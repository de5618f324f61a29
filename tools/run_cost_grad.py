#!/usr/bin/env python3
"""Three cost+gradient evaluations of BASELINE config 3's shape at B=131072 (a short, fixed workload
for counter collection: tools/pmc_cost_grad.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import allocnet_amd as aa
    from tools.bench_configs import synth, to_bm
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    B, s, c, N, M = int(os.environ.get("ANET_RUN_BATCH", str(1 << 17))), 4, 3, 8, 16   # ANET_RUN_BATCH: another batch size
    ld = aa.recommended_ld(B)
    head, tail, wps, T, hp = synth(np.random.default_rng(1), B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0,
                          max_acc=6.0, res=int(os.environ.get('ANET_RES', '20')), poly_rows=M)   # ANET_RES=1: loads and stores only
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    for _ in range(2):
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost,
                               gradP=gP, gradT=gT, ctx=ctx)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
