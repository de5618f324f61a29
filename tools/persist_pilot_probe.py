#!/usr/bin/env python3
"""Experiment (follows persist_order_probe.py): does a short PILOT run predict a problem's evaluation count?  Runs the
config-4 batch for K evaluations, then correlates what is known at that point with the count of the full run.
    gpurun -- 'python tools/persist_pilot_probe.py > gpurun_out/persist_pilot.json'"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import to_bm  # noqa: E402
from allocnet_amd.synth import corridor_problem as synth  # noqa: E402


def main():
    import torch
    import allocnet_amd as aa
    from scipy.stats import spearmanr
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    B, s, c, N, M = 4096, 3, 3, 16, 16
    ld = aa.recommended_ld(B)
    rng = np.random.default_rng(2)
    data = synth(rng, B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=20, poly_rows=M)
    prm = aa.lbfgs_parameter_t()

    def run(cap):
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in data)
        r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=cap, opt=3, ctx=ctx)
        cost, gP, gT, _ = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
        e0, *_ = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, ctx=ctx)              # energy only
        gn = torch.sqrt((gP[:, :B] ** 2).sum(dim=0) + (gT[:, :B] ** 2).sum(dim=0)).cpu().numpy()
        Tsum = tT[:, :B].sum(dim=0).cpu().numpy()
        return (r["evals"].cpu().numpy()[:B], r["iters"].cpu().numpy()[:B], cost[:B].cpu().numpy(), gn,
                e0[:B].cpu().numpy(), Tsum)

    ev, *_ = run(30000)
    out = {}
    for K in (100, 300, 1000):
        evk, itk, ck, gk, ek, Tk = run(K)
        alive = evk >= K
        penalty = ck - ek - 50.0 * Tk
        feats = {"cost": ck, "gradient_norm": gk, "penalty_part": penalty, "penalty_over_cost": penalty / ck,
                 "evals_per_iteration": evk / np.maximum(itk, 1), "gn_over_cost": gk / ck}
        out[f"pilot_{K}"] = {k: float(spearmanr(f[alive], ev[alive])[0]) for k, f in feats.items()}
        out[f"pilot_{K}"]["still_running"] = int(alive.sum())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
