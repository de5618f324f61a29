#!/usr/bin/env python3
"""Experiment: how much of config 4's one-launch time is the ORDER in which problems get their wave?  The run is as long
as the last problem to finish; a long problem that starts in the second round of workgroups finishes late.  Times the
same batch (a) as given, (b) sorted by the true evaluation count of a first run, longest first (the bound of what any
predictor could reach), (c) sorted by predictors available before the run (initial cost, gradient norm, ...).
    gpurun -- 'python tools/persist_order_probe.py > gpurun_out/persist_order.json'"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import to_bm  # noqa: E402
from allocnet_amd.synth import corridor_problem as synth  # noqa: E402


def main():
    import torch
    import allocnet_amd as aa
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    B, s, c, N, M = 4096, 3, 3, 16, 16
    ld = aa.recommended_ld(B)
    rng = np.random.default_rng(2)
    data = synth(rng, B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=20, poly_rows=M)
    prm = aa.lbfgs_parameter_t()

    def run(perm, reps=3):
        best = None
        for _ in range(reps):
            th, tt, tw, tT, thp = (to_bm(torch, x[perm], B, ld, dev) for x in data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=30000,
                                     opt=3, ctx=ctx)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, res["evals"].cpu().numpy()[:B]

    ident = np.arange(B)
    run(ident, 1)
    t_id, ev = run(ident)
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in data)
    cost, gP, gT, _ = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    c0 = cost[:B].cpu().numpy()
    gn = np.sqrt((gP[:, :B].cpu().numpy() ** 2).sum(axis=0) + (gT[:, :B].cpu().numpy() ** 2).sum(axis=0))
    feats = {"initial_cost": c0, "gradient_norm": gn, "total_time": data[3].sum(axis=1),
             "path_length": np.linalg.norm(np.diff(np.concatenate([data[0][:, None, :, 0] if data[0].ndim == 3 else data[0][:, None, :3],
                                                                    data[2].reshape(B, N - 1, 3)], axis=1), axis=1), axis=2).sum(axis=1)
             if False else c0 * 0}
    if "--dump" in sys.argv:
        np.save(os.path.join(ROOT, "gpurun_out", "config4_evals.npy"), ev)
    out = {"as_given_s": t_id, "evals_mean": float(ev.mean()), "evals_max": int(ev.max())}
    t_lpt, ev2 = run(np.argsort(-ev))
    out["longest_first_by_true_count_s"] = t_lpt
    out["counts_identical_after_permutation"] = bool((np.sort(ev2) == np.sort(ev)).all())
    t_spt, _ = run(np.argsort(ev))
    out["shortest_first_by_true_count_s"] = t_spt
    from scipy.stats import spearmanr
    for k, f in feats.items():
        if not np.any(f):
            continue
        rho = float(spearmanr(f, ev)[0])
        t_f, _ = run(np.argsort(-f) if rho > 0 else np.argsort(f))
        out[f"by_{k}"] = {"spearman": rho, "seconds": t_f}
    # a re-solve scenario: the counts of a PERTURBED copy of the batch (waypoints moved by ~1 cm, durations by 1 %) as the predictor
    for label, sw, sT in (("perturbed_1cm_1pct", 0.01, 0.01), ("perturbed_10cm_5pct", 0.1, 0.05)):
        rng2 = np.random.default_rng(99)
        pert = [x.copy() for x in data]
        pert[2] = pert[2] + sw * rng2.standard_normal(pert[2].shape)
        pert[3] = pert[3] * (1.0 + sT * rng2.uniform(-1, 1, size=pert[3].shape))
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in pert)
        r2 = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=30000, opt=3, ctx=ctx)
        evp = r2["evals"].cpu().numpy()[:B]
        t_p, _ = run(np.argsort(-evp))
        out[f"by_counts_of_{label}"] = {"spearman": float(spearmanr(evp, ev)[0]), "seconds": t_p}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
