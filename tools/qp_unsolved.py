#!/usr/bin/env python3
"""Which problems does the default QP method leave unsolved although they are feasible?  
The feasible set is established independently of the method under test: a problem counts as feasible if the
interior-point method OR the ADMM method run to tight tolerances with a 25x larger iteration budget solves it.
    gpurun -- 'python tools/qp_unsolved.py'"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import allocnet_amd as aa
    from allocnet_amd.synth import corridor_problem
    ctx = aa.Context(0)
    out = {}
    for (s, N, M, B, sc, seed) in [(4, 8, 16, 4096, 1.5, 1), (4, 8, 16, 512, 1.0, 5), (3, 5, 16, 2048, 1.5, 1), (4, 5, 16, 2048, 1.0, 2)]:
        head, tail, wps, T, hp = corridor_problem(np.random.default_rng(seed), B, N, 3, M)
        T = T * sc
        kw = dict(res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
        runs = {
            "admm_default": aa.qp_solve(s, head, tail, hp, T, settings=aa.qp_settings(method=aa.qp.QP_METHOD_ADMM), **kw),
            "ipm": aa.qp_solve(s, head, tail, hp, T, settings=aa.qp_settings(method=aa.qp.QP_METHOD_INTERIOR_POINT), **kw),
            "admm_tight": aa.qp_solve(s, head, tail, hp, T, settings=aa.qp_settings(method=aa.qp.QP_METHOD_ADMM, eps_abs=1e-6, eps_rel=1e-6,
                                                                                   max_iter=100000), **kw),
            "default": aa.qp_solve(s, head, tail, hp, T, **kw),
        }
        ok = {k: v["status"] == 1 for k, v in runs.items()}
        feas = ok["ipm"] | ok["admm_tight"]
        rec = {"batch": B, "feasible": int(feas.sum())}
        for k in runs:
            rec[k] = {"solved": int(ok[k].sum()), "feasible_but_unsolved": int((feas & ~ok[k]).sum()),
                      "feasible_but_unsolved_frac": float((feas & ~ok[k]).sum() / max(1, feas.sum())),
                      "iters_mean": float(runs[k]["iters"].mean()),
                      "statuses": {str(a): int(b) for a, b in zip(*np.unique(runs[k]["status"], return_counts=True))}}
        both = ok["ipm"] & ok["admm_default"]
        rec["obj_rel_diff_admm_default_vs_ipm_median"] = float(np.median(np.abs(runs["ipm"]["obj"][both] - runs["admm_default"]["obj"][both]) /
                                                                 np.maximum(1e-9, np.abs(runs["ipm"]["obj"][both])))) if both.any() else None
        out[f"s{s}_N{N}_M{M}_B{B}_Tx{sc}"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
