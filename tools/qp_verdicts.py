#!/usr/bin/env python3
"""Feasibility verdicts of the two QP methods on the same random problems: every problem the ADMM method
solves must be solved by the interior-point method, with the same objective."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import allocnet_amd as aa
    from allocnet_amd.synth import corridor_problem
    ctx = aa.Context(0)
    for (s, N, M, sc) in [(4, 8, 16, 0.7), (4, 8, 16, 1.5), (4, 8, 16, 4.0), (3, 5, 16, 0.5), (3, 5, 16, 10.0), (4, 5, 16, 0.2)]:
        B = 512
        head, tail, wps, T, hp = corridor_problem(np.random.default_rng(11), B, N, 3, M)
        kw = dict(res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
        ipm = aa.qp_solve(s, head, tail, hp, T * sc, settings=aa.qp_settings(method=1), **kw)
        adm = aa.qp_solve(s, head, tail, hp, T * sc, settings=aa.qp_settings(method=aa.qp.QP_METHOD_ADMM, eps_abs=1e-7, eps_rel=1e-7, max_iter=100000), **kw)
        si, sa = ipm["status"] == 1, adm["status"] == 1
        both = si & sa
        rel = np.abs(ipm["obj"][both] - adm["obj"][both]) / np.maximum(1e-3, np.abs(adm["obj"][both]))
        print(s, N, "T x", sc, "| ipm solved %d admm solved %d | admm-only %d ipm-only %d | obj rel diff max %.1e | ipm iters mean %.1f max %d | admm statuses %s" %
              (si.sum(), sa.sum(), (sa & ~si).sum(), (si & ~sa).sum(), rel.max() if both.any() else 0.0, ipm["iters"][si].mean() if si.any() else 0,
               ipm["iters"][si].max() if si.any() else 0, dict(zip(*np.unique(adm["status"], return_counts=True)))))


if __name__ == "__main__":
    main()
