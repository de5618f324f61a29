#!/usr/bin/env python3
"""Interior-point method vs tightly converged ADMM on the same problems (GPU)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import allocnet_amd as aa
    from allocnet_amd.synth import qp_corridor_problem as _corridor_problem
    from allocnet_amd.synth import corridor_problem
    ctx = aa.Context(0)
    for (s, N, M, res) in [(4, 3, 9, 6), (3, 4, 8, 5), (4, 5, 12, 10), (4, 1, 7, 8)]:
        rng = np.random.default_rng(10 * s + N)
        probs = [_corridor_problem(rng, N, M, margin=0.6) for _ in range(6)]
        ini = np.array([p[0] for p in probs]); fin = np.array([p[1] for p in probs])
        hp = np.array([p[2] for p in probs]); T = np.array([p[3] for p in probs])
        kw = dict(res=res, max_vel=3.0, max_acc=4.0, ctx=ctx)
        ref = aa.qp_solve(s, ini, fin, hp, T, settings=aa.qp_settings(method=aa.qp.QP_METHOD_ADMM, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000), **kw)
        ipm = aa.qp_solve(s, ini, fin, hp, T, settings=aa.qp_settings(method=1), **kw)
        print(s, N, "admm status", ref["status"], "ipm status", ipm["status"], "iters", ipm["iters"])
        print("   obj admm", np.round(ref["obj"], 6), "\n   obj ipm ", np.round(ipm["obj"], 6), "res", ipm["residuals"].max(axis=0))
        print("   coeff diff", np.abs(ref["coeffs"] - ipm["coeffs"]).max(axis=(1, 2, 3)) / np.abs(ref["coeffs"]).max(axis=(1, 2, 3)))
    for (s, N, M, B) in [(4, 8, 16, 4096), (3, 5, 16, 4096), (3, 16, 16, 1024), (4, 5, 16, 1)]:
        rng = np.random.default_rng(1)
        head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, M)
        T = T * 1.5
        st = aa.qp_settings(method=1)
        aa.qp_solve(s, head[:2], tail[:2], hp[:2], T[:2], settings=st, ctx=ctx)
        t0 = time.perf_counter()
        r = aa.qp_solve(s, head, tail, hp, T, res=20, max_vel=4.0, max_acc=6.0, settings=st, ctx=ctx)
        dt = time.perf_counter() - t0
        print("bench", s, N, M, B, "seconds %.4f" % dt, "solves/s %.0f" % (B / dt), "status", dict(zip(*np.unique(r["status"], return_counts=True))),
              "iters mean %.1f max %d" % (r["iters"].mean(), r["iters"].max()), "obj median %.5g" % np.median(r["obj"]))


if __name__ == "__main__":
    main()
