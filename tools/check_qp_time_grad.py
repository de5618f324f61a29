#!/usr/bin/env python3
"""d(optimal QP cost)/dT from anet_qp_solve_time_grad against central differences of the optimal cost
itself (tight tolerances).  Prints the worst relative error per problem."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import allocnet_amd as aa
    from allocnet_amd.synth import qp_corridor_problem as _corridor_problem
    ctx = aa.Context(0)
    for (s, N, M, res, vmax, amax) in [(4, 3, 9, 6, 3.0, 4.0), (3, 4, 8, 5, 3.0, 4.0), (3, 2, 7, 10, 1.0, 1.5), (4, 5, 12, 10, 2.0, 2.5)]:
        rng = np.random.default_rng(10 * s + N)
        B = 6
        probs = [_corridor_problem(rng, N, M, margin=0.6) for _ in range(B)]
        ini = np.array([p[0] for p in probs]); fin = np.array([p[1] for p in probs])
        hp = np.array([p[2] for p in probs]); T = np.array([p[3] for p in probs])
        st = aa.qp_settings(method=aa.qp.QP_METHOD_ADMM, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000)
        out = aa.qp_solve(s, ini, fin, hp, T, res=res, max_vel=vmax, max_acc=amax, settings=st, time_grad=True, ctx=ctx)
        g = out["grad_T"]
        fd = np.zeros_like(g)
        h = 1e-5
        for i in range(N):
            Tp = T.copy(); Tp[:, i] += h; Tm = T.copy(); Tm[:, i] -= h
            op = aa.qp_solve(s, ini, fin, hp, Tp, res=res, max_vel=vmax, max_acc=amax, settings=st, ctx=ctx)["obj"]
            om = aa.qp_solve(s, ini, fin, hp, Tm, res=res, max_vel=vmax, max_acc=amax, settings=st, ctx=ctx)["obj"]
            fd[:, i] = (op - om) / (2 * h)
        eff = aa.traj_cost_grad_T(out["coeffs"], T, m34=1400.0, ctx=ctx)
        for b in range(B):
            sc = np.abs(fd[b]).max()
            print(s, N, "status", out["status"][b], "iters", out["iters"][b], "obj %.4g" % out["obj"][b],
                  "err %.2e" % (np.abs(g[b] - fd[b]).max() / sc), "effective-grad err %.2e" % (np.abs(eff[b] - fd[b]).max() / sc),
                  "g", np.round(g[b], 4), "fd", np.round(fd[b], 4))
        # default tolerance
        outd = aa.qp_solve(s, ini, fin, hp, T, res=res, max_vel=vmax, max_acc=amax, time_grad=True, ctx=ctx)
        print("   default tol: err", ["%.1e" % (np.abs(outd["grad_T"][b] - fd[b]).max() / np.abs(fd[b]).max()) for b in range(B)])


if __name__ == "__main__":
    main()
