#!/usr/bin/env python3
"""anet_qp_solve_vjp against central differences of a smooth loss of the optimal coefficients, at several tolerances
and difference steps (tests/test_qp_solve_gpu.py::test_backward_pass_through_the_qp is the asserted version)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import allocnet_amd as aa
from allocnet_amd.synth import qp_corridor_problem
ctx = aa.Context(0)
s,N,M,res = 4,3,9,8
rng = np.random.default_rng(70 + 10 * s + N)
B=8
probs = [qp_corridor_problem(rng, N, M, margin=1.2) for _ in range(B)]
ini = np.array([p[0] for p in probs]); fin = np.array([p[1] for p in probs])
hp = np.array([p[2] for p in probs]); T = np.array([p[3] for p in probs])
kw = dict(res=res, max_vel=3.0, max_acc=4.0, ctx=ctx)
D=2*s
w1 = rng.normal(size=(N, 3, D)); w2 = rng.uniform(0.0, 1.0, size=(N, 3, D))
loss=lambda z: (w1 * z).sum(axis=(1, 2, 3)) + 0.5 * (w2 * z * z).sum(axis=(1, 2, 3))
def st(e): return aa.qp_settings(method=1, eps_abs=e, eps_rel=e)
base = aa.qp_solve(s, ini, fin, hp, T, settings=st(1e-11), **kw)
print("status", base["status"], "iters", base["iters"])
gz = w1[None] + w2[None] * base["coeffs"]
for e in (1e-7, 1e-9, 1e-11, 1e-13):
    out = aa.qp_solve_vjp(s, ini, fin, hp, T, gz, settings=st(e), **kw)
    print("vjp eps", e, "iters", out["iters"]); print(np.array2string(out["grad_T"], precision=6))
for h in (1e-4, 1e-5, 1e-6):
    fd = np.zeros((B, N))
    for i in range(N):
        Tp = T.copy(); Tp[:, i] += h
        Tm = T.copy(); Tm[:, i] -= h
        fd[:, i] = (loss(aa.qp_solve(s, ini, fin, hp, Tp, settings=st(1e-12), **kw)["coeffs"]) - loss(aa.qp_solve(s, ini, fin, hp, Tm, settings=st(1e-12), **kw)["coeffs"])) / (2 * h)
    print("fd h", h); print(np.array2string(fd, precision=6))
