#!/usr/bin/env python3
"""Latency of the host (trajectory-major) entry points for ONE trajectory -- how the reference's planner
would call them (learning_planner.hpp:196-233, learning_planning.cpp:217-304)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import allocnet_amd as aa
from allocnet_amd.synth import random_problem
ctx = aa.Context(0)
rng = np.random.default_rng(0)
head, tail, wps, T = random_problem(rng, 1, 5, 3)
co, en = aa.minco_solve(head, tail, wps, T, 3, ctx=ctx)
traj = aa.Trajectory(list(T[0]), list(co[0]), ctx=ctx)
def timeit(f, n=300):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("minco_solve  B=1 : %.1f us" % timeit(lambda: aa.minco_solve(head, tail, wps, T, 3, ctx=ctx)))
print("traj.getPos  B=1 : %.1f us" % timeit(lambda: traj.getPos(1.234)))
print("traj_cost    B=1 : %.1f us" % timeit(lambda: traj.getTrajCost(3)))
pen = aa.make_penalty(rho=10.0, w_corridor=100.0, w_vel=10.0, w_acc=10.0, res=20, poly_rows=0)
print("cost_grad    B=1 : %.1f us" % timeit(lambda: aa.minco_cost_grad(head, tail, wps, T, 3, penalty=pen, ctx=ctx)))
# QPSolver::solve as the planner calls it (learning_planner.hpp:196): one problem, host pointers in and out
from allocnet_amd.synth import corridor_problem
for (s, N) in ((3, 5), (4, 5), (4, 8)):
    h1, t1, w1, T1, hp1 = corridor_problem(np.random.default_rng(1), 1, N, 3, 16)
    r = aa.qp_solve(s, h1, t1, hp1, T1 * 1.5, res=20, max_vel=4.0, max_acc=6.0, ctx=ctx)
    print("qp_solve     B=1 : %.1f us  (s=%d, %d pieces, 16 rows, res 20; status %d, %d Newton steps)" % (
        timeit(lambda: aa.qp_solve(s, h1, t1, hp1, T1 * 1.5, res=20, max_vel=4.0, max_acc=6.0, ctx=ctx), 100), s, N,
        int(r["status"][0]), int(r["iters"][0])))
