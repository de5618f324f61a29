#!/usr/bin/env python3
"""One pass of the reference's online flow (LearningPlanner::plan + callModel, learning_planner.hpp:240-300, 140-236) on the
MI355X path, with a constant-speed time allocation standing in for the network:

    route -> convexCover -> shortCut -> planner form -> [network: segment times] -> QPSolver::solve -> Trajectory

    python examples/plan_once.py          # needs a GPU; prints the stages and their wall times
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import allocnet_amd as aa  # noqa: E402


def main():
    rng = np.random.default_rng(17)
    route = [np.array(p, dtype=float) for p in ([0, 0, 1], [3.5, 1, 1.5], [6, 4, 1], [9, 4.5, 2])]
    lo, hi = [-3, -3, 0], [12, 8, 4]
    pts = rng.uniform(lo, hi, size=(4000, 3))                       # the map's surface points (voxel_map getSurf)
    keep = np.ones(len(pts), dtype=bool)
    for p0, p1 in zip(route[:-1], route[1:]):
        d = p1 - p0
        t = np.clip(((pts - p0) @ d) / (d @ d), 0, 1)
        keep &= np.linalg.norm(pts - (p0 + t[:, None] * d), axis=1) > 0.7
    pts = pts[keep]
    aa.convex_cover(route[:2], pts, lo, hi, progress=7.0, rng_range=3.0)            # warm-up (module load)
    t0 = time.perf_counter()
    polys = aa.convex_cover(route, pts, lo, hi, progress=7.0, rng_range=3.0)        # learning_planner.hpp:274-280
    t1 = time.perf_counter()
    polys = aa.short_cut(polys)                                                     # :282
    t2 = time.perf_counter()
    seg = len(polys)
    if seg > 5:
        print("give up this try, long corridor")                                   # :286-290 (modelMaxSeg)
        return 1
    H = max(len(p) for p in polys)
    raw = np.zeros((seg, H, 4)); rows = [len(p) for p in polys]
    for i, p in enumerate(polys):
        raw[i, :len(p)] = p
    hp = aa.to_planner_form(raw, rows)                                              # :293-299
    ini = np.zeros((3, 3)); fin = np.zeros((3, 3))
    ini[:, 0] = route[0]; fin[:, 0] = route[-1]
    state, corridor = aa.pack_model_inputs(ini, fin, [hp[i, :rows[i]] for i in range(seg)])   # what callModel feeds the network
    length = np.linalg.norm(np.diff(np.array(route), axis=0), axis=1).sum()
    times = np.full(seg, length / seg / 1.0, dtype=np.float32)                      # <- minsnap_conv_lstm_network.forward(inputs)
    t3 = time.perf_counter()
    solver = aa.QPSolver(aa.QPConfig(MaxVelBox=4.0, MaxAccBox=6.0, ConstRes=20))
    solver.setOrder(4)
    ok, flat = solver.solve(ini, fin, [hp[i, :rows[i]] for i in range(seg)], times)  # :196
    t4 = time.perf_counter()
    print(f"convexCover {1e3 * (t1 - t0):.2f} ms -> shortCut {1e3 * (t2 - t1):.2f} ms -> {seg} polytopes of {rows} rows; "
          f"network inputs {state.shape} {corridor.shape}; QP {1e3 * (t4 - t3):.2f} ms, solved {ok}, cost {solver.getObjCost():.3f}")
    if not ok:
        return 1
    traj = aa.Trajectory()
    co = np.asarray(flat).reshape(seg, 3, 8)                                        # :219-231
    for i in range(seg):
        traj.emplace_back(float(times[i]), co[i])
    T = traj.getTotalDuration()
    print(f"trajectory: {T:.2f} s, start {traj.getPos(0.0)}, end {traj.getPos(T)}, max |v| {traj.getMaxVelRate():.2f}, "
          f"max |a| {traj.getMaxAccRate():.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
