"""The online flow of the reference end to end on the GPU path: route -> convexCover (FIRI) -> the planner's
normalise-and-negate of the polytopes -> QPSolver::solve -> Trajectory (sfc_gen.hpp:116-186,
learning_planner.hpp:196-233, 293-299).  Checks what the planner relies on: the trajectory starts and ends
where asked, stays inside the corridor it was given and inside the velocity / acceleration boxes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_route_to_corridor_to_trajectory(anet_ctx):
    import allocnet_amd as aa
    from allocnet_amd.firi import to_planner_form
    rng = np.random.default_rng(17)
    route = [np.array([0.0, 0.0, 1.0]), np.array([3.5, 1.0, 1.5]), np.array([6.0, 4.0, 1.0]), np.array([9.0, 4.5, 2.0])]
    pts = rng.uniform([-3, -3, 0], [12, 8, 4], size=(4000, 3))
    keep = np.ones(len(pts), dtype=bool)
    for p0, p1 in zip(route[:-1], route[1:]):
        d = p1 - p0
        t = np.clip(((pts - p0) @ d) / (d @ d), 0, 1)
        keep &= np.linalg.norm(pts - (p0 + t[:, None] * d), axis=1) > 0.7
    pts = pts[keep]
    polys = aa.convex_cover(route, pts, [-3, -3, 0], [12, 8, 4], progress=2.5, rng_range=3.0, ctx=anet_ctx)
    N = len(polys)
    assert 4 <= N <= 12
    M = max(p.shape[0] for p in polys)
    raw = np.zeros((N, M, 4)); nrows = np.array([p.shape[0] for p in polys], dtype=np.int32)
    for i, p in enumerate(polys):
        raw[i, :p.shape[0]] = p
    hp = to_planner_form(raw, nrows)                      # rows a.x <= b, unit normals (learning_planner.hpp:293-299)
    for i in range(N):
        k = nrows[i]
        assert np.abs(np.linalg.norm(hp[i, :k, :3], axis=1) - 1.0).max() < 1e-12
        xs = rng.uniform([-3, -3, 0], [12, 8, 4], size=(50, 3))            # same half-spaces in both forms
        assert np.array_equal((xs @ raw[i, :k, :3].T + raw[i, :k, 3]) <= 0.0, (xs @ hp[i, :k, :3].T - hp[i, :k, 3]) <= 0.0)
    # time allocation: a constant speed along the chord between consecutive overlap points (the network's job in the
    # reference); generous enough for the limits below
    ini = np.zeros((3, 3)); fin = np.zeros((3, 3))
    ini[:, 0] = route[0]; fin[:, 0] = route[-1]
    T = np.full(N, np.linalg.norm(np.diff(np.array(route), axis=0), axis=1).sum() / N / 0.8)
    vmax, amax, res = 3.0, 4.0, 20
    for s in (3, 4):
        out = aa.qp_solve(s, ini[None], fin[None], hp[None], T[None], res=res, max_vel=vmax, max_acc=amax,
                          settings=aa.qp_settings(method=aa.qp.QP_METHOD_INTERIOR_POINT), ctx=anet_ctx)
        assert out["status"][0] == 1, out["status"]
        traj = aa.Trajectory()
        for i in range(N):
            traj.emplace_back(T[i], out["coeffs"][0, i])
        assert np.abs(traj.getPos(0.0) - route[0]).max() < 1e-8
        assert np.abs(traj.getPos(traj.getTotalDuration()) - route[-1]).max() < 1e-8
        assert np.abs(traj.getVel(0.0)).max() < 1e-8 and np.abs(traj.getAcc(traj.getTotalDuration())).max() < 1e-7
        t0 = 0.0
        for i in range(N):
            k = nrows[i]
            for j in range(res):                          # the samples the QP constrains (qp_solver.hpp:244-296)
                t = t0 + j * T[i] / res
                p = traj.getPos(t)
                assert (hp[i, :k, :3] @ p - hp[i, :k, 3]).max() <= 1e-6, (s, i, j)
                assert np.abs(traj.getVel(t)).max() <= vmax + 1e-6 and np.abs(traj.getAcc(t)).max() <= amax + 1e-6
            t0 += T[i]
        # every obstacle point is outside the polytope of the piece it would otherwise hit: sample the trajectory densely
        ts = np.linspace(0.0, traj.getTotalDuration(), 400)
        P = np.array([traj.getPos(t) for t in ts])
        dmin = np.min(np.linalg.norm(P[:, None, :] - pts[None, ::7, :], axis=2))
        assert dmin > 0.05


def test_route_to_corridor_to_minco_lbfgs(anet_ctx):
    """The north-star formulation on the same corridor: waypoints and durations are the variables, corridor and
    limit rows a smoothed penalty, L-BFGS outer loop (anet_lbfgs_minco).  From the junction points of the corridor
    the optimiser lowers the cost, and the result stays within a few centimetres of the corridor."""
    import allocnet_amd as aa
    from allocnet_amd.firi import to_planner_form
    rng = np.random.default_rng(17)
    route = [np.array([0.0, 0.0, 1.0]), np.array([3.5, 1.0, 1.5]), np.array([6.0, 4.0, 1.0]), np.array([9.0, 4.5, 2.0])]
    pts = rng.uniform([-3, -3, 0], [12, 8, 4], size=(4000, 3))
    keep = np.ones(len(pts), dtype=bool)
    for p0, p1 in zip(route[:-1], route[1:]):
        d = p1 - p0
        t = np.clip(((pts - p0) @ d) / (d @ d), 0, 1)
        keep &= np.linalg.norm(pts - (p0 + t[:, None] * d), axis=1) > 0.7
    pts = pts[keep]
    # one polytope per straight step of the route (no gap polytopes needed for this check): FIRI directly
    steps = []
    b = route[0]
    i = 1
    while i < len(route):
        a = b
        if np.linalg.norm(a - route[i]) > 2.5:
            b = (route[i] - a) / np.linalg.norm(route[i] - a) * 2.5 + a
        else:
            b = route[i]; i += 1
        steps.append((a, b))
    N = len(steps)
    bd = np.zeros((N, 6, 4)); A = np.array([s_[0] for s_ in steps]); Bv = np.array([s_[1] for s_ in steps])
    for k in range(N):
        hi = np.minimum(np.maximum(A[k], Bv[k]) + 3.0, [12, 8, 4]); lo = np.maximum(np.minimum(A[k], Bv[k]) - 3.0, [-3, -3, 0])
        for ax in range(3):
            bd[k, 2 * ax, ax] = 1.0; bd[k, 2 * ax, 3] = -hi[ax]; bd[k, 2 * ax + 1, ax] = -1.0; bd[k, 2 * ax + 1, 3] = lo[ax]
    out = aa.firi(bd, np.broadcast_to(pts, (N,) + pts.shape).copy(), A, Bv, max_rows=64, ctx=anet_ctx)
    assert (out["ok"] == 1).all()
    M = int(out["n_rows"].max())
    hp = to_planner_form(out["hpoly"][:, :M], out["n_rows"])[None]            # (1, N, M, 4)
    head = np.zeros((1, 3, 3)); tail = np.zeros((1, 3, 3))
    head[0, :, 0] = route[0]; tail[0, :, 0] = route[-1]
    wps = Bv[None, :-1].copy()                                                   # junction points: inside both neighbours
    T = np.full((1, N), 2.5 / 1.0)
    pen = aa.make_penalty(rho=30.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=20, poly_rows=M)
    c0 = aa.minco_cost_grad(head, tail, wps, T, 4, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
    res = aa.lbfgs_minco(head, tail, wps, T, 4, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=3000,
                         ctx=anet_ctx)
    assert res["status"][0] >= 0 and res["cost"][0] < c0[0]
    traj = aa.Trajectory()
    for i in range(N):
        traj.emplace_back(res["T"][0, i], res["coeffs"][0, i])
    assert np.abs(traj.getPos(0.0) - route[0]).max() < 1e-9 and np.abs(traj.getPos(traj.getTotalDuration()) - route[-1]).max() < 1e-9
    t0 = 0.0
    worst = 0.0
    for i in range(N):
        k = out["n_rows"][i]
        for j in range(20):
            p = traj.getPos(t0 + j * res["T"][0, i] / 20)
            worst = max(worst, float((hp[0, i, :k, :3] @ p - hp[0, i, :k, 3]).max()))
        t0 += res["T"][0, i]
    assert worst <= 0.05, worst                                                  # soft constraint: centimetres, not metres


def test_example_script_runs():
    """examples/plan_once.py (the reference's online flow, README) exits 0 and reports a solved QP."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "examples", "plan_once.py")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "solved True" in res.stdout and "trajectory:" in res.stdout
