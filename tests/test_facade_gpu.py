"""C++ facade (include/allocnet_amd/*.hpp) end to end on the GPU: the program in tests/cpp mirrors how
learning_planner.hpp:203-233 consumes solver output; its numbers are checked against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import minco_np as onp
from tests.util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_program():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_facade")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout)
    N, s = 8, 4
    goal = np.array([8.0, 3.0, 1.0])
    head = np.zeros((3, 3)); tail = np.zeros((3, 3)); tail[:, 0] = goal
    wps = np.array([goal * (k + 1) / N for k in range(N - 1)]).T
    T = np.ones(N)
    co, e, *_ = onp.minco_dense_solve(s, head, tail, wps, T)
    assert rel_err(np.array(out["coeffs"]).reshape(N, 3, 8), co) < 1e-9
    assert abs(out["energy"] - e) <= 1e-9 * e
    # MINCO_S4NU::sampleTimeAllocations: candidate 0 is the durations of the solve above (cost = energy + rho * sum T), the
    # others against the oracle
    from oracle import cbind
    sc = np.array(out["sample_costs"])
    assert sc.shape == (6,) and abs(sc[0] - (e + 2.0 * 8.0)) <= 1e-9 * sc[0]
    for k in range(1, 6):
        Tk = np.array([1.0 + 0.15 * k * (1.0 if i % 2 else -0.5) for i in range(8)])
        _, ek = cbind.minco_solve_batch(4, head[None], tail[None], wps.T[None], Tk[None], want_coeffs=False)
        assert abs(sc[k] - (ek[0] + 2.0 * Tk.sum())) <= 1e-9 * sc[k], k
    assert abs(out["traj_cost_1440"] - 0.5 * e) <= 1e-9 * e
    assert abs(out["traj_cost_1400"] - onp.traj_cost(co, T, s, 1400.0)) <= 1e-9 * e
    eC, eT = onp.energy_partials(s, co, T)
    gP, gT = onp.minco_dense_propagate(s, head, tail, wps, T, eC, eT)
    assert np.abs(np.array(out["gdT"]) - eT).max() <= 1e-8 * np.abs(eT).max()
    assert np.abs(np.array(out["gradP"]).reshape(N - 1, 3).T - gP).max() <= 1e-7 * max(1.0, np.abs(gP).max())
    assert np.abs(np.array(out["gradT"]) - gT).max() <= 1e-7 * max(1.0, np.abs(gT).max())
    for key, d in (("pos", 0), ("vel", 1), ("acc", 2), ("jer", 3)):
        ref = onp.traj_eval(co, T, 3.5, d)
        assert np.abs(np.array(out[key]) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(np.array(out["endp"]) - onp.traj_eval(co, T, 8.25, 0)).max() < 1e-9
    assert np.abs(np.array(out["junc_vel_3"]) - onp.piece_eval(co[3], 0.0, 1)).max() < 1e-12
    for key, d in (("norm_pos", 0), ("norm_vel", 1), ("norm_acc", 2)):
        assert np.array_equal(np.array(out[key]).reshape(3, 8 - d), onp.piece_normalized_coeffs(np.array(out["coeffs"]).reshape(N, 3, 8)[3], 1.0, d))
    assert out["locate"][0] == 2 and abs(out["locate"][1] - 0.25) < 1e-15
    assert out["pieces"] == 8 and out["total"] == 8.0
    assert abs(out["max_vel"] - max(onp.piece_max_rate(co[i], 1.0, 1) for i in range(N))) <= 1e-9 * out["max_vel"]
    assert abs(out["max_acc"] - max(onp.piece_max_rate(co[i], 1.0, 2) for i in range(N))) <= 1e-9 * out["max_acc"]
    assert out["check_vel"] == 1
    assert out["lbfgs_default_mem"] == 8 and out["strerror"].startswith("Line search reaches")
    # lbfgs::lbfgs_optimize(x, minCost, &costMVIE, nullptr, nullptr, optData, paramsMVIE) as firi.hpp:221-227 writes it:
    # largest ellipsoid in the unit cube = the unit ball at the origin; same outcome as the C restatement
    from oracle import cbind
    cube = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
    x0 = np.array([0.05, -0.02, 0.01, 0.5, 0.5, 0.5, 0.0, 0.0, 0.0])
    ret, xo, fo, it, ev = cbind.lbfgs_mvie(cube, 1e-2, 1e3, x0, cbind.lbfgs_default_param(
        mem_size=18, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7))
    assert out["mvie_ret"] >= 0 and ret >= 0
    # same stop (delta = 1e-7 on the cost over three iterations) on both sides: the costs agree to 1e-9 of their scale,
    # the variables to 1e-5 (measured: 1e-11, 3e-8); the optimum itself is the unit ball up to the penalty's bias (1e-4)
    assert out["mvie_ret"] == ret
    assert abs(out["mvie_cost"] - fo) <= 1e-9 * max(1.0, abs(fo))
    xm = np.array(out["mvie_x"])
    assert np.abs(xm - xo).max() <= 1e-5
    assert np.abs(xm[:3]).max() < 1e-6 and np.abs(xm[3:6] ** 2 - 1.0).max() < 1e-3 and np.abs(xm[6:]).max() < 1e-6

    # lbfgs::lbfgs_optimize with HOST callbacks (include/allocnet_amd/lbfgs.hpp -> anet_lbfgs_optimize_host): the extended
    # Rosenbrock function, plain and with a step bound + a progress monitor that cancels at k = 12, against the C restatement
    # of lbfgs_optimize (lbfgs.hpp:434-717) driving the same function through Python callbacks
    def rosen(x):
        g = np.zeros_like(x)
        t1 = 1.0 - x[0::2]; t2 = 10.0 * (x[1::2] - x[0::2] ** 2)
        g[1::2] = 20.0 * t2
        g[0::2] = -2.0 * (x[0::2] * g[1::2] + t1)
        return float((t1 * t1 + t2 * t2).sum()), g
    x0 = np.where(np.arange(10) % 2 == 1, 1.0, -1.2)
    prm = cbind.lbfgs_default_param(g_epsilon=1e-8, delta=1e-10)
    ret0, xo0, fo0, it0, ev0 = cbind.lbfgs_optimize(x0, rosen, prm)
    assert out["rosen0_ret"] == ret0 and ret0 in (0, 1) and out["rosen0_evals"] == ev0
    assert abs(out["rosen0_f"] - fo0) <= 1e-12 and np.abs(np.array(out["rosen0_x"]) - xo0).max() <= 1e-6
    assert np.abs(np.array(out["rosen0_x"]) - 1.0).max() < 1e-4 and out["rosen0_bounds"] == 0 and out["rosen0_reports"] == 0
    seen = []
    ret1, xo1, fo1, it1, ev1 = cbind.lbfgs_optimize(x0, rosen, prm, stepbound=lambda xp, d: 0.5 / np.abs(d).max(),
                                                  progress=lambda x, g, fx, step, k, ls: (seen.append(fx), k >= 12)[1])
    assert ret1 == 2 and out["rosen1_ret"] == 2 and out["rosen1_evals"] == ev1      # LBFGS_CANCELED at the 12th report
    assert out["rosen1_reports"] == len(seen) == 12 and out["rosen1_bounds"] == 12
    assert np.allclose(out["rosen1_fx_seen"], seen, rtol=1e-9, atol=1e-12)
    assert abs(out["rosen1_f"] - fo1) <= 1e-9 * max(1.0, abs(fo1)) and np.abs(np.array(out["rosen1_x"]) - xo1).max() <= 1e-8
    # the same run with an evaluate callback that calls a host-staged entry point on the SAME context (a coefficient solve whose
    # batch grows per call, so the context's scratch is re-allocated under the live optimiser): bit for bit the plain run
    assert out["rosen2_ret"] == out["rosen0_ret"] and out["rosen2_evals"] == out["rosen0_evals"]
    assert out["rosen2_f"] == out["rosen0_f"] and out["rosen2_x"] == out["rosen0_x"]
    assert out["rosen_bad_ret"] == -1016 and out["rosen_bad_evals"] == 0 and out["rosen_bad_f"] == 123.0

    # lbfgs::lbfgs_optimize_batched (anet_lbfgs_optimize_dev through the facade; the status row is complete on return): five
    # Rosenbrock problems, each against the restatement from its own start point
    assert len(out["batched_status"]) == 5
    for bq in range(5):
        xq = np.where(np.arange(6) % 2 == 1, 1.0, -1.2) + 0.1 * bq
        retq, xoq, foq, _, _ = cbind.lbfgs_optimize(xq, rosen, prm)
        assert out["batched_status"][bq] in (0, 1) and retq in (0, 1)
        assert abs(out["batched_f"][bq] - foq) <= 1e-10 and np.abs(np.array(out["batched_x"]).reshape(5, 6)[bq] - 1.0).max() < 1e-3

    # QPSolver facade: solved, ends where asked, inside the velocity box, objective == 1/2 z'Qz of its coefficients
    assert out["qp_ok"] == 1 and out["qp_iters"] > 0
    assert np.abs(np.array(out["qp_end"]) - np.array([6.0, 3.0, 1.0])).max() < 5e-2
    assert np.abs(np.array(out["qp_vel"])).max() <= 3.0 + 5e-2
    zc = np.array(out["qp_coeffs"]).reshape(3, 3, 6)
    assert abs(onp.traj_cost(zc, np.array([2.0, 1.5, 2.0]), 3) - out["qp_obj"]) <= 1e-9 * max(1.0, out["qp_obj"])
    # getTimeGrad (extension): one entry per segment; giving a rest-to-rest trajectory more time lowers its cost
    gT = np.array(out["qp_time_grad"])
    assert gT.shape == (3,) and np.isfinite(gT).all() and gT.sum() < 0
    # ... asked for after a solve that did not compute it (re-solve of the remembered problem) = carried by the next solve
    assert np.array_equal(gT, np.array(out["qp_time_grad_inline"]))

    # get_t_state<T> (qp_solver.hpp:88-116): row k = k-th derivative of (t^(d-1) ... t 1), evaluated in T with the reference's
    # multiplication tree for the powers; the float instantiation bit for bit against a float32 restatement
    def t_state(t, order, ft):
        t = ft(t)
        p = [ft(1), t, t * t]
        p.append(t * p[2]); p.append(p[2] * p[2]); p.append(p[2] * p[3]); p.append(p[3] * p[3]); p.append(p[4] * p[3])
        d = 2 * order
        A = np.zeros((order, d))
        for k in range(order):
            for j in range(d):
                e = d - 1 - j
                if e >= k:
                    ff = int(np.prod([e - q for q in range(k)])) if k else 1
                    A[k, j] = ff if e == k else float(ft(ff) * p[e - k])
        return A
    assert np.array_equal(np.array(out["qp_tstate_f3"]).reshape(3, 6), t_state(0.37, 3, np.float32))
    assert np.array_equal(np.array(out["qp_tstate_d3"]).reshape(3, 6), t_state(0.37, 3, np.float64))
    assert out["qp_tstate_f4_shape"] == [4, 8]
    assert np.array_equal(np.array(out["qp_tstate_f4"]).reshape(4, 8), t_state(1.7, 4, np.float32))
    # (and it is what the oracle's float assembly has as its rows: the d = 0 row is the position row of get_t_state)
    assert abs(t_state(0.37, 3, np.float64)[1, 0] - 5 * 0.37 ** 4) < 1e-15
    # setMethod(interior point): same optimum (OSQP's 1e-3 tolerances leave the ADMM objective within a few 1e-3 of it)
    assert out["qp_ipm_ok"] == 1 and out["qp_ipm_iters"] <= 40
    assert abs(out["qp_ipm_obj"] - out["qp_obj"]) <= 2e-2 * max(1.0, out["qp_obj"])
    # sfc_gen::convexCover + shortCut + geo_utils facade: the same corridor through the Python mirror (same library,
    # so identical), the kept indices against the restated shortCut walk, and the corridor's guarantees
    import allocnet_amd as aa
    cpts = np.array(out["cover_pts"]).reshape(-1, 3)
    rows = out["cover_rows"]; flat = np.array(out["cover_hpolys"]).reshape(-1, 4)
    cover = [flat[sum(rows[:k]):sum(rows[:k + 1])] for k in range(len(rows))]
    route = [np.array(w) for w in ([0.0, 0.0, 1.0], [4.0, 1.0, 1.5], [6.0, 4.0, 1.0], [9.0, 4.5, 2.0])]
    ref_cover = aa.convex_cover(route, cpts, [-3, -3, 0], [12, 8, 4], progress=2.0, rng_range=3.0)
    assert out["cover_n"] == len(ref_cover) and all(np.array_equal(a, b) for a, b in zip(cover, ref_cover))
    for hpk in cover:
        assert ((cpts @ hpk[:, :3].T + hpk[:, 3]).max(axis=1) > -2e-6).all()
    from oracle import firi_np as F
    idx = F.short_cut(cover, 0.1)
    srows = out["short_rows"]; sflat = np.array(out["short_hpolys"]).reshape(-1, 4)
    short = [sflat[sum(srows[:k]):sum(srows[:k + 1])] for k in range(len(srows))]
    assert len(short) == len(idx) and all(np.array_equal(a, cover[k]) for a, k in zip(short, idx))
    assert out["interior_found"] == 1 and (short[0] @ np.r_[out["interior"], 1.0]).max() < 0.0
    assert out["overlap_first_two"] == int(F.overlap(short[0], short[1]))
    assert out["overlap_ends"] == int(F.overlap(short[0], short[-1], 0.1))
    mid = np.r_[out["overlap_pt"], 1.0]
    assert out["overlap_pt_ok"] == 1 and (short[0] @ mid).max() < 0.0 and (short[1] @ mid).max() < 0.0
    # firi::firi facade: polytope around the segment (0,0,1)-(2,.5,1.2), lattice points outside, a outside bd -> false
    assert out["firi_ok"] == 1 and out["firi_rows"] >= 6 and out["firi_outside"] == 0
    hp = np.array(out["firi_hpoly"]).reshape(-1, 4); pts = np.array(out["firi_pts"]).reshape(-1, 3)
    assert (hp @ np.array([0.0, 0.0, 1.0, 1.0])).max() <= 1e-6 and (hp @ np.array([2.0, 0.5, 1.2, 1.0])).max() <= 1e-6
    assert ((pts @ hp[:, :3].T + hp[:, 3]).max(axis=1) > -2e-6).all()
    bd = np.zeros((6, 4)); lo = [-3.0, -3.0, -2.0]; hi = [5.0, 3.5, 4.0]
    for ax in range(3):
        bd[2 * ax, ax] = 1.0; bd[2 * ax, 3] = -hi[ax]; bd[2 * ax + 1, ax] = -1.0; bd[2 * ax + 1, 3] = lo[ax]
    # The four passes of firi::firi differ between the two sides only through the MVIE optimiser's stopping point, so
    # the comparison that can be exact is made exact: ONE pass (no optimiser between the seed ellipsoid and the planes)
    # through the same library gives the restatement's rows, same count, same order, to rounding ...
    a0, b0 = np.array([0.0, 0.0, 1.0]), np.array([2.0, 0.5, 1.2])
    ok1, hp1 = F.firi(bd, pts, a0, b0, iterations=1)
    one = aa.firi(bd[None], pts[None], a0[None], b0[None], iterations=np.array([1], dtype=np.int32))
    n1 = int(one["n_rows"][0])
    assert ok1 and one["ok"][0] >= 1 and n1 == hp1.shape[0]
    assert np.abs(one["hpoly"][0, :n1] - hp1).max() <= 1e-9 * np.abs(hp1).max()
    # ... and after the four passes both polytopes hold the segment, exclude every point, and have the same volume scale
    # (the facade's is the library's own 4-pass result: identical to the Python mirror)
    ok0, hp0 = F.firi(bd, pts, a0, b0)
    four = aa.firi(bd[None], pts[None], a0[None], b0[None])
    n4 = int(four["n_rows"][0])
    assert ok0 and n4 == hp.shape[0] and np.array_equal(four["hpoly"][0, :n4], hp)
    assert ((pts @ hp0[:, :3].T + hp0[:, 3]).max(axis=1) > -2e-6).all()
    # (row COUNTS after four passes may differ by a plane or two: a plane that is just redundant for one side's ellipsoid
    #  is just not for the other's, whose MVIE optimisation stopped 1e-3 away -- hence no equality here)
    assert abs(n4 - hp0.shape[0]) <= 2
