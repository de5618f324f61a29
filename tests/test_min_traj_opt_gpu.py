"""The Python training-side mirrors (MinTrajOpt.update, OsqpLayer.forward) against the fixtures made by
importing the reference's MinTrajOpt with the same inputs (50x4xseq_len zero-padded polytopes)."""
import numpy as np
import pytest

from tests.util import golden_files
from tests.golden.make_golden import make_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", golden_files())
def test_min_traj_opt_update_matches_reference(anet_ctx, path):
    import allocnet_amd as aa
    from tests.test_qp_assembly_gpu import _expand
    d = np.load(path)
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    hp50 = np.zeros((50, 4, N)); hp50[:16] = d["hpolys"]
    opt = aa.MinTrajOpt(make_params(s, res), ctx=anet_ctx)
    opt.update(d["state"], hp50, d["T"], phase=phase, seq_len=N)
    assert opt.seg == N and opt.var_num == 3 * 2 * s * N
    assert abs(opt.path_length - float(d["path_length"])) < 1e-12
    Q, A, b, G1, h1, G2, h2 = opt.params
    Gref, href = _expand(d, N, 2 * s)
    n1 = d["h1"].shape[0]
    for got, ref in [(Q, d["Q"]), (A, d["A"]), (b, d["b"]), (G1, Gref[:n1]), (h1, d["h1"]), (G2, Gref[n1:]), (h2, d["h2"])]:
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["tf_snap_n3", "tf_jerk_n5", "tf_snap_n2"])
def test_min_traj_opt_use_time_factor_matches_reference(anet_ctx, name):
    """use_time_factor = True (min_traj_opt.py:113-136, 185-296): waypoints, time lower bounds, Times, ref_time_factor,
    path length and the matrices assembled with those times, against fixtures made by the imported reference."""
    import os
    import allocnet_amd as aa
    from tests.util import GOLDEN
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    hp50 = np.zeros((50, 4, 5)); hp50[:16] = d["hpolys"]
    opt = aa.MinTrajOpt(make_params(s, res, vmax=4.5, amax=7.0, use_time_factor=True), ctx=anet_ctx)
    opt.update(d["state"], hp50, d["factor"], phase=phase, traj_times=d["ref_times"], seq_len=5)
    assert opt.seg == N
    assert np.abs(opt.waypts - d["waypts"]).max() <= 1e-9
    assert np.array_equal(opt.time_lb, d["time_lb"])                 # float32 roundings of the reference reproduced
    assert np.abs(opt.Times - d["Times"]).max() <= 1e-15
    assert abs(opt.path_length - float(d["path_length"])) <= 1e-12
    assert np.allclose(opt.ref_time_factor[:N], d["ref_time_factor"][:N], rtol=1e-15, atol=0)
    Q, A, b, G1, h1, G2, h2 = opt.params
    for got, ref in [(Q, d["Q"]), (A, d["A"]), (b, d["b"]), (h1, d["h1"]), (h2, d["h2"])]:
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max())
    assert abs(G1.sum() - float(d["G1_sum"])) <= 1e-12 * max(1.0, abs(float(d["G1_sum"])))
    assert abs(G2.sum() - float(d["G2_sum"])) <= 1e-12 * max(1.0, abs(float(d["G2_sum"])))


def test_osqp_layer_forward(anet_ctx):
    import allocnet_amd as aa
    from tests.test_qp_solve_gpu import _corridor_problem
    rng = np.random.default_rng(21)
    ini, fin, hp, T = _corridor_problem(rng, 3, 8)
    state = np.zeros((9, 2))
    state[:, 0] = ini.reshape(-1); state[:, 1] = fin.reshape(-1)
    hp50 = np.zeros((50, 4, 5))
    for i in range(3):
        rows = hp[i][np.abs(hp[i]).sum(axis=1) > 0]
        hp50[:rows.shape[0], :, i] = rows
    times = np.r_[T, 0.0, 0.0]
    opt = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx)
    opt.update(state, hp50, times, phase=2, seq_len=5)
    assert opt.seg == 3
    layer = aa.OsqpLayer(ctx=anet_ctx)
    z, obj1, objt, objc, pad = layer.forward(opt)
    assert z is not None and z.shape == (3 * 3 * 8,) and objt is None
    assert abs(obj1 - T.sum() / 3) < 1e-12 and pad == 0.0
    Q = opt.params[0]
    assert abs(objc - 0.5 * z @ Q @ z / opt.path_length) <= 1e-9 * max(1.0, objc)
    # time gradient == finite difference of 1/2 z'Q(T)z / path_length with z held fixed (the reference's backward)
    h = 1e-6
    for i in range(3):
        tp = times.copy(); tp[i] += h; tm = times.copy(); tm[i] -= h
        op = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx); op.update(state, hp50, tp, phase=2, seq_len=5)
        om = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx); om.update(state, hp50, tm, phase=2, seq_len=5)
        fd = (0.5 * z @ op.params[0] @ z - 0.5 * z @ om.params[0] @ z) / (2 * h) / opt.path_length
        assert abs(fd - layer.time_grad[i]) <= 1e-5 * max(1.0, abs(fd))
    # implicit gradient == finite difference of the OPTIMAL objc (z re-solved), to the solve tolerance
    imp = layer.implicit_time_grad.copy()
    assert imp.shape == times.shape and (imp[3:] == 0).all()
    h = 1e-3
    for i in range(3):
        vals = []
        for sg in (+1, -1):
            tt = times.copy(); tt[i] += sg * h
            o = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx); o.update(state, hp50, tt, phase=2, seq_len=5)
            vals.append(aa.OsqpLayer(ctx=anet_ctx).forward(o)[3])
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - imp[i]) <= 2e-3 * np.abs(imp[:3]).max() + 1e-6, (i, fd, imp)      # (10 % with the ADMM default of round 1)
    # OsqpLayer.backward: the pass the KKT hook of layers.py:129-141 is after, carried to the durations.  With
    # grad_z = Q z / path_length (the gradient of objc w.r.t. the solution) plus the explicit part (time_grad) it is
    # the implicit gradient of objc; and the position of the middle knot, a loss that is NOT the objective, moves
    # with the durations as central differences of re-solved QPs say
    gz = (Q @ z) / opt.path_length
    total = layer.backward(opt, gz) + layer.time_grad
    assert np.abs(total[:3] - imp[:3]).max() <= 1e-3 * np.abs(imp[:3]).max() + 1e-6 and (total[3:] == 0).all()
    wsel = np.zeros(3 * 3 * 8); wsel[(1 * 3 + 0) * 8 + 7] = 1.0          # x-position at the start of piece 1 (constant term)
    gb = layer.backward(opt, wsel)
    for i in range(3):
        vals = []
        for sg in (+1, -1):
            tt = times.copy(); tt[i] += sg * 1e-4
            o = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx); o.update(state, hp50, tt, phase=2, seq_len=5)
            lay = aa.OsqpLayer(ctx=anet_ctx); lay._settings = aa.qp_settings(eps_abs=1e-10, eps_rel=1e-10)
            vals.append(lay.forward(o)[0] @ wsel)
        fd = (vals[0] - vals[1]) / 2e-4
        assert abs(fd - gb[i]) <= 1e-4 * np.abs(gb[:3]).max() + 1e-7, (i, fd, gb)
    # the interior-point method behind the same layer: same objective (to OSQP's tolerance), sharper gradient
    lay2 = aa.OsqpLayer(ctx=anet_ctx, method=aa.qp.QP_METHOD_INTERIOR_POINT)
    zi, _, _, objc_i, _ = lay2.forward(opt)
    assert zi is not None and abs(objc_i - objc) <= 2e-2 * max(1.0, objc)
    assert np.abs(lay2.implicit_time_grad[:3] - imp[:3]).max() <= 0.1 * np.abs(imp[:3]).max() + 1e-3
    z2, o1, ot, oc, stl = layer.forward4lstm(opt, np.array([0.1, 0.2, 0.9, 0.95, 0.99]), seq_len=5)
    assert z2 is not None and abs(oc - objc) <= 1e-2 * max(1.0, objc)
    # stop-token loss (layers.py:186-204): BCE against [0]*(seg-1) + [1]*(seq_len-seg+1) plus 5.0 per premature / late
    # token at threshold 0.42 -- three segments here, so the ground truth is [0, 0, 1, 1, 1]
    import torch
    gt = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0], dtype=torch.float64)
    hand = -(np.log(0.9) + np.log(0.8) + np.log(0.9) + np.log(0.95) + np.log(0.99)) / 5.0
    assert abs(stl - hand) <= 1e-14
    for pred, extra in (([0.1, 0.2, 0.9, 0.95, 0.99], 0.0), ([0.5, 0.2, 0.9, 0.3, 0.99], 10.0),     # one premature, one late
                        ([0.43, 0.9, 0.41, 0.1, 0.2], 25.0), ([0.0, 0.42, 0.42, 1.0, 1.0], 0.0)):    # at the threshold: neither
        want = float(torch.nn.BCELoss()(torch.tensor(pred, dtype=torch.float64), gt)) + extra
        got = layer.forward4lstm(opt, np.array(pred), seq_len=5)[4]
        assert abs(got - want) <= 1e-12 * max(1.0, want), (pred, got, want)


def test_osqp_layer_forward_batch(anet_ctx):
    """forward_batch == forward sample by sample (mixed segment counts are grouped, order preserved)."""
    import allocnet_amd as aa
    from tests.util import qp_corridor_problem
    rng = np.random.default_rng(33)
    opts = []
    for seg in (3, 2, 3, 4, 2):
        ini, fin, hp, T = qp_corridor_problem(rng, seg, 8)
        state = np.zeros((9, 2)); state[:, 0] = ini.reshape(-1); state[:, 1] = fin.reshape(-1)
        hp50 = np.zeros((50, 4, 5))
        for i in range(seg):
            rows = hp[i][np.abs(hp[i]).sum(axis=1) > 0]
            hp50[:rows.shape[0], :, i] = rows
        times = np.r_[T, np.zeros(5 - seg)]
        o = aa.MinTrajOpt(make_params(4, 10, vmax=3.0, amax=4.0), ctx=anet_ctx)
        o.update(state, hp50, times, phase=2, seq_len=5)
        opts.append(o)
    layer = aa.OsqpLayer(ctx=anet_ctx, method=aa.qp.QP_METHOD_INTERIOR_POINT)
    res, tg, itg = layer.forward_batch(opts)
    assert len(res) == len(opts)
    for i, o in enumerate(opts):
        one = aa.OsqpLayer(ctx=anet_ctx, method=aa.qp.QP_METHOD_INTERIOR_POINT)
        z, o1, ot, oc, pad = one.forward(o)
        zb, o1b, otb, ocb, padb = res[i]
        assert (z is None) == (zb is None) and o1 == o1b and pad == padb
        if z is not None:
            assert np.abs(z - zb).max() <= 1e-7 * max(1.0, np.abs(z).max()) and abs(oc - ocb) <= 1e-9 * max(1.0, abs(oc))
            assert np.abs(one.time_grad - tg[i]).max() <= 1e-7 * max(1.0, np.abs(tg[i]).max())
            assert np.abs(one.implicit_time_grad - itg[i]).max() <= 1e-6 * max(1.0, np.abs(itg[i]).max())
    # backward_batch == backward sample by sample (a skipped sample stays None)
    gzs = [None if (r[0] is None or i == 1) else np.cos(np.arange(r[0].size) * 0.37) for i, r in enumerate(res)]
    gb = layer.backward_batch(opts, gzs)
    for i, o in enumerate(opts):
        if gzs[i] is None:
            assert gb[i] is None
        else:
            one = layer.backward(o, gzs[i])
            assert np.abs(one - gb[i]).max() <= 1e-7 * max(1.0, np.abs(one).max())


@pytest.mark.parametrize("name", ["layers_snap_n3", "layers_jerk_n4", "layers_snap_n5"])
def test_osqp_layer_against_the_references_own_layers_py(anet_ctx, name):
    """Fixtures made by RUNNING network/utils/learning/layers.py (OsqpLayer.forward :51-151, forward4lstm :153-247, its KKT hook
    and torch.autograd through the reference's Q(T)) on qp_traj objects built by the reference's MinTrajOpt.update, with
    `osqp.OSQP` replaced by an injector that returns the float64 optimum (tests/golden/make_golden.py main_layers: this pins the
    code around the solve, not OSQP's iterates).  Compared: every loss term of both entry points in the solved and the unsolved
    branch, the solution, and d objc / d Times as the reference's backward pass delivers it."""
    import os
    import allocnet_amd as aa
    from tests.util import GOLDEN
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    hp50 = np.zeros((50, 4, 5)); hp50[:16] = d["hpolys"]
    opt = aa.MinTrajOpt(make_params(s, res), ctx=anet_ctx)
    opt.update(d["state"], hp50, d["Times"], phase=phase, traj_times=d["ref_times"], seq_len=5)
    assert opt.seg == N and abs(opt.path_length - float(d["path_length"])) <= 1e-12
    layer = aa.OsqpLayer(ctx=anet_ctx)                      # interior point: the optimum, as the injector returns it
    z, obj1, objt, objc, pad = layer.forward(opt)
    assert z is not None and objt is None
    assert abs(obj1 - float(d["forward_solved_obj1"])) <= 1e-15 * max(1.0, abs(obj1))
    assert abs(pad - float(d["forward_solved_last"])) <= 1e-15
    zr = d["forward_solved_z"]
    assert np.abs(z - zr).max() <= 2e-5 * np.abs(zr).max()                     # two interior-point solves of the same QP
    assert abs(objc - float(d["forward_solved_objc"])) <= 1e-6 * float(d["forward_solved_objc"])
    # the reference's backward pass: d objc / d Times = 1/2 z'(dQ/dT)z / path_length, z detached (zeros beyond the segments used)
    gr = d["forward_solved_dobjc_dT"]
    assert layer.time_grad.shape == gr.shape and (layer.time_grad[N:] == 0).all() and (gr[N:] == 0).all()
    assert np.abs(layer.time_grad - gr).max() <= 2e-5 * np.abs(gr).max()
    # forward4lstm: same solve, the stop-token loss instead of the padding loss
    pred = d["pred_stop_tokens"]
    z4, o1, ot, oc, stl = layer.forward4lstm(opt, pred.astype(np.float32), seq_len=5)
    assert z4 is not None and ot is None and abs(o1 - float(d["forward4lstm_solved_obj1"])) <= 1e-15 * max(1.0, abs(o1))
    assert abs(oc - float(d["forward4lstm_solved_objc"])) <= 1e-6 * float(d["forward4lstm_solved_objc"])
    # (the reference's BCELoss runs in float32 on float32 tokens: agreement to float32 rounding)
    assert abs(stl - float(d["forward4lstm_solved_last"])) <= 2e-6 * max(1.0, abs(stl)), (stl, float(d["forward4lstm_solved_last"]))
    assert np.abs(layer.time_grad - d["forward4lstm_solved_dobjc_dT"]).max() <= 2e-5 * np.abs(gr).max()
    # the unsolved branches (layers.py:98-116, 206-215): an infeasible twin of the problem -- limits no trajectory of these
    # durations can meet -- gives None / None and the reference-time loss; the loss terms do not depend on WHY the solve failed
    bad = aa.MinTrajOpt(make_params(s, res, vmax=1e-3, amax=1e-3, vmax1=1e-3, amax1=1e-3), ctx=anet_ctx)
    bad.update(d["state"], hp50, d["Times"], phase=phase, traj_times=d["ref_times"], seq_len=5)
    zb, b1, bt, bc, bpad = layer.forward(bad)
    assert zb is None and bc is None and layer.time_grad is None
    assert abs(b1 - float(d["forward_unsolved_obj1"])) <= 1e-15 * max(1.0, abs(b1))
    assert abs(bt - float(d["forward_unsolved_objt"])) <= 1e-14 * max(1.0, abs(bt)) and abs(bpad - float(d["forward_unsolved_last"])) <= 1e-15
    zb, b1, bt, bc, bstl = layer.forward4lstm(bad, pred.astype(np.float32), seq_len=5)
    assert zb is None and bc is None
    assert abs(bt - float(d["forward4lstm_unsolved_objt"])) <= 1e-14 * max(1.0, abs(bt))
    assert abs(bstl - float(d["forward4lstm_unsolved_last"])) <= 2e-6 * max(1.0, abs(bstl))
    # analytic gradients of the plain loss terms, as torch.autograd gave them through the reference's code
    g1 = np.zeros(5); g1[:N] = 1.0 / N
    assert np.array_equal(d["forward_solved_dobj1_dT"], g1)
    if N < 5:
        gp = np.zeros(5); gp[N:] = 2.0 * d["Times"][N:] / (5 - N)
        assert np.abs(d["dpadding_dT"] - gp).max() <= 1e-15
