"""Edge cases of the C ABI on the GPU: empty and ragged batches, maximum sizes (16 pieces, 50-row
polytopes = learning_planner.hpp:40), strides larger than the batch, argument validation."""
import ctypes

import numpy as np
import pytest

from oracle import cbind
from tests.util import random_problem, rel_err

pytestmark = pytest.mark.gpu


def test_empty_batch_is_a_noop(anet_ctx):
    import allocnet_amd as aa
    co, en = aa.minco_solve(np.zeros((0, 3, 3)), np.zeros((0, 3, 3)), np.zeros((0, 7, 3)), np.zeros((0, 8)), 4, ctx=anet_ctx)
    assert co.shape == (0, 8, 3, 8) and en.shape == (0,)
    cost, gP, gT = aa.minco_cost_grad(np.zeros((0, 3, 3)), np.zeros((0, 3, 3)), np.zeros((0, 4, 3)), np.zeros((0, 5)), 3,
                                      penalty=aa.make_penalty(), ctx=anet_ctx)
    assert cost.shape == (0,) and gP.shape == (0, 4, 3)
    assert aa.traj_cost(np.zeros((0, 2, 3, 6)), np.zeros((0, 2)), 3, ctx=anet_ctx).shape == (0,)


@pytest.mark.parametrize("B", [1, 63, 64, 65, 129])
def test_ragged_batches(anet_ctx, B):
    import allocnet_amd as aa
    rng = np.random.default_rng(B)
    head, tail, wps, T = random_problem(rng, B, 8, 3)
    co, en = aa.minco_solve(head, tail, wps, T, 4, ctx=anet_ctx)
    cc, ec = cbind.minco_solve_batch(4, head, tail, wps, T)
    assert rel_err(co, cc) < 1e-9 and rel_err(en, ec) < 1e-9


def test_maximum_sizes(anet_ctx):
    """16 pieces (ANET_MAX_PIECES) for both orders and all boundary conventions; 50-row polytopes."""
    import allocnet_amd as aa
    rng = np.random.default_rng(16)
    for s in (3, 4):
        for c in range(1, s + 1):
            head, tail, wps, T = random_problem(rng, 70, 16, c)
            co, en = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
            cc, ec = cbind.minco_solve_batch(s, head, tail, wps, T)
            assert rel_err(co, cc) < 1e-8 and rel_err(en, ec) < 1e-8, (s, c)
    head, tail, wps, T = random_problem(rng, 9, 5, 3)
    hp = rng.normal(size=(9, 5, 50, 4)); hp[..., 3] += 8.0
    pen = aa.make_penalty(rho=1.0, w_corridor=10.0, w_vel=1.0, w_acc=1.0, res=20, poly_rows=50)
    cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, 4, hpolys=hp, penalty=pen, ctx=anet_ctx)
    assert np.isfinite(cost).all() and np.isfinite(gP).all() and np.isfinite(gT).all()
    with pytest.raises(aa.AnetError):
        aa.minco_cost_grad(head, tail, wps, T, 4, hpolys=np.zeros((9, 5, 51, 4)),
                           penalty=aa.make_penalty(poly_rows=51), ctx=anet_ctx)


def test_argument_validation(anet_ctx):
    import allocnet_amd as aa
    from allocnet_amd import _lib
    h = np.zeros((2, 3, 3)); T = np.ones((2, 17))
    with pytest.raises(aa.AnetError) as ei:                       # > ANET_MAX_PIECES
        aa.minco_solve(h, h, np.zeros((2, 16, 3)), T, 4, ctx=anet_ctx)
    assert ei.value.code == _lib.ANET_ERR_INVALID
    with pytest.raises(aa.AnetError):                             # order 5 does not exist
        aa.minco_solve(h, h, np.zeros((2, 1, 3)), np.ones((2, 2)), 5, ctx=anet_ctx)
    with pytest.raises(aa.AnetError):                             # c > s
        aa.minco_solve(np.zeros((2, 3, 4)), np.zeros((2, 3, 4)), np.zeros((2, 1, 3)), np.ones((2, 2)), 3, ctx=anet_ctx)
    with pytest.raises(aa.AnetError):                             # smoothing must be positive
        aa.minco_cost_grad(h, h, np.zeros((2, 1, 3)), np.ones((2, 2)), 3, penalty=aa.make_penalty(smooth_mu=0.0),
                           ctx=anet_ctx)
    lib = _lib.load()
    rc = lib.anet_minco_solve(anet_ctx.handle, 4, 3, 2, 2, None, None, None, None, None, None)     # NULL inputs
    assert rc == _lib.ANET_ERR_INVALID and b"NULL" in lib.anet_last_error(anet_ctx.handle)
    # polytope depth: negative batch, no rows, NULL; an empty batch is a no-op; a polytope of padding rows only is empty
    import ctypes
    d = np.zeros(1)
    assert lib.anet_polytope_depth(anet_ctx.handle, -1, 4, None, 1, None, None) == _lib.ANET_ERR_INVALID
    assert lib.anet_polytope_depth(anet_ctx.handle, 1, 0, None, 1, None, None) == _lib.ANET_ERR_INVALID
    assert lib.anet_polytope_depth(anet_ctx.handle, 1, 4, None, 1, d.ctypes.data_as(ctypes.c_void_p), None) == _lib.ANET_ERR_INVALID
    assert lib.anet_polytope_depth(anet_ctx.handle, 0, 4, None, 1, None, None) == _lib.ANET_OK
    depth, _ = aa.polytope_depth(np.zeros((2, 5, 4)), ctx=anet_ctx)
    assert np.isneginf(depth).all()
    # a launch order with out-of-range entries: those workgroups do nothing, the problems they would have taken keep
    # whatever the caller put into status (torch.empty here, so only the solved ones are checked)
    import torch
    from tools.bench_configs import to_bm
    rng = np.random.default_rng(0)
    head, tail, wps, T = random_problem(rng, 8, 4, 3, rest=True)
    dev = torch.device("cuda", 0)
    ld = aa.recommended_ld(8)
    th, tt, tw, tT = (to_bm(torch, x, 8, ld, dev) for x in (head, tail, wps, T))
    order = torch.tensor([0, 1, 2, 3, 4, 5, 99, -7], dtype=torch.int32, device=dev)
    r = aa.lbfgs_minco_dev(th, tt, tw, tT, 3, 3, 4, 8, max_evals=50, opt=1, launch_order=order, ctx=anet_ctx)
    torch.cuda.synchronize()
    assert (r["evals"][:6].cpu().numpy() >= 1).all() and (r["status"][:6].cpu().numpy() != 0x7ffffff0).all()


def test_stride_larger_than_batch(anet_ctx):
    """Device entry point with ld > batch (the recommended padded stride) leaves the padding untouched."""
    import torch
    import allocnet_amd as aa
    B, N, s, c = 300, 8, 4, 3
    ld = aa.recommended_ld(B) + 128
    assert ld % 64 == 0 and ld >= B
    rng = np.random.default_rng(1)
    head, tail, wps, T = random_problem(rng, B, N, c)
    dev = torch.device("cuda", 0)

    def bm(a):
        t = torch.full((a.reshape(B, -1).shape[1], ld), -7.0, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(np.ascontiguousarray(a.reshape(B, -1).T)).to(dev)
        return t
    co = torch.full((N * 3 * 8, ld), -7.0, device=dev, dtype=torch.float64)
    en = torch.full((ld,), -7.0, device=dev, dtype=torch.float64)
    aa.minco_solve_dev(bm(head), bm(tail), bm(wps), bm(T), s, c, N, B, coeffs=co, energy=en, ctx=anet_ctx)
    torch.cuda.synchronize()
    cc, ec = cbind.minco_solve_batch(s, head, tail, wps, T)
    got = co[:, :B].T.cpu().numpy().reshape(B, N, 3, 8)
    assert rel_err(got, cc) < 1e-9 and rel_err(en[:B].cpu().numpy(), ec) < 1e-9
    assert (co[:, B:] == -7.0).all() and (en[B:] == -7.0).all()


@pytest.mark.parametrize("s,c,N", [(4, 3, 5), (3, 3, 5), (3, 3, 12), (4, 4, 16), (3, 2, 3), (4, 3, 8)])
def test_lane_per_trajectory_kernels_above_the_axis_threshold(anet_ctx, s, c, N):
    """Batches above 16384 use the lane-per-trajectory kernels (generic and specialised instantiations);
    smaller ones the axis-parallel kernels.  Both must agree with the oracle and with each other."""
    import allocnet_amd as aa
    rng = np.random.default_rng(7 * N + s)
    B = 16384 + 700
    head, tail, wps, T = random_problem(rng, B, N, c)
    co, en = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    idx = np.arange(0, B, 997)
    cc, ec = cbind.minco_solve_batch(s, head[idx], tail[idx], wps[idx], T[idx])
    assert rel_err(co[idx], cc) < 1e-9 and rel_err(en[idx], ec) < 1e-9
    co2, en2 = aa.minco_solve(head[:3000], tail[:3000], wps[:3000], T[:3000], s, ctx=anet_ctx)   # axis-parallel path
    assert rel_err(co2, co[:3000]) < 1e-11 and rel_err(en2, en[:3000]) < 1e-11


@pytest.mark.parametrize("s,c,N", [(4, 3, 8), (3, 3, 16), (4, 3, 5), (3, 3, 5), (4, 4, 3), (3, 2, 6), (2, 2, 12)])
def test_gradient_propagation_lane_and_axis_kernels_agree(anet_ctx, s, c, N):
    """propogateGrad has the same two launch shapes as the solve: lane per trajectory above 16384,
    lane per (trajectory, axis) below.  The small-batch result is pinned against the numpy oracle in
    test_grad_gpu.py; here the two shapes are compared on the same trajectories."""
    import allocnet_amd as aa
    rng = np.random.default_rng(91 * N + s)
    B = 16384 + 300
    head, tail, wps, T = random_problem(rng, B, N, c)
    pen = aa.make_penalty(rho=1.3, w_vel=20.0, w_acc=8.0, smooth_mu=0.05, max_vel=1.0, max_acc=1.5, res=5)
    cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, penalty=pen, ctx=anet_ctx)            # lane kernels
    n = 2000
    cost2, gP2, gT2 = aa.minco_cost_grad(head[:n], tail[:n], wps[:n], T[:n], s, penalty=pen, ctx=anet_ctx)
    assert np.isfinite(cost).all() and np.isfinite(gT).all()
    assert rel_err(cost2, cost[:n]) < 1e-12
    assert np.abs(gP2 - gP[:n]).max() <= 1e-10 * max(1.0, np.abs(gP).max())
    assert np.abs(gT2 - gT[:n]).max() <= 1e-10 * max(1.0, np.abs(gT).max())
    # rows past the batch in the last, partly filled wave stay untouched (idle lanes store nothing)
    odd = 21 * 5 + 4
    c3, gP3, gT3 = aa.minco_cost_grad(head[:odd], tail[:odd], wps[:odd], T[:odd], s, penalty=pen, ctx=anet_ctx)
    assert np.abs(gT3 - gT[:odd]).max() <= 1e-10 * max(1.0, np.abs(gT).max())
    assert np.abs(gP3 - gP[:odd]).max() <= 1e-10 * max(1.0, np.abs(gP).max())


@pytest.mark.parametrize("s,c,N,M", [(4, 3, 8, 16), (3, 3, 5, 10), (3, 3, 16, 9)])
def test_penalty_kernel_launch_shapes_agree(anet_ctx, s, c, N, M):
    """k_piece_grad has three launch shapes: one lane per (trajectory, piece) above 16384 trajectories, two lanes
    per pair below, and -- up to 16384 pairs -- the samples spread over the four waves of a workgroup as well.
    Corridor rows, box rows and the energy part are split differently in each; the sums must agree."""
    import allocnet_amd as aa
    from tests.util import corridor_problem
    rng = np.random.default_rng(5 * N + s + M)
    B = 16384 + 200
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    hp[:, :, :, 3] -= 0.8          # tighten the corridors so that rows are active
    pen = aa.make_penalty(rho=2.0, w_corridor=300.0, w_vel=20.0, w_acc=8.0, smooth_mu=0.05, max_vel=1.5, max_acc=2.0,
                          res=7, poly_rows=M)
    full = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)            # lane per pair
    n2 = 16384 // N + 40                                                                              # two lanes
    n4 = 16384 // N - 40                                                                              # + sample split
    assert (full[0] > 0).all()
    for n in (n2, n4, 1, 33):
        part = aa.minco_cost_grad(head[:n], tail[:n], wps[:n], T[:n], s, hpolys=hp[:n], penalty=pen, ctx=anet_ctx)
        assert rel_err(part[0], full[0][:n]) < 1e-12
        for g, gf in zip(part[1:], full[1:]):
            assert np.abs(g - gf[:n]).max() <= 1e-10 * max(1.0, np.abs(gf).max())


def test_device_entry_points_capture_into_a_hip_graph(anet_ctx):
    """The *_dev entry points only enqueue work on the caller's stream (no allocation, no synchronisation once the
    basis table of an (order, res) exists), so a caller can capture them into a hipGraph -- what bench.py does for
    the literal BASELINE configs[1] batch of 1024 trajectories, where launches, not bytes, are the cost.  Replays
    reproduce the eager results bit for bit: coefficient solve and cost + gradient evaluation on parallel chains."""
    import torch
    import allocnet_amd as aa
    from tests.util import corridor_problem
    dev = torch.device("cuda", 0)
    s, c, N, M, B = 4, 3, 8, 16, 1024
    ld = aa.recommended_ld(B)
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(9), B, N, c, M)

    def bm(x):
        f = np.ascontiguousarray(x.reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(dev)
        return t
    th, tt, tw, tT, thp = (bm(x) for x in (head, tail, wps, T, hp))
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=20, poly_rows=M)
    co = torch.zeros(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64); en = torch.zeros(ld, device=dev, dtype=torch.float64)
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=anet_ctx)   # eager: builds the table
    aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=anet_ctx)
    torch.cuda.synchronize()
    ref = (co.clone(), en.clone(), cost.clone(), gP.clone(), gT.clone())
    for t in (co, en, cost, gP, gT):
        t.zero_()
    side = [torch.cuda.Stream(device=dev) for _ in range(2)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream(device=dev)):
        cur = torch.cuda.current_stream(dev)
        for st in side:
            st.wait_stream(cur)
        aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, stream=side[0].cuda_stream, ctx=anet_ctx)
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT,
                               stream=side[1].cuda_stream, ctx=anet_ctx)
        for st in side:
            cur.wait_stream(st)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for got, want in zip((co, en, cost, gP, gT), ref):
        assert torch.equal(got[..., :B], want[..., :B])


def test_one_launch_lbfgs_captures_into_a_hip_graph(anet_ctx):
    """The one-launch L-BFGS (anet_lbfgs_minco_dev without ANET_OPT_LOCKSTEP) only enqueues kernels -- no host test for
    completion, no allocation -- so a receding-horizon loop can capture "restore the initial guess, optimise" once and
    replay it; every replay reproduces the eager run bit for bit."""
    import torch
    import allocnet_amd as aa
    from tests.util import corridor_problem
    dev = torch.device("cuda", 0)
    s, c, N, M, B = 3, 3, 6, 8, 200
    ld = aa.recommended_ld(B)
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(4), B, N, c, M)

    def bm(x):
        f = np.ascontiguousarray(x.reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(dev)
        return t
    th, tt, tw0, tT0, thp = (bm(x) for x in (head, tail, wps, T, hp))
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=10, poly_rows=M)
    tw, tT = tw0.clone(), tT0.clone()
    eager = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, max_evals=300, opt=3, ctx=anet_ctx)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in eager.items()}
    ref_w, ref_T = tw.clone(), tT.clone()
    cap_stream = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap_stream):
        tw.copy_(tw0); tT.copy_(tT0)
        out = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, max_evals=300, opt=3,
                                 stream=torch.cuda.current_stream(dev).cuda_stream, ctx=anet_ctx)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(tw[:, :B], ref_w[:, :B]) and torch.equal(tT[:, :B], ref_T[:, :B])
    assert (ref["evals"] > 1).all()


def test_randomised_soak_of_cost_and_gradients(anet_ctx):
    """tests/soak/soak_cost_grad.py: 80 random configurations -- orders 3 / 4, 2..s boundary derivatives, 1..16 pieces, 0..16 corridor
    rows, 1..33 samples per piece, batch sizes on both sides of every launch-shape threshold (63 / 64 / 65, 2047 / 2048 / 2049,
    16384 / 16385), random weights, limits and smoothing -- each against the C restatement (classic banded LU + adjoint) on a
    random sample of its trajectories: cost 1e-9, gradients 1e-7."""
    from tests.soak.soak_cost_grad import run
    worst = run(80, seed=2024, ctx=anet_ctx, verbose=False)
    assert worst["cost"] <= 1e-10 and worst["gT"] <= 1e-9 and worst["gP"] <= 1e-9, worst
