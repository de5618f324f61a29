"""GPU parity: Trajectory/Piece evaluation and getTrajCost vs the reference-generated fixtures
(network/utils/trajectory.py outputs) and the C/numpy oracle."""
import numpy as np
import pytest

from oracle import minco_np as onp
from oracle import cbind
from tests.util import golden_files, random_problem, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", golden_files())
def test_eval_matches_reference_fixture(anet_ctx, path):
    import allocnet_amd as aa
    d = np.load(path)
    s, N = int(d["order"]), int(d["N"]); D = 2 * s
    z = d["z_eq"].reshape(N, 3, D)
    traj = aa.Trajectory(list(d["T"]), list(z), ctx=anet_ctx)
    assert traj.getPieceNum() == N
    assert abs(traj.getTotalDuration() - d["T"].sum()) < 1e-13
    for k, key in [(0, "eval_pos"), (1, "eval_vel"), (2, "eval_acc")]:
        got = [traj.getPos, traj.getVel, traj.getAcc][k](d["eval_t"])
        ref = d[key]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # getTrajCost with the reference constant: 1/2 z'Qz of the fixture
    assert abs(traj.getTrajCost(s) - d["e_eq"]) <= 1e-12 * max(1.0, abs(d["e_eq"]))


@pytest.mark.parametrize("s,N", [(4, 8), (3, 16), (3, 5), (4, 1), (2, 3)])
def test_eval_batched_vs_oracle(anet_ctx, s, N):
    import allocnet_amd as aa
    rng = np.random.default_rng(7 + s + N)
    B, D, nq = 130, 2 * s, 9
    coeffs = rng.normal(size=(B, N, 3, D))
    T = rng.uniform(0.4, 2.0, size=(B, N))
    # include t = 0, exact knots, the end, and beyond the end (clamp branch of locatePieceIdx)
    tq = rng.uniform(0.0, 1.0, size=(B, nq)) * T.sum(axis=1, keepdims=True)
    tq[:, 0] = 0.0
    tq[:, 1] = T[:, 0]
    tq[:, 2] = T.sum(axis=1)
    tq[:, 3] = T.sum(axis=1) + 0.37
    for d in range(4):
        got = aa.traj_eval(coeffs, T, tq, d, ctx=anet_ctx)
        for b in range(0, B, 13):
            for q in range(nq):
                ref = onp.traj_eval(coeffs[b], T[b], tq[b, q], d)
                assert np.abs(got[b, q] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
                idx, tl = onp.locate(T[b], tq[b, q])
                refc = cbind.piece_eval(coeffs[b, idx], tl, d)
                assert np.abs(got[b, q] - refc).max() <= 1e-11 * max(1.0, np.abs(refc).max())
    for m34 in (1400.0, 1440.0):
        got = aa.traj_cost(coeffs, T, s, m34, ctx=anet_ctx)
        for b in range(0, B, 7):
            if s > 2:
                assert abs(got[b] - onp.traj_cost(coeffs[b], T[b], s, m34)) <= 1e-11 * abs(got[b])
                assert abs(got[b] - cbind.traj_cost(s, coeffs[b], T[b], m34)) <= 1e-11 * abs(got[b])


@pytest.mark.parametrize("s", [2, 3, 4])
def test_normalized_coefficient_matrices(anet_ctx, s):
    """Piece::normalizePosCoeffMat / normalizeVelCoeffMat / normalizeAccCoeffMat (trajectory.hpp:135-171): a batch of pieces against
    the loop-for-loop restatement, bit for bit (the same products in the same order), and what they mean: evaluated at
    tau = t / duration the normalised polynomials are the position, duration * velocity and duration^2 * acceleration."""
    import allocnet_amd as aa
    from allocnet_amd.trajectory import piece_normalized_coeffs, Piece
    rng = np.random.default_rng(40 + s)
    P, D = 300, 2 * s
    cm = rng.normal(size=(P, 3, D))
    T = rng.uniform(0.2, 3.0, size=P)
    for d in range(3):
        got = piece_normalized_coeffs(cm, T, d, ctx=anet_ctx)
        assert got.shape == (P, 3, D - d)
        for p in range(P):
            assert np.array_equal(got[p], onp.piece_normalized_coeffs(cm[p], T[p], d)), (d, p)
    pc = Piece(T[7], cm[7], ctx=anet_ctx)
    tau = 0.37
    for d, mat in ((0, pc.normalizePosCoeffMat()), (1, pc.normalizeVelCoeffMat()), (2, pc.normalizeAccCoeffMat())):
        val = np.array([np.polyval(mat[ax], tau) for ax in range(3)])
        ref = onp.piece_eval(cm[7], tau * T[7], d) * T[7] ** d
        assert np.abs(val - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert piece_normalized_coeffs(np.zeros((0, 3, D)), np.zeros(0), 1, ctx=anet_ctx).shape == (0, 3, D - 1)
    with pytest.raises(Exception):
        piece_normalized_coeffs(cm, T, 3, ctx=anet_ctx)


def test_solve_then_cost_consistency(anet_ctx):
    """energy from the solve == 2 x getTrajCost(1440) of its coefficients; junction accessors."""
    import allocnet_amd as aa
    rng = np.random.default_rng(3)
    head, tail, wps, T = random_problem(rng, 300, 8, 3)
    coeffs, energy = aa.minco_solve(head, tail, wps, T, 4, ctx=anet_ctx)
    cost = aa.traj_cost(coeffs, T, 4, 1440.0, ctx=anet_ctx)
    assert rel_err(2.0 * cost, energy) < 1e-10
    traj = aa.Trajectory(list(T[0]), list(coeffs[0]), ctx=anet_ctx)
    pos = traj.getPositions()
    assert np.abs(pos[:, 0] - head[0, :, 0]).max() < 1e-9
    assert np.abs(pos[:, 1:8] - wps[0].T).max() < 1e-9
    assert np.abs(pos[:, 8] - tail[0, :, 0]).max() < 1e-8
    assert np.abs(traj.getJuncVel(0) - head[0, :, 1]).max() < 1e-9
    assert np.abs(traj.getJuncAcc(8) - tail[0, :, 2]).max() < 1e-7
    idx, tl = traj.locatePieceIdx(T[0, 0] + 0.25 * T[0, 1])
    assert idx == 1 and abs(tl - 0.25 * T[0, 1]) < 1e-12


@pytest.mark.parametrize("path", golden_files())
def test_time_gradient_matches_reference_autograd(anet_ctx, path):
    """d(1/2 z'Q(T)z)/dT with z detached: the fixture value comes from torch.autograd through the
    REFERENCE's own Q(T) assembly (tests/golden/make_golden.py step 4), i.e. the gradient its training
    sends to the segment times (layers.py:121,143-147)."""
    import allocnet_amd as aa
    d = np.load(path)
    s, N = int(d["order"]), int(d["N"]); D = 2 * s
    z = d["z_eq"].reshape(1, N, 3, D)
    g = aa.traj_cost_grad_T(z, d["T"][None], m34=1400.0, ctx=anet_ctx)[0]
    ref = d["dcost_dT"]
    assert np.abs(g - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
