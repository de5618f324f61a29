"""GPU getMaxVelRate / getMaxAccRate / checkMax*Rate (trajectory.hpp:177-314, 576-630) vs the numpy
restatement (companion-matrix roots) and vs dense sampling of the GPU's own evaluation."""
import numpy as np
import pytest

from oracle import minco_np as onp
from tests.util import random_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s,N", [(4, 8), (3, 5), (3, 16), (2, 3)])
def test_max_rate_matches_oracle(anet_ctx, s, N):
    import allocnet_amd as aa
    rng = np.random.default_rng(3 * s + N)
    B = 70
    head, tail, wps, T = random_problem(rng, B, N, min(3, s))
    coeffs, _ = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    # make it harder: add random high-order wiggles to some trajectories (more interior extrema)
    coeffs[::3] += rng.normal(size=coeffs[::3].shape) * 0.05
    for which in (1, 2):
        got = aa.traj_max_rate(coeffs, T, which, ctx=anet_ctx)
        for b in range(0, B, 5):
            for i in range(N):
                ref = onp.piece_max_rate(coeffs[b, i], T[b, i], which)
                assert abs(got[b, i] - ref) <= 1e-8 * max(1.0, ref), (b, i, which, got[b, i], ref)
        # never below a dense sampling of the same polynomial
        for b in range(0, B, 9):
            tq = np.linspace(0, T[b].sum(), 400)[None]
            v = aa.traj_eval(coeffs[b:b + 1], T[b:b + 1], tq, which, ctx=anet_ctx)[0]
            assert np.linalg.norm(v, axis=1).max() <= got[b].max() * (1 + 1e-9) + 1e-12


def test_max_rate_edge_cases_and_class_methods(anet_ctx):
    import allocnet_amd as aa
    D = 6
    cm = np.zeros((3, D))
    cm[:, D - 2] = [1.0, -2.0, 0.5]          # constant velocity, zero acceleration
    cm[:, D - 1] = [0.3, 0.1, 0.0]
    piece = aa.Piece(1.7, cm, ctx=anet_ctx)
    assert abs(piece.getMaxVelRate() - np.linalg.norm([1.0, -2.0, 0.5])) < 1e-12
    assert piece.getMaxAccRate() == 0.0
    assert piece.checkMaxVelRate(2.5) and not piece.checkMaxVelRate(2.0)
    # accelerate-then-brake: the speed maximum is strictly inside the piece
    rng = np.random.default_rng(1)
    head, tail, wps, T = random_problem(rng, 1, 4, 3, rest=True)
    coeffs, _ = aa.minco_solve(head, tail, wps, T, 3, ctx=anet_ctx)
    traj = aa.Trajectory(list(T[0]), list(coeffs[0]), ctx=anet_ctx)
    vmax = traj.getMaxVelRate(); amax = traj.getMaxAccRate()
    ref_v = max(onp.piece_max_rate(coeffs[0, i], T[0, i], 1) for i in range(4))
    ref_a = max(onp.piece_max_rate(coeffs[0, i], T[0, i], 2) for i in range(4))
    assert abs(vmax - ref_v) <= 1e-9 * ref_v and abs(amax - ref_a) <= 1e-9 * ref_a
    assert traj.checkMaxVelRate(vmax * 1.001) and not traj.checkMaxVelRate(vmax * 0.999)
    assert traj.checkMaxAccRate(amax * 1.001) and not traj.checkMaxAccRate(amax * 0.999)
