"""CPU suite: the numpy restatement of the QP backward pass (oracle/qp_vjp_np.py: layers.py:129-139's KKT hook carried
through to the durations) against the fixtures generated THROUGH THE REFERENCE'S OWN matrices (tests/golden/vjp_*.npz,
make_golden.py main_vjp: torch.autograd of w'F(y; T) w.r.t. MinTrajOpt's Times)."""
import glob
import os

import numpy as np
import pytest

from oracle import qp_vjp_np
from tests.util import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "vjp_*.npz")))
LIMITS = {1: (5.0, 8.0), 2: (5.0, 7.0)}          # make_golden.make_params: phase-1 / phase-2 limits


def test_fixtures_present():
    assert len(FILES) == 4


@pytest.mark.parametrize("path", FILES)
def test_numpy_backward_pass_matches_the_reference_autograd_fixture(path):
    d = np.load(path)
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    vmax, amax = LIMITS[phase]
    w1, w2 = d["w1"], d["w2"]
    out = qp_vjp_np.qp_vjp(s, d["state"], d["hpolys"], d["m_rows"], d["T"], res, vmax, amax, lambda z: w1 + w2 * z)
    ref = d["dloss_dT"]
    assert np.abs(out["z"] - d["z"]).max() <= 1e-8 * np.abs(d["z"]).max()
    assert abs(out["obj"] - float(d["obj"])) <= 1e-9 * abs(float(d["obj"]))
    assert int((out["lam"] > 1e-6).sum()) == int(d["n_active"]) >= 2
    # the hook's output on z (layers.py:139) and the time gradient; cond(J) is 1e6 - 4e10, float64 leaves 1e-6
    assert np.abs(out["hook"][:len(w1)] - d["hook_grad_z"]).max() <= 1e-5 * np.abs(d["hook_grad_z"]).max()
    assert np.abs(out["grad_T"] - ref).max() <= 2e-6 * np.abs(ref).max(), (out["grad_T"], ref)


@pytest.mark.parametrize("path", FILES[:1])
def test_backward_pass_is_the_derivative_of_the_loss(path):
    """independent of the KKT algebra: central differences of loss(z*(T)) through re-solves of the oracle"""
    d = np.load(path)
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    vmax, amax = LIMITS[phase]
    w1, w2 = d["w1"], d["w2"]
    from oracle.qp_np import qp_ipm

    def loss(T):
        Q, A, b, G, h = qp_vjp_np.assemble_dense(s, d["state"], d["hpolys"], d["m_rows"], T, res, vmax, amax)
        z = qp_ipm(Q, A, b, G, h, tol=1e-13, max_iter=300)[0]
        return w1 @ z + 0.5 * (w2 * z * z).sum()
    T = d["T"].astype(float)
    fd = np.zeros(N)
    for i in range(N):
        h = 1e-5 * T[i]
        Tp = T.copy(); Tp[i] += h
        Tm = T.copy(); Tm[i] -= h
        fd[i] = (loss(Tp) - loss(Tm)) / (2 * h)
    assert np.abs(fd - d["dloss_dT"]).max() <= 1e-4 * np.abs(fd).max(), (fd, d["dloss_dT"])
