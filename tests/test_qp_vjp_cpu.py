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


@pytest.mark.parametrize("name", ["layers_snap_n3", "layers_jerk_n4", "layers_snap_n5"])
def test_numpy_hook_matches_what_the_references_own_hook_left(name):
    """tests/golden/layers_*.npz (make_golden.py main_layers) hold z.grad after `objc.backward()` through the reference's OWN
    layers.py code -- its J (layers.py:129-134) and hook (:136-141) executed, not restated.  The numpy backward pass with
    dloss/dz = Q z / path_length (objc's gradient) must leave the same vector on z, and the same optimum and objc."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    vmax, amax = LIMITS[phase]
    hp = d["hpolys"][:, :, :N]
    m_rows = np.array([int(np.sum(np.linalg.norm(hp[:, :, i], axis=1) > 0)) for i in range(N)])
    T = d["Times"][:N]
    pl = float(d["path_length"])
    Q = qp_vjp_np.assemble_dense(s, d["state"], hp, m_rows, T, res, vmax, amax)[0]
    out = qp_vjp_np.qp_vjp(s, d["state"], hp, m_rows, T, res, vmax, amax, lambda z: (Q @ z) / pl)
    zr = d["forward_solved_z"]
    assert np.abs(out["z"] - zr).max() <= 1e-8 * np.abs(zr).max()
    assert abs(out["obj"] / pl - float(d["forward_solved_objc"])) <= 1e-9 * float(d["forward_solved_objc"])
    # For THIS loss the hook's exact output on z is zero: J w = -[Qz; 0; 0] is solved by w = (0, 1 on the active rows, nu)
    # because Qz = -G'lam - A'nu at the optimum (the envelope theorem: through z the optimal objective has no first-order
    # sensitivity) -- what the reference's executed hook left is rounding at 1e-12 of the incoming gradient, and so is the
    # restatement's.  (A loss other than objc is never back-propagated by the reference; vjp_*.npz cover that case.)
    hk = d["forward_solved_hook_grad_z"]
    scale = np.abs(Q @ zr).max() / pl
    assert np.abs(hk).max() <= 1e-9 * scale and np.abs(out["hook"][:zr.size]).max() <= 1e-9 * scale
    assert np.array_equal(hk, d["forward4lstm_solved_hook_grad_z"])
