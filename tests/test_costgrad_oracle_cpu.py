"""CPU suite: the C cost + gradient restatement (oracle/minco_costgrad.c: classic banded LU + adjoint through the same
factors) against the numpy dense-adjoint oracle (oracle/minco_np.py), finite differences, and the C L-BFGS driver
against the same loop driven through the Python callback.  No GPU."""
import numpy as np
import pytest

from oracle import cbind
from oracle import minco_np as onp
from tests.util import corridor_problem

KW = dict(res=20, vmax=4.0, amax=6.0, wc=1e4, wv=1e3, wa=1e3, mu=1e-2)
RHO = 50.0


def _np_cost_grad(s, head, tail, wps, T, hp):
    hpb = np.transpose(hp, (1, 2, 0))
    co, e, *_ = onp.minco_dense_solve(s, head, tail, wps.T, T)
    jp, gC, gTp, _ = onp.penalty_partials(s, co, T, hpb, **KW)
    eC, eT = onp.energy_partials(s, co, T)
    gP, gT = onp.minco_dense_propagate(s, head, tail, wps.T, T, gC + eC, gTp + eT + RHO)
    return e + RHO * T.sum() + jp, gP, gT


@pytest.mark.parametrize("s,c,N,M", [(4, 3, 8, 16), (3, 3, 16, 16), (4, 4, 5, 12), (3, 2, 3, 6), (3, 3, 1, 8)])
def test_c_cost_grad_matches_numpy_dense_adjoint(s, c, N, M):
    rng = np.random.default_rng(100 * s + N)
    B = 6
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    cost, gP, gT = cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, RHO, nthreads=2, **KW)
    active = 0
    for b in range(B):
        c0, gP0, gT0 = _np_cost_grad(s, head[b], tail[b], wps[b], T[b], hp[b])
        e0 = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b])[1]
        active += c0 - e0 - RHO * T[b].sum() > 1e-9 * c0
        assert abs(cost[b] - c0) <= 1e-10 * abs(c0)
        if N > 1:
            assert np.abs(gP[b].T - gP0).max() <= 1e-8 * max(1.0, np.abs(gP0).max())
        assert np.abs(gT[b] - gT0).max() <= 1e-8 * max(1.0, np.abs(gT0).max())
    assert active >= 1


def test_c_cost_grad_directional_derivative():
    s, c, N, M = 4, 3, 8, 16
    rng = np.random.default_rng(5)
    B = 64
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    cost, gP, gT = cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, RHO, **KW)
    dT = rng.normal(size=T.shape) * 1e-6
    dW = rng.normal(size=wps.shape) * 1e-6
    cp = cbind.minco_cost_grad_batch(s, head, tail, wps + dW, T + dT, hp, RHO, **KW)[0]
    cm = cbind.minco_cost_grad_batch(s, head, tail, wps - dW, T - dT, hp, RHO, **KW)[0]
    dd = (gT * dT).sum(axis=1) + (gP * dW).sum(axis=(1, 2))
    assert np.abs((cp - cm) / 2 - dd).max() <= 1e-5 * np.abs(dd).max()


def _fwd(tau):
    return np.where(tau > 0, (0.5 * tau + 1) * tau + 1, 1.0 / ((0.5 * tau - 1) * tau + 1))


def _dfwd(tau):
    den = (0.5 * tau - 1) * tau + 1
    return np.where(tau > 0, tau + 1, (1 - tau) / den ** 2)


def _bwd(T):
    with np.errstate(invalid="ignore"):
        return np.where(T > 1, np.sqrt(2 * T - 1) - 1, 1 - np.sqrt(2 / T - 1))


def test_c_lbfgs_driver_equals_the_callback_driven_loop():
    """oracle_lbfgs_minco_batch == oracle_lbfgs_optimize driven from Python with the same C objective: same counters,
    same iterates (the batch driver adds the duration map and threading, nothing else)."""
    s, c, N, M = 3, 3, 6, 8
    rng = np.random.default_rng(11)
    B = 5
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    prm = cbind.lbfgs_default_param(max_iterations=25)
    out = cbind.lbfgs_minco_batch(s, head, tail, wps, T, hp, RHO, param=prm, nthreads=3, **KW)
    nw = 3 * (N - 1)
    for b in range(B):
        def fun(x, b=b):
            w = x[:nw].reshape(1, N - 1, 3)
            tau = x[nw:]
            f, gP, gT = cbind.minco_cost_grad_batch(s, head[b:b + 1], tail[b:b + 1], w, _fwd(tau)[None], hp[b:b + 1], RHO, **KW)
            return f[0], np.r_[gP.reshape(-1), gT[0] * _dfwd(tau)]
        x0 = np.r_[wps[b].reshape(-1), _bwd(T[b])]
        ret, xo, fo, it, ev = cbind.lbfgs_optimize(x0, fun, cbind.lbfgs_default_param(max_iterations=25))
        assert (out["status"][b], out["iters"][b], out["evals"][b]) == (ret, it, ev)
        # (numpy's duration map and the C one differ in the last bit, so iterates agree to rounding, not bit for bit)
        assert abs(out["cost"][b] - fo) <= 1e-9 * abs(fo)
        assert np.abs(out["wps"][b].reshape(-1) - xo[:nw]).max() <= 1e-7 * np.abs(xo[:nw]).max()
        assert np.abs(out["T"][b] - _fwd(xo[nw:])).max() <= 1e-7 * out["T"][b].max()
    assert (out["cost"] < cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, RHO, **KW)[0]).all()


def test_configs3_objective_amplifies_rounding_in_the_restatement_alone():
    """No GPU involved: the C restatement of lbfgs_optimize (lbfgs.hpp:434-717) on the C cost + gradient, 64 strided problems of
    BASELINE configs[3], against ITSELF from a start point perturbed by 1e-15 relative (one unit in the last place).  Up to 50
    iterations every problem keeps identical (status, iterations, evaluations); by 200 iterations most do not, and the costs of
    those that do are apart by more than 1e-4: the L-BFGS on this objective (smoothed-L1 penalties of weight 1e4 at mu = 1e-2)
    amplifies a last-bit difference roughly 10^5-fold per 25 iterations.  This is why the converged end points of the device
    run and the restatement can only be compared statistically (tests/test_baseline_configs_gpu.py has the device side of the
    same profile, asserted to be no worse than this control)."""
    import bench
    from allocnet_amd.synth import corridor_problem
    B, s, c, N, M = 4096, 3, 3, 16, 16
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(2), B, N, c, M)
    idx = np.linspace(0, B - 1, 64).astype(int)
    prof = bench.lbfgs_divergence_profile(None, cbind, s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], None, 4,
                                          budgets=(25, 50, 200), eps=1e-15)
    ctl = prof["cpu_vs_cpu_perturbed"]
    assert "gpu_vs_cpu" not in prof
    assert ctl["same"][0] == 1.0 and ctl["max_rel_same"][0] <= 1e-8
    assert ctl["same"][1] >= 0.95
    assert ctl["same"][2] <= 0.5 and ctl["median_rel"][2] >= 1e-5
