"""The CPU port of the structured interior point (oracle/qp_ipm_port.c -- bench.py's like-for-like CPU figure beside
k_qp_ipm) pinned to the dense numpy interior point (oracle/qp_np.py) on the REFERENCE-ASSEMBLED matrices: the fixtures under
tests/golden/ hold Q, A, b, G, h as the imported reference built them (tests/golden/make_golden.py); the port gets only the
problem data (boundary states, durations, corridor rows) and must arrive at the same optimum."""
import numpy as np
import pytest

from oracle import cbind, qp_np
from tests.util import golden_files


def _dense_rows(d):
    """full-width G, h of a fixture from its compact row blocks (as tests/test_qp_solve_gpu.py::_dense does)"""
    s, N, res = int(d["order"]), int(d["N"]), int(d["res"])
    D = 2 * s
    n = 3 * D * N
    G1c, G2c, m_rows = d["G1c"], d["G2c"], d["m_rows"].astype(int)
    G = np.zeros((G1c.shape[0] + G2c.shape[0], n))
    r = 0
    for i in range(N):
        for _ in range(res):
            G[r:r + m_rows[i], i * 3 * D:(i + 1) * 3 * D] = G1c[r:r + m_rows[i]]
            r += m_rows[i]
    r2 = 0
    for i in range(N):
        for _ in range(res):
            for j in range(3):
                G[r + r2:r + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2c[r2:r2 + 4]
                r2 += 4
    return G, np.r_[d["h1"], d["h2"]]


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_port_reaches_the_optimum_of_the_reference_assembled_qp(path):
    d = np.load(path)
    s, N, res = int(d["order"]), int(d["N"]), int(d["res"])
    G, h = _dense_rows(d)
    z, lam, nu, fo, it = qp_np.qp_ipm(d["Q"], d["A"], d["b"], G, h, tol=1e-10)
    state = np.zeros((1, 2, 3, 3))
    for ax in range(3):
        state[0, 0, ax] = d["state"][3 * ax:3 * ax + 3, 0]
        state[0, 1, ax] = d["state"][3 * ax:3 * ax + 3, 1]
    hp = np.ascontiguousarray(np.transpose(d["hpolys"], (2, 0, 1)))[None]          # (1, N, M, 4)
    vmax, amax = float(d["h2"][0]), float(d["h2"][1])       # the phase's limits (rows +v, +a, -v, -a: min_traj_opt.py:535-613)
    out = cbind.qp_ipm_batch(s, state, d["T"][None], hp, res=res, vmax=vmax, amax=amax, tol=1e-9)
    if it >= 200:                                   # infeasible for the dense oracle: the port must not call it solved
        assert out["status"][0] < 1
        return
    assert out["status"][0] >= 1, (out["status"], out["iters"])    # (2: stalled above 1e-9 at its rounding floor -- the optimum below still holds)
    assert abs(out["obj"][0] - fo) <= 1e-6 * max(1.0, abs(fo)), (out["obj"][0], fo)
    zc = out["coeffs"][0].reshape(-1)               # [piece][axis][D] = the reference's flattening
    # the optimum of a convex QP is unique in the objective; the minimiser itself to the accuracy its flat directions allow
    assert np.abs(d["A"] @ zc - d["b"]).max() <= 1e-7 * max(1.0, np.abs(d["b"]).max())        # equality rows hold by construction
    assert (G @ zc - h).max() <= 1e-6 * max(1.0, np.abs(h).max())                             # feasible
    assert abs(0.5 * zc @ d["Q"] @ zc - fo) <= 1e-6 * max(1.0, abs(fo))                      # objective through the reference's Q
    assert np.abs(zc - z).max() <= 1e-3 * max(1.0, np.abs(z).max())


def test_port_on_the_bench_generator_and_threads():
    """The generator of bench.py's qp_solve leg: 8-segment snap, 16 rows; the port's optimum equals the dense oracle's on a
    few problems, and the threaded batch equals the single-threaded one."""
    from allocnet_amd.synth import corridor_problem
    from oracle import minco_np as onp
    s, N, M, B = 4, 8, 16, 6
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    T = T * 1.5
    state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
    a = cbind.qp_ipm_batch(s, state, T, hp, nthreads=1)
    b = cbind.qp_ipm_batch(s, state, T, hp, nthreads=3)
    assert np.array_equal(a["obj"], b["obj"]) and np.array_equal(a["status"], b["status"])
    D = 2 * s
    n = 3 * D * N
    for k in range(2):
        st9 = np.zeros((9, 2))
        for ax in range(3):
            st9[3 * ax:3 * ax + 3, 0] = state[k, 0, ax]
            st9[3 * ax:3 * ax + 3, 1] = state[k, 1, ax]
        Q, A, bb, G1, h1, G2, h2 = onp.qp_assemble(s, st9, np.transpose(hp[k], (1, 2, 0)), np.full(N, M), T[k], 20, 4.0, 6.0)
        G = np.zeros((G1.shape[0] + G2.shape[0], n))
        r = 0
        for i in range(N):
            for _ in range(20):
                G[r:r + M, i * 3 * D:(i + 1) * 3 * D] = G1[r:r + M]
                r += M
        r2 = 0
        for i in range(N):
            for _ in range(20):
                for j in range(3):
                    G[r + r2:r + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2[r2:r2 + 4]
                    r2 += 4
        hh = np.r_[h1, h2]
        keep = (np.abs(G).sum(axis=1) > 0) | (hh != 0)
        z, lam, nu, fo, it = qp_np.qp_ipm(Q, A, bb, G[keep], hh[keep], tol=1e-9)
        if it < 200:
            assert a["status"][k] == 1
            assert abs(a["obj"][k] - fo) <= 1e-6 * max(1.0, abs(fo)), (k, a["obj"][k], fo)
