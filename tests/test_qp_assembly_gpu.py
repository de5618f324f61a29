"""GPU parity of the QP assembly against the fixtures produced by importing the reference's
MinTrajOpt (tests/golden/make_golden.py): Q, A, b, G1, h1, G2, h2."""
import numpy as np
import pytest

from oracle import minco_np as onp
from tests.util import golden_files

pytestmark = pytest.mark.gpu


def _expand(d, N, D):
    """dense G = [G1; G2] (python order) from the compact fixture blocks"""
    n = 3 * D * N
    res = int(d["res"]); m_rows = d["m_rows"]
    G1 = np.zeros((d["G1c"].shape[0], n)); G2 = np.zeros((d["G2c"].shape[0], n))
    r = 0
    for i in range(N):
        for _ in range(res):
            G1[r:r + m_rows[i], i * 3 * D:(i + 1) * 3 * D] = d["G1c"][r:r + m_rows[i]]
            r += m_rows[i]
    r = 0
    for i in range(N):
        for _ in range(res):
            for j in range(3):
                G2[r:r + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = d["G2c"][r:r + 4]
                r += 4
    return np.vstack([G1, G2]), np.r_[d["h1"], d["h2"]]


@pytest.mark.parametrize("path", golden_files())
def test_assembly_matches_reference_fixture(anet_ctx, path):
    import allocnet_amd as aa
    d = np.load(path)
    s, N, res, phase = int(d["order"]), int(d["N"]), int(d["res"]), int(d["phase"])
    D = 2 * s
    vmax, amax = (5.0, 8.0) if phase == 1 else (5.0, 7.0)
    st = d["state"]
    ini = np.array([st[3 * j:3 * j + 3, 0] for j in range(3)]); fin = np.array([st[3 * j:3 * j + 3, 1] for j in range(3)])
    polys = [d["hpolys"][:d["m_rows"][i], :, i] for i in range(N)]
    Q, A, b, G, h = aa.qp_assemble(s, ini, fin, polys, d["T"], res=res, max_vel=vmax, max_acc=amax,
                                   row_order=aa.qp.ORDER_PYTHON, ctx=anet_ctx)
    Gref, href = _expand(d, N, D)
    for got, ref, name in [(Q, d["Q"], "Q"), (A, d["A"], "A"), (b, d["b"], "b"), (G, Gref, "G"), (h, href, "h")]:
        assert got.shape == ref.shape, name
        # same multiplication order as the reference -> equal to the last bit or one ulp
        assert np.abs(got - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max()), name
    # C++ row order: same rows, interleaved per sample (qp_solver.hpp:258-294)
    Qc, Ac, bc, Gc, hc = aa.qp_assemble(s, ini, fin, polys, d["T"], res=res, max_vel=vmax, max_acc=amax,
                                        row_order=aa.qp.ORDER_CPP, ctx=anet_ctx)
    m_rows = d["m_rows"]; n1 = int(res * m_rows.sum())
    r = 0; r1 = 0; r2 = n1
    for i in range(N):
        for j in range(res):
            assert np.array_equal(Gc[r:r + m_rows[i]], G[r1:r1 + m_rows[i]]); assert np.array_equal(hc[r:r + m_rows[i]], h[r1:r1 + m_rows[i]])
            r += m_rows[i]; r1 += m_rows[i]
            assert np.array_equal(Gc[r:r + 12], G[r2:r2 + 12]); assert np.array_equal(hc[r:r + 12], h[r2:r2 + 12])
            r += 12; r2 += 12
    assert np.array_equal(Qc, Q) and np.array_equal(Ac, A) and np.array_equal(bc, b)


def test_float_time_mode_and_batch(anet_ctx):
    """The C++ planner forms time powers in float (qp_solver.hpp:90-116 with T=float, :183-236): the
    float_time outputs must equal a numpy float32 restatement bit for bit; batched zero-padded input."""
    import allocnet_amd as aa
    rng = np.random.default_rng(9)
    B, N, M, s, res = 5, 4, 6, 4, 3
    D = 2 * s
    ini = rng.normal(size=(B, 3, 3)); fin = rng.normal(size=(B, 3, 3))
    hp = rng.normal(size=(B, N, M, 4)); hp[:, :, 4:] = 0.0
    T = rng.uniform(0.5, 2.0, size=(B, N)).astype(np.float32).astype(np.float64)
    Q, A, b, G, h = aa.qp_assemble(s, ini, fin, hp, T, res=res, float_time=True, ctx=anet_ctx)
    n = 3 * D * N
    assert Q.shape == (B, n, n) and G.shape == (B, res * N * (M + 12), n)
    f32 = np.float32
    for bb in range(B):
        for i in range(N):
            t = f32(T[bb, i]); t2 = t * t; t3 = t * t2; t4 = t2 * t2; t5 = t2 * t3; t6 = t3 * t3; t7 = t4 * t3
            blk = Q[bb, i * 3 * D:i * 3 * D + 4, i * 3 * D:i * 3 * D + 4]
            assert blk[0, 0] == np.float64(f32(100800) * t7) and blk[2, 3] == np.float64(f32(1400) * t2)
            assert blk[3, 3] == np.float64(f32(576) * t) and blk[1, 2] == np.float64(f32(10800) * t4)
        # one inequality row: piece 1, sample 2, corridor row 1 (C++ order)
        i, j, q = 1, 2, 1
        r = res * (M + 12) * i + (M + 12) * j + q
        step = f32(T[bb, i]) / f32(res); t = step * f32(j)
        t2 = t * t; t3 = t * t2; t4 = t2 * t2; t5 = t2 * t3; t6 = t3 * t3; t7 = t4 * t3
        basis = np.array([t7, t6, t5, t4, t3, t2, t, f32(1)], dtype=np.float64)
        for ax in range(3):
            assert np.array_equal(G[bb, r, i * 3 * D + ax * D:i * 3 * D + (ax + 1) * D], hp[bb, i, q, ax] * basis)
        assert h[bb, r] == hp[bb, i, q, 3]
        # box rows follow: +v,+a,-v,-a of axis 0
        r0 = res * (M + 12) * i + (M + 12) * j + M
        vrow = np.array([f32(7) * t6, f32(6) * t5, f32(5) * t4, f32(4) * t3, f32(3) * t2, f32(2) * t, 1, 0], dtype=np.float64)
        assert np.array_equal(G[bb, r0, i * 3 * D:i * 3 * D + D], vrow)
        assert np.array_equal(G[bb, r0 + 2, i * 3 * D:i * 3 * D + D], -vrow)
        assert h[bb, r0] == 4.0 and h[bb, r0 + 1] == 6.0
    # double mode equals the numpy oracle restatement
    Qd, Ad, bd, Gd, hd = aa.qp_assemble(s, ini, fin, hp, T, res=res, row_order=aa.qp.ORDER_PYTHON, ctx=anet_ctx)
    for bb in range(B):
        state = np.zeros((9, 2))
        for ax in range(3):
            state[3 * ax:3 * ax + 3, 0] = ini[bb, ax]; state[3 * ax:3 * ax + 3, 1] = fin[bb, ax]
        Qo, Ao, bo, G1, h1, G2, h2 = onp.qp_assemble(s, state, np.transpose(hp[bb], (1, 2, 0)), np.full(N, M), T[bb], res, 4.0, 6.0)
        assert np.abs(Qd[bb] - Qo).max() <= 1e-13 * np.abs(Qo).max()
        assert np.abs(Ad[bb] - Ao).max() <= 1e-13 * np.abs(Ao).max() and np.abs(bd[bb] - bo).max() == 0
