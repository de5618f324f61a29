"""GPU parity of the cost/gradient path: partial gradients + penalty functional + propogateGrad
vs the numpy oracle (classic dense adjoint), which itself is pinned by finite differences and by
the G z - h rows of the reference-pinned QP assembly (tests/test_oracle_cpu.py)."""
import numpy as np
import pytest

from oracle import minco_np as onp
from tests.util import random_problem, rel_err

pytestmark = pytest.mark.gpu


def make_corridors(rng, head, tail, wps, M, tight=1.0):
    B, Nm1, _ = wps.shape
    N = Nm1 + 1
    pts = np.concatenate([head[:, None, :, 0], wps, tail[:, None, :, 0]], axis=1)   # (B,N+1,3)
    hp = np.zeros((B, N, M, 4))
    for b in range(B):
        for i in range(N):
            k = rng.integers(max(1, M - 3), M + 1)          # some zero-padded rows
            for r in range(k):
                a = rng.normal(size=3); a /= np.linalg.norm(a)
                hp[b, i, r, :3] = a
                hp[b, i, r, 3] = a @ (0.5 * (pts[b, i] + pts[b, i + 1])) + rng.uniform(0.2, 1.0) * tight
    return hp


@pytest.mark.parametrize("s,c,N,M", [(4, 3, 8, 16), (4, 4, 5, 7), (3, 3, 16, 12), (3, 3, 2, 5), (4, 3, 1, 6),
                                     (3, 2, 4, 0), (2, 2, 3, 4)])
def test_cost_grad_matches_oracle(anet_ctx, s, c, N, M):
    import allocnet_amd as aa
    rng = np.random.default_rng(1000 + 10 * N + s)
    B = 37
    head, tail, wps, T = random_problem(rng, B, N, c)
    hp = make_corridors(rng, head, tail, wps, M) if M else None
    kw = dict(res=7, vmax=1.5, amax=2.5, wc=60.0, wv=25.0, wa=9.0, mu=0.05)
    pen = aa.make_penalty(rho=0.8, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"],
                          max_vel=kw["vmax"], max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    cost, gP, gT, co = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, want_coeffs=True,
                                          ctx=anet_ctx)
    if s == 2:
        return      # numpy oracle covers s >= 3 (the reference's orders)
    nz = 0
    for b in range(0, B, 3):
        hpb = np.transpose(hp[b], (1, 2, 0)) if M else np.zeros((1, 4, N))
        co0, e0, *_ = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b])
        jp, gC, gTp, pc = onp.penalty_partials(s, co0, T[b], hpb, **kw)
        eC, eT = onp.energy_partials(s, co0, T[b])
        gP0, gT0 = onp.minco_dense_propagate(s, head[b], tail[b], wps[b].T, T[b], gC + eC, gTp + eT + 0.8)
        c0 = e0 + 0.8 * T[b].sum() + jp
        nz += jp > 0
        assert rel_err(co[b], co0) < 1e-9
        assert abs(cost[b] - c0) <= 1e-9 * abs(c0)
        if N > 1:
            assert np.abs(gP[b].T - gP0).max() <= 1e-7 * max(1.0, np.abs(gP0).max())
        assert np.abs(gT[b] - gT0).max() <= 1e-7 * max(1.0, np.abs(gT0).max())
    assert nz > 0       # the penalty really was active in the compared samples


def test_energy_only_gradient_finite_difference(anet_ctx):
    """No penalty: cost == energy + rho sum T, gradient vs central differences of the GPU cost itself."""
    import allocnet_amd as aa
    rng = np.random.default_rng(77)
    s, c, N, B = 4, 3, 6, 8
    head, tail, wps, T = random_problem(rng, B, N, c)
    pen = aa.make_penalty(rho=2.0, res=4)
    cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, penalty=pen, ctx=anet_ctx)
    _, energy = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    assert rel_err(cost, energy + 2.0 * T.sum(axis=1)) < 1e-12
    h = 1e-6
    for (k, ax) in [(0, 0), (2, 1), (4, 2)]:
        wp = wps.copy(); wp[:, k, ax] += h; wm = wps.copy(); wm[:, k, ax] -= h
        fd = (aa.minco_cost_grad(head, tail, wp, T, s, penalty=pen, ctx=anet_ctx)[0]
              - aa.minco_cost_grad(head, tail, wm, T, s, penalty=pen, ctx=anet_ctx)[0]) / (2 * h)
        assert np.abs(fd - gP[:, k, ax]).max() <= 2e-5 * max(1.0, np.abs(gP[:, k, ax]).max())
    for i in (0, 3, 5):
        tp = T.copy(); tp[:, i] += h; tm = T.copy(); tm[:, i] -= h
        fd = (aa.minco_cost_grad(head, tail, wps, tp, s, penalty=pen, ctx=anet_ctx)[0]
              - aa.minco_cost_grad(head, tail, wps, tm, s, penalty=pen, ctx=anet_ctx)[0]) / (2 * h)
        assert np.abs(fd - gT[:, i]).max() <= 2e-5 * max(1.0, np.abs(gT[:, i]).max())


def test_random_shapes_against_both_oracles(anet_ctx):
    """Every (order, boundary count, piece count) combination the kernels are instantiated for is reachable
    here: 30 random shapes with random batch sizes, coefficients and energy against the C restatement for the whole
    batch, cost and gradients against the numpy adjoint on two members of each."""
    import allocnet_amd as aa
    from oracle import cbind
    rng = np.random.default_rng(4242)
    seen = set()
    for trial in range(30):
        s = int(rng.integers(3, 5))
        c = int(rng.integers(1, s + 1))
        N = int(rng.integers(1, 17))
        M = int(rng.choice([0, 3, 8, 11, 16]))
        B = int(rng.choice([1, 2, 20, 21, 22, 63, 64, 65, 130]))
        seen.add((s, c, N))
        head, tail, wps, T = random_problem(rng, B, N, c)
        hp = make_corridors(rng, head, tail, wps, M) if M else None
        kw = dict(res=int(rng.integers(2, 9)), vmax=1.5, amax=2.5, wc=60.0, wv=25.0, wa=9.0, mu=0.05)
        pen = aa.make_penalty(rho=0.8, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"],
                              max_vel=kw["vmax"], max_acc=kw["amax"], res=kw["res"], poly_rows=M)
        cost, gP, gT, co = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, want_coeffs=True,
                                              ctx=anet_ctx)
        cc, ec = cbind.minco_solve_batch(s, head, tail, wps, T)
        assert rel_err(co, cc) < 1e-8, (s, c, N, B)
        for b in {0, B - 1}:
            hpb = np.transpose(hp[b], (1, 2, 0)) if M else np.zeros((1, 4, N))
            co0, e0, *_ = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b])
            jp, gC, gTp, pc = onp.penalty_partials(s, co0, T[b], hpb, **kw)
            eC, eT = onp.energy_partials(s, co0, T[b])
            gP0, gT0 = onp.minco_dense_propagate(s, head[b], tail[b], wps[b].T, T[b], gC + eC, gTp + eT + 0.8)
            c0 = e0 + 0.8 * T[b].sum() + jp
            assert abs(cost[b] - c0) <= 1e-8 * abs(c0), (s, c, N, B)
            if N > 1:
                assert np.abs(gP[b].T - gP0).max() <= 1e-6 * max(1.0, np.abs(gP0).max()), (s, c, N, B)
            assert np.abs(gT[b] - gT0).max() <= 1e-6 * max(1.0, np.abs(gT0).max()), (s, c, N, B)
    assert len(seen) >= 20


@pytest.mark.parametrize("s,c,N,M,res", [(4, 3, 8, 16, 9), (4, 4, 8, 9, 9), (4, 3, 5, 16, 9), (3, 3, 8, 12, 9), (3, 3, 5, 7, 9),
                                         (4, 3, 1, 6, 9), (4, 3, 2, 6, 9), (3, 2, 3, 0, 9),
                                         # res = 20, orders 3 / 4: k_piece_grad_mx (the table contractions on the matrix instructions) --
                                         # one full row block, two blocks with a ragged second, a partial group of four, no rows
                                         (4, 3, 8, 16, 20), (4, 4, 5, 21, 20), (4, 2, 3, 7, 20), (4, 3, 2, 0, 20), (4, 3, 1, 37, 20),
                                         (3, 3, 8, 12, 20)])
def test_large_batch_shapes_agree_with_small_batch_shapes_and_oracle(anet_ctx, s, c, N, M, res):
    """Batches above 16384 run the lane-per-trajectory solve / adjoint and the lane-per-piece penalty kernel (orders 3 / 4 at 20 samples
    per piece: four lanes per piece and the FP64 matrix instructions, csrc/piece_grad_mx.h), batches up to
    16384 the lane-per-(trajectory, axis) and two-lanes-per-piece shapes: the same 16384 + 257 trajectories (a ragged last
    workgroup) through both agree to rounding -- whole batch against chunks of 8192 -- and a strided sample agrees with the C
    restatement (classic banded LU + adjoint) to the tolerance of the other parity tests."""
    import allocnet_amd as aa
    from oracle import cbind
    from tests.util import corridor_problem
    rng = np.random.default_rng(7000 + 100 * s + 10 * N + c)
    B = 16384 + 257
    if M:
        head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    else:
        head, tail, wps, T = random_problem(rng, B, N, c)
        hp = None
    kw = dict(res=res, vmax=2.5, amax=3.5, wc=1e3, wv=40.0, wa=15.0, mu=0.03)
    pen = aa.make_penalty(rho=3.0, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"], max_vel=kw["vmax"],
                          max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)          # large-batch shapes
    assert np.isfinite(cost).all() and np.isfinite(gP).all() and np.isfinite(gT).all()
    # the small-batch shapes: chunks of at most 8192 trajectories (and, separately, the whole batch with coefficients asked for)
    c2, gP2, gT2 = np.empty_like(cost), np.empty_like(gP), np.empty_like(gT)
    for b0 in range(0, B, 8192):
        sl = slice(b0, min(B, b0 + 8192))
        c2[sl], gP2[sl], gT2[sl] = aa.minco_cost_grad(head[sl], tail[sl], wps[sl], T[sl], s, hpolys=None if hp is None else hp[sl],
                                                      penalty=pen, ctx=anet_ctx)
    c3, gP3, gT3, _ = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, want_coeffs=True, ctx=anet_ctx)
    for other in ((c2, gP2, gT2), (c3, gP3, gT3)):
        assert np.abs(cost - other[0]).max() <= 1e-11 * np.abs(other[0]).max()
        sc = np.maximum(1.0, np.abs(other[2]).max(axis=1, keepdims=True))
        assert (np.abs(gT - other[2]) <= 1e-9 * sc).all(), np.abs(gT - other[2]).max()
        if N > 1:
            sp = np.maximum(1.0, np.abs(other[1]).reshape(B, -1).max(axis=1))[:, None, None]
            assert (np.abs(gP - other[1]) <= 1e-9 * sp).all(), np.abs(gP - other[1]).max()
    idx = np.arange(0, B, B // 48)
    cc, cgP, cgT = cbind.minco_cost_grad_batch(s, head[idx], tail[idx], wps[idx], T[idx], None if hp is None else hp[idx], 3.0,
                                               nthreads=4, **kw)
    assert np.abs(cost[idx] - cc).max() <= 1e-9 * np.abs(cc).max()
    assert np.abs(gT[idx] - cgT).max() <= 1e-7 * max(1.0, np.abs(cgT).max())
    if N > 1:
        assert np.abs(gP[idx] - cgP).max() <= 1e-7 * max(1.0, np.abs(cgP).max())
    assert (cost - 3.0 * T.sum(axis=1) > 0).all()


@pytest.mark.parametrize("s,c,N,M,res", [(4, 3, 8, 16, 20), (4, 4, 8, 50, 7), (4, 3, 5, 20, 64), (4, 2, 3, 6, 1), (4, 3, 1, 6, 3),
                                         (3, 3, 16, 16, 20), (3, 3, 13, 9, 10), (3, 3, 5, 16, 20), (3, 2, 8, 0, 5), (3, 1, 2, 7, 2),
                                         (4, 3, 8, 16, 65),
                                         # full groups of 8-piece snap at 20 samples: phase 2 on the matrix instructions (MX) -- two row
                                         # blocks with a ragged second, a partial group of four rows, no rows
                                         (4, 3, 8, 21, 20), (4, 3, 8, 7, 20), (4, 3, 8, 0, 20)])
def test_one_launch_evaluation_at_every_group_size(anet_ctx, s, c, N, M, res):
    """Batches of up to one workgroup per CU (two rounds of them for <= 8 pieces) are evaluated in ONE launch (csrc/minco_fused_kernel.h: solve, penalty / energy
    partial gradients and adjoint of a group of G trajectories in one workgroup, G = 16 ... 1 by batch, the lanes a smaller group
    frees splitting the samples of a piece).  Every group size, ragged last groups, 1 ... 16 pieces, 0 ... 50 corridor rows, 1 ... 64
    samples per piece (65: the three-launch path), coefficients asked for or not -- the whole batch against the C restatement
    (classic banded LU + adjoint through the same factors), to the tolerance of the other parity tests."""
    import allocnet_amd as aa
    from oracle import cbind
    from tests.util import corridor_problem
    rng = np.random.default_rng(9000 + 1000 * s + 50 * N + M + res)
    kw = dict(res=res, vmax=2.5, amax=3.5, wc=1e3, wv=40.0, wa=15.0, mu=0.03)
    pen = aa.make_penalty(rho=3.0, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"], max_vel=kw["vmax"],
                          max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    gmax = 16 if N <= 8 else 8
    # (up to 256 workgroups the batch picks G = gmax, gmax / 2, ..., 1; +-1 / odd sizes leave the last group ragged; beyond, two rounds of
    #  groups of gmax)
    for B in sorted({1, 2, 3, 255, 256, 257, 300, 512, 513, 1000, 1024 + 7, 2048 - 5, 256 * gmax - 3, 256 * gmax, 512 * gmax - 3}):
        if M:
            head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
        else:
            head, tail, wps, T = random_problem(rng, B, N, c)
            hp = None
        want = bool(B % 2)
        out = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, want_coeffs=want, ctx=anet_ctx)
        cost, gP, gT = out[:3]
        cc, cgP, cgT = cbind.minco_cost_grad_batch(s, head, tail, wps, T, hp, 3.0, nthreads=8, **kw)
        assert np.abs(cost - cc).max() <= 1e-9 * np.abs(cc).max(), (B, np.abs(cost - cc).max())
        assert np.abs(gT - cgT).max() <= 1e-7 * max(1.0, np.abs(cgT).max()), B
        if N > 1:
            assert np.abs(gP - cgP).max() <= 1e-7 * max(1.0, np.abs(cgP).max()), B
        if want:
            co, _ = cbind.minco_solve_batch(s, head, tail, wps, T)
            assert rel_err(out[3], co) < 1e-9, B
        if M and res >= 5:
            assert (cost - 3.0 * T.sum(axis=1) > 0).all()


@pytest.mark.parametrize("s,N", [(4, 8), (3, 16), (3, 5), (4, 3), (4, 6), (3, 10), (4, 2), (3, 1)])
def test_time_reversal_is_a_symmetry_of_the_evaluation(anet_ctx, s, N):
    """A size-independent property of the path: the trajectory run backwards in time -- waypoints and durations reversed, head and
    tail swapped with the sign of every odd derivative changed -- has the same control effort, and the gradients come out reversed.
    The one-launch evaluation of the exact shapes RELIES on it (the second half of its chain is eliminated on the reversed data,
    csrc/minco_fused_kernel.h); here it is checked from outside on both families of kernels (4096 trajectories: one launch; 20000: the
    streaming kernels), with the penalty weights at zero -- the penalty's samples tau_j = j / res, j < res, are not symmetric."""
    import allocnet_amd as aa
    c = 3
    pen = aa.make_penalty(rho=2.5, w_corridor=0.0, w_vel=0.0, w_acc=0.0, smooth_mu=0.05, max_vel=2.0, max_acc=3.0, res=8, poly_rows=0)
    sign = np.array([1.0, -1.0, 1.0, -1.0])[:c]
    for B in (4096, 20000):
        rng = np.random.default_rng(31 * s + N + B)
        head, tail, wps, T = random_problem(rng, B, N, c)
        cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, penalty=pen, ctx=anet_ctx)
        rc, rgP, rgT = aa.minco_cost_grad(tail * sign, head * sign, wps[:, ::-1].copy(), T[:, ::-1].copy(), s, penalty=pen, ctx=anet_ctx)
        assert np.abs(rc - cost).max() <= 1e-11 * np.abs(cost).max(), (B, np.abs(rc - cost).max())
        st = np.maximum(1.0, np.abs(gT).max(axis=1, keepdims=True))
        assert (np.abs(rgT[:, ::-1] - gT) <= 1e-9 * st).all(), (B, (np.abs(rgT[:, ::-1] - gT) / st).max())
        if N > 1:
            sp = np.maximum(1.0, np.abs(gP).reshape(B, -1).max(axis=1))[:, None, None]
            assert (np.abs(rgP[:, ::-1] - gP) <= 1e-9 * sp).all(), (B, (np.abs(rgP[:, ::-1] - gP) / sp).max())


def test_matrix_instruction_kernels_agree_with_the_vector_kernels(anet_ctx, tmp_path):
    """The matrix-instruction forms of the penalty arithmetic (round 6: k_piece_grad_mx, phase 2 of the one-launch kernel on a wave
    per column set) against the vector forms they replace, same inputs: the vector side runs in a process of its own with
    ANET_PG_MX=0 ANET_FUSED_MX=0 (the library reads the switches once).  Batches that take every matrix-instruction shape: groups
    of 2 / 8 / 16 trajectories in one launch (300, 2047, 4096 + 3 ragged), six rounds of groups (20 000), the streaming kernel with
    four and with eight column sets per wave (33 000, 70 001), 8-piece snap and 16-piece jerk.  Cost to 1e-12, gradients to 1e-10
    of their scale -- two orders inside the parity bar against the C restatement, which both sides meet on their own."""
    import subprocess
    import sys
    import os
    import allocnet_amd as aa
    from tests.util import corridor_problem
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(4, 8, 300), (4, 8, 2047), (4, 8, 4099), (4, 8, 20000), (4, 8, 33000), (4, 8, 70001), (3, 16, 1024), (3, 16, 9000), (3, 16, 40000)]
    script = f"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import allocnet_amd as aa
from tests.util import corridor_problem
out = {{}}
for k, (s, N, B) in enumerate({cases!r}):
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(500 + k), B, N, 3, 16)
    pen = aa.make_penalty(rho=3.0, w_corridor=1e3, w_vel=40.0, w_acc=15.0, smooth_mu=0.03, max_vel=2.5, max_acc=3.5, res=20, poly_rows=16)
    c, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen)
    out[f"c{{k}}"], out[f"p{{k}}"], out[f"t{{k}}"] = c, gP, gT
np.savez({str(tmp_path / 'vector.npz')!r}, **out)
"""
    env = dict(os.environ, ANET_PG_MX="0", ANET_FUSED_MX="0")
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    ref = np.load(tmp_path / "vector.npz")
    for k, (s, N, B) in enumerate(cases):
        head, tail, wps, T, hp = corridor_problem(np.random.default_rng(500 + k), B, N, 3, 16)
        pen = aa.make_penalty(rho=3.0, w_corridor=1e3, w_vel=40.0, w_acc=15.0, smooth_mu=0.03, max_vel=2.5, max_acc=3.5, res=20, poly_rows=16)
        c, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)
        assert np.abs(c - ref[f"c{k}"]).max() <= 1e-12 * np.abs(c).max(), (s, N, B)
        assert np.abs(gT - ref[f"t{k}"]).max() <= 1e-10 * max(1.0, np.abs(gT).max()), (s, N, B)
        assert np.abs(gP - ref[f"p{k}"]).max() <= 1e-10 * max(1.0, np.abs(gP).max()), (s, N, B)
    # (the switches are honoured: the default side really took the matrix-instruction shapes)
    if os.environ.get("ANET_PG_MX", "1") != "0":
        assert aa.minco_piece_grad_shape(4, 8, 70001, penalty=pen, ctx=anet_ctx) == 3
