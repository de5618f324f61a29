"""GPU parity: HIP MINCO solve vs the numpy oracle (classic dense collocation) and vs the golden
KKT solutions computed from the reference-assembled QP matrices."""
import numpy as np
import pytest

from oracle import cbind
from oracle import minco_np as onp
from tests.util import golden_files, random_problem, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-6   # north-star tolerance (relative, coefficients and energy); we assert much tighter
TIGHT = 1e-9


@pytest.mark.parametrize("s,c,N", [(4, 3, 8), (4, 4, 8), (3, 3, 16), (3, 3, 5), (4, 3, 5), (4, 4, 1),
                                   (4, 3, 1), (3, 3, 1), (3, 2, 3), (2, 2, 6), (4, 1, 4), (4, 2, 11),
                                   (3, 1, 9), (2, 1, 1), (4, 3, 16), (4, 4, 13)])
def test_solve_matches_oracle(anet_ctx, s, c, N):
    import allocnet_amd as aa
    rng = np.random.default_rng(100 * s + 10 * c + N)
    B = 67          # ragged: not a multiple of the wave size
    head, tail, wps, T = random_problem(rng, B, N, c)
    coeffs, energy = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    for b in range(B):
        co, e, *_ = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b]) if s > 2 else (None, None)
        if co is None:
            continue
        assert rel_err(coeffs[b], co) < TIGHT
        assert abs(energy[b] - e) / e < TIGHT


@pytest.mark.parametrize("path", golden_files())
def test_solve_matches_reference_kkt(anet_ctx, path):
    """z_wp_* are minimisers of the REFERENCE-ASSEMBLED Q(1440),A,b with waypoint rows appended."""
    import allocnet_amd as aa
    d = np.load(path)
    s, N = int(d["order"]), int(d["N"]); D = 2 * s
    st = d["state"]
    head = np.array([st[3 * j:3 * j + 3, 0] for j in range(3)])[None]
    tail = np.array([st[3 * j:3 * j + 3, 1] for j in range(3)])[None]
    wps = d["pts"][1:N][None]
    T = d["T"][None]
    coeffs, energy = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    assert rel_err(coeffs.reshape(-1), d["z_wp_c3"]) < TOL
    assert abs(0.5 * energy[0] - d["e_wp_c3"]) / d["e_wp_c3"] < TOL
    if s == 4:
        head4 = np.concatenate([head, d["jerk_head"][None, :, None]], axis=2)
        tail4 = np.concatenate([tail, d["jerk_tail"][None, :, None]], axis=2)
        coeffs, energy = aa.minco_solve(head4, tail4, wps, T, s, ctx=anet_ctx)
        assert rel_err(coeffs.reshape(-1), d["z_wp_cs"]) < TOL
        assert abs(0.5 * energy[0] - d["e_wp_cs"]) / d["e_wp_cs"] < TOL


def test_class_mirror(anet_ctx):
    import allocnet_amd as aa
    rng = np.random.default_rng(5)
    head, tail, wps, T = random_problem(rng, 1, 8, 4)
    m = aa.MINCO_S4NU(ctx=anet_ctx)
    m.setConditions(head[0], tail[0], 8)
    m.setParameters(wps[0], T[0])
    co, e, *_ = onp.minco_dense_solve(4, head[0], tail[0], wps[0].T, T[0])
    assert rel_err(m.getCoeffs(), co) < TIGHT
    assert abs(m.getEnergy() - e) / e < TIGHT


def test_large_batch_properties(anet_ctx):
    """Full-size batch (config 2 x 64): size-independent properties instead of the oracle --
    waypoint interpolation, C^(2s-2) continuity at knots, boundary conditions, energy == cost
    recomputed from the coefficients."""
    import allocnet_amd as aa
    s, c, N, B = 4, 3, 8, 65536
    rng = np.random.default_rng(0)
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    coeffs, energy = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    D = 2 * s
    pw = np.arange(D - 1, -1, -1)

    def deriv_at(cm, t, d):      # cm (B,3,D), t (B,)
        k = pw[None, None, :]
        f = np.ones_like(k, dtype=float)
        for i in range(d):
            f = f * np.clip(k - i, 0, None)
        tt = np.where(k - d >= 0, t[:, None, None] ** np.clip(k - d, 0, None), 0.0)
        return np.sum(cm * f * tt, axis=2)
    scale = np.abs(coeffs).max()
    zero = np.zeros(B)
    for i in range(N):
        p0 = deriv_at(coeffs[:, i], zero, 0)
        target = head[:, :, 0] if i == 0 else wps[:, i - 1]
        assert np.abs(p0 - target).max() < 1e-9 * scale
        if i + 1 < N:
            for d in range(2 * s - 1):
                a = deriv_at(coeffs[:, i], T[:, i], d); b = deriv_at(coeffs[:, i + 1], zero, d)
                assert np.abs(a - b).max() < 1e-7 * max(1.0, np.abs(a).max())
    pe = deriv_at(coeffs[:, N - 1], T[:, N - 1], 0)
    assert np.abs(pe - tail[:, :, 0]).max() < 1e-9 * scale
    # energy recomputed from coefficients with the true-integral cost block
    e2 = np.zeros(B)
    for i in range(N):
        t = T[:, i]
        Q = np.array([[100800 * t**7, 50400 * t**6, 20160 * t**5, 5040 * t**4],
                      [50400 * t**6, 25920 * t**5, 10800 * t**4, 2880 * t**3],
                      [20160 * t**5, 10800 * t**4, 4800 * t**3, 1440 * t**2],
                      [5040 * t**4, 2880 * t**3, 1440 * t**2, 576 * t]])      # (4,4,B)
        z = coeffs[:, i, :, :4]                                               # (B,3,4)
        e2 += np.einsum("bak,klb,bal->b", z, Q, z)
    assert np.abs(e2 - energy).max() / energy.max() < 1e-9


@pytest.mark.parametrize("s,c,N", [(4, 3, 8), (3, 3, 16), (4, 4, 8)])
def test_duration_spread_accuracy_envelope(anet_ctx, s, c, N):
    """Durations spread over a factor 25 and 100 INSIDE one trajectory (the planner's network outputs stay within
    ~20).  The Hermite / block-LDL' form has no pivoting, so its error grows with the spread faster than the
    oracle's pivoted banded LU; this pins the envelope: well inside the north star's 1e-6 on coefficients and
    energy up to a spread of 100 (beyond ~10^3 the snap coefficients drift to 1e-4..1e-3)."""
    import allocnet_amd as aa
    rng = np.random.default_rng(100 + s + N)
    B = 300
    head, tail, wps, T = random_problem(rng, B, N, c)
    for half_decades, tol in ((0.7, 1e-8), (1.0, 1e-6)):
        Tm = 10.0 ** rng.uniform(-half_decades, half_decades, size=T.shape)
        co, en = aa.minco_solve(head, tail, wps, Tm, s, ctx=anet_ctx)
        cc, ec = cbind.minco_solve_batch(s, head, tail, wps, Tm)
        err = np.array([np.abs(co[b] - cc[b]).max() / np.abs(cc[b]).max() for b in range(B)])
        assert err.max() <= tol, (half_decades, err.max())
        assert np.abs(en - ec).max() <= 1e-9 * np.abs(ec).max()
    # beyond a spread of 50 the host entry point redoes the trajectory with the pivoted collocation solve
    # (anet_minco_solve_wide_spread_dev): 1e-6 on the coefficients holds at spreads of 10^3 and 10^4 as well
    for half_decades in (1.5, 2.0):
        Tm = 10.0 ** rng.uniform(-half_decades, half_decades, size=T.shape)
        co, en = aa.minco_solve(head, tail, wps, Tm, s, ctx=anet_ctx)
        cc, ec = cbind.minco_solve_batch(s, head, tail, wps, Tm)
        err = np.array([np.abs(co[b] - cc[b]).max() / np.abs(cc[b]).max() for b in range(B)])
        assert err.max() <= 1e-6, (half_decades, err.max())
        assert np.abs(en - ec).max() <= 1e-6 * np.abs(ec).max()


def test_wide_spread_solve_touches_only_wide_trajectories(anet_ctx):
    """Device entry point: trajectories below the spread threshold keep the fast kernel's output bit for bit, the
    others are overwritten with the collocation solve (checked against the oracle); min_spread <= 1 redoes all and
    agrees with the fast kernel where that one is accurate."""
    import torch
    import allocnet_amd as aa
    rng = np.random.default_rng(77)
    s, c, N, B = 4, 3, 8, 130
    head, tail, wps, T = random_problem(rng, B, N, c)
    wide = np.arange(B) % 3 == 0
    T[wide] = 10.0 ** rng.uniform(-1.5, 1.5, size=(int(wide.sum()), N))
    ld = aa.recommended_ld(B)
    dev = torch.device("cuda", 0)

    def bm(x):
        f = np.ascontiguousarray(x.reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(dev)
        return t
    th, tt, tw, tT = bm(head), bm(tail), bm(wps), bm(T)
    co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64); en = torch.empty(ld, device=dev, dtype=torch.float64)
    aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=anet_ctx)
    fast = co[:, :B].T.cpu().numpy().reshape(B, N, 3, 2 * s).copy(); efast = en[:B].cpu().numpy().copy()
    aa.minco.minco_solve_wide_spread_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, min_spread=50.0, ctx=anet_ctx)
    after = co[:, :B].T.cpu().numpy().reshape(B, N, 3, 2 * s); eafter = en[:B].cpu().numpy()
    spread = T.max(axis=1) / T.min(axis=1)
    keep = spread <= 50.0
    assert keep.sum() > 40 and (~keep).sum() > 20
    assert np.array_equal(after[keep], fast[keep]) and np.array_equal(eafter[keep], efast[keep])
    cc, ec = cbind.minco_solve_batch(s, head, tail, wps, T)
    err = np.array([np.abs(after[b] - cc[b]).max() / np.abs(cc[b]).max() for b in range(B)])
    assert err[~keep].max() <= 1e-7 and not np.array_equal(after[~keep], fast[~keep])
    assert np.abs(eafter - ec)[~keep].max() <= 1e-7 * np.abs(ec).max()
    aa.minco.minco_solve_wide_spread_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, min_spread=0.0, ctx=anet_ctx)
    allc = co[:, :B].T.cpu().numpy().reshape(B, N, 3, 2 * s)
    assert np.abs(allc[keep] - fast[keep]).max() <= 1e-8 * np.abs(fast[keep]).max()


@pytest.mark.parametrize("s,c,N", [(4, 3, 5), (3, 3, 4), (4, 4, 3)])
def test_torch_minco_layer_gradients(anet_ctx, s, c, N):
    """allocnet_amd.torch_layer.minco_layer: a loss of the coefficients AND the energy back-propagated to the waypoints and
    durations (propogateGrad fed by torch) against central differences of the same loss through re-solved trajectories."""
    import torch
    import allocnet_amd as aa
    from allocnet_amd.torch_layer import minco_layer
    rng = np.random.default_rng(50 + s + N)
    B = 7
    head, tail, wps, T = random_problem(rng, B, N, c)
    dev = torch.device("cuda", 0)
    ld = aa.recommended_ld(B)

    def bm(a):
        f = np.ascontiguousarray(a.reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(dev)
        return t
    th, tt, tw, tT = bm(head), bm(tail), bm(wps), bm(T)
    tT[:, B:] = 1.0                                                       # (padding columns: any positive duration)
    D = 2 * s
    w = rng.normal(size=(N, 3, D))
    wb = torch.zeros(N * 3 * D, ld, device=dev, dtype=torch.float64); wb[:, :B] = torch.from_numpy(np.tile(w.reshape(-1, 1), (1, B))).to(dev)
    tw.requires_grad_(True); tT.requires_grad_(True)
    co, en = minco_layer(tw, tT, th, tt, s, c, N, B, ctx=anet_ctx)
    loss = (wb * co)[:, :B].sum() + 0.01 * en[:B].sum()
    loss.backward()
    gP = tw.grad[:, :B].cpu().numpy().T.reshape(B, N - 1, 3); gT = tT.grad[:, :B].cpu().numpy().T

    def loss_np(wp_, T_):
        cc, ee = aa.minco_solve(head, tail, wp_, T_, s, ctx=anet_ctx)
        return (w[None] * cc).sum(axis=(1, 2, 3)) + 0.01 * ee
    dP = rng.normal(size=wps.shape); dT = rng.normal(size=T.shape) * T * 0.2
    h = 1e-6
    fd = (loss_np(wps + h * dP, T + h * dT) - loss_np(wps - h * dP, T - h * dT)) / (2 * h)
    an = (gP * dP).sum(axis=(1, 2)) + (gT * dT).sum(axis=1)
    sc = np.abs(gP * dP).sum(axis=(1, 2)) + np.abs(gT * dT).sum(axis=1)
    assert (np.abs(an - fd) <= 1e-6 * sc).all(), np.abs(an - fd) / sc


@pytest.mark.parametrize("s,c,N", [(4, 3, 8), (3, 3, 16), (4, 3, 5), (4, 4, 7), (3, 2, 3), (2, 2, 4), (3, 3, 1)])
def test_time_allocation_sampling_matches_the_replicated_solve(anet_ctx, s, c, N):
    """anet_minco_sample_costs[_dev]: K candidate duration vectors for P problems in one launch; the cost of every sample
    equals energy + rho * sum T of a full solve of the replicated problem (same arithmetic: to rounding), the C oracle
    agrees to 1e-9, and the argmin over a problem's samples is the oracle's."""
    import torch
    import allocnet_amd as aa
    rng = np.random.default_rng(300 + 10 * N + s)
    P, K, rho = 5, 1000, 3.0
    head, tail, wps, _ = random_problem(rng, P, N, c)
    T = rng.uniform(0.4, 2.5, size=(P, K, N))
    # host variant, one problem
    cost0 = aa.minco_sample_costs(head[0], tail[0], wps[0], T[0], s, rho=rho, ctx=anet_ctx)
    rep = lambda x: np.repeat(x[0:1], K, axis=0)
    _, e0 = aa.minco_solve(rep(head), rep(tail), rep(wps), T[0], s, ctx=anet_ctx)
    ref0 = e0 + rho * T[0].sum(axis=1)
    assert np.abs(cost0 - ref0).max() <= 1e-13 * np.abs(ref0).max()
    _, eo = cbind.minco_solve_batch(s, rep(head), rep(tail), rep(wps), T[0], want_coeffs=False)
    oc = eo + rho * T[0].sum(axis=1)
    assert np.abs(cost0 - oc).max() <= 1e-9 * np.abs(oc).max()
    assert int(np.argmin(cost0)) == int(np.argmin(oc))
    # device variant, P problems x K samples, problem data with its own row stride
    dev = torch.device("cuda", 0)
    ld, ldp = aa.recommended_ld(P * K), 8
    def bm(a, n, stride):
        f = np.ascontiguousarray(a.reshape(n, -1).T)
        t = torch.zeros(f.shape[0], stride, device=dev, dtype=torch.float64)
        t[:, :n] = torch.from_numpy(f).to(dev)
        return t
    th, tt = bm(head, P, ldp), bm(tail, P, ldp)
    tw = bm(wps, P, ldp) if N > 1 else None
    tT = bm(T.reshape(P * K, N), P * K, ld)
    cost = aa.minco_sample_costs_dev(th, tt, tw, tT, s, c, N, P, K, rho=rho, ctx=anet_ctx).cpu().numpy().reshape(P, K)
    for p in range(P):
        repp = lambda x: np.repeat(x[p:p + 1], K, axis=0)
        _, ep = aa.minco_solve(repp(head), repp(tail), repp(wps), T[p], s, ctx=anet_ctx)
        refp = ep + rho * T[p].sum(axis=1)
        assert np.abs(cost[p] - refp).max() <= 1e-13 * np.abs(refp).max(), p
    assert np.array_equal(cost[0], cost0)


@pytest.mark.gpu
def test_bound_solve_call_is_the_same_launch(anet_ctx):
    """bind_minco_solve: the arguments of minco_solve_dev checked and converted once; calling the bound object gives the same bits
    as the wrapper, sees new CONTENTS of the bound tensors, runs on another stream through with_stream, and a tensor of the wrong
    kind is refused when it is bound (not at the launch)."""
    import torch
    import allocnet_amd as aa
    rng = np.random.default_rng(5)
    dev = torch.device("cuda", 0)
    for s, c, N, B in ((4, 3, 8, 1024), (3, 3, 16, 100), (4, 4, 5, 37)):
        head, tail, wps, T = random_problem(rng, B, N, c)
        ld = aa.recommended_ld(B)

        def bm(x):
            f = np.ascontiguousarray(x.reshape(B, -1).T)
            t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
            t[:, :B] = torch.from_numpy(f).to(dev)
            return t
        th, tt, tw, tT = bm(head), bm(tail), bm(wps), bm(T)
        co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64); en = torch.empty(ld, device=dev, dtype=torch.float64)
        aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=anet_ctx)
        torch.cuda.synchronize()
        ref_c, ref_e = co[:, :B].clone(), en[:B].clone()
        co2 = torch.zeros_like(co); en2 = torch.zeros_like(en)
        call = aa.bind_minco_solve(th, tt, tw, tT, s, c, N, B, coeffs=co2, energy=en2, ctx=anet_ctx)
        call()
        torch.cuda.synchronize()
        assert torch.equal(co2[:, :B], ref_c) and torch.equal(en2[:B], ref_e)
        tT[:, :B] *= 1.25                                         # new contents, same storage
        aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=anet_ctx)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        call.with_stream(side.cuda_stream)()
        side.synchronize(); torch.cuda.synchronize()
        assert torch.equal(co2[:, :B], co[:, :B]) and torch.equal(en2[:B], en[:B]) and not torch.equal(en2[:B], ref_e)
        cc, ec = cbind.minco_solve_batch(s, head, tail, wps, T * 1.25)
        assert rel_err(en2[:B].cpu().numpy(), ec) < 1e-10
    with pytest.raises(ValueError):
        aa.bind_minco_solve(th.float(), tt, tw, tT, s, c, N, B, coeffs=co2, energy=en2, ctx=anet_ctx)
    with pytest.raises(ValueError):
        aa.bind_minco_solve(th, tt, tw, tT, s, c, N, B, coeffs=co2, energy=en2[: B - 1], ctx=anet_ctx)
