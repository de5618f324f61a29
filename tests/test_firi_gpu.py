"""Batched FIRI on the GPU vs the numpy restatement of firi.hpp (oracle/firi_np.py; parity unpinned by the
reference, see its header) and through the properties the algorithm guarantees."""
import numpy as np
import pytest

from oracle import firi_np as F
from tests.util import firi_scene as make_case, firi_pack as pack

pytestmark = pytest.mark.gpu


def check_properties(hp, bd, pts, a, b, eps=1e-6):
    assert (hp @ np.r_[a, 1.0]).max() <= eps and (hp @ np.r_[b, 1.0]).max() <= eps         # contains the segment
    if len(pts):
        assert ((pts @ hp[:, :3].T + hp[:, 3]).max(axis=1) > -2 * eps).all()              # no obstacle point strictly inside


def test_planes_match_oracle_exactly(anet_ctx):
    """iterations = 1: the polytope around the unit ball at the segment midpoint -- no optimisation involved,
    so rows, their order and their count must agree to rounding."""
    import allocnet_amd as aa
    rng = np.random.default_rng(5)
    cases = [make_case(rng, n) for n in (0, 1, 7, 60, 300, 300, 1000, 257)]
    bd, pc, npts, a, b = pack(cases)
    out = aa.firi(bd, pc, a, b, n_points=npts, max_rows=96, params=aa.firi_params(iterations=1), ctx=anet_ctx)
    assert (out["ok"] == 1).all()
    for i, c in enumerate(cases):
        ok, hp0 = F.firi(c[0], c[1], c[2], c[3], iterations=1)
        k = out["n_rows"][i]
        assert ok and k == hp0.shape[0], (i, k, hp0.shape)
        assert np.abs(out["hpoly"][i, :k] - hp0).max() <= 1e-12 * max(1.0, np.abs(hp0).max()), i
        assert (out["hpoly"][i, k:] == 0).all()
        check_properties(hp0, *c)


def test_full_firi_against_oracle_and_properties(anet_ctx):
    import allocnet_amd as aa
    rng = np.random.default_rng(11)
    cases = [make_case(rng, n) for n in (40, 120, 400, 400, 800, 25)]
    bd, pc, npts, a, b = pack(cases)
    vols = []
    for iters in (2, 4):
        out = aa.firi(bd, pc, a, b, n_points=npts, max_rows=96, params=aa.firi_params(iterations=iters), ctx=anet_ctx)
        assert (out["ok"] == 1).all()
        v = []
        for i, c in enumerate(cases):
            k = out["n_rows"][i]
            hp = out["hpoly"][i, :k]
            check_properties(hp, *c)
            tr = []
            ok, hp0 = F.firi(c[0], c[1], c[2], c[3], iterations=iters, trace=tr)
            R = out["ellipsoid"][i, :9].reshape(3, 3); p = out["ellipsoid"][i, 9:12]; r = out["ellipsoid"][i, 12:15]
            assert abs(np.linalg.det(R) - 1.0) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9
            # the ellipsoid the last polytope was built around is the oracle's, to the optimiser's tolerance
            R0, p0, r0 = tr[-1]["R"], tr[-1]["p"], tr[-1]["r"]
            Q = R @ np.diag(r * r) @ R.T; Q0 = R0 @ np.diag(r0 * r0) @ R0.T
            assert np.abs(Q - Q0).max() <= 2e-2 * np.abs(Q0).max(), (iters, i)
            assert np.abs(p - p0).max() <= 2e-2 * max(1.0, np.abs(r0).max()), (iters, i)
            assert abs(np.prod(r) - np.prod(r0)) <= 2e-2 * np.prod(r0)
            # the inscribed ellipsoid lies inside the polytope it was inscribed in (previous pass): support function
            if iters == 2:
                hprev = tr[0]["hPoly"]
                nrm = np.linalg.norm(hprev[:, :3] @ R @ np.diag(r), axis=1)
                assert (nrm + hprev[:, :3] @ p + hprev[:, 3]).max() <= 2e-2
            v.append(np.prod(r))
        vols.append(np.array(v))
    assert (vols[1] >= vols[0] * (1 - 1e-3)).all()  # inflation is monotone


def test_failure_codes_and_edge_cases(anet_ctx):
    import allocnet_amd as aa
    rng = np.random.default_rng(2)
    cases = [make_case(rng, 50) for _ in range(4)]
    bd, pc, npts, a, b = pack(cases)
    a2 = a.copy(); a2[1] += np.array([100.0, 0, 0])            # a outside the bounding box: firi returns false
    out = aa.firi(bd, pc, a2, b, n_points=npts, max_rows=64, ctx=anet_ctx)
    assert list(out["ok"]) == [1, 0, 1, 1] and out["n_rows"][1] == 0
    out = aa.firi(bd, pc, a, b, n_points=npts, max_rows=6, params=aa.firi_params(iterations=1), ctx=anet_ctx)
    assert (out["ok"] == -1).all()                              # 6 box planes + at least one point: does not fit
    # no obstacle points at all: the polytope is the bounding box
    out = aa.firi(bd, np.zeros((4, 0, 3)), a, b, max_rows=16, ctx=anet_ctx)
    assert (out["ok"] == 1).all() and (out["n_rows"] == 6).all()
    for i in range(4):
        got = out["hpoly"][i, :6]
        nrm = np.linalg.norm(got[:, :3], axis=1, keepdims=True)
        want = {tuple(np.round(r, 9)) for r in bd[i]}
        assert {tuple(np.round(r, 9)) for r in got / nrm} == want
    with pytest.raises(aa.AnetError):
        aa.firi(bd, pc, a, b, max_rows=2, ctx=anet_ctx)


def test_convex_cover_batches_the_path(anet_ctx):
    """sfc_gen::convexCover mirror: consecutive polytopes overlap (share the junction point), every
    polytope contains its own segment, obstacle points stay outside."""
    import allocnet_amd as aa
    rng = np.random.default_rng(3)
    path = [np.array([0.0, 0.0, 1.0]), np.array([4.0, 1.0, 1.5]), np.array([6.0, 4.0, 1.0]), np.array([9.0, 4.5, 2.0])]
    pts = rng.uniform([-3, -3, 0], [12, 8, 4], size=(3000, 3))
    keep = np.ones(len(pts), dtype=bool)
    for p0, p1 in zip(path[:-1], path[1:]):
        d = p1 - p0
        t = np.clip(((pts - p0) @ d) / (d @ d), 0, 1)
        keep &= np.linalg.norm(pts - (p0 + t[:, None] * d), axis=1) > 0.6
    pts = pts[keep]
    polys = aa.convex_cover(path, pts, [-3, -3, 0], [12, 8, 4], progress=2.0, rng_range=3.0, ctx=anet_ctx)
    assert len(polys) >= 6
    for hp in polys:
        assert hp.shape[1] == 4 and hp.shape[0] >= 4
        assert ((pts @ hp[:, :3].T + hp[:, 3]).max(axis=1) > -2e-6).all()
    # the walk of convexCover: every junction point lies in some pair of consecutive polytopes
    q = path[0]
    for hp in polys[:1]:
        assert (hp @ np.r_[q, 1.0]).max() <= 1e-6
    assert (polys[-1] @ np.r_[path[-1], 1.0]).max() <= 1e-6


def test_random_scenes_keep_the_guarantees(anet_ctx):
    """300 random scenes, sparse to dense, obstacles from 5 cm off the segment outwards: every polytope
    contains its segment, excludes every obstacle point, and is bounded by the box."""
    import allocnet_amd as aa
    rng = np.random.default_rng(99)
    cases = []
    for i in range(300):
        n = int(rng.choice([3, 20, 100, 400, 1200]))
        cases.append(make_case(rng, n, clearance=float(rng.choice([0.05, 0.2, 0.6])), box=float(rng.uniform(1.0, 3.0))))
    bd, pc, npts, a, b = pack(cases)
    out = aa.firi(bd, pc, a, b, n_points=npts, max_rows=128, ctx=anet_ctx)
    assert (out["ok"] == 1).all(), np.unique(out["ok"], return_counts=True)
    for i, c in enumerate(cases):
        k = out["n_rows"][i]
        assert 4 <= k <= 128
        hp = out["hpoly"][i, :k]
        assert np.isfinite(hp).all()
        check_properties(hp, *c)
        r = out["ellipsoid"][i, 12:15]
        assert np.isfinite(out["ellipsoid"][i]).all() and (r > 0).all()


def test_device_pointer_entry_point(anet_ctx):
    """anet_firi_dev: device tensors in and out, caller-provided workspace, asynchronous -- same polytopes as the host
    entry point (which is a thin copy-in / copy-out wrapper around it)."""
    import torch
    import allocnet_amd as aa
    rng = np.random.default_rng(23)
    cases = [make_case(rng, n) for n in (30, 200, 600, 0, 90)]
    bd, pc, npts, a, b = pack(cases)
    host = aa.firi(bd, pc, a, b, n_points=npts, max_rows=80, ctx=anet_ctx)
    dev = torch.device("cuda", 0)
    t = lambda x, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(x)).to(dev, dtype=dt)
    out = aa.firi_dev(t(bd), t(pc), t(npts, torch.int32), t(a), t(b), max_rows=80, ctx=anet_ctx)
    torch.cuda.synchronize()
    assert np.array_equal(out["ok"].cpu().numpy(), host["ok"]) and np.array_equal(out["n_rows"].cpu().numpy(), host["n_rows"])
    hp = out["hpoly"].cpu().numpy()
    for i in range(len(cases)):
        k = host["n_rows"][i]
        # same kernels, same inputs; the L-BFGS history of the MVIE stage lives in registers: identical results
        assert np.abs(hp[i, :k] - host["hpoly"][i, :k]).max() <= 1e-9 * max(1.0, np.abs(host["hpoly"][i, :k]).max())
        assert (hp[i, k:] == 0).all()
    assert np.abs(out["ellipsoid"].cpu().numpy() - host["ellipsoid"]).max() <= 1e-9


def _random_polytope(rng, centre, n_extra):
    """A box about `centre` cut by random planes that keep the centre inside; raw form, rows NOT normalised."""
    half = rng.uniform(0.5, 2.0, size=3)
    rows = []
    for ax in range(3):
        e = np.zeros(3); e[ax] = 1.0
        rows.append(np.r_[e, -(centre[ax] + half[ax])]); rows.append(np.r_[-e, centre[ax] - half[ax]])
    for _ in range(n_extra):
        n = rng.standard_normal(3) * rng.uniform(0.3, 3.0)
        off = rng.uniform(0.2, 1.5) * np.linalg.norm(n)
        rows.append(np.r_[n, -(n @ centre) - off])
    return np.array(rows)


def test_polytope_depth_matches_the_linear_programme(anet_ctx):
    """geo_utils::findInterior / overlap: the LP optimum against HiGHS, normalised and raw rows, empty polytopes,
    padded batches."""
    import allocnet_amd as aa
    rng = np.random.default_rng(21)
    polys = [_random_polytope(rng, rng.uniform(-3, 3, size=3), n) for n in (0, 1, 5, 12, 30, 58)]
    # two empty ones: opposite half-spaces that exclude each other, and a box with a cut beyond it
    empty = np.array([[1.0, 0, 0, 1.0], [-1.0, 0, 0, 1.0], [0, 1, 0, -1], [0, -1, 0, -1], [0, 0, 1, -1], [0, 0, -1, -1]])
    polys += [empty, np.vstack([_random_polytope(rng, np.zeros(3), 0), [[1.0, 0, 0, 50.0]]])]
    for normalise in (True, False):
        d, x = aa.polytope_depth(polys, normalise=normalise, ctx=anet_ctx)
        for i, hp in enumerate(polys):
            d0, _ = F.polytope_depth(hp, normalise)
            # feasible with depth < 0 still has an LP optimum: compare wherever HiGHS found one
            assert np.isfinite(d[i]) == np.isfinite(d0), (i, d[i], d0)
            if np.isfinite(d0):
                assert abs(d[i] - d0) <= 1e-9 * max(1.0, abs(d0)), (i, normalise, d[i], d0)
                nrm = np.linalg.norm(hp[:, :3], axis=1) if normalise else 1.0
                slack = -(hp[:, :3] @ x[i] + hp[:, 3]) / nrm
                assert slack.min() >= d[i] - 1e-9 * max(1.0, abs(d[i]))           # the point attains the depth
    ok, pt = aa.find_interior(polys[3], ctx=anet_ctx)
    assert ok and (polys[3] @ np.r_[pt, 1.0]).max() < 0.0
    assert not aa.find_interior(polys[-1], ctx=anet_ctx)[0] and not aa.find_interior(empty, ctx=anet_ctx)[0]
    assert aa.overlap(polys[2], polys[2], ctx=anet_ctx)
    ok2, mid = aa.overlap_pt(polys[2], polys[2], ctx=anet_ctx)
    assert ok2 and (polys[2] @ np.r_[mid, 1.0]).max() < 0.0
    far = _random_polytope(rng, np.array([40.0, 0, 0]), 3)
    assert not aa.overlap(polys[2], far, ctx=anet_ctx) and not F.overlap(polys[2], far)


def test_short_cut_follows_the_reference_walk(anet_ctx):
    """sfc_gen::shortCut: the kept indices against the restated loop (oracle/firi_np.py) on chains of polytopes with
    varying overlap, the single-polytope duplication, and a corridor out of convex_cover."""
    import allocnet_amd as aa
    rng = np.random.default_rng(8)
    for trial in range(6):
        M = int(rng.integers(2, 12))
        step = rng.uniform(0.3, 2.5)
        centres = np.cumsum(rng.uniform(0.2, 1.0, size=(M, 3)) * step, axis=0)
        chain = [_random_polytope(rng, c, int(rng.integers(0, 8))) for c in centres]
        idx = F.short_cut(chain, 0.1)
        out = aa.short_cut(chain, ctx=anet_ctx)
        assert len(out) == len(idx) and all((o == chain[k]).all() for o, k in zip(out, idx)), (trial, idx)
        assert idx[-1] == M - 1 and idx[0] == 0 and all(a < b for a, b in zip(idx[:-1], idx[1:]))
    one = [_random_polytope(rng, np.zeros(3), 4)]
    out = aa.short_cut(one, ctx=anet_ctx)
    assert len(out) == 2 and (out[0] == one[0]).all() and (out[1] == one[0]).all()
    # a real corridor: convexCover then shortCut, as LearningPlanner does (learning_planner.hpp:274-283)
    path = [np.array([0.0, 0.0, 1.0]), np.array([4.0, 1.0, 1.5]), np.array([6.0, 4.0, 1.0]), np.array([9.0, 4.5, 2.0])]
    pts = rng.uniform([-3, -3, 0], [12, 8, 4], size=(2000, 3))
    keep = np.ones(len(pts), dtype=bool)
    for p0, p1 in zip(path[:-1], path[1:]):
        dd = p1 - p0
        t = np.clip(((pts - p0) @ dd) / (dd @ dd), 0, 1)
        keep &= np.linalg.norm(pts - (p0 + t[:, None] * dd), axis=1) > 0.6
    polys = aa.convex_cover(path, pts[keep], [-3, -3, 0], [12, 8, 4], progress=2.0, rng_range=3.0, ctx=anet_ctx)
    short = aa.short_cut(polys, ctx=anet_ctx)
    idx = F.short_cut(polys, 0.1)
    assert len(short) == len(idx) <= len(polys) and all((o == polys[k]).all() for o, k in zip(short, idx))
    for h0, h1 in zip(short[:-1], short[1:]):                     # consecutive polytopes of the result still meet
        assert np.isfinite(F.polytope_depth(np.vstack([h0, h1]), False)[0])


def test_polytope_depth_ascent_enumeration_and_highs_agree_on_generated_polytopes(anet_ctx, monkeypatch):
    """The certified active-set ascent (default), the vertex enumeration (fallback; forced through its A/B switch) and HiGHS
    on polytopes chosen to be awkward: cubes (degenerate vertices), duplicated / rescaled rows, many cuts, empty sets, stacked
    boxes, cuts through the centre.  An unbounded set gives +inf, which findInterior / overlap read as false."""
    import allocnet_amd as aa
    rng = np.random.default_rng(3)

    def box(c, half):
        rows = []
        for ax in range(3):
            e = np.zeros(3); e[ax] = 1
            rows += [np.r_[e, -(c[ax] + half[ax])], np.r_[-e, c[ax] - half[ax]]]
        return rows
    polys = []
    for t in range(150):
        kind = t % 6
        c = rng.uniform(-20, 20, size=3); half = rng.uniform(0.3, 4, size=3)
        rows = box(c, half)
        if kind == 1:
            rows += [r * rng.uniform(0.1, 10) for r in rows[:3]]
        elif kind == 2:
            for _ in range(rng.integers(1, 60)):
                n = rng.standard_normal(3) * rng.uniform(0.2, 5)
                rows.append(np.r_[n, -(n @ c) - rng.uniform(0.05, 2) * np.linalg.norm(n)])
        elif kind == 3:
            n = rng.standard_normal(3); n /= np.linalg.norm(n); rows.append(np.r_[n, -(n @ c) + 10.0])
        elif kind == 4:
            rows += box(c + rng.uniform(-1, 1, size=3) * half, rng.uniform(0.3, 4, size=3))
        elif kind == 5:
            for _ in range(rng.integers(1, 6)):
                n = rng.standard_normal(3); rows.append(np.r_[n, -(n @ c)])
        polys.append(np.array(rows))
    for normalise in (True, False):
        monkeypatch.delenv("ANET_POLYTOPE_DEPTH_ENUMERATE", raising=False)
        d, x = aa.polytope_depth(polys, normalise=normalise, ctx=anet_ctx)
        monkeypatch.setenv("ANET_POLYTOPE_DEPTH_ENUMERATE", "1")
        d_e, _ = aa.polytope_depth(polys, normalise=normalise, ctx=anet_ctx)
        monkeypatch.delenv("ANET_POLYTOPE_DEPTH_ENUMERATE")
        for i, hp in enumerate(polys):
            d0, _ = F.polytope_depth(hp, normalise)
            tol = 1e-8 * max(1.0, abs(d0))
            assert abs(d[i] - d0) <= tol and abs(d_e[i] - d0) <= tol, (i, normalise, d[i], d_e[i], d0)
            nrm = np.linalg.norm(hp[:, :3], axis=1) if normalise else 1.0
            assert (-(hp[:, :3] @ x[i] + hp[:, 3]) / nrm).min() >= d[i] - tol
    half_space = np.array([[1.0, 0.0, 0.0, -1.0], [0.0, 1.0, 0.0, -1.0]])
    d, _ = aa.polytope_depth([half_space], ctx=anet_ctx)
    assert np.isposinf(d[0]) and not aa.find_interior(half_space, ctx=anet_ctx)[0] and not aa.overlap(half_space, half_space, ctx=anet_ctx)


def test_pass_count_per_corridor_equals_separate_calls(anet_ctx):
    """anet_firi_var: corridors with their own pass count in one batch (what convexCover needs: 4 passes for a segment,
    1 for a gap polytope) give exactly what separate calls with a common count give."""
    import allocnet_amd as aa
    rng = np.random.default_rng(23)
    cases = [make_case(rng, n) for n in (60, 200, 400, 120, 300, 80)]
    bd, pc, npts, a, b = pack(cases)
    its = np.array([4, 1, 2, 1, 4, 3], dtype=np.int32)
    mixed = aa.firi(bd, pc, a, b, n_points=npts, max_rows=96, iterations=its, ctx=anet_ctx)
    assert (mixed["ok"] >= 1).all()
    for k in sorted(set(its.tolist())):
        sel = np.where(its == k)[0]
        sep = aa.firi(bd[sel], pc[sel], a[sel], b[sel], n_points=npts[sel], max_rows=96, params=aa.firi_params(iterations=int(k)),
                      ctx=anet_ctx)
        assert np.array_equal(sep["n_rows"], mixed["n_rows"][sel]), k
        assert np.array_equal(sep["hpoly"], mixed["hpoly"][sel]), k
    with pytest.raises(ValueError):
        aa.firi(bd, pc, a, b, n_points=npts, iterations=np.zeros(6, dtype=np.int32), ctx=anet_ctx)
