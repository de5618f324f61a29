"""Synthetic problem generators of SURVEY.md section 8(d): random-walk waypoints (config 2), random SFC-like
corridors around the segments (configs 3-5), single corridor QPs and FIRI scenes.  Shared by bench.py, tools/ and
tests/ (tests/util.py re-exports them); no arithmetic of the solver lives here."""
import numpy as np


def random_problem(rng, B, N, c, rest=False):
    """Random-walk waypoints (SURVEY.md 8(d) config 2 generator)."""
    d = rng.normal(size=(B, N, 3))
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    d *= rng.uniform(1.0, 3.0, size=(B, N, 1))
    pts = np.concatenate([np.zeros((B, 1, 3)), np.cumsum(d, axis=1)], axis=1)
    pts[:, :, 2] = np.clip(pts[:, :, 2] + 1.0, 0.0, 5.0)
    head = np.zeros((B, 3, c)); tail = np.zeros((B, 3, c))
    head[:, :, 0] = pts[:, 0]; tail[:, :, 0] = pts[:, N]
    if not rest:
        head[:, :, 1:] = rng.normal(size=(B, 3, c - 1)) * 0.5
        tail[:, :, 1:] = rng.normal(size=(B, 3, c - 1)) * 0.5
    wps = pts[:, 1:N].copy()
    T = rng.uniform(0.5, 2.0, size=(B, N))
    return head, tail, wps, T


def corridor_problem(rng, B, N, c, M):
    """SURVEY 8(d) config 3 generator: random-walk waypoints, axis-aligned box around each segment
    inflated by U(0.5,3) plus k~U{0,6} random tangent half-spaces, rows normalised, a.x <= b, padded to M."""
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    pts = np.concatenate([head[:, None, :, 0], wps, tail[:, None, :, 0]], axis=1)
    hp = np.zeros((B, N, M, 4))
    lo = np.minimum(pts[:, :-1], pts[:, 1:]) - rng.uniform(0.5, 3.0, size=(B, N, 3))
    hi = np.maximum(pts[:, :-1], pts[:, 1:]) + rng.uniform(0.5, 3.0, size=(B, N, 3))
    for ax in range(3):
        hp[:, :, 2 * ax, ax] = 1.0; hp[:, :, 2 * ax, 3] = hi[:, :, ax]
        hp[:, :, 2 * ax + 1, ax] = -1.0; hp[:, :, 2 * ax + 1, 3] = -lo[:, :, ax]
    k = rng.integers(0, min(6, M - 6) + 1, size=(B, N))
    mid = 0.5 * (pts[:, :-1] + pts[:, 1:])
    for r in range(min(6, M - 6)):
        a = rng.normal(size=(B, N, 3)); a /= np.linalg.norm(a, axis=2, keepdims=True)
        b = np.einsum("bnk,bnk->bn", a, mid) + rng.uniform(1.0, 3.0, size=(B, N))
        use = (k > r)[..., None]
        hp[:, :, 6 + r, :3] = np.where(use, a, 0.0)
        hp[:, :, 6 + r, 3] = np.where(use[..., 0], b, 0.0)
    return head, tail, wps, T, hp


def qp_corridor_problem(rng, N, M, margin=1.0):
    pts = np.cumsum(np.vstack([np.zeros(3), rng.normal(size=(N, 3)) * 1.5]), axis=0)
    hp = np.zeros((N, M, 4))
    for i in range(N):
        lo = np.minimum(pts[i], pts[i + 1]) - margin
        hi = np.maximum(pts[i], pts[i + 1]) + margin
        for ax in range(3):
            hp[i, 2 * ax, ax] = 1.0; hp[i, 2 * ax, 3] = hi[ax]
            hp[i, 2 * ax + 1, ax] = -1.0; hp[i, 2 * ax + 1, 3] = -lo[ax]
        mid = 0.5 * (pts[i] + pts[i + 1])
        for r in range(6, M - 1):
            a = rng.normal(size=3); a /= np.linalg.norm(a)
            hp[i, r, :3] = a; hp[i, r, 3] = max(a @ pts[i], a @ pts[i + 1]) + rng.uniform(0.3, 1.5)
    ini = np.zeros((3, 3)); fin = np.zeros((3, 3))
    ini[:, 0] = pts[0]; fin[:, 0] = pts[-1]
    T = rng.uniform(1.5, 2.5, size=N)
    return ini, fin, hp, T


def firi_scene(rng, n_pts, clearance=0.5, box=3.0):
    a = rng.uniform(-2, 2, size=3)
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    b = a + d * rng.uniform(0.5, 2.5)
    lo = np.minimum(a, b) - box; hi = np.maximum(a, b) + box
    bd = np.zeros((6, 4))
    for ax in range(3):
        bd[2 * ax, ax] = 1.0; bd[2 * ax, 3] = -hi[ax]
        bd[2 * ax + 1, ax] = -1.0; bd[2 * ax + 1, 3] = lo[ax]
    pts = rng.uniform(lo + 1e-3, hi - 1e-3, size=(4 * n_pts, 3))
    dd = b - a
    t = np.clip(((pts - a) @ dd) / (dd @ dd), 0, 1)
    dist = np.linalg.norm(pts - (a + t[:, None] * dd), axis=1)
    pts = pts[dist > clearance][:n_pts]
    return bd, pts, a, b


def firi_pack(cases):
    B = len(cases)
    Np = max(1, max(len(c[1]) for c in cases))
    bd = np.array([c[0] for c in cases]); a = np.array([c[2] for c in cases]); b = np.array([c[3] for c in cases])
    pc = np.zeros((B, Np, 3)); npts = np.zeros(B, dtype=np.int32)
    for i, c in enumerate(cases):
        pc[i, :len(c[1])] = c[1]; npts[i] = len(c[1])
    return bd, pc, npts, a, b
