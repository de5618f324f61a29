import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "qp_*.npz")))


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


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


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
