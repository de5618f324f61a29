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
