#!/usr/bin/env python3
"""Prototype (CPU, numpy): primal-dual interior-point method on the reference's QP after (i) time
normalisation and (ii) elimination of the equality constraints -- how many Newton steps does it need,
how accurate is it, what happens on infeasible instances?  Decides whether a GPU IPM is worth building
next to the ADMM kernel (which needs ~1200 iterations on the 8-piece snap problems)."""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import minco_np as onp  # noqa: E402  (prototype / analysis script, not product code)


def dense(s, ini, fin, hp, T, res, vmax, amax):
    N, M = hp.shape[0], hp.shape[1]
    state = np.zeros((9, 2))
    for ax in range(3):
        state[3 * ax:3 * ax + 3, 0] = ini[ax]; state[3 * ax:3 * ax + 3, 1] = fin[ax]
    Q, A, b, G1, h1, G2, h2 = onp.qp_assemble(s, state, np.transpose(hp, (1, 2, 0)), np.full(N, M), T, res, vmax, amax)
    D = 2 * s; n = 3 * D * N
    G = np.zeros((G1.shape[0] + G2.shape[0], n)); r = 0
    for i in range(N):
        for _ in range(res):
            G[r:r + M, i * 3 * D:(i + 1) * 3 * D] = G1[r:r + M]; r += M
    r2 = 0
    for i in range(N):
        for _ in range(res):
            for j in range(3):
                G[r + r2:r + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2[r2:r2 + 4]; r2 += 4
    hh = np.r_[h1, h2]
    keep = (np.abs(G).sum(axis=1) > 0) | (hh != 0)
    return Q, A, b, G[keep], hh[keep]


def normalise(s, N, T, Q, A, b, G, h):
    """c~_k = c_k T^k (column scaling) and unit-ish rows."""
    D = 2 * s
    col = np.zeros(3 * D * N)
    for i in range(N):
        for ax in range(3):
            for cc in range(D):
                col[i * 3 * D + ax * D + cc] = T[i] ** (D - 1 - cc)
    Qn = Q / col[:, None] / col[None, :]
    An = A / col[None, :]; Gn = G / col[None, :]
    ra = 1.0 / np.maximum(np.abs(An).max(axis=1), 1e-300)
    rg = 1.0 / np.maximum(np.abs(Gn).max(axis=1), 1e-300)
    return Qn, An * ra[:, None], b * ra, Gn * rg[:, None], h * rg, col


def ipm(P, q, G, h, tol=1e-9, max_iter=60):
    """min 1/2 y'Py + q'y  s.t. Gy <= h   (Mehrotra predictor-corrector)."""
    n, m = P.shape[0], G.shape[0]
    y = np.zeros(n)
    sl = np.maximum(h - G @ y, 1.0); lam = np.ones(m)
    hist = []
    for it in range(max_iter):
        rd = P @ y + q + G.T @ lam
        rg = G @ y + sl - h
        mu = sl @ lam / m
        pres = np.abs(rg).max() / max(1.0, np.abs(h).max())
        dres = np.abs(rd).max() / max(1.0, np.abs(P @ y).max(), np.abs(G.T @ lam).max())
        hist.append((mu, pres, dres))
        if pres < tol and dres < tol and mu < tol * max(1.0, abs(y @ P @ y)):
            return y, lam, it, True, hist
        W = lam / sl
        H = P + G.T @ (W[:, None] * G)
        H = 0.5 * (H + H.T) + 1e-11 * np.trace(H) / n * np.eye(n)
        cf = sla.cho_factor(H)

        def step(rc):
            dy = sla.cho_solve(cf, -rd - G.T @ ((lam * rg - rc) / sl))
            ds = -rg - G @ dy
            dl = (-rc - lam * ds) / sl
            return dy, ds, dl

        def mx(v, dv, fr):
            neg = dv < 0
            return min(1.0, fr * (-v[neg] / dv[neg]).min()) if neg.any() else 1.0
        dy, ds, dl = step(sl * lam)
        al = min(mx(sl, ds, 1.0), mx(lam, dl, 1.0))
        sig = ((sl + al * ds) @ (lam + al * dl) / m / mu) ** 3
        dy, ds, dl = step(sl * lam + ds * dl - sig * mu)
        al = min(mx(sl, ds, 0.99), mx(lam, dl, 0.99))
        y = y + al * dy; sl = sl + al * ds; lam = lam + al * dl
        if not np.isfinite(y).all():
            break
    return y, lam, max_iter, False, hist


def solve_one(s, ini, fin, hp, T, res, vmax, amax):
    N = hp.shape[0]
    Q, A, b, G, h = dense(s, ini, fin, hp, T, res, vmax, amax)
    Qn, An, bn, Gn, hn, col = normalise(s, N, T, Q, A, b, G, h)
    # equality elimination: c~ = Z y + c0
    c0 = np.linalg.lstsq(An, bn, rcond=None)[0]
    Z = sla.null_space(An)
    P = Z.T @ Qn @ Z; q = Z.T @ Qn @ c0
    y, lam, it, ok, hist = ipm(P, q, Gn @ Z, hn - Gn @ c0)
    ct = Z @ y + c0
    z = ct / col
    return z, 0.5 * z @ Q @ z, it, ok, hist, (Q, A, b, G, h)


def main():
    from tests.util import corridor_problem
    rng = np.random.default_rng(1)
    for (s, N, M, scale) in [(4, 8, 16, 1.5), (4, 8, 16, 0.4), (4, 8, 16, 5.0), (3, 16, 16, 1.5), (3, 5, 16, 1.5)]:
        head, tail, wps, T, hp = corridor_problem(rng, 6, N, 3, M)
        its = []
        for bb in range(6):
            z, obj, it, ok, hist, mats = solve_one(s, head[bb], tail[bb], hp[bb], T[bb] * scale, 20, 4.0, 6.0)
            Q, A, b, G, h = mats
            viol = max(np.abs(A @ z - b).max(), (G @ z - h).max())
            its.append((it, ok, "%.4g" % obj, "viol %.1e" % viol, "mu %.1e" % hist[-1][0]))
        print(s, N, "T x", scale, its)


if __name__ == "__main__":
    main()
