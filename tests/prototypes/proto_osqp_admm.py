#!/usr/bin/env python3
"""Dense numpy restatement of OSQP's published algorithm (Stellato et al., "OSQP: an operator splitting solver for quadratic
programs", Math. Prog. Comp. 2020: Algorithm 1, section 5.1 modified Ruiz equilibration, section 5.2 rho estimate) on the
reference-assembled matrices of QPSolver::solve (oracle/minco_np.qp_assemble) -- what k_qp_admm's iteration counts and
unsolved rates are held against, OSQP itself not being in the image.  CPU only (test side: it uses the oracle).

    python tests/prototypes/proto_osqp_admm.py <order> <pieces> <problems> <seed> <duration scale> [variant]

variant: "osqp" (default): 10 Ruiz passes on the raw matrices, cost scale c = 1 / max(mean column norm of P, 1), termination on the
UNSCALED residuals, rho estimate from the SCALED residuals (osqp/src/auxil.c compute_rho_estimate), rho_eq = 1e3 rho, adaptation
tested every 100 iterations with tolerance 5;  "raw": no scaling at all.
Round 5 (profiles/r05_qp_unsolved_admm.txt): 8-piece snap, seed 1, durations x 1.5: 0 of 30 unsolved at 4000 iterations, mean 540;
seed 5, durations x 1.0: 18 of 120 unsolved, mean 1380 (k_qp_admm on that set: 49 of 512 at max_iter, 21 of them infeasible)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import minco_np as onp
from allocnet_amd.synth import corridor_problem


def dense_problem(s, head, tail, hp, T, res=20, vmax=4.0, amax=6.0):
    N = len(T); D = 2 * s; n = 3 * D * N; M = hp.shape[1]
    st9 = np.zeros((9, 2))
    for ax in range(3):
        st9[3 * ax:3 * ax + 3, 0] = head[ax, :3]; st9[3 * ax:3 * ax + 3, 1] = tail[ax, :3]
    Q, A, b, G1, h1, G2, h2 = onp.qp_assemble(s, st9, np.transpose(hp, (1, 2, 0)), np.full(N, M), T, res, vmax, amax)
    G = np.zeros((G1.shape[0] + G2.shape[0], n)); r = 0
    for i in range(N):
        for _ in range(res):
            G[r:r + M, i * 3 * D:(i + 1) * 3 * D] = G1[r:r + M]; r += M
    r2 = 0
    for i in range(N):
        for _ in range(res):
            for j in range(3):
                G[r + r2:r + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2[r2:r2 + 4]; r2 += 4
    hh = np.r_[h1, h2]; keep = np.abs(G).sum(axis=1) > 0
    return Q, A, b, G[keep], hh[keep]


def ruiz(P, A, passes=10):
    """osqp/src/scaling.c scale_data with q = 0"""
    n, m = P.shape[0], A.shape[0]
    Dv, Ev, c = np.ones(n), np.ones(m), 1.0
    P, A = P.copy(), A.copy()
    lim = lambda v: np.where(v < 1e-4, 1.0, np.minimum(v, 1e4))
    for _ in range(passes):
        dd = 1.0 / np.sqrt(lim(np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0))))
        de = 1.0 / np.sqrt(lim(np.abs(A).max(axis=1)))
        P = dd[:, None] * P * dd[None, :]; A = de[:, None] * A * dd[None, :]
        Dv *= dd; Ev *= de
        g = 1.0 / max(float(lim(np.abs(P).max(axis=0).mean())), 1.0)      # (|q|_inf = 0 -> limit_scaling -> 1)
        P *= g; c *= g
    return P, A, Dv, Ev, c


def admm(P, A, l, u, is_eq, Dv, Ev, c, rho=0.1, sigma=1e-6, alpha=1.6, eps=1e-3, max_iter=4000, check=25, adapt=100):
    n, m = P.shape[0], A.shape[0]
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)

    def fac(rho):
        rv = np.where(is_eq, 1e3 * rho, rho)
        return np.linalg.cholesky(P + sigma * np.eye(n) + A.T @ (rv[:, None] * A)), rv
    L, rv = fac(rho)
    for it in range(1, max_iter + 1):
        xt = np.linalg.solve(L.T, np.linalg.solve(L, sigma * x + A.T @ (rv * z - y)))
        zt = A @ xt
        x = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        zn = np.clip(zr + y / rv, l, u)
        y = y + rv * (zr - zn); z = zn
        if it % check == 0:
            Ax, Px, Aty = A @ x, P @ x, A.T @ y
            rp = np.abs((Ax - z) / Ev).max(); rd = np.abs((Px + Aty) / Dv).max() / c
            npr = max(np.abs(Ax / Ev).max(), np.abs(z / Ev).max()); ndr = max(np.abs(Px / Dv).max(), np.abs(Aty / Dv).max()) / c
            if rp <= eps + eps * npr and rd <= eps + eps * ndr:
                return it
            if it % adapt == 0:
                a_ = np.abs(Ax - z).max() / (max(np.abs(Ax).max(), np.abs(z).max()) + 1e-10)
                b_ = np.abs(Px + Aty).max() / (max(np.abs(Px).max(), np.abs(Aty).max()) + 1e-10)
                rn = float(np.clip(rho * np.sqrt(a_ / (b_ + 1e-10)), 1e-6, 1e6))
                if rn > 5 * rho or rn < 0.2 * rho:
                    rho = rn; L, rv = fac(rho)
    return -1


if __name__ == "__main__":
    s, N, nprob, seed, sc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    variant = sys.argv[6] if len(sys.argv) > 6 else "osqp"
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(seed), max(nprob, 512), N, 3, 16)
    T = T * sc
    its = []
    for b in range(nprob):
        Q, A, bb, G, h = dense_problem(s, head[b], tail[b], hp[b], T[b])
        Aall = np.vstack([A, G]); l = np.r_[bb, np.full(G.shape[0], -np.inf)]; u = np.r_[bb, h]
        is_eq = np.r_[np.ones(len(bb), bool), np.zeros(G.shape[0], bool)]
        if variant == "osqp":
            Ps, As, Dr, Er, c = ruiz(Q, Aall)
            its.append(admm(Ps, As, l * Er, u * Er, is_eq, Dr, Er, c))
        else:
            its.append(admm(Q, Aall, l, u, is_eq, np.ones(Q.shape[0]), np.ones(Aall.shape[0]), 1.0))
        if b % 10 == 9:
            a = np.array(its)
            print(b + 1, variant, "unsolved at 4000:", int((a < 0).sum()), "iterations mean", np.where(a < 0, 4000, a).mean().round(0),
                  "median", np.median(np.where(a < 0, 4000, a)), flush=True)
    print("unsolved:", [i for i, v in enumerate(its) if v < 0])
