"""The problems of the soak (tests/soak/soak_qp.py, same random stream) that the C port solves and the kernel does not: the
kernel's verdict and step count, the port's, and the dense oracle's.   gpurun -- 'python tests/soak/qp_port_only.py 1500'"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import allocnet_amd as aa
from oracle import cbind, qp_np, minco_np as onp
from allocnet_amd.synth import corridor_problem
ctx = aa.Context(0)
rng = np.random.default_rng(777)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for case in range(n_cases):
    s = int(rng.choice([3, 4]))
    N = int(rng.choice([1, 2, 3, 5, 8, 11, 16] if s == 3 else [1, 2, 3, 5, 8]))
    M = int(rng.choice([6, 8, 12, 16]))
    res = int(rng.choice([3, 8, 20]))
    B = int(rng.choice([1, 7, 64]))
    head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, M)
    tsc = float(rng.choice([0.3, 0.7, 1.5, 4.0]))
    T = T * tsc
    vmax, amax = float(rng.uniform(2.0, 6.0)), float(rng.uniform(3.0, 9.0))
    g = aa.qp_solve(s, head, tail, hp, T, res=res, max_vel=vmax, max_acc=amax, ctx=ctx)
    state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
    p = cbind.qp_ipm_batch(s, state, T, hp, res=res, vmax=vmax, amax=amax, tol=1e-9, want_coeffs=False, nthreads=4)
    D = 2 * s; n = 3 * D * N
    for b in np.nonzero((p["status"] >= 1) & (g["status"] != 1))[0]:
        st9 = np.zeros((9, 2))
        for ax in range(3):
            st9[3 * ax:3 * ax + 3, 0] = state[b, 0, ax]; st9[3 * ax:3 * ax + 3, 1] = state[b, 1, ax]
        Q, A, bb, G1, h1, G2, h2 = onp.qp_assemble(s, st9, np.transpose(hp[b], (1, 2, 0)), np.full(N, M), T[b], res, vmax, amax)
        G = np.zeros((G1.shape[0] + G2.shape[0], n)); r = 0
        for i in range(N):
            for _ in range(res):
                G[r:r + M, i * 3 * D:(i + 1) * 3 * D] = G1[r:r + M]; r += M
        r2 = 0
        for i in range(N):
            for _ in range(res):
                for j in range(3):
                    G[r + r2:r + r2 + 4, i * 3 * D + j * D:i * 3 * D + (j + 1) * D] = G2[r2:r2 + 4]; r2 += 4
        hh = np.r_[h1, h2]; keep = (np.abs(G).sum(axis=1) > 0) | (hh != 0)
        z, lam, nu, fo, it = qp_np.qp_ipm(Q, A, bb, G[keep], hh[keep], tol=1e-10)
        print(f"case {case} s={s} N={N} M={M} res={res} B={B} T x {tsc} b {b}: kernel st {g['status'][b]} it {g['iters'][b]} | port st {p['status'][b]} it {p['iters'][b]} obj {p['obj'][b]:.6e} | dense it {it} obj {fo:.6e} maxviol {(G[keep] @ z - hh[keep]).max():.1e}", flush=True)
