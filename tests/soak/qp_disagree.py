"""Where the interior-point kernel and its C port (oracle/qp_ipm_port.c) disagree on the verdict, who is right: the dense
interior point of oracle/qp_np.py on the reference-assembled problem decides (objective, worst row violation of either
answer).  Durations scaled to 0.7 of the generator's -- close to infeasibly short, optimal costs up to 1e9.
   gpurun -- 'python tests/soak/qp_disagree.py'"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import allocnet_amd as aa
from oracle import cbind, qp_np, minco_np as onp
from allocnet_amd.synth import corridor_problem
ctx = aa.Context(0)
rng = np.random.default_rng(9)
vmax, amax = 4.0, 6.0
for (s, N, M, res) in ((4, 2, 6, 3), (4, 2, 16, 4), (4, 3, 16, 4), (3, 5, 16, 20), (4, 8, 16, 20)):
    B = 64
    head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, M)
    T = T * 0.7
    g = aa.qp_solve(s, head, tail, hp, T, res=res, max_vel=vmax, max_acc=amax, ctx=ctx)
    state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
    p = cbind.qp_ipm_batch(s, state, T, hp, res=res, vmax=vmax, amax=amax, tol=1e-9, want_coeffs=False, nthreads=8)
    p2 = cbind.qp_ipm_batch(s, state, T, hp, res=res, vmax=vmax, amax=amax, tol=1e-9, want_coeffs=False, nthreads=8, max_iter=200)
    gs, ps = g["status"] == 1, p["status"] >= 1
    D = 2 * s; n = 3 * D * N
    print(f"--- s={s} N={N} M={M} res={res}: gpu {gs.sum()} port {ps.sum()} port(200 it) {(p2['status'] >= 1).sum()}")
    for b in np.nonzero(gs != ps)[0]:
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
        viol = float((G[keep] @ z - hh[keep]).max())
        zg = g["coeffs"][b].reshape(-1) if gs[b] else None
        gv = float((G[keep] @ zg - hh[keep]).max()) if zg is not None and zg.size == n else float('nan')
        print(f"  b {b}: gpu st {g['status'][b]} it {g['iters'][b]} obj {g['obj'][b]:.6e} maxviol {gv:.2e} | port st {p['status'][b]} it {p['iters'][b]} / st {p2['status'][b]} it {p2['iters'][b]} obj {p2['obj'][b]:.6e} | dense it {it} obj {fo:.6e} maxviol {viol:.2e}")
