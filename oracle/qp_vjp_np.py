"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the backward pass through the reference's inequality QP.

network/utils/learning/layers.py:120-147 (and :225-243) installs, on y = (z, lam, nu) of the solved QP, the hook
    grad <- -J^-1 grad,   J = [[Q, G' diag(lam), A'], [G, diag(Gz - h), 0], [A, 0, 0]]            (layers.py:129-139)
J is the transpose of the Jacobian dF/dy of the KKT residual F(y; T) = (Qz + G'lam + A'nu, lam * (Gz - h), Az - b), so
w = -J^-1 [dloss/dz; 0; 0] is the adjoint vector and, F(y*(T), T) = 0 holding identically,  dloss/dT = w' dF/dT at fixed y.
The reference stops at its detached leaf z; this carries the hook through to the durations.  dF/dT is taken by central
differences of the oracle's own assembly (oracle/minco_np.py qp_assemble, pinned entry by entry to the reference's
MinTrajOpt.update by tests/golden/qp_*.npz); the entries are polynomials in T, so the differences are exact to ~1e-10.

PINNED by tests/golden/vjp_*.npz: the same quantity with dF/dT from torch.autograd through the REFERENCE-assembled
matrices (tests/golden/make_golden.py main_vjp; tests/test_qp_vjp_cpu.py)."""
import numpy as np

from . import minco_np as onp
from .qp_np import qp_ipm


def dense_ineq(order, N, m_rows, res, G1c, G2c):
    """qp_assemble keeps the inequality rows compact (the piece's 3D columns / the axis' D columns); the dense
    m_g x n matrices of min_traj_opt.py:535-613, corridor rows first, then the box rows (layers.py:68-70)."""
    D = 2 * order; n = 3 * D * N
    G1 = np.zeros((G1c.shape[0], n)); G2 = np.zeros((G2c.shape[0], n))
    r = 0
    for i in range(N):
        for _ in range(res):
            m = int(m_rows[i])
            G1[r:r + m, i * 3 * D:(i + 1) * 3 * D] = G1c[r:r + m]
            r += m
    r = 0
    for i in range(N):
        for _ in range(res):
            for ax in range(3):
                c0 = i * 3 * D + ax * D
                G2[r:r + 4, c0:c0 + D] = G2c[r:r + 4]
                r += 4
    return np.vstack([G1, G2])


def assemble_dense(order, state, hpolys, m_rows, T, res, vmax, amax):
    Q, A, b, G1c, h1, G2c, h2 = onp.qp_assemble(order, state, hpolys, m_rows, T, res, vmax, amax)
    return Q, A, b, dense_ineq(order, len(T), m_rows, res, G1c, G2c), np.r_[h1, h2]


def kkt_residual(mats, z, lam, nu):
    Q, A, b, G, h = mats
    return np.r_[Q @ z + G.T @ lam + A.T @ nu, lam * (G @ z - h), A @ z - b]


def qp_vjp(order, state, hpolys, m_rows, T, res, vmax, amax, grad_z_of, rel_step=1e-6):
    """grad_z_of(z) -> dloss/dz at the optimum.  Returns dict(z, lam, nu, obj, hook (= -J^-1 [gz;0;0], layers.py:139),
    grad_T (= dloss/dT), cond)."""
    T = np.asarray(T, dtype=float)
    Q, A, b, G, h = assemble_dense(order, state, hpolys, m_rows, T, res, vmax, amax)
    z, lam, nu, obj, it = qp_ipm(Q, A, b, G, h, tol=1e-12, max_iter=300)
    if it >= 300:
        raise RuntimeError("qp_vjp: the QP oracle did not converge (infeasible problem?)")
    n, mg, me = Q.shape[0], G.shape[0], A.shape[0]
    g = G @ z - h
    J = np.block([[Q, G.T * lam[None, :], A.T],
                  [G, np.diag(g), np.zeros((mg, me))],
                  [A, np.zeros((me, mg + me))]])                     # layers.py:129-134
    w = -np.linalg.solve(J, np.r_[grad_z_of(z), np.zeros(mg + me)])  # layers.py:139
    gT = np.zeros(len(T))
    for i in range(len(T)):
        hstep = rel_step * T[i]
        Tp = T.copy(); Tp[i] += hstep
        Tm = T.copy(); Tm[i] -= hstep
        Fp = kkt_residual(assemble_dense(order, state, hpolys, m_rows, Tp, res, vmax, amax), z, lam, nu)
        Fm = kkt_residual(assemble_dense(order, state, hpolys, m_rows, Tm, res, vmax, amax), z, lam, nu)
        gT[i] = w @ (Fp - Fm) / (2.0 * hstep)
    return dict(z=z, lam=lam, nu=nu, obj=obj, hook=w, grad_T=gT, cond=float(np.linalg.cond(J)))
