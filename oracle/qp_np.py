"""TEST INFRASTRUCTURE ONLY -- dense primal-dual interior-point solver for the reference's QP
    min 1/2 z'Qz  s.t.  A z = b,  G z <= h
used as the high-accuracy oracle for the GPU ADMM solve.  It is a textbook Mehrotra predictor-
corrector method (Nocedal & Wright, Numerical Optimization, Alg. 16.4), deliberately a different
algorithm from both OSQP and the GPU kernel.  OSQP itself (the reference's solver, third-party,
release-0.6.3) is not available in this image: parity with OSQP iterates is UNPINNED; what is checked
is that the GPU solution satisfies the same problem's KKT conditions to OSQP's own tolerances and
agrees with this oracle's optimum."""
import numpy as np
from scipy.linalg import lu_factor, lu_solve


def qp_ipm(Q, A, b, G, h, tol=1e-10, max_iter=200):
    """Returns z, lam, nu, objective, iterations (iterations == max_iter means no convergence: the
    problem is probably infeasible).  Degenerate problems can reach their rounding floor above `tol` (the 1e-10
    regularisation of H caps the stationarity residual) and then drift: the iterate of smallest scaled residual is
    kept, and returned as converged when it is within 1e-7 once the iteration stalls for 15 steps."""
    n = Q.shape[0]; me = A.shape[0]; mg = G.shape[0]
    z = np.linalg.lstsq(A, b, rcond=None)[0]
    s = np.maximum(h - G @ z, 1.0); lam = np.ones(mg); nu = np.zeros(me)
    reg = 1e-10 * max(1.0, np.abs(Q).max())
    hs = max(1.0, np.abs(h).max())
    it = 0
    best = (np.inf, 0, None)
    for it in range(max_iter):
        rd = Q @ z + A.T @ nu + G.T @ lam
        rp = A @ z - b
        rg = G @ z + s - h
        mu = s @ lam / mg
        merit = max(np.abs(rd).max() / max(1.0, np.abs(Q @ z).max(), np.abs(G.T @ lam).max()), np.abs(rp).max() / hs,
                    np.abs(rg).max() / hs, mu / max(1.0, abs(z @ Q @ z)))
        if merit <= tol:
            break
        if merit < best[0]:
            best = (merit, it, (z.copy(), lam.copy(), nu.copy()))
        elif it - best[1] >= 15:
            break
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            W = lam / s
            H = Q + G.T @ (W[:, None] * G) + reg * np.eye(n)
        if not np.isfinite(H).all():      # an infeasible problem: slacks underflow while the multipliers grow
            it = max_iter
            break
        K = np.block([[H, A.T], [A, -1e-13 * np.eye(me)]])
        lu = lu_factor(K)

        def step(rc):
            rhs1 = -rd - G.T @ ((lam * rg - rc) / s)
            sol = lu_solve(lu, np.r_[rhs1, -rp])
            dz = sol[:n]; dnu = sol[n:]
            ds = -rg - G @ dz
            dlam = (-rc - lam * ds) / s
            return dz, dnu, ds, dlam

        def maxstep(v, dv, frac):
            neg = dv < 0
            return min(1.0, frac * (-v[neg] / dv[neg]).min()) if neg.any() else 1.0
        dz, dnu, ds, dlam = step(s * lam)
        al = min(maxstep(s, ds, 1.0), maxstep(lam, dlam, 1.0))
        mu_aff = (s + al * ds) @ (lam + al * dlam) / mg
        sig = (mu_aff / mu) ** 3
        dz, dnu, ds, dlam = step(s * lam + ds * dlam - sig * mu)
        al = min(maxstep(s, ds, 0.995), maxstep(lam, dlam, 0.995))
        z = z + al * dz; s = s + al * ds; lam = lam + al * dlam; nu = nu + al * dnu
        if not np.isfinite(z).all():
            it = max_iter
            break
    else:
        it = max_iter
    if merit > tol:                   # stalled or out of iterations: the best iterate decides
        if best[0] <= 1e-7:
            (z, lam, nu), it = best[2], best[1]
        else:
            it = max_iter
    return z, lam, nu, 0.5 * z @ Q @ z, it


def kkt_violation(Q, A, b, G, h, z):
    """Primal feasibility of z (max violation) -- duals are not needed for this check."""
    return max(np.abs(A @ z - b).max(), np.maximum(G @ z - h, 0.0).max())
