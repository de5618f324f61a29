"""TEST INFRASTRUCTURE -- CPU restatement of firi::firi and firi::maxVolInsEllipsoid
(src/planner/include/gcopter/firi.hpp:159-416).  Only tests/ may import this module.

PARITY UNPINNED: the reference holds no test or golden vector for FIRI and its C++ cannot be built here
(Eigen absent).  The restatement follows the header statement by statement; two third-party pieces
are replaced by mathematically equivalent ones and checked independently in tests/test_oracle_cpu.py:
  * sdlp::linprog<4> (Seidel's LP, gcopter/sdlp.hpp, firi.hpp:181) -> exact enumeration of the 4-subsets
    of constraints (the optimum of the Chebyshev-centre LP is a vertex), checked against scipy's HiGHS;
  * Eigen::JacobiSVD of the 3x3 factor L (firi.hpp:247) -> numpy.linalg.svd.
The L-BFGS stage is the C restatement of lbfgs.hpp + costMVIE (oracle/lbfgs_oracle.c via cbind).
"""
import itertools

import numpy as np


def chol3d(A):
    """firi.hpp:45-58"""
    L = np.zeros((3, 3))
    L[0, 0] = np.sqrt(A[0, 0])
    L[1, 0] = 0.5 * (A[0, 1] + A[1, 0]) / L[0, 0]
    L[1, 1] = np.sqrt(A[1, 1] - L[1, 0] * L[1, 0])
    L[2, 0] = 0.5 * (A[0, 2] + A[2, 0]) / L[0, 0]
    L[2, 1] = (0.5 * (A[1, 2] + A[2, 1]) - L[2, 0] * L[1, 0]) / L[1, 1]
    L[2, 2] = np.sqrt(A[2, 2] - L[2, 0] * L[2, 0] - L[2, 1] * L[2, 1])
    return L


def chebyshev_centre(Alp3, blp):
    """max t  s.t.  Alp3 x + t <= blp (rows of Alp3 unit length): deepest interior point and its depth
    (the LP of firi.hpp:170-186).  Exact vertex enumeration; returns (depth, x) or (-inf, None)."""
    M = Alp3.shape[0]
    A4 = np.c_[Alp3, np.ones(M)]
    best, xbest = -np.inf, None
    combos = np.array(list(itertools.combinations(range(M), 4)))
    if combos.size == 0:
        return best, xbest
    for lo in range(0, len(combos), 20000):
        cb = combos[lo:lo + 20000]
        K = A4[cb]                                    # (k,4,4)
        rhs = blp[cb]
        det = np.linalg.det(K)
        ok = np.abs(det) > 1e-12
        if not ok.any():
            continue
        sol = np.linalg.solve(K[ok], rhs[ok][..., None])[..., 0]      # (k,4)
        slack = blp[None, :] - sol @ A4.T
        feas = (slack >= -1e-9 * np.maximum(1.0, np.abs(blp))[None, :]).all(axis=1)
        if feas.any():
            t = np.where(feas, sol[:, 3], -np.inf)
            j = int(np.argmax(t))
            if t[j] > best:
                best, xbest = float(t[j]), sol[j, :3].copy()
    return best, xbest


def mvie_setup(hPoly, R, p, r):
    """firi.hpp:170-222: interior point, normalised rows A (a.x <= 1 about the interior point), x0."""
    hn = np.linalg.norm(hPoly[:, :3], axis=1)
    Alp3 = hPoly[:, :3] / hn[:, None]
    blp = -hPoly[:, 3] / hn
    depth, interior = chebyshev_centre(Alp3, blp)
    if not (depth > 0.0) or np.isinf(depth):
        return None
    A = Alp3 / (blp - Alp3 @ interior)[:, None]
    Q = R @ np.diag(r * r) @ R.T
    L = chol3d(Q)
    x0 = np.r_[p - interior, np.sqrt(L[0, 0]), np.sqrt(L[1, 1]), np.sqrt(L[2, 2]), L[1, 0], L[2, 1], L[2, 0]]
    return depth, interior, A, x0


def mvie_finish(x, interior):
    """firi.hpp:235-265: ellipsoid centre, rotation and radii from the optimiser's variables."""
    p = x[:3] + interior
    L = np.array([[x[3] * x[3], 0.0, 0.0], [x[6], x[4] * x[4], 0.0], [x[8], x[7], x[5] * x[5]]])
    U, S, _ = np.linalg.svd(L)
    if np.linalg.det(U) < 0.0:
        R = U[:, [1, 0, 2]].copy()
        r = S[[1, 0, 2]].copy()
    else:
        R, r = U, S
    return R, p, r, L


def max_vol_ins_ellipsoid(hPoly, R, p, r):
    from oracle import cbind
    st = mvie_setup(hPoly, R, p, r)
    if st is None:
        return False, R, p, r
    depth, interior, A, x0 = st
    prm = cbind.lbfgs_default_param(mem_size=18, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7)   # firi.hpp:212-217
    ret, x, f, it, ev = cbind.lbfgs_mvie(A, 1e-2, 1e3, x0, prm)
    R, p, r, _ = mvie_finish(x, interior)
    return ret >= 0, R, p, r


def firi_planes(bd, pc, a, b, R, p, r, epsilon=1e-6):
    """One pass of the loop body of firi::firi (firi.hpp:297-405): the polytope for the current ellipsoid."""
    M, N = bd.shape[0], pc.shape[0]
    forward = np.diag(1.0 / r) @ R.T
    backward = R @ np.diag(r)
    forwardB = bd[:, :3] @ backward
    forwardD = bd[:, 3] + bd[:, :3] @ p
    forwardPC = (forward @ (pc - p).T).T                    # (N,3)
    fa, fb = forward @ (a - p), forward @ (b - p)
    distDs = np.abs(forwardD) / np.linalg.norm(forwardB, axis=1)
    tang = np.zeros((N, 4))
    distRs = np.zeros(N)
    for i in range(N):
        q = forwardPC[i]
        distRs[i] = np.linalg.norm(q)
        tang[i, 3] = -distRs[i]
        tang[i, :3] = q / distRs[i]
        if tang[i, :3] @ fa + tang[i, 3] > epsilon:
            d = q - fa
            tang[i, :3] = fa - (d @ fa / (d @ d)) * d
            distRs[i] = np.linalg.norm(tang[i, :3])
            tang[i, 3] = -distRs[i]
            tang[i, :3] /= distRs[i]
        if tang[i, :3] @ fb + tang[i, 3] > epsilon:
            d = q - fb
            tang[i, :3] = fb - (d @ fb / (d @ d)) * d
            distRs[i] = np.linalg.norm(tang[i, :3])
            tang[i, 3] = -distRs[i]
            tang[i, :3] /= distRs[i]
        if tang[i, :3] @ fa + tang[i, 3] > epsilon:
            n = np.cross(fa - q, fb - q)
            tang[i, :3] = n / np.linalg.norm(n)
            tang[i, 3] = -tang[i, :3] @ fa
            tang[i] *= -1.0 if tang[i, 3] > 0.0 else 1.0
    bdF = np.ones(M, dtype=bool)
    pcF = np.ones(N, dtype=bool)
    H = []
    bdMin = int(np.argmin(distDs))
    minD = distDs[bdMin]
    pcMin, minR = 0, np.inf
    if N:
        pcMin = int(np.argmin(distRs))
        minR = distRs[pcMin]
    completed = False
    i = 0
    while not completed and i < M + N:
        if minD < minR:
            row = np.r_[forwardB[bdMin], forwardD[bdMin]]
            bdF[bdMin] = False
        else:
            row = tang[pcMin].copy()
            pcF[pcMin] = False
        H.append(row)
        completed = True
        minD = np.inf
        for j in range(M):
            if bdF[j]:
                completed = False
                if minD > distDs[j]:
                    bdMin, minD = j, distDs[j]
        minR = np.inf
        if N:
            outside = pcF & (forwardPC @ row[:3] + row[3] > -epsilon)
            pcF &= ~outside
            if pcF.any():
                completed = False
                idx = np.flatnonzero(pcF)
                k = idx[np.argmin(distRs[idx])]           # first minimum, like the scan with a strict >
                pcMin, minR = int(k), distRs[k]
        i += 1
    H = np.array(H)
    hPoly = np.zeros_like(H)
    hPoly[:, :3] = H[:, :3] @ forward
    hPoly[:, 3] = H[:, 3] - hPoly[:, :3] @ p
    return hPoly


def firi(bd, pc, a, b, iterations=4, epsilon=1e-6, trace=None):
    """firi::firi (firi.hpp:268-416).  bd (M,4) rows h.[x;1] <= 0, pc (N,3) obstacle points, a, b the
    segment the polytope must contain.  Returns (ok, hPoly (nH,4))."""
    bd = np.asarray(bd, dtype=float); pc = np.asarray(pc, dtype=float).reshape(-1, 3)
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    if (bd @ np.r_[a, 1.0]).max() > 0.0 or (bd @ np.r_[b, 1.0]).max() > 0.0:
        return False, None
    R, p, r = np.eye(3), 0.5 * (a + b), np.ones(3)
    hPoly = None
    for loop in range(iterations):
        hPoly = firi_planes(bd, pc, a, b, R, p, r, epsilon)
        if trace is not None:
            trace.append(dict(R=R.copy(), p=p.copy(), r=r.copy(), hPoly=hPoly.copy()))
        if loop == iterations - 1:
            break
        _, R, p, r = max_vol_ins_ellipsoid(hPoly, R, p, r)
    return True, hPoly


def polytope_depth(hPoly, normalise):
    """geo_utils.hpp:43-85: max t s.t. n.x + t <= -h3 (the LP findInterior and overlap hand to sdlp::linprog<4>),
    here by scipy's HiGHS.  Returns (depth, x); depth = -inf when infeasible."""
    from scipy.optimize import linprog
    h = np.asarray(hPoly, dtype=np.float64)
    nrm = np.linalg.norm(h[:, :3], axis=1) if normalise else np.ones(len(h))
    A = np.c_[h[:, :3] / nrm[:, None], np.ones(len(h))]
    r = linprog(c=[0, 0, 0, -1.0], A_ub=A, b_ub=-h[:, 3] / nrm, bounds=[(None, None)] * 4, method="highs")
    if r.status != 0:
        return -np.inf, None
    return float(r.x[3]), r.x[:3]


def find_interior(hPoly):
    """geo_utils.hpp:43-62"""
    d, x = polytope_depth(hPoly, True)
    return d > 0.0 and np.isfinite(d), x


def overlap(hPoly0, hPoly1, eps=1.0e-6):
    """geo_utils.hpp:64-85"""
    d, _ = polytope_depth(np.vstack([hPoly0, hPoly1]), False)
    return d > eps and np.isfinite(d)


def short_cut(hpolys, eps=0.1):
    """sfc_gen.hpp:188-226, the loop as written there (deque of indices, i reset inside the inner loop)."""
    htemp = list(hpolys)
    if len(htemp) == 1:
        htemp.insert(0, htemp[0])
    M = len(htemp)
    idices = [M - 1]
    i = M - 1
    while i >= 0:
        for j in range(i):
            ov = overlap(htemp[i], htemp[j], eps) if j < i - 1 else True
            if ov:
                idices.insert(0, j)
                i = j + 1
                break
        i -= 1
    return idices
