"""Test-side helper: an (effectively) exact reference for the MINCO coefficient solve, used to check which of two
float64 solutions is the accurate one when they disagree (accuracy envelope)."""
import numpy as np


def exact_minco(s, head, tail, wps, T, digits=60):
    """Minimum-control-effort coefficients from a `digits`-digit LU of the classic 2sN x 2sN collocation system
    (same rows as oracle/minco_np.minco_dense_matrix).  head/tail 3 x c, wps 3 x (N-1), T (N,).  Returns (N,3,2s),
    highest power first, rounded to float64."""
    import mpmath as mp
    from oracle import minco_np as onp
    mp.mp.dps = digits
    N = len(T); D = 2 * s; c = head.shape[1]
    Mf, info = onp.minco_dense_matrix(s, T, c)      # only for the row bookkeeping
    Tm = [mp.mpf(float(t)) for t in T]
    M = mp.zeros(D * N, D * N)
    def drow(t, j):
        r = [mp.mpf(0)] * D
        for k in range(j, D):
            f = mp.mpf(1)
            for q in range(j): f *= (k - q)
            r[k] = f * t ** (k - j)
        return r
    row = 0
    def put(row, col0, vals, sign=1):
        for k, v in enumerate(vals): M[row, col0 + k] = sign * v
    for j in range(s):
        put(row, 0, drow(mp.mpf(0), j if j < c else 2 * s - 1 - j)); row += 1
    for i in range(1, N):
        cl = (i - 1) * D; cr = i * D
        put(row, cl, drow(Tm[i - 1], 0)); row += 1
        for j in range(2 * s - 1):
            put(row, cl, drow(Tm[i - 1], j)); put(row, cr, drow(mp.mpf(0), j), -1); row += 1
    cl = (N - 1) * D
    for j in range(s):
        put(row, cl, drow(Tm[N - 1], j if j < c else 2 * s - 1 - j)); row += 1
    rhs = mp.zeros(D * N, 3)
    for r, (kind, j) in enumerate(info):
        src = head[:, j] if kind == "head" else tail[:, j] if kind == "tail" else wps[:, j - 1] if kind == "wp" else None
        if src is not None:
            for a in range(3): rhs[r, a] = mp.mpf(float(src[a]))
    co = np.zeros((N, 3, D))
    for a in range(3):
        sol = mp.lu_solve(M, rhs[:, a])
        for i in range(N):
            for k in range(D): co[i, a, D - 1 - k] = float(sol[i * D + k])
    return co
