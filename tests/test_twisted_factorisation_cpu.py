"""The twisted block Cholesky of csrc/qp_ipm.h (twisted_factor / twisted_solve), restated in numpy: for an SPD block-tridiagonal
K with diagonal blocks D_k and sub-diagonal blocks A_{k+1,k} the chain of knots is eliminated from both ends towards a middle
knot p,

    k < p :  L_k L_k' = D_k - L_{k,k-1} L_{k,k-1}',      L_{k+1,k} = A_{k+1,k} L_k^-T          (wave 0, top down)
    k > p :  L_k L_k' = D_k - M_k M_k',                  M_{k-1}   = A_{k,k-1}' L_k^-T         (wave 1, bottom up)
    k = p :  L_p L_p' = D_p - L_{p,p-1} L_{p,p-1}' - M_p M_p'                                   (both updates)

so that K = T T' with T lower bidiagonal above the middle knot and upper bidiagonal below it -- no fill-in, half the sequential
depth -- and K x = r is  T z = r  from both ends to the middle,  T' x = z  from the middle outwards.  This test pins the algebra
the kernel relies on (every block size and piece count it is instantiated for, p = N / 2 and the one-chain order p = N)."""
import numpy as np
import pytest


def _random_block_tridiagonal(rng, n_knots, bk):
    n = n_knots * bk
    B = np.zeros((n, n))
    for k in range(n_knots):
        B[k * bk:(k + 1) * bk, k * bk:(k + 1) * bk] = rng.normal(size=(bk, bk))
        if k + 1 < n_knots:
            B[(k + 1) * bk:(k + 2) * bk, k * bk:(k + 1) * bk] = rng.normal(size=(bk, bk))
    K = B @ B.T + 0.5 * np.eye(n)                    # SPD, block tridiagonal
    D = [K[k * bk:(k + 1) * bk, k * bk:(k + 1) * bk].copy() for k in range(n_knots)]
    A = [K[(k + 1) * bk:(k + 2) * bk, k * bk:(k + 1) * bk].copy() for k in range(n_knots - 1)]     # block (k+1, k)
    return K, D, A


def _twisted_factor(D, A, p):
    N = len(D) - 1
    L = [None] * (N + 1)
    Off = [None] * N                                  # Off[k]: L_{k+1,k} for k < p, M_k for k >= p
    for k in range(0, p):                             # top chain
        S = D[k] - (Off[k - 1] @ Off[k - 1].T if k > 0 else 0.0)
        L[k] = np.linalg.cholesky(S)
        Off[k] = np.linalg.solve(L[k], A[k].T).T      # A_{k+1,k} L_k^-T
    for k in range(N, p, -1):                         # bottom chain
        S = D[k] - (Off[k] @ Off[k].T if k < N else 0.0)
        L[k] = np.linalg.cholesky(S)
        Off[k - 1] = np.linalg.solve(L[k], A[k - 1]).T    # A_{k,k-1}' L_k^-T
    S = D[p].copy()
    if p > 0:
        S -= Off[p - 1] @ Off[p - 1].T
    if p < N:
        S -= Off[p] @ Off[p].T
    L[p] = np.linalg.cholesky(S)
    return L, Off


def _twisted_solve(L, Off, p, r):
    N = len(L) - 1
    bk = L[0].shape[0]
    x = [r[k * bk:(k + 1) * bk].copy() for k in range(N + 1)]
    for k in range(0, p):                             # T z = r, top down
        if k > 0:
            x[k] -= Off[k - 1] @ x[k - 1]
        x[k] = np.linalg.solve(L[k], x[k])
    for k in range(N, p, -1):                         # ... and bottom up
        if k < N:
            x[k] -= Off[k] @ x[k + 1]
        x[k] = np.linalg.solve(L[k], x[k])
    if p > 0:
        x[p] -= Off[p - 1] @ x[p - 1]
    if p < N:
        x[p] -= Off[p] @ x[p + 1]
    x[p] = np.linalg.solve(L[p].T, np.linalg.solve(L[p], x[p]))   # middle: forwards, then the first of T' x = z
    for k in range(p - 1, -1, -1):                    # outwards, upwards
        x[k] = np.linalg.solve(L[k].T, x[k] - Off[k].T @ x[k + 1])
    for k in range(p + 1, N + 1):                     # outwards, downwards
        x[k] = np.linalg.solve(L[k].T, x[k] - Off[k - 1].T @ x[k - 1])
    return np.concatenate(x)


@pytest.mark.parametrize("bk", [9, 12])
@pytest.mark.parametrize("n_pieces", [1, 2, 5, 8, 16])
def test_twisted_block_cholesky_is_an_exact_factorisation(bk, n_pieces):
    rng = np.random.default_rng(100 * bk + n_pieces)
    K, D, A = _random_block_tridiagonal(rng, n_pieces + 1, bk)
    r = rng.normal(size=K.shape[0])
    ref = np.linalg.solve(K, r)
    for p in (n_pieces // 2, n_pieces):              # the kernel's middle knot, and the one-chain (classic) order
        L, Off = _twisted_factor(D, A, p)
        # K = T T'
        n = K.shape[0]
        T = np.zeros((n, n))
        for k in range(n_pieces + 1):
            T[k * bk:(k + 1) * bk, k * bk:(k + 1) * bk] = L[k]
        for k in range(n_pieces):
            if k < p:
                T[(k + 1) * bk:(k + 2) * bk, k * bk:(k + 1) * bk] = Off[k]         # below the diagonal
            else:
                T[k * bk:(k + 1) * bk, (k + 1) * bk:(k + 2) * bk] = Off[k]         # above it
        assert np.abs(T @ T.T - K).max() <= 1e-11 * np.abs(K).max()
        x = _twisted_solve(L, Off, p, r)
        assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
