"""Exact (rational) constant tables for the normalised-time Hermite form of a
minimum-control-effort piece of order s (degree 2s-1).

For one piece of duration T, with tau = t/T and u = [P(0),P'(0),..,P^(s-1)(0), P(1),..,P^(s-1)(1)]
(P(tau) = p(T tau), so P^(k) = T^k p^(k)):
    a  = B u          normalised ascending monomial coefficients, P(tau) = sum_k a_k tau^k
    int_0^T (p^(s))^2 dt = T^(1-2s) u' M u,   M = B' Qbar B
Both B and M are constant.  Derived from first principles (no reference source exists for
this: SURVEY.md section 0).
"""
from fractions import Fraction
from math import factorial


def falling(k, j):
    r = 1
    for i in range(j):
        r *= (k - i)
    return r


def inv_exact(A):
    n = len(A)
    M = [[Fraction(x) for x in row] + [Fraction(int(i == j)) for j in range(n)]
         for i, row in enumerate(A)]
    for c in range(n):
        p = next(r for r in range(c, n) if M[r][c] != 0)
        M[c], M[p] = M[p], M[c]
        pv = M[c][c]
        M[c] = [x / pv for x in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                f = M[r][c]
                M[r] = [x - f * y for x, y in zip(M[r], M[c])]
    return [row[n:] for row in M]


def tables(s):
    D = 2 * s
    H = [[Fraction(0)] * D for _ in range(D)]
    for j in range(s):
        H[j][j] = Fraction(factorial(j))               # P^(j)(0) = j! a_j
        for k in range(j, D):
            H[s + j][k] = Fraction(falling(k, j))      # P^(j)(1) = sum k!/(k-j)! a_k
    B = inv_exact(H)
    Qb = [[Fraction(0)] * D for _ in range(D)]
    for j in range(s, D):
        for k in range(s, D):
            Qb[j][k] = Fraction(falling(j, s) * falling(k, s), j + k - 2 * s + 1)
    QB = [[sum(Qb[i][l] * B[l][j] for l in range(D)) for j in range(D)] for i in range(D)]
    M = [[sum(B[l][i] * QB[l][j] for l in range(D)) for j in range(D)] for i in range(D)]
    return B, M, Qb


if __name__ == "__main__":
    for s in (2, 3, 4):
        B, M, Qb = tables(s)
        print("s =", s)
        print(" B hi rows:")
        for r in B[s:]:
            print("  ", [str(x) for x in r])
        print(" M:")
        for r in M:
            print("  ", [str(x) for x in r])
