#!/usr/bin/env python3
"""Second, independent pin of the L-BFGS oracle (oracle/lbfgs_oracle.c).

The reference's gcopter/lbfgs.hpp needs Eigen and cannot be compiled in this image, and no reference test holds a
golden run of it, so `lbfgs_oracle.c` -- the checker of every GPU L-BFGS test -- was a single restatement.  This script
is a SECOND restatement that shares no code with it: pure Python, its own data structures (a deque of (s, y, ys)
records instead of ring-buffer columns, a line-search generator instead of a loop with in/out arguments), written from
the statements and comments of lbfgs.hpp:276-384 (line_search_lewisoverton) and :434-717 (lbfgs_optimize), and a pure-
Python costMVIE written from firi.hpp:60-157.  It writes the traces

    tests/golden/lbfgs_traces.json     (status, k, evaluations, x, f) per problem and iteration budget

which tests/test_lbfgs_cpu.py requires `lbfgs_oracle.c` to reproduce: counters and return codes exactly, x and f to
1e-9.  Two restatements agreeing is weaker than the reference compiled, and DESIGN.md section 6 says so.

Dot products are summed left to right in double precision, the way a scalar loop does; `--fsum` repeats every run with
exactly rounded sums (math.fsum); where the counters of the two differ (only the runs to convergence on the non-smooth
MVIE objective and the ten-variable Rosenbrock function) the trace is marked `order_sensitive` and the test pins its
outcome only (status, final f), not its counters.

    python tests/golden/make_lbfgs_traces.py
"""
import json
import math
import os
import sys
from collections import deque

HERE = os.path.dirname(os.path.abspath(__file__))

# return codes, lbfgs.hpp:135-184
CONVERGENCE, STOP = 0, 1
ERR_INVALID_FUNCVAL, ERR_MINIMUMSTEP, ERR_MAXIMUMSTEP, ERR_MAXIMUMLINESEARCH = -1012, -1011, -1010, -1009
ERR_MAXIMUMITERATION, ERR_WIDTHTOOSMALL, ERR_INVALIDPARAMETERS, ERR_INCREASEGRADIENT = -1008, -1007, -1006, -1005

DEFAULTS = dict(mem_size=8, g_epsilon=1.0e-5, past=3, delta=1.0e-6, max_iterations=0, max_linesearch=64,
                min_step=1.0e-20, max_step=1.0e+20, f_dec_coeff=1.0e-4, s_curv_coeff=0.9, cautious_factor=1.0e-6,
                machine_prec=1.0e-16)          # lbfgs.hpp:25-128

EXACT = False


def dot(a, b):
    if EXACT:
        return math.fsum(u * v for u, v in zip(a, b))
    acc = 0.0
    for u, v in zip(a, b):
        acc += u * v
    return acc


def inf_norm(a):
    return max(abs(v) for v in a)


class Counted:
    def __init__(self, fun):
        self.fun, self.n = fun, 0

    def __call__(self, x):
        self.n += 1
        return self.fun(x)


def lewis_overton(fun, xp, gp, fx, d, step, prm):
    """lbfgs.hpp:276-384.  Returns (code_or_count, x, f, g, step)."""
    if not step > 0.0:
        return ERR_INVALIDPARAMETERS, xp, fx, gp, step
    slope0 = dot(gp, d)
    if slope0 > 0.0:
        return ERR_INCREASEGRADIENT, xp, fx, gp, step
    armijo, wolfe = prm["f_dec_coeff"] * slope0, prm["s_curv_coeff"] * slope0
    lo, hi = 0.0, prm["max_step"]
    bracketed = tried_max = False
    trials = 0
    f0 = fx
    while True:
        x = [a + step * b for a, b in zip(xp, d)]
        f, g = fun(x)
        trials += 1
        if math.isinf(f) or math.isnan(f):
            return ERR_INVALID_FUNCVAL, x, f, g, step
        if f > f0 + step * armijo:
            hi, bracketed = step, True
        elif dot(g, d) < wolfe:
            lo = step
        else:
            return trials, x, f, g, step
        if prm["max_linesearch"] <= trials:
            return ERR_MAXIMUMLINESEARCH, x, f, g, step
        if bracketed and (hi - lo) < prm["machine_prec"] * hi:
            return ERR_WIDTHTOOSMALL, x, f, g, step
        step = 0.5 * (lo + hi) if bracketed else 2.0 * step
        if step < prm["min_step"]:
            return ERR_MINIMUMSTEP, x, f, g, step
        if step > prm["max_step"]:
            if tried_max:
                return ERR_MAXIMUMSTEP, x, f, g, step
            tried_max, step = True, prm["max_step"]


def lbfgs(fun, x0, **over):
    """lbfgs.hpp:434-717 (no step-bound / progress callbacks: the reference's only caller passes nullptr for both,
    firi.hpp:221-227).  Returns dict(status, k, evals, x, f)."""
    prm = dict(DEFAULTS, **over)
    fun = Counted(fun)
    x = list(x0)
    fx, g = fun(x)
    recent = [fx] + [0.0] * (max(1, prm["past"]) - 1)
    d = [-v for v in g]
    k = 0
    if inf_norm(g) / max(1.0, inf_norm(x)) < prm["g_epsilon"]:
        return dict(status=CONVERGENCE, k=k, evals=fun.n, x=x, f=fx)
    step = 1.0 / math.sqrt(dot(d, d))
    k = 1
    pairs = deque()                      # newest first: (s, y, y.s)
    while True:
        xp, gp = x, g
        ls, x, f_trial, g, step = lewis_overton(fun, xp, gp, fx, d, step, prm)
        if ls < 0:
            # the point is reverted, the reported value is the last trial's (lbfgs.hpp:570-577 with fx passed by reference)
            return dict(status=ls, k=k, evals=fun.n, x=xp, f=f_trial)
        fx = f_trial
        if inf_norm(g) / max(1.0, inf_norm(x)) < prm["g_epsilon"]:
            return dict(status=CONVERGENCE, k=k, evals=fun.n, x=x, f=fx)
        if prm["past"] > 0:
            slot = k % prm["past"]
            if prm["past"] <= k and abs(recent[slot] - fx) / max(1.0, abs(fx)) < prm["delta"]:
                return dict(status=STOP, k=k, evals=fun.n, x=x, f=fx)
            recent[slot] = fx
        if prm["max_iterations"] != 0 and prm["max_iterations"] <= k:
            return dict(status=ERR_MAXIMUMITERATION, k=k, evals=fun.n, x=x, f=fx)
        k += 1
        s = [a - b for a, b in zip(x, xp)]
        y = [a - b for a, b in zip(g, gp)]
        ys, yy = dot(y, s), dot(y, y)
        d = [-v for v in g]
        if ys > dot(s, s) * math.sqrt(dot(gp, gp)) * prm["cautious_factor"]:
            pairs.appendleft((s, y, ys))
            while len(pairs) > prm["mem_size"]:
                pairs.pop()
            alphas = []
            for (sj, yj, ysj) in pairs:                       # newest to oldest
                a = dot(sj, d) / ysj
                alphas.append(a)
                d = [u + (-a) * v for u, v in zip(d, yj)]
            scale = ys / yy
            d = [u * scale for u in d]
            for (sj, yj, ysj), a in zip(reversed(pairs), reversed(alphas)):   # oldest to newest
                b = dot(yj, d) / ysj
                d = [u + (a - b) * v for u, v in zip(d, sj)]
        step = 1.0


# ---- objectives ----------------------------------------------------------------------------------------------------
def quadratic(diag, shift):
    def fun(x):
        r = [a - b for a, b in zip(x, shift)]
        return 0.5 * sum(w * v * v for w, v in zip(diag, r)), [w * v for w, v in zip(diag, r)]
    return fun


def rosenbrock(x):
    f, g = 0.0, [0.0] * len(x)
    for i in range(len(x) - 1):
        a, b = x[i + 1] - x[i] * x[i], 1.0 - x[i]
        f += 100.0 * a * a + b * b
        g[i] += -400.0 * a * x[i] - 2.0 * b
        g[i + 1] += 200.0 * a
    return f, g


def nonsmooth(x):
    """|x_0| + 3 |x_1 - 1| + a smooth bowl: the kink is what the Lewis-Overton search is for."""
    f = abs(x[0]) + 3.0 * abs(x[1] - 1.0) + 0.5 * sum((v - 0.3) ** 2 for v in x)
    g = [v - 0.3 for v in x]
    g[0] += (1.0 if x[0] > 0 else -1.0)
    g[1] += (3.0 if x[1] > 1.0 else -3.0)
    return f, g


def smoothed_l1(mu, x):                 # firi.hpp:60-84
    if x < 0.0:
        return None
    if x > mu:
        return x - 0.5 * mu, 1.0
    r = x / mu
    rem = mu - 0.5 * x
    return rem * r * r * r, r * r * (-0.5 * r + 3.0 * rem / mu)


def cost_mvie(A, eps, wt):              # firi.hpp:86-157
    tiny = sys.float_info.epsilon

    def fun(x):
        p, rt, od = x[0:3], x[3:6], x[6:9]
        L = [[rt[0] * rt[0] + tiny, 0.0, 0.0], [od[0], rt[1] * rt[1] + tiny, 0.0], [od[2], od[1], rt[2] * rt[2] + tiny]]
        cost, gp_, gr, gc = 0.0, [0.0] * 3, [0.0] * 3, [0.0] * 3
        for a in A:
            al = [sum(a[i] * L[i][j] for i in range(3)) for j in range(3)]
            nrm = math.sqrt(sum(v * v for v in al))
            pen = smoothed_l1(eps, nrm + sum(u * v for u, v in zip(a, p)) - 1.0)
            if pen is None:
                continue
            c, dc = pen
            unit = [v / nrm for v in al]
            vec = [dc * v for v in a]
            cost += c
            for i in range(3):
                gp_[i] += vec[i]
                gr[i] += unit[i] * vec[i]
            gc[0] += unit[0] * vec[1]
            gc[1] += unit[1] * vec[2]
            gc[2] += unit[0] * vec[2]
        cost = cost * wt - (math.log(L[0][0]) + math.log(L[1][1]) + math.log(L[2][2]))
        g = [wt * v for v in gp_]
        g += [(wt * gr[i] - 1.0 / L[i][i]) * 2.0 * rt[i] for i in range(3)]
        g += [wt * v for v in gc]
        return cost, g
    return fun


class Lcg:
    """Tiny deterministic generator (no numpy: the test side regenerates the same inputs from the same integers)."""

    def __init__(self, seed):
        self.s = seed

    def u(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return ((self.s >> 11) & ((1 << 53) - 1)) / float(1 << 53)


def mvie_rows(seed, M):
    """A bounded polytope a.x <= 1 around the origin: the six box rows plus random unit directions, scaled."""
    r = Lcg(seed)
    rows = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    while len(rows) < M:
        v = [2.0 * r.u() - 1.0 for _ in range(3)]
        n = math.sqrt(sum(t * t for t in v))
        if n > 0.2:
            rows.append([t / n for t in v])
    return [[t / (0.8 + 1.7 * r.u()) for t in row] for row in rows]


def problems():
    out = []
    r = Lcg(7)
    out.append(("quadratic_n6", quadratic([1.0, 4.0, 25.0, 100.0, 400.0, 2500.0], [1.0, -2.0, 0.5, 3.0, -1.0, 0.25]),
                [0.0] * 6, {}))
    out.append(("quadratic_n12_mem3", quadratic([1.0 + 40.0 * r.u() for _ in range(12)], [r.u() for _ in range(12)]),
                [2.0 * r.u() - 1.0 for _ in range(12)], dict(mem_size=3)))
    out.append(("rosenbrock_n2", rosenbrock, [-1.2, 1.0], dict(g_epsilon=1e-8, delta=1e-12)))
    out.append(("rosenbrock_n10", rosenbrock, [-1.2, 1.0] * 5, dict(g_epsilon=1e-8, delta=1e-12)))
    out.append(("nonsmooth_n5", nonsmooth, [1.5, -0.5, 2.0, -1.0, 0.7], dict(g_epsilon=0.0, delta=1e-10)))
    call_site = dict(mem_size=18, g_epsilon=0.0, min_step=1.0e-32, past=3, delta=1.0e-7)       # firi.hpp:212-217
    for seed, M in ((11, 8), (12, 14), (13, 22), (14, 30)):
        A = mvie_rows(seed, M)
        x0 = [0.01, -0.02, 0.015] + [math.sqrt(0.3)] * 3 + [0.0] * 3
        out.append((f"mvie_seed{seed}_M{M}", cost_mvie(A, 1.0e-2, 1.0e3), x0, dict(call_site), dict(A=A, eps=1e-2, wt=1e3)))
    out.append(("already_stationary", quadratic([1.0, 2.0], [0.5, -0.5]), [0.5, -0.5], {}))
    out.append(("uphill_direction", lambda x: (-(x[0] ** 2), [2.0 * x[0]]), [1.0], {}))        # g has the wrong sign
    return out


def main():
    global EXACT
    traces = []
    for entry in problems():
        name, fun, x0, over = entry[:4]
        data = entry[4] if len(entry) > 4 else None
        for budget in (1, 3, 10, 0):
            EXACT = False
            run = lbfgs(fun, x0, **dict(over, max_iterations=budget))
            rec = dict(problem=name, max_iterations=budget, params=over, x0=x0, **run)
            if data:
                rec["mvie"] = data
            traces.append(rec)
            # the same run with exactly rounded dot products: where its counters differ the trace hinges on the
            # summation order (long runs on the non-smooth objectives) and only its outcome is pinned
            EXACT = True
            alt = lbfgs(fun, x0, **dict(over, max_iterations=budget))
            EXACT = False
            rec["order_sensitive"] = (alt["status"], alt["k"], alt["evals"]) != (run["status"], run["k"], run["evals"])
    with open(os.path.join(HERE, "lbfgs_traces.json"), "w") as fh:
        json.dump(traces, fh, indent=0)
    print(f"{len(traces)} traces ->", os.path.join(HERE, "lbfgs_traces.json"))
    for t in traces:
        print(f"{t['problem']:22s} budget {t['max_iterations']:3d}: status {t['status']:6d} k {t['k']:4d} evals {t['evals']:4d} "
              f"f {t['f']:.12g}" + ("   (order sensitive)" if t["order_sensitive"] else ""))


if __name__ == "__main__":
    main()
