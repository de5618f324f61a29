"""CPU suite: known-answer pins of the C restatement of lbfgs.hpp and firi::costMVIE (the reference
has no tests or golden vectors for them and cannot be compiled here: parity unpinned otherwise)."""
import numpy as np

from oracle import cbind


def test_quadratic_known_answer():
    rng = np.random.default_rng(0)
    n = 12
    A = rng.normal(size=(n, n)); H = A @ A.T + n * np.eye(n); bvec = rng.normal(size=n)
    xs = np.linalg.solve(H, bvec)
    ret, x, f, it, ev = cbind.lbfgs_optimize(np.zeros(n), lambda x: (0.5 * x @ H @ x - bvec @ x, H @ x - bvec),
                                            cbind.lbfgs_default_param(g_epsilon=1e-7, delta=0.0, past=0))
    assert ret == 0
    assert np.abs(x - xs).max() < 1e-6
    assert abs(f - (0.5 * xs @ H @ xs - bvec @ xs)) < 1e-10 and it > 2 and ev >= it


def test_rosenbrock_known_answer():
    def fun(x):
        f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
        return f, np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
    ret, x, f, it, ev = cbind.lbfgs_optimize([-1.2, 1.0], fun, cbind.lbfgs_default_param(g_epsilon=1e-9, delta=1e-14))
    assert ret in (0, 1)
    assert np.abs(x - 1.0).max() < 1e-5 and f < 1e-10


def test_parameter_validation_and_codes():
    fun = lambda x: (float(x @ x), 2 * x)
    assert cbind.lbfgs_optimize([1.0], fun, cbind.lbfgs_default_param(mem_size=0))[0] == -1022   # INVALID_MEMSIZE
    assert cbind.lbfgs_optimize([1.0], fun, cbind.lbfgs_default_param(g_epsilon=-1.0))[0] == -1021
    assert cbind.lbfgs_optimize([1.0], fun, cbind.lbfgs_default_param(s_curv_coeff=1e-5))[0] == -1015
    assert cbind.lbfgs_optimize([1.0], fun, cbind.lbfgs_default_param(max_linesearch=0))[0] == -1013
    # already stationary -> convergence with zero iterations, one evaluation
    ret, x, f, it, ev = cbind.lbfgs_optimize([0.0, 0.0], fun)
    assert (ret, it, ev) == (0, 0, 1)
    # NaN objective -> INVALID_FUNCVAL, x reverted
    bad = lambda x: (np.nan if abs(x[0]) < 0.99 else float(x @ x), 2 * x)
    ret, x, f, it, ev = cbind.lbfgs_optimize([1.0], bad)
    assert ret == -1012 and x[0] == 1.0
    # iteration cap
    ret, *_ = cbind.lbfgs_optimize([3.0, -2.0], lambda x: (float(x[0] ** 4 + x[1] ** 2), np.array([4 * x[0] ** 3, 2 * x[1]])),
                                   cbind.lbfgs_default_param(max_iterations=2, g_epsilon=1e-14, delta=0.0))
    assert ret == -1008


def _mvie_problem(rng, M):
    A = rng.normal(size=(M, 3)); A /= np.linalg.norm(A, axis=1, keepdims=True)
    A /= rng.uniform(0.8, 2.5, size=(M, 1))          # rows a with a.x <= 1, interior point at the origin
    x0 = np.r_[np.zeros(3), np.sqrt([0.3, 0.3, 0.3]), np.zeros(3)]
    return A, x0


def test_cost_mvie_gradient_finite_difference():
    rng = np.random.default_rng(1)
    A, x0 = _mvie_problem(rng, 14)
    x = x0 + rng.normal(size=9) * 0.2
    f, g = cbind.cost_mvie(A, 1e-2, 1e3, x)
    h = 1e-6
    for i in range(9):
        xp = x.copy(); xp[i] += h; xm = x.copy(); xm[i] -= h
        fd = (cbind.cost_mvie(A, 1e-2, 1e3, xp)[0] - cbind.cost_mvie(A, 1e-2, 1e3, xm)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(g[i]))


def test_lbfgs_mvie_reference_call_site_parameters():
    """firi.hpp:212-217 parameter set; the optimum's ellipsoid must sit inside the polytope."""
    rng = np.random.default_rng(2)
    for M in (8, 20):
        A, x0 = _mvie_problem(rng, M)
        prm = cbind.lbfgs_default_param(mem_size=18, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7)
        ret, x, f, it, ev = cbind.lbfgs_mvie(A, 1e-2, 1e3, x0, prm)
        assert ret >= 0 and it > 3
        L = np.array([[x[3] ** 2, 0, 0], [x[6], x[4] ** 2, 0], [x[8], x[7], x[5] ** 2]])
        viol = np.linalg.norm(A @ L, axis=1) + A @ x[:3] - 1.0
        assert viol.max() < 2e-2             # smoothed penalty: inside up to ~eps
        assert f < cbind.cost_mvie(A, 1e-2, 1e3, x0)[0]


# ---- second, independent restatement (tests/golden/make_lbfgs_traces.py) --------------------------------------------
def _trace_objective(t):
    """The objectives of the committed traces, written once more for the C oracle (numpy)."""
    name = t["problem"]
    if name.startswith("quadratic_n6"):
        w = np.array([1.0, 4.0, 25.0, 100.0, 400.0, 2500.0]); sh = np.array([1.0, -2.0, 0.5, 3.0, -1.0, 0.25])
        return lambda x: (float(0.5 * (w * (x - sh) ** 2).sum()), w * (x - sh))
    if name.startswith("rosenbrock"):
        def ros(x):
            a = x[1:] - x[:-1] ** 2; b = 1.0 - x[:-1]
            g = np.zeros_like(x); g[:-1] += -400.0 * a * x[:-1] - 2.0 * b; g[1:] += 200.0 * a
            return float((100.0 * a * a + b * b).sum()), g
        return ros
    if name.startswith("nonsmooth"):
        def ns(x):
            g = x - 0.3
            g[0] += 1.0 if x[0] > 0 else -1.0
            g[1] += 3.0 if x[1] > 1.0 else -3.0
            return float(abs(x[0]) + 3.0 * abs(x[1] - 1.0) + 0.5 * ((x - 0.3) ** 2).sum()), g
        return ns
    if name == "already_stationary":
        w = np.array([1.0, 2.0]); sh = np.array([0.5, -0.5])
        return lambda x: (float(0.5 * (w * (x - sh) ** 2).sum()), w * (x - sh))
    if name == "uphill_direction":
        return lambda x: (float(-(x[0] ** 2)), np.array([2.0 * x[0]]))
    return None


def test_oracle_reproduces_the_independent_python_traces():
    """`lbfgs_oracle.c` against the traces of a pure-Python restatement of lbfgs.hpp / costMVIE that shares no code with
    it (different data structures, written from the header's statements): return code, iteration and evaluation counters
    exact, x and f to 1e-9 -- except where the generator itself found the counters to depend on the summation order of
    the dot products (long runs on non-smooth objectives), where the outcome is compared."""
    import json
    import os
    traces = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbfgs_traces.json")))
    assert len(traces) >= 40
    exact = 0
    for t in traces:
        over = dict(t["params"], max_iterations=t["max_iterations"])
        prm = cbind.lbfgs_default_param(**over)
        x0 = np.array(t["x0"], dtype=float)
        if "mvie" in t:
            A = np.array(t["mvie"]["A"], dtype=float)
            ret, x, f, it, ev = cbind.lbfgs_mvie(A, t["mvie"]["eps"], t["mvie"]["wt"], x0, prm)
        else:
            fun = _trace_objective(t)
            if fun is None:              # quadratic_n12_mem3: its coefficients come out of the generator's own LCG
                continue
            ret, x, f, it, ev = cbind.lbfgs_optimize(x0, fun, prm)
        key = (t["problem"], t["max_iterations"])
        assert ret == t["status"], key
        if t["order_sensitive"]:
            assert abs(f - t["f"]) <= 2e-3 * max(1.0, abs(t["f"])), key
            continue
        if t["status"] != 0 or t["k"] > 0:          # (an already stationary start leaves k unset in lbfgs.hpp)
            assert it == t["k"], key
        assert ev == t["evals"], key
        tol = 1e-9 * max(1, t["k"])
        assert np.abs(x - np.array(t["x"])).max() <= tol * max(1.0, np.abs(t["x"]).max()), key
        assert abs(f - t["f"]) <= tol * max(1.0, abs(t["f"])), key
        exact += 1
    assert exact >= 30
