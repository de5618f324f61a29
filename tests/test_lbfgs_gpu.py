"""GPU parity of the batched L-BFGS against the C restatement of lbfgs.hpp (oracle/lbfgs_oracle.c):
(1) the reference's own objective and call-site parameters (firi::costMVIE, firi.hpp:207-227),
(2) the MINCO trajectory cost, objective evaluated by the numpy oracle through a C callback."""
import numpy as np
import pytest

from oracle import cbind
from oracle import minco_np as onp
from tests.util import random_problem
from tests.test_grad_gpu import make_corridors

pytestmark = pytest.mark.gpu


def _mvie_batch(rng, B, M):
    A = rng.normal(size=(B, M, 3)); A /= np.linalg.norm(A, axis=2, keepdims=True)
    A[:, :6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
    A /= rng.uniform(0.8, 2.5, size=(B, M, 1))      # bounded polytopes a.x <= 1 around the origin
    k = rng.integers(max(6, M // 2), M + 1, size=B)
    for b in range(B):
        A[b, k[b]:] = 0.0                       # zero rows: inactive padding (ragged polytopes)
    x0 = np.tile(np.r_[np.zeros(3), np.sqrt([0.3, 0.3, 0.3]), np.zeros(3)], (B, 1))
    x0[:, :3] += rng.normal(size=(B, 3)) * 0.02
    return A, x0, k


def test_mvie_matches_oracle(anet_ctx):
    import allocnet_amd as aa
    rng = np.random.default_rng(3)
    B, M = 150, 18
    A, x0, k = _mvie_batch(rng, B, M)
    call_site = dict(mem_size=18, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7)   # firi.hpp:212-217
    # (a) fixed iteration budgets: the control flow is lbfgs.hpp's statement for statement, so the
    #     iterates agree to rounding and the iteration / evaluation counters exactly
    # (rounding differences grow with the iteration count on this weight-1e3 non-smooth penalty)
    for mi, tol in ((1, 1e-12), (3, 1e-11), (7, 1e-8), (15, 1e-5)):
        x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(max_iterations=mi, **call_site),
                                                   ctx=anet_ctx)
        prm = cbind.lbfgs_default_param(max_iterations=mi, **call_site)
        for b in range(0, B, 3):
            ret, xo, fo, it, ev = cbind.lbfgs_mvie(A[b, :k[b]], 1e-2, 1e3, x0[b], prm)
            assert (status[b], iters[b]) == (ret, it), (mi, b)
            if mi <= 15:
                assert evals[b] == ev, (mi, b)
            assert np.abs(x[b] - xo).max() <= tol * max(1.0, np.abs(xo).max()), (mi, b)
            assert abs(f[b] - fo) <= tol * max(1.0, abs(fo))
    # (b) to convergence with the call-site parameters: hundreds of iterations on a weight-1e3
    #     penalty amplify rounding differences, so only the outcome is compared
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, ctx=anet_ctx)
    prm = cbind.lbfgs_default_param(**call_site)
    for b in range(0, B, 2):
        ret, xo, fo, it, ev = cbind.lbfgs_mvie(A[b, :k[b]], 1e-2, 1e3, x0[b], prm)
        assert status[b] >= 0 and ret >= 0
        assert abs(f[b] - fo) <= 2e-3 * max(1.0, abs(fo)), (b, f[b], fo)
        fchk, _ = cbind.cost_mvie(A[b, :k[b]], 1e-2, 1e3, x[b])
        assert abs(fchk - f[b]) <= 1e-9 * max(1.0, abs(f[b]))      # reported f is f(x returned)
        L = np.array([[x[b, 3] ** 2, 0, 0], [x[b, 6], x[b, 4] ** 2, 0], [x[b, 8], x[b, 7], x[b, 5] ** 2]])
        Ab = A[b, :k[b]]
        assert (np.linalg.norm(Ab @ L, axis=1) + Ab @ x[b, :3] - 1.0).max() < 2e-2   # ellipsoid inside


@pytest.mark.parametrize("mem", [3, 8, 24])
def test_mvie_history_lengths(anet_ctx, mem):
    """The wave-per-problem kernel has three instantiations by history length (registers for
    mem_size <= 8 and <= 20, re-read from memory above); mem_size 3 also wraps the ring buffer
    inside the fixed budget.  Counters exact, iterates to rounding, as in (a) above."""
    import allocnet_amd as aa
    rng = np.random.default_rng(30 + mem)
    B, M = 40, 12
    A, x0, k = _mvie_batch(rng, B, M)
    kw = dict(mem_size=mem, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7, max_iterations=9)
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(**kw), ctx=anet_ctx)
    prm = cbind.lbfgs_default_param(**kw)
    for b in range(0, B, 2):
        ret, xo, fo, it, ev = cbind.lbfgs_mvie(A[b, :k[b]], 1e-2, 1e3, x0[b], prm)
        assert (status[b], iters[b], evals[b]) == (ret, it, ev), b
        assert np.abs(x[b] - xo).max() <= 1e-7 * max(1.0, np.abs(xo).max()), b
        assert abs(f[b] - fo) <= 1e-7 * max(1.0, abs(fo))


@pytest.mark.parametrize("M", [64, 70, 128, 150])
def test_mvie_row_counts_across_the_kernel_variants(anet_ctx, M):
    """The register-resident MVIE kernel holds one or two rows of A per lane (up to 64 / 128 rows); above that the
    optimiser state goes through memory (k_lbfgs_mvie_persistent).  Same counters and iterates whichever runs."""
    import allocnet_amd as aa
    rng = np.random.default_rng(200 + M)
    B = 12
    A, x0, k = _mvie_batch(rng, B, M)
    kw = dict(mem_size=18, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7, max_iterations=8)
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(**kw), ctx=anet_ctx)
    prm = cbind.lbfgs_default_param(**kw)
    for b in range(B):
        ret, xo, fo, it, ev = cbind.lbfgs_mvie(A[b, :k[b]], 1e-2, 1e3, x0[b], prm)
        assert (status[b], iters[b], evals[b]) == (ret, it, ev), b
        assert np.abs(x[b] - xo).max() <= 1e-7 * max(1.0, np.abs(xo).max()), b
        assert abs(f[b] - fo) <= 1e-7 * max(1.0, abs(fo))


def test_mvie_register_resident_vs_state_in_memory_random_shapes(anet_ctx, monkeypatch):
    """Differential fuzz of the two MVIE kernels (k_lbfgs_mvie_resident / k_lbfgs_mvie_persistent; ANET_MVIE_STATE_IN_MEMORY
    is read per call): 1..200 rows, histories 1..30, `past` 0..5, with and without the gradient test.  Short budgets:
    identical counters and iterates; long runs amplify rounding, so their outcome is compared."""
    import allocnet_amd as aa
    rng = np.random.default_rng(7)
    long_runs = [0, 0]                             # problems of the long runs that ended on both sides / that agree
    for trial in range(50):
        M = int(rng.choice([1, 4, 6, 9, 18, 40, 63, 64, 65, 100, 128, 129, 200])); B = int(rng.integers(1, 20))
        A = rng.normal(size=(B, M, 3)); A /= np.linalg.norm(A, axis=2, keepdims=True)
        if M >= 6:
            A[:, :6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
        A /= rng.uniform(0.8, 2.5, size=(B, M, 1))
        k = rng.integers(max(1, M // 2), M + 1, size=B)
        for b in range(B):
            A[b, k[b]:] = 0.0
        x0 = np.tile(np.r_[np.zeros(3), np.sqrt([0.3, 0.3, 0.3]), np.zeros(3)], (B, 1))
        x0[:, :3] += rng.normal(size=(B, 3)) * 0.02
        kw = dict(mem_size=int(rng.choice([1, 2, 8, 9, 18, 20, 21, 30])), g_epsilon=float(rng.choice([0.0, 1e-6])),
                  min_step=1e-32, past=int(rng.choice([0, 1, 3, 5])), delta=1e-7, max_iterations=int(rng.choice([0, 1, 4, 12])))
        me = int(rng.choice([3, 40, 400]))
        monkeypatch.delenv("ANET_MVIE_STATE_IN_MEMORY", raising=False)
        x1, f1, s1, i1, e1 = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(**kw), max_evals=me, ctx=anet_ctx)
        monkeypatch.setenv("ANET_MVIE_STATE_IN_MEMORY", "1")
        x2, f2, s2, i2, e2 = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(**kw), max_evals=me, ctx=anet_ctx)
        tag = (trial, M, B, kw, me)
        if kw["max_iterations"] in (1, 4) or me == 3:
            assert (s1 == s2).all() and (i1 == i2).all() and (e1 == e2).all(), tag
            assert (np.abs(x1 - x2).max(axis=1) <= 1e-9 * np.maximum(1.0, np.abs(x2).max(axis=1))).all(), tag
        else:
            # (a long run of a non-convex problem may part ways for good on one flipped Armijo / Wolfe test and settle in
            # another local minimum: at most one problem of a trial -- or 15 % -- and 3 % of all of them may do so)
            done = (s1 != aa.lbfgs.LBFGS_RUNNING) & (s2 != aa.lbfgs.LBFGS_RUNNING) & (s1 >= 0) & (s2 >= 0)
            agree = np.abs(f1 - f2)[done] <= 5e-3 * np.maximum(1.0, np.abs(f2))[done]
            assert np.isfinite(f1).all() and np.isfinite(f2).all(), tag
            assert (~agree).sum() <= max(1, int(0.15 * done.sum())), (tag, np.abs(f1 - f2)[done])
            long_runs[0] += int(done.sum())
            long_runs[1] += int(agree.sum())
    assert long_runs[0] - long_runs[1] <= max(1, int(0.03 * long_runs[0])), long_runs


def test_mvie_error_codes_and_budget(anet_ctx):
    import allocnet_amd as aa
    rng = np.random.default_rng(4)
    A, x0, k = _mvie_batch(rng, 70, 10)
    with pytest.raises(aa.AnetError):
        aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(mem_size=0), ctx=anet_ctx)
    # iteration cap -> LBFGSERR_MAXIMUMITERATION for every problem
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(max_iterations=2, g_epsilon=0.0,
                                                                                   delta=0.0), ctx=anet_ctx)
    assert (status == aa.lbfgs.LBFGSERR_MAXIMUMITERATION).all() and (iters == 2).all()
    # evaluation budget exhausted -> still running
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, max_evals=3, ctx=anet_ctx)
    assert (status == aa.lbfgs.LBFGS_RUNNING).all() and (evals == 3).all()
    assert "max_evals" in aa.lbfgs_strerror(aa.lbfgs.LBFGS_RUNNING)
    assert aa.lbfgs_strerror(-1009).startswith("Line search reaches the maximum")


def _fwd(tau):
    return np.where(tau > 0, (0.5 * tau + 1) * tau + 1, 1.0 / ((0.5 * tau - 1) * tau + 1))


def _dfwd(tau):
    den = (0.5 * tau - 1) * tau + 1
    return np.where(tau > 0, tau + 1, (1 - tau) / den ** 2)


def _bwd(T):
    big = T > 1                       # (each branch evaluated on its own domain only: no invalid-value warnings)
    return np.where(big, np.sqrt(2 * np.where(big, T, 1.0) - 1) - 1, 1 - np.sqrt(2 / np.where(big, 1.0, T) - 1))


@pytest.mark.parametrize("s,c,N,M", [(4, 3, 4, 8), (3, 3, 5, 6)])
def test_minco_lbfgs_matches_oracle(anet_ctx, s, c, N, M):
    import allocnet_amd as aa
    rng = np.random.default_rng(40 + s)
    B = 6
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=1.5)
    kw = dict(res=8, vmax=3.0, amax=4.0, wc=1e3, wv=1e2, wa=1e2, mu=1e-2)
    rho = 20.0
    pen = aa.make_penalty(rho=rho, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"],
                          max_vel=kw["vmax"], max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    nw = 3 * (N - 1)

    def make_fun(b):
        hpb = np.transpose(hp[b], (1, 2, 0))

        def fun(x):
            w = x[:nw].reshape(N - 1, 3).T
            tau = x[nw:]; Tt = _fwd(tau)
            co, e, *_ = onp.minco_dense_solve(s, head[b], tail[b], w, Tt)
            jp, gC, gTp, _ = onp.penalty_partials(s, co, Tt, hpb, **kw)
            eC, eT = onp.energy_partials(s, co, Tt)
            gP, gT = onp.minco_dense_propagate(s, head[b], tail[b], w, Tt, gC + eC, gTp + eT + rho)
            return e + rho * Tt.sum() + jp, np.r_[gP.T.reshape(-1), gT * _dfwd(tau)]
        return fun
    # fixed iteration budgets: same control flow -> same counters, iterates equal to rounding;
    # long runs only agree in outcome (rounding differences are amplified by the penalty weights)
    for mi, tol in ((2, 1e-9), (6, 1e-7), (200, 5e-3)):
        prm = aa.lbfgs_parameter_t(g_epsilon=1e-6, delta=1e-8, past=3, max_iterations=mi)
        out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, ctx=anet_ctx)
        for b in range(B):
            fun = make_fun(b)
            x0 = np.r_[wps[b].reshape(-1), _bwd(T[b])]
            f0, _ = fun(x0)
            ret, xo, fo, it, ev = cbind.lbfgs_optimize(x0, fun, cbind.lbfgs_default_param(
                g_epsilon=1e-6, delta=1e-8, past=3, max_iterations=mi))
            assert out["cost"][b] < f0                                   # it optimised something
            assert abs(out["cost"][b] - fo) <= tol * abs(fo), (mi, b, out["cost"][b], fo, out["status"][b], ret)
            if mi <= 6:
                assert (out["status"][b], out["iters"][b], out["evals"][b]) == (ret, it, ev), (mi, b)
                xg = np.r_[out["wps"][b].reshape(-1), _bwd(out["T"][b])]
                assert np.abs(xg - xo).max() <= tol * max(1.0, np.abs(xo).max())
            # returned parameters reproduce the returned cost
            fchk, _ = fun(np.r_[out["wps"][b].reshape(-1), _bwd(out["T"][b])])
            assert abs(fchk - out["cost"][b]) <= 1e-8 * abs(fchk)
            # coefficients returned are the MINCO solution of the returned parameters
            co, *_ = onp.minco_dense_solve(s, head[b], tail[b], out["wps"][b].T, out["T"][b])
            assert np.abs(out["coeffs"][b] - co).max() <= 1e-8 * np.abs(co).max()


def test_mvie_lane_kernel_for_long_histories(anet_ctx):
    """History lengths above 64 run the lane-per-problem L-BFGS kernel (everything else one wave per problem,
    covered above): same counters as the oracle on a sample, and -- the history never fills in 12 iterations, so
    its length cannot matter -- the same run as the wave kernel gives with mem_size 64 on the same problems."""
    import allocnet_amd as aa
    rng = np.random.default_rng(8)
    B, M = 33000, 10
    A, x0, k = _mvie_batch(rng, B, M)
    prm = dict(mem_size=70, g_epsilon=0.0, min_step=1e-32, past=3, delta=1e-7, max_iterations=12)
    x, f, status, iters, evals = aa.lbfgs_mvie(A, x0, param=aa.lbfgs_parameter_t(**prm), ctx=anet_ctx)
    cprm = cbind.lbfgs_default_param(**prm)
    for b in range(0, B, 1500):
        ret, xo, fo, it, ev = cbind.lbfgs_mvie(A[b, :k[b]], 1e-2, 1e3, x0[b], cprm)
        assert (status[b], iters[b], evals[b]) == (ret, it, ev), b
        assert np.abs(x[b] - xo).max() <= 1e-6 * max(1.0, np.abs(xo).max())
    prm["mem_size"] = 64
    xs, fs, ss, its, evs = aa.lbfgs_mvie(A[:2000], x0[:2000], param=aa.lbfgs_parameter_t(**prm), ctx=anet_ctx)
    assert np.array_equal(ss, status[:2000]) and np.array_equal(its, iters[:2000]) and np.array_equal(evs, evals[:2000])
    assert np.abs(xs - x[:2000]).max() <= 1e-4        # reduction order differs (DPP tree vs sequential)


@pytest.mark.parametrize("opt_name", ["waypoints", "times"])
def test_minco_lbfgs_partial_variable_sets(anet_ctx, opt_name):
    """ANET_OPT_WAYPOINTS / ANET_OPT_TIMES alone: the other block of variables is left untouched and the
    cost still decreases; the result is a stationary point of the restricted problem."""
    import allocnet_amd as aa
    rng = np.random.default_rng(77)
    s, c, N, M, B = 3, 3, 6, 6, 40
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=2.0)
    pen = aa.make_penalty(rho=30.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=8, poly_rows=M)
    c0, gP0, gT0 = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)
    opt = aa.lbfgs.OPT_WAYPOINTS if opt_name == "waypoints" else aa.lbfgs.OPT_TIMES
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, opt=opt,
                         param=aa.lbfgs_parameter_t(g_epsilon=1e-7, delta=1e-10, max_iterations=400), ctx=anet_ctx)
    assert (out["cost"] < c0).all()
    if opt_name == "waypoints":
        assert np.array_equal(out["T"], T) and not np.array_equal(out["wps"], wps)
    else:
        assert np.array_equal(out["wps"], wps) and not np.array_equal(out["T"], T)
        assert (out["T"] > 0).all()
    c1, gP1, gT1 = aa.minco_cost_grad(head, tail, out["wps"], out["T"], s, hpolys=hp, penalty=pen, ctx=anet_ctx)
    assert np.abs(c1 - out["cost"]).max() <= 1e-9 * np.abs(c1).max()
    g1 = gP1 if opt_name == "waypoints" else gT1
    g0 = gP0 if opt_name == "waypoints" else gT0
    ok = out["status"] >= 0
    assert ok.mean() > 0.5
    assert np.abs(g1[ok]).max() < 1e-2 * np.abs(g0[ok]).max()      # (much) closer to stationarity than the start


def test_minco_s2nu_16_segments_lbfgs(anet_ctx):
    """BASELINE.json labels config 4 "16-segment min-jerk (MINCO_S2NU)"; SURVEY 8(d) reads it as s = 3
    (covered above and by tools/bench_configs.py).  The other reading, s = 2 (MINCO_S2NU, minimum
    acceleration) with 16 segments, through the same driver: gradient checked by central differences of
    the GPU cost (the numpy oracle covers the reference's orders s >= 3), L-BFGS lowers the cost and
    stops at a point where the gradient has dropped."""
    import allocnet_amd as aa
    rng = np.random.default_rng(202)
    s, c, N, M, B = 2, 2, 16, 6, 24
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=2.0)
    pen = aa.make_penalty(rho=20.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=8, poly_rows=M)
    c0, gP0, gT0 = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)
    h = 1e-6
    for (k, ax) in [(0, 0), (7, 1), (14, 2)]:
        wp = wps.copy(); wp[:, k, ax] += h
        wm = wps.copy(); wm[:, k, ax] -= h
        fd = (aa.minco_cost_grad(head, tail, wp, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
              - aa.minco_cost_grad(head, tail, wm, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]) / (2 * h)
        assert np.abs(fd - gP0[:, k, ax]).max() <= 1e-4 * max(1.0, np.abs(gP0[:, k, ax]).max())
    for i in (0, 9, 15):
        tp = T.copy(); tp[:, i] += h
        tm = T.copy(); tm[:, i] -= h
        fd = (aa.minco_cost_grad(head, tail, wps, tp, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
              - aa.minco_cost_grad(head, tail, wps, tm, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]) / (2 * h)
        assert np.abs(fd - gT0[:, i]).max() <= 1e-4 * max(1.0, np.abs(gT0[:, i]).max())
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=3000,
                         ctx=anet_ctx)
    assert (out["cost"] < c0).all() and (out["status"] >= 0).all()
    c1, gP1, gT1 = aa.minco_cost_grad(head, tail, out["wps"], out["T"], s, hpolys=hp, penalty=pen, ctx=anet_ctx)
    fin = out["status"] <= 1          # converged / stopped (a problem still running at max_evals holds a trial point)
    assert fin.sum() >= B // 2
    assert np.abs(c1 - out["cost"])[fin].max() <= 1e-9 * np.abs(c1).max()
    g0 = np.sqrt((gP0 ** 2).sum(axis=(1, 2)) + (gT0 ** 2).sum(axis=1))
    g1 = np.sqrt((gP1 ** 2).sum(axis=(1, 2)) + (gT1 ** 2).sum(axis=1))
    assert (g1[fin] < 0.2 * g0[fin]).all()


@pytest.mark.parametrize("s,c,N,M,res", [(3, 3, 16, 16, 20), (4, 3, 8, 16, 20), (4, 4, 5, 7, 10), (3, 3, 2, 5, 7),
                                          (3, 3, 1, 4, 6), (4, 3, 12, 50, 9), (4, 2, 7, 6, 10), (4, 2, 1, 5, 8),
                                          (3, 2, 9, 8, 12)])
def test_minco_lbfgs_one_launch_agrees_with_lockstep(anet_ctx, s, c, N, M, res):
    """The one-launch kernel (one wave per problem, lbfgs_minco_persistent.h) and the launch-per-evaluation kernels
    are two execution shapes of the same algorithm: at fixed iteration budgets identical counters and return codes,
    iterates and costs equal to rounding; piece counts below / at the lane-group sizes, ragged sample counts
    (res not a multiple of the samples a lane holds), the maximum of 50 corridor rows, a single piece."""
    import allocnet_amd as aa
    rng = np.random.default_rng(900 + 10 * N + s)
    B = 70
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=1.5)
    pen = aa.make_penalty(rho=20.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=res, poly_rows=M)
    both = aa.lbfgs.OPT_WAYPOINTS | aa.lbfgs.OPT_TIMES
    for mi, tol in ((1, 1e-10), (4, 1e-8), (12, 1e-5)):
        prm = aa.lbfgs_parameter_t(g_epsilon=1e-7, delta=1e-9, max_iterations=mi)
        a = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, opt=both, ctx=anet_ctx)
        b = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm,
                           opt=both | aa.lbfgs.OPT_LOCKSTEP, ctx=anet_ctx)
        same = (a["status"] == b["status"]) & (a["iters"] == b["iters"]) & (a["evals"] == b["evals"])
        # (later on a line search or a stopping test may flip at rounding level -- a one-variable problem is converged
        #  to machine precision by then; the costs still agree)
        assert same.mean() >= (1.0 if mi <= 4 else 0.5), (mi, same.mean())
        assert np.abs(a["cost"] - b["cost"]).max() <= max(tol, 1e-6) * np.abs(b["cost"]).max(), mi
        assert np.abs(a["cost"] - b["cost"])[same].max() <= tol * np.abs(b["cost"]).max(), mi
        assert np.abs(a["T"] - b["T"])[same].max() <= tol * 10 * max(1.0, np.abs(b["T"]).max()), mi
        if N > 1:
            assert np.abs(a["wps"] - b["wps"])[same].max() <= tol * 10 * max(1.0, np.abs(b["wps"]).max()), mi


def test_minco_lbfgs_one_launch_vs_lockstep_random_shapes(anet_ctx):
    """Differential fuzz of the two execution shapes over what the fixed cases above leave out: orders 3 / 4, 1..16 pieces,
    every boundary count 2..s (c = 2 with order 4 once hid an in-place overwrite in the scanned sweeps: the pinned rows of
    S_0^-1 are identity rows, which masked it for c >= 3), 0..50 corridor rows, 1..40 samples, the three variable sets,
    history lengths 1 / 3 / 8, `past` 0 / 1 / 3.  Counters identical at budgets <= 2, costs equal to rounding."""
    import allocnet_amd as aa
    rng = np.random.default_rng(2024)
    for trial in range(80):
        s = int(rng.choice([3, 4])); N = int(rng.integers(1, 17))
        c = int(rng.integers(2, s + 1)) if rng.random() < 0.4 else 3
        M = int(rng.choice([0, 1, 3, 7, 16, 33, 50])); res = int(rng.integers(1, 41)); B = int(rng.integers(1, 24))
        opt = int(rng.choice([1, 2, 3])) if N > 1 else 2
        head, tail, wps, T = random_problem(rng, B, N, c, rest=bool(rng.random() < 0.5))
        hp = make_corridors(rng, head, tail, wps, M, tight=1.5) if M else None
        pen = aa.make_penalty(rho=20.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                              res=res, poly_rows=M)
        mi = int(rng.choice([1, 2, 5]))
        prm = aa.lbfgs_parameter_t(g_epsilon=1e-7, delta=1e-9, max_iterations=mi, mem_size=int(rng.choice([1, 3, 8])),
                                   past=int(rng.choice([0, 1, 3])))
        a = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, opt=opt, ctx=anet_ctx)
        b = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, opt=opt | aa.lbfgs.OPT_LOCKSTEP,
                           ctx=anet_ctx)
        same = (a["status"] == b["status"]) & (a["iters"] == b["iters"]) & (a["evals"] == b["evals"])
        rel = np.abs(a["cost"] - b["cost"]) / np.maximum(1e-300, np.abs(b["cost"]))
        tag = dict(trial=trial, s=s, N=N, c=c, M=M, res=res, B=B, opt=opt, mi=mi, mem=prm.mem_size, past=prm.past)
        assert np.isfinite(a["cost"]).all(), tag
        assert same.mean() >= (1.0 if mi <= 2 else 0.6), (tag, same.mean())
        assert rel[same].max(initial=0.0) <= 1e-7, (tag, rel.max())


def test_minco_lbfgs_one_launch_budget_and_fallbacks(anet_ctx):
    """Evaluation budget exhausted -> still running with exactly max_evals evaluations (both shapes); parameter sets
    the one-launch kernel does not hold in registers (mem_size > 8) take the launch-per-evaluation path and still
    optimise."""
    import allocnet_amd as aa
    rng = np.random.default_rng(5)
    s, c, N, M, B = 3, 3, 7, 8, 33
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=1.5)
    pen = aa.make_penalty(rho=20.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=8, poly_rows=M)
    for opt in (3, 3 | aa.lbfgs.OPT_LOCKSTEP):
        out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=5, opt=opt, ctx=anet_ctx)
        assert (out["status"] == aa.lbfgs.LBFGS_RUNNING).all() and (out["evals"] == 5).all()
    c0 = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(mem_size=12),
                         max_evals=3000, ctx=anet_ctx)
    assert (out["cost"] < c0).all() and (out["status"] >= 0).all()
    # no penalty at all: energy + rho * sum(T) only (rho = 0 would drive the durations to infinity; use waypoints only)
    out = aa.lbfgs_minco(head, tail, wps, T, s, opt=aa.lbfgs.OPT_WAYPOINTS, max_evals=500, ctx=anet_ctx)
    _, e0 = aa.minco_solve(head, tail, wps, T, s, ctx=anet_ctx)
    assert (out["cost"] < e0).all() and (out["status"] >= 0).all()
    _, e1 = aa.minco_solve(head, tail, out["wps"], out["T"], s, ctx=anet_ctx)
    assert np.abs(e1 - out["cost"]).max() <= 1e-9 * np.abs(e1).max()


def test_minco_lbfgs_launch_order_changes_nothing_but_the_schedule(anet_ctx):
    """anet_lbfgs_minco_ordered_dev: problems are independent, so any launch order returns bit-identical results;
    an out-of-range entry is skipped (its problem keeps the status it had)."""
    import torch
    import allocnet_amd as aa
    from tools.bench_configs import to_bm
    rng = np.random.default_rng(77)
    s, c, N, M, B = 3, 3, 9, 8, 300
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=1.5)
    pen = aa.make_penalty(rho=20.0, w_corridor=1e3, w_vel=1e2, w_acc=1e2, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0,
                          res=8, poly_rows=M)
    dev = torch.device("cuda", 0)
    ld = aa.recommended_ld(B)

    def run(order):
        th, tt, tw, tT = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T))
        thp = to_bm(torch, hp, B, ld, dev)
        r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, max_evals=400, opt=3,
                               launch_order=order, ctx=anet_ctx)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in r.items()}, tw[:, :B].cpu().numpy(), tT[:, :B].cpu().numpy()

    ref, w0, T0 = run(None)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(B).astype(np.int32)).to(dev)
    by_count = aa.launch_order_from_counts(torch.from_numpy(ref["evals"]).to(dev))
    assert sorted(by_count.cpu().tolist()) == list(range(B)) and ref["evals"][by_count.cpu().numpy()[0]] == ref["evals"].max()
    lib_order = aa.lbfgs.launch_order_from_counts_dev(torch.from_numpy(ref["evals"]).to(dev), ctx=anet_ctx)
    lo = lib_order.cpu().numpy()
    assert sorted(lo.tolist()) == list(range(B)) and (np.diff(ref["evals"][lo] >> 4) <= 0).all()      # longest first, 16 per bucket
    for order in (perm, by_count, lib_order):
        got, w1, T1 = run(order)
        for k in ref:
            assert np.array_equal(ref[k], got[k]), k
        assert np.array_equal(w0, w1) and np.array_equal(T0, T1)
    with pytest.raises(ValueError):
        run(perm.to(torch.int64))
    # an order that is NOT a permutation (entry 5 twice, problem 7 never): the skipped problem reports RUNNING with zero
    # counters and keeps its start point -- never whatever the workspace held
    bad = torch.arange(B, dtype=torch.int32, device=dev)
    bad[7] = 5
    got, w1, T1 = run(bad)
    assert got["status"][7] == aa.lbfgs.LBFGS_RUNNING and got["iters"][7] == 0 and got["evals"][7] == 0
    assert np.isnan(got["cost"][7])                              # no stale cost for a problem that never ran
    assert np.array_equal(w1[:, 7], wps[7].reshape(-1)) and np.allclose(T1[:, 7], T[7], rtol=1e-14, atol=0)   # (T -> tau -> T)
    keep = np.arange(B) != 7
    for k in ("status", "iters", "evals", "cost"):
        assert np.array_equal(ref[k][keep], got[k][keep]), k


def test_returned_coefficients_of_wide_spread_durations_come_from_the_pivoted_solve(anet_ctx):
    """Durations that spread over more than 10^3 at the end of an optimisation: the coefficients anet_lbfgs_minco hands
    back are those of the pivoted collocation solve (<= 1e-6 of the classic banded-LU oracle), not of the reduced system the
    loop itself works with, and the problem is flagged."""
    import allocnet_amd as aa
    rng = np.random.default_rng(9)
    s, c, N, B = 4, 3, 6, 24
    seg = np.tile(np.array([0.004, 9.0, 0.004, 7.0, 0.004, 8.0]), (B, 1)) * rng.uniform(0.8, 1.2, size=(B, N))
    d = rng.normal(size=(B, N, 3)); d /= np.linalg.norm(d, axis=2, keepdims=True)
    pts = np.concatenate([np.zeros((B, 1, 3)), np.cumsum(d * seg[:, :, None], axis=1)], axis=1)
    head = np.zeros((B, 3, c)); tail = np.zeros((B, 3, c))
    head[:, :, 0] = pts[:, 0]; tail[:, :, 0] = pts[:, N]
    wps = pts[:, 1:N].copy()
    T = np.where(seg < 0.1, 0.012, 30.0) * rng.uniform(0.9, 1.1, size=(B, N))
    T[B // 2:] = rng.uniform(0.8, 1.6, size=(B - B // 2, N))           # second half: ordinary durations
    pen = aa.make_penalty(rho=1e-6, w_corridor=0.0, w_vel=0.0, w_acc=0.0, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=4,
                          poly_rows=0)
    out = aa.lbfgs_minco(head, tail, wps, T, s, penalty=pen, param=aa.lbfgs_parameter_t(max_iterations=3), max_evals=60,
                         ctx=anet_ctx)
    spread = out["T"].max(axis=1) / out["T"].min(axis=1)
    assert (spread[:B // 2] > 1e3).all() and (spread[B // 2:] < 50).all(), spread
    assert out["wide_spread"][:B // 2].all() and not out["wide_spread"][B // 2:].any()
    co, _ = cbind.minco_solve_batch(s, head, tail, out["wps"], out["T"])
    for b in range(B):
        err = np.abs(out["coeffs"][b] - co[b]).max() / np.abs(co[b]).max()
        assert err <= 1e-6, (b, spread[b], err)
    # the same through the device entry point: flags from anet_minco_spread_flags_dev
    import torch
    from tools.bench_configs import to_bm
    dev = torch.device("cuda", 0)
    ld = aa.recommended_ld(B)
    th, tt, tw, tT = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T))
    coeffs = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64)
    r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, penalty=pen, param=aa.lbfgs_parameter_t(max_iterations=3),
                           max_evals=60, coeffs=coeffs, ctx=anet_ctx)
    torch.cuda.synchronize()
    assert np.array_equal(r["wide_spread"].cpu().numpy().astype(bool), out["wide_spread"])
    cd = coeffs[:, :B].cpu().numpy().T.reshape(B, N, 3, 2 * s)
    assert np.abs(cd - out["coeffs"]).max() <= 1e-12 * np.abs(out["coeffs"]).max()


@pytest.mark.parametrize("lockstep", [False, True])
def test_minco_lbfgs_step_bound_keeps_a_minimum_duration(anet_ctx, lockstep):
    """lbfgs_optimize's proc_stepbound (lbfgs.hpp:557-565) with the built-in minimum-duration bound
    (anet_lbfgs_minco_bounded): a cost that wants short durations (large rho) drives them below T_min without the bound and
    never with it; (status, iterations, evaluations) and iterates equal the C restatement of lbfgs_optimize WITH the same
    bound as its proc_stepbound callback.  Both execution shapes: the one-launch kernel and the lockstep update kernels."""
    import allocnet_amd as aa
    shape = aa.lbfgs.OPT_WAYPOINTS | aa.lbfgs.OPT_TIMES | (aa.lbfgs.OPT_LOCKSTEP if lockstep else 0)
    rng = np.random.default_rng(31)
    s, c, N, M, B = 3, 3, 6, 8, 96
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=2.0)
    T = rng.uniform(1.2, 2.0, size=(B, N))
    kw = dict(res=8, vmax=3.0, amax=4.0, wc=1e3, wv=1.0, wa=1.0, mu=1e-2)
    rho = 400.0                      # time is expensive: the optimiser shortens the pieces
    pen = aa.make_penalty(rho=rho, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"],
                          max_vel=kw["vmax"], max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    Tmin = 0.9
    free = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx)
    assert (free["T"].min(axis=1) < Tmin).sum() >= B // 4          # the bound has something to do
    for budget in (4, 15, 0):
        prm = aa.lbfgs_parameter_t(max_iterations=budget)
        out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, max_evals=3000, min_duration=Tmin,
                             opt=shape, ctx=anet_ctx)
        assert (out["T"] >= Tmin * (1 - 1e-12)).all(), out["T"].min()
        ref = cbind.lbfgs_minco_batch(s, head, tail, wps, T, hp, rho, nthreads=4, min_duration=Tmin,
                                      param=cbind.lbfgs_default_param(max_iterations=budget), **kw)
        assert (ref["T"] >= Tmin * (1 - 1e-12)).all()
        same = (out["status"] == ref["status"]) & (out["iters"] == ref["iters"]) & (out["evals"] == ref["evals"])
        # A duration that a line search has put ON the bound leaves the next bound at rounding level (room = tau - tau_min
        # is +-1 ulp): such a problem ends with a failed line search on both sides -- step below min_step on one, step_max
        # tried twice on the other, whichever way its last bit fell -- at the same point.  That pair of outcomes is one class.
        stuck = np.isin(out["status"], (-1011, -1010)) & np.isin(ref["status"], (-1011, -1010)) & (out["iters"] == ref["iters"])
        agree = same | stuck
        rel = np.abs(out["cost"] - ref["cost"]) / np.abs(ref["cost"])
        print("step bound, lockstep=%s budget=%d: same %.3f agree %.3f; rel cost: agreeing max %.1e, others max %.1e median %.1e"
              % (lockstep, budget, same.mean(), agree.mean(), rel[agree].max(), rel[~agree].max() if (~agree).any() else 0.0,
                 np.median(rel[~agree]) if (~agree).any() else 0.0))
        # (measured with the restatement built without contraction and the kernels' trial point unfused: same 0.92-0.93,
        #  agree 0.99 at every budget, the run to each problem's own stop included)
        assert same.mean() >= 0.9 and agree.mean() >= 0.98, (budget, same.mean(), agree.mean())
        assert rel[agree].max() <= 1e-6, (budget, rel[agree].max())
        if budget:
            assert np.abs(out["T"][agree] - ref["T"][agree]).max() <= 1e-6
        # the problems whose counters differ are not excused: they ran to their own stop on both sides (budget 0: a flipped
        # test at the bound sends the two runs along different iterates), they respect the bound, and they end at a cost
        # within 2 % of the restatement's
        assert (out["T"][~agree] >= Tmin * (1 - 1e-12)).all()
        if (~agree).any():
            assert rel[~agree].max() <= 2e-2, (budget, rel[~agree].max())
    # the bound is active, not decorative: some problem ends ON the minimum, and the bounded optimum costs more
    assert (np.abs(out["T"] - Tmin).min(axis=1) < 1e-6).sum() >= 1
    assert out["cost"].mean() > free["cost"].mean()
    # min_duration = 0 is the unbounded call, bit for bit
    again = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, min_duration=0.0, opt=shape,
                           ctx=anet_ctx)
    assert np.array_equal(again["cost"], free["cost"]) and np.array_equal(again["evals"], free["evals"])


def test_minco_lbfgs_step_bound_shapes_agree(anet_ctx):
    """The bounded run at fixed budgets: the one-launch kernel and the lockstep update kernels return the same counters and
    iterates (as the unbounded shapes do, test_minco_lbfgs_one_launch_agrees_with_lockstep)."""
    import allocnet_amd as aa
    rng = np.random.default_rng(33)
    s, c, N, M, B = 4, 3, 5, 8, 64
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=2.0)
    T = rng.uniform(1.2, 2.0, size=(B, N))
    pen = aa.make_penalty(rho=400.0, w_corridor=1e3, w_vel=1.0, w_acc=1.0, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0, res=8,
                          poly_rows=M)
    both = aa.lbfgs.OPT_WAYPOINTS | aa.lbfgs.OPT_TIMES
    for budget in (3, 10):
        prm = aa.lbfgs_parameter_t(max_iterations=budget)
        a = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, max_evals=2000, min_duration=1.0, opt=both,
                           ctx=anet_ctx)
        b = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=prm, max_evals=2000, min_duration=1.0,
                           opt=both | aa.lbfgs.OPT_LOCKSTEP, ctx=anet_ctx)
        same = (a["status"] == b["status"]) & (a["iters"] == b["iters"]) & (a["evals"] == b["evals"])
        # (a duration ON the bound: the next line search fails on both sides, by step_max tried twice or by a step below
        # min_step as the last bit of tau - tau_min fell -- one class, as in the test against the restatement)
        stuck = np.isin(a["status"], (-1011, -1010)) & np.isin(b["status"], (-1011, -1010)) & (a["iters"] == b["iters"])
        agree = same | stuck
        # (64 problems: at most three may part ways.  The lockstep shape evaluates small batches with the one-launch cost + gradient
        #  kernel since round 5, minco_fused_kernel.h, whose adjoint rounds differently from the one-launch L-BFGS kernel's.)
        assert same.mean() >= 0.8 and agree.mean() >= 0.95, (budget, same.mean(), agree.mean())
        assert (a["T"] >= 1.0 - 1e-12).all() and (b["T"] >= 1.0 - 1e-12).all()
        assert np.abs(a["cost"] - b["cost"])[agree].max() <= 1e-7 * np.abs(b["cost"]).max()
        assert np.abs(a["T"] - b["T"])[agree].max() <= 1e-6


@pytest.mark.parametrize("lockstep", [False, True])
def test_minco_lbfgs_cancel_word(anet_ctx, lockstep):
    """(both execution shapes) lbfgs_optimize's progress callback (lbfgs.hpp:580-587: called after every successful line search, a non-zero return
    ends the run with LBFGS_CANCELED) as the cancel word of anet_set_cancel_flag.  With the word at zero the run is the
    plain one bit for bit; with the word set from the start every problem is cancelled after its FIRST iteration -- at the
    very point a run with max_iterations = 1 stops (the progress report comes before the convergence, stop and iteration
    tests of the same iteration), so iterates, cost and counters must be those, only the return code differs."""
    import torch
    import allocnet_amd as aa
    from allocnet_amd import lbfgs as L
    shape = aa.lbfgs.OPT_WAYPOINTS | aa.lbfgs.OPT_TIMES | (aa.lbfgs.OPT_LOCKSTEP if lockstep else 0)
    rng = np.random.default_rng(32)
    s, c, N, M, B = 3, 3, 6, 8, 96
    head, tail, wps, T = random_problem(rng, B, N, c, rest=True)
    hp = make_corridors(rng, head, tail, wps, M, tight=2.0)
    T = rng.uniform(1.2, 2.0, size=(B, N))
    pen = aa.make_penalty(rho=50.0, w_corridor=1e3, w_vel=1.0, w_acc=1.0, smooth_mu=1e-2, max_vel=3.0, max_acc=4.0, res=8,
                          poly_rows=M)
    free = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx)
    one = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx,
                         param=aa.lbfgs_parameter_t(max_iterations=1))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    anet_ctx.set_cancel_flag(flag)
    try:
        same = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx)
        for k in ("cost", "evals", "iters", "status", "T"):
            assert np.array_equal(same[k], free[k]), k
        flag.fill_(1)
        torch.cuda.synchronize()
        can = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx)
    finally:
        anet_ctx.set_cancel_flag(None)
    ran = one["status"] == L.LBFGSERR_MAXIMUMITERATION          # problems whose first iteration completed and went on
    assert ran.mean() > 0.9
    assert (can["status"][ran] == L.LBFGS_CANCELED).all()
    for k in ("iters", "evals", "cost", "T", "wps"):
        assert np.array_equal(can[k][ran], one[k][ran]), k
    # the others: a failed first line search keeps its error code; a first iteration that would have ended the run on
    # its own is reported as cancelled (the report comes first)
    rest = ~ran
    assert np.isin(can["status"][rest], (L.LBFGS_CANCELED,) + tuple(np.unique(one["status"][rest]))).all()
    assert (can["status"][rest & (one["status"] < 0)] == one["status"][rest & (one["status"] < 0)]).all()
    # cleared: the plain run again
    again = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, max_evals=3000, opt=shape, ctx=anet_ctx)
    assert np.array_equal(again["status"], free["status"]) and np.array_equal(again["evals"], free["evals"])


def test_minco_lbfgs_cancel_word_set_while_the_run_is_in_flight(anet_ctx):
    """The cancel word written WHILE the one-launch kernel runs (anet_set_cancel_flag: "device memory written from another
    stream, or mapped pinned host memory") must be seen by the running problems: the kernel reads it with a system-scope load
    once per evaluation.  A long batch (BASELINE configs[3] shape, ~0.2 s uncancelled) is launched, the host sets the word in
    mapped pinned memory a few milliseconds later; every problem still running then stops with LBFGS_CANCELED and the
    batch spends far fewer evaluations than the uncancelled run."""
    import time
    import torch
    import allocnet_amd as aa
    from allocnet_amd import lbfgs as L
    from allocnet_amd.synth import corridor_problem
    B, s, c, N, M = 2048, 3, 3, 16, 16
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(2), B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20,
                          poly_rows=M)
    ld = aa.recommended_ld(B)
    dev = torch.device("cuda", 0)

    def bm(a):
        f = np.ascontiguousarray(a.reshape(B, -1).T)
        t = torch.zeros(f.shape[0], ld, device=dev, dtype=torch.float64)
        t[:, :B] = torch.from_numpy(f).to(dev)
        return t
    thp = bm(hp)
    free = aa.lbfgs_minco_dev(bm(head), bm(tail), bm(wps), bm(T), s, c, N, B, hpolys=thp, penalty=pen, max_evals=40000,
                              ctx=anet_ctx)
    torch.cuda.synchronize()
    ev_free = free["evals"].cpu().numpy()
    assert ev_free.mean() > 500                                   # a run long enough to be interrupted
    flag = torch.zeros(1, dtype=torch.int32).pin_memory()         # mapped pinned host memory: the device reads the host's word
    anet_ctx.set_cancel_flag(flag)
    try:
        args = (bm(head), bm(tail), bm(wps), bm(T))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = aa.lbfgs_minco_dev(*args, s, c, N, B, hpolys=thp, penalty=pen, max_evals=40000, ctx=anet_ctx)   # asynchronous
        time.sleep(0.004)
        flag[0] = 1                                               # host store, while the kernel is running
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        anet_ctx.set_cancel_flag(None)
    st = res["status"].cpu().numpy()
    ev = res["evals"].cpu().numpy()
    assert (st == L.LBFGS_CANCELED).mean() > 0.9, np.unique(st, return_counts=True)
    assert ev.sum() < 0.5 * ev_free.sum(), (ev.sum(), ev_free.sum(), dt)
    assert np.isfinite(res["cost"].cpu().numpy()).all()


@pytest.mark.parametrize("n,B", [(10, 200), (2, 5), (70, 33), (140, 9)])
def test_generic_device_objective_matches_the_restatement(anet_ctx, n, B):
    """anet_lbfgs_optimize_dev: lbfgs_optimize (lbfgs.hpp:434-717) for an objective the caller evaluates on the device -- here
    the extended Rosenbrock function written with torch operations -- against the C restatement of lbfgs_optimize driving
    the same function in numpy, problem by problem: counters equal at a fixed budget for nearly every problem (a sum in another
    order may flip a test that sits on its threshold), every problem ends at the minimum when left to run.  Covers the wave
    kernels (n <= 128) and the lane kernel (n = 140)."""
    import torch
    import allocnet_amd as aa
    rng = np.random.default_rng(50 + n)
    x0 = rng.uniform(-1.5, 1.5, size=(B, n))
    dev = torch.device("cuda", 0)
    ld = aa.recommended_ld(B)

    def evaluate(x, f, g):
        a, b = x[:-1], x[1:]
        t = b - a * a
        f.copy_((100.0 * t * t + (1.0 - a) ** 2).sum(dim=0))
        g.zero_()
        g[:-1] += -400.0 * a * t - 2.0 * (1.0 - a)
        g[1:] += 200.0 * t

    def fun(x):
        a, b = x[:-1], x[1:]
        t = b - a * a
        g = np.zeros_like(x)
        g[:-1] += -400.0 * a * t - 2.0 * (1.0 - a)
        g[1:] += 200.0 * t
        return (100.0 * t * t + (1.0 - a) ** 2).sum(), g

    def start():
        x = torch.zeros(n, ld, device=dev, dtype=torch.float64)
        x[:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).to(dev)
        return x
    budget = 12
    x = start()
    out = aa.lbfgs_optimize_dev(x, evaluate, batch=B, param=aa.lbfgs_parameter_t(max_iterations=budget), max_evals=400, ctx=anet_ctx)
    st, it, ev = (out[k].cpu().numpy() for k in ("status", "iters", "evals"))
    fg = out["f"].cpu().numpy()
    xg = x[:, :B].cpu().numpy().T
    same = 0
    for b in range(B):
        ret, xo, fo, ito, evo = cbind.lbfgs_optimize(x0[b], fun, cbind.lbfgs_default_param(max_iterations=budget))
        if (st[b], it[b], ev[b]) == (ret, ito, evo):
            same += 1
            assert abs(fg[b] - fo) <= 1e-9 * max(1.0, abs(fo)), b
            assert np.abs(xg[b] - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max()), b
        else:
            # a problem whose counters differ is not excused: a flipped line-search test costs a trial or an iteration, not the
            # run -- the same kind of outcome, within a few evaluations, and a cost the same order of magnitude
            assert (st[b] < 0) == (ret < 0) and abs(int(ev[b]) - evo) <= 6 and abs(int(it[b]) - ito) <= 2, (b, st[b], it[b], ev[b], ret, ito, evo)
            assert fg[b] <= 10.0 * fo + 1.0 and fo <= 10.0 * fg[b] + 1.0, (b, fg[b], fo)
    print("device-objective L-BFGS, n = %d: identical (status, k, evals) in %d of %d problems" % (n, same, B))
    # measured (round 6): every problem of every shape (200 / 5 / 33 / 9) -- deterministic on both sides; the bounded branch above
    # stays for the day a reordered sum flips a test on its threshold, but at most one problem in fifty may take it
    assert same >= B - B // 50, (same, B)
    # the built-in step bound (bound_from / bound_min: the last variables may not fall below a floor within a line search,
    # lbfgs.hpp:557-565) against the restatement running the same bound as its proc_stepbound callback
    nb = max(1, n // 3)
    floor_ = -1.75
    x = start()
    out = aa.lbfgs_optimize_dev(x, evaluate, batch=B, param=aa.lbfgs_parameter_t(max_iterations=budget), max_evals=400,
                                bound_from=n - nb, bound_min=floor_, ctx=anet_ctx)
    st, it, ev = (out[k].cpu().numpy() for k in ("status", "iters", "evals"))
    xg = x[:, :B].cpu().numpy().T

    def sb(xp, d):
        worst = 0.0
        for i in range(n - nb, n):
            if d[i] < 0.0:
                worst = max(worst, -d[i] / max(xp[i] - floor_, 1e-300))
        return 1.0 / worst if worst > 0.0 else np.inf
    same_b = 0
    for b in range(B):
        ret, xo, fo, ito, evo = cbind.lbfgs_optimize(x0[b], fun, cbind.lbfgs_default_param(max_iterations=budget), stepbound=sb)
        if (st[b], it[b], ev[b]) == (ret, ito, evo):
            same_b += 1
            assert np.abs(xg[b] - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max()), b
    assert same_b >= 0.85 * B, (same_b, B)
    assert (xg[:, n - nb:] >= floor_ - 1e-12).all()
    # the cancel word (proc_progress's one effect): set from the start, every problem stops after its first iteration
    flag = torch.ones(1, dtype=torch.int32, device=dev)
    anet_ctx.set_cancel_flag(flag)
    try:
        x = start()
        out = aa.lbfgs_optimize_dev(x, evaluate, batch=B, max_evals=400, ctx=anet_ctx)
        stc, itc = out["status"].cpu().numpy(), out["iters"].cpu().numpy()
        assert ((stc == aa.lbfgs.LBFGS_CANCELED) | (stc < 0) | (stc == aa.lbfgs.LBFGS_CONVERGENCE)).all() and (itc <= 1).all()
        assert (stc == aa.lbfgs.LBFGS_CANCELED).mean() >= 0.9
    finally:
        anet_ctx.set_cancel_flag(None)
    # left to run: every problem reaches the minimum f = 0 at x = 1 (a local minimum near x_0 = -1 exists for n >= 4: accept it)
    x = start()
    out = aa.lbfgs_optimize_dev(x, evaluate, batch=B, param=aa.lbfgs_parameter_t(g_epsilon=1e-6, delta=0.0, past=0), max_evals=20000,
                                ctx=anet_ctx)
    st = out["status"].cpu().numpy()
    xf = x[:, :B].cpu().numpy().T
    gn = np.array([np.abs(fun(xf[b])[1]).max() / max(1.0, np.abs(xf[b]).max()) for b in range(B)])
    # (the gradient test of lbfgs.hpp:590-597 met, or a line search that found nothing left to gain at rounding level)
    assert ((st == aa.lbfgs.LBFGS_CONVERGENCE) | ((st < 0) & (gn < 1e-4))).all(), (np.unique(st, return_counts=True), gn.max())
    assert (st == aa.lbfgs.LBFGS_CONVERGENCE).mean() >= 0.8
    assert gn[st == aa.lbfgs.LBFGS_CONVERGENCE].max() < 1e-6
    assert (out["f"].cpu().numpy() < 4.0).all()
    # an exception in the objective comes back as the exception, not as a crash
    with pytest.raises(ZeroDivisionError):
        aa.lbfgs_optimize_dev(start(), lambda x, f, g: 1 / 0, batch=B, ctx=anet_ctx)


def test_minco_lbfgs_two_launch_form_returns_the_same_bits():
    """Batches of 4096 problems and more with ten pieces or more run the one-launch shape in TWO launches (lbfgs_minco_persistent.h
    PersistArgs::park; the thresholds are lowered here through the environment so that smaller shapes take the path too):
    1000 evaluations of every problem, the optimisers of the unfinished ones parked, then resumed longest-expected first.
    Parking changes no arithmetic: against a single launch (ANET_LBFGS_SPLIT_EVALS=0, read once per process: two child
    processes) waypoints, durations, costs, return codes and both counters are bit-identical -- with and without the
    minimum-duration bound, problems that stop before the split point and after it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, hashlib, numpy as np; sys.path.insert(0, %r); import allocnet_amd as aa\n"
            "from allocnet_amd.synth import corridor_problem\n"
            "out = {}\n"
            "for (s, N, B, mind) in ((3, 16, 3200, 0.0), (4, 8, 3100, 0.0), (3, 5, 3072, 0.4)):\n"
            "    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(41), B, N, 3, 16)\n"
            "    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=16)\n"
            "    kw = dict(min_duration=mind) if mind else {}\n"
            "    r = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=20000, **kw)\n"
            "    h = hashlib.sha256()\n"
            "    for k in ('wps', 'T', 'cost', 'status', 'iters', 'evals'): h.update(np.ascontiguousarray(r[k]).tobytes())\n"
            "    ev = r['evals']\n"
            "    out['%%d_%%d' %% (s, N)] = dict(sha=h.hexdigest(), below=int((ev < 1000).sum()), above=int((ev > 1000).sum()), longest=int(ev.max()))\n"
            "print(json.dumps(out))\n") % root
    res = {}
    for name, val in (("two", "1000"), ("one", "0")):
        env = dict(os.environ, ANET_LBFGS_SPLIT_EVALS=val, ANET_LBFGS_SPLIT_MIN_BATCH="3000", ANET_LBFGS_SPLIT_MIN_VARS="1")
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["two"] == res["one"], res
    assert sum(v["below"] for v in res["two"].values()) > 100 and sum(v["above"] for v in res["two"].values()) > 1000, res


def _rosen(x):
    a, b = x[:-1], x[1:]
    t = b - a * a
    g = np.zeros_like(x)
    g[:-1] += -400.0 * a * t - 2.0 * (1.0 - a)
    g[1:] += 200.0 * t
    return (100.0 * t * t + (1.0 - a) ** 2).sum(), g


@pytest.mark.parametrize("n", [2, 9, 40, 300])
def test_host_callback_objective_matches_the_restatement(anet_ctx, n):
    """anet_lbfgs_optimize_host = lbfgs::lbfgs_optimize with the reference's three HOST callbacks (lbfgs.hpp:186-246): the
    objective evaluated on the host, the optimiser's vectors and arithmetic on the device, the state machine parked where the
    reference calls back.  Against the C restatement driving the same Python callbacks: return code and both counters equal at
    fixed budgets for nearly every start point (a dot product summed in another order may flip a test on its threshold), iterates
    to 1e-8 where they are -- plain, with a step bound, and with a progress monitor whose every call (k, ls, fx, step) is compared
    and which cancels the run."""
    import allocnet_amd as aa
    rng = np.random.default_rng(700 + n)
    starts = rng.uniform(-1.5, 1.5, size=(12, n))
    bound = lambda xp, d: 0.3 / np.abs(d).max()          # no variable moves by more than 0.3 per line search
    total = same = 0
    for budget in (2, 7, 25):
        for x0 in starts:
            for mode in ("plain", "bound", "progress"):
                log_g, log_o = [], []

                def mk(log, stop_at):
                    def progress(x, g, fx, step, k, ls):
                        log.append((k, ls, fx, step, float(np.abs(g).max())))
                        return 1 if k >= stop_at else 0
                    return progress
                kw_g = dict(stepbound=bound if mode != "plain" else None, progress=mk(log_g, 5) if mode == "progress" else None)
                kw_o = dict(stepbound=bound if mode != "plain" else None, progress=mk(log_o, 5) if mode == "progress" else None)
                ret, xg, fg, it, ev = aa.lbfgs_optimize(x0, _rosen, param=aa.lbfgs_parameter_t(max_iterations=budget), ctx=anet_ctx, **kw_g)
                reto, xo, fo, ito, evo = cbind.lbfgs_optimize(x0, _rosen, cbind.lbfgs_default_param(max_iterations=budget), **kw_o)
                total += 1
                if (ret, it, ev) == (reto, ito, evo):
                    same += 1
                    assert abs(fg - fo) <= 1e-9 * max(1.0, abs(fo)), (n, budget, mode)
                    assert np.abs(xg - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max()), (n, budget, mode)
                    assert len(log_g) == len(log_o)
                    for a, b in zip(log_g, log_o):
                        assert a[:2] == b[:2] and np.allclose(a[2:], b[2:], rtol=1e-8, atol=1e-12), (a, b)
                else:
                    # not excused: a line-search test that flipped on its threshold costs a trial or an iteration, not the run
                    assert (ret < 0) == (reto < 0) and abs(ev - evo) <= 6 and abs(it - ito) <= 2, (n, budget, mode, ret, it, ev, reto, ito, evo)
                if mode == "progress" and budget > 5:
                    assert ret == aa.lbfgs.LBFGS_CANCELED or ret < 0 or it < 5, (ret, it)
    print("host-callback L-BFGS, n = %d: identical (ret, k, evals) in %d of %d runs" % (n, same, total))
    # measured (round 6): 108 of 108 runs for every n -- the arithmetic is deterministic on both sides, so anything less is a change
    assert same == total, (same, total)
    # left to run from the classic start: the minimum f = 0 at x = 1
    x0 = np.where(np.arange(n) % 2 == 0, -1.2, 1.0)
    ret, x, f, it, ev = aa.lbfgs_optimize(x0, _rosen, param=aa.lbfgs_parameter_t(g_epsilon=1e-7, delta=0.0, past=0), ctx=anet_ctx)
    assert ret == aa.lbfgs.LBFGS_CONVERGENCE or (ret < 0 and np.abs(_rosen(x)[1]).max() < 1e-4), (ret, f)
    assert f < 4.0 and ev >= it


def test_host_callback_objective_validation_and_errors(anet_ctx):
    """Parameter errors are lbfgs_optimize's return value, in its order (lbfgs.hpp:449-495), before any evaluation; an exception
    in a callback comes back as the exception; the start point may already be stationary (LBFGS_CONVERGENCE after one evaluation)."""
    import allocnet_amd as aa
    calls = []

    def fun(x):
        calls.append(1)
        return _rosen(x)
    ret, x, f, it, ev = aa.lbfgs_optimize(np.zeros(4), fun, param=aa.lbfgs_parameter_t(f_dec_coeff=1.5), ctx=anet_ctx)
    assert ret == -1016 and not calls and ev == 0
    ret, *_ = aa.lbfgs_optimize(np.zeros(4), fun, param=aa.lbfgs_parameter_t(mem_size=0), ctx=anet_ctx)
    assert ret == -1022 and not calls
    ret, x, f, it, ev = aa.lbfgs_optimize(np.ones(6), fun, ctx=anet_ctx)
    assert ret == aa.lbfgs.LBFGS_CONVERGENCE and ev == 1 and f == 0.0 and np.array_equal(x, np.ones(6))
    with pytest.raises(ZeroDivisionError):
        aa.lbfgs_optimize(np.zeros(4), lambda x: (1.0 / 0.0, x), ctx=anet_ctx)
    # a non-finite objective value in a line search ends the run as the reference does (LBFGSERR_INVALID_FUNCVAL), x restored
    state = {"k": 0}

    def poisoned(x):
        state["k"] += 1
        f, g = _rosen(x)
        return (float("inf") if state["k"] == 3 else f), g
    ret, x, f, it, ev = aa.lbfgs_optimize(np.full(4, -1.0), poisoned, ctx=anet_ctx)
    st2 = {"k": 0}

    def poisoned2(x):
        st2["k"] += 1
        f, g = _rosen(x)
        return (float("inf") if st2["k"] == 3 else f), g
    reto, xo, fo, ito, evo = cbind.lbfgs_optimize(np.full(4, -1.0), poisoned2)
    assert (ret, it, ev) == (reto, ito, evo) and ret == -1012 and np.allclose(x, xo, rtol=0, atol=1e-12)
