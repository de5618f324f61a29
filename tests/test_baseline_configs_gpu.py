"""BASELINE.json configs[2] and configs[3] at their full sizes, with the generators and seeds of SURVEY.md 8(d):

  configs[2]  Batch=4096 x 8-segment min-snap, SFC corridor penalties + time-allocation gradients   (seed 1)
  configs[3]  Batch=4096 x 16-segment min-jerk, full L-BFGS to convergence, lbfgs_parameter_t defaults (seed 2)

The whole batch runs on the GPU; a strided sample is compared with the oracles (the numpy dense adjoint for cost and
gradients, the C restatement of lbfgs.hpp:434-717 driving that numpy objective for the L-BFGS counters)."""
import numpy as np
import pytest

from oracle import cbind
from oracle import minco_np as onp
from tests.util import corridor_problem

pytestmark = pytest.mark.gpu

KW = dict(res=20, vmax=4.0, amax=6.0, wc=1e4, wv=1e3, wa=1e3, mu=1e-2)      # planner.yaml:17-21 limits, res
RHO = 50.0


def _penalty(aa, M):
    return aa.make_penalty(rho=RHO, w_corridor=KW["wc"], w_vel=KW["wv"], w_acc=KW["wa"], smooth_mu=KW["mu"],
                           max_vel=KW["vmax"], max_acc=KW["amax"], res=KW["res"], poly_rows=M)


def _fwd(tau):
    return np.where(tau > 0, (0.5 * tau + 1) * tau + 1, 1.0 / ((0.5 * tau - 1) * tau + 1))


def _dfwd(tau):
    den = (0.5 * tau - 1) * tau + 1
    return np.where(tau > 0, tau + 1, (1 - tau) / den ** 2)


def _bwd(T):
    with np.errstate(invalid="ignore"):          # (np.where evaluates both branches)
        return np.where(T > 1, np.sqrt(2 * T - 1) - 1, 1 - np.sqrt(2 / T - 1))


def _oracle_cost_grad(s, head, tail, wps, T, hp):
    """cost, dJ/dwaypoints (3, N-1), dJ/dT (N,) of one trajectory by the numpy oracle (classic dense adjoint)."""
    hpb = np.transpose(hp, (1, 2, 0))
    co, e, *_ = onp.minco_dense_solve(s, head, tail, wps.T, T)
    jp, gC, gTp, _ = onp.penalty_partials(s, co, T, hpb, **KW)
    eC, eT = onp.energy_partials(s, co, T)
    gP, gT = onp.minco_dense_propagate(s, head, tail, wps.T, T, gC + eC, gTp + eT + RHO)
    return e + RHO * T.sum() + jp, gP, gT, co


def test_config2_corridor_cost_and_time_gradients_full_batch(anet_ctx):
    """configs[2]: 4096 x 8-segment min-snap, corridor + limit penalties, gradients w.r.t. waypoints and durations."""
    import allocnet_amd as aa
    B, s, c, N, M = 4096, 4, 3, 8, 16
    rng = np.random.default_rng(1)
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    pen = _penalty(aa, M)
    cost, gP, gT, co = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, want_coeffs=True, ctx=anet_ctx)
    assert np.isfinite(cost).all() and np.isfinite(gP).all() and np.isfinite(gT).all()
    active = 0
    for b in range(0, B, 512):
        c0, gP0, gT0, co0 = _oracle_cost_grad(s, head[b], tail[b], wps[b], T[b], hp[b])
        e0 = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b])[1]
        active += c0 - e0 - RHO * T[b].sum() > 1e-9 * c0
        assert np.abs(co[b] - co0).max() <= 1e-9 * np.abs(co0).max(), b
        assert abs(cost[b] - c0) <= 1e-9 * abs(c0), b
        assert np.abs(gP[b].T - gP0).max() <= 1e-7 * max(1.0, np.abs(gP0).max()), b
        assert np.abs(gT[b] - gT0).max() <= 1e-7 * max(1.0, np.abs(gT0).max()), b
    assert active >= 4          # the penalty rows really are violated in the compared sample
    # size-independent property over the whole batch: the directional derivative along a random direction of the
    # durations agrees with a central difference of the cost (one extra pair of evaluations for all 4096)
    dT = rng.normal(size=T.shape) * 1e-6
    cp = aa.minco_cost_grad(head, tail, wps, T + dT, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
    cm = aa.minco_cost_grad(head, tail, wps, T - dT, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
    dd = (gT * dT).sum(axis=1)
    assert np.abs((cp - cm) / 2 - dd).max() <= 1e-5 * np.abs(dd).max()


def test_config3_lbfgs_to_convergence_full_batch(anet_ctx):
    """configs[3]: 4096 x 16-segment min-jerk, L-BFGS with lbfgs_parameter_t defaults until every problem stops on
    its own (lbfgs.hpp:434-717); the evaluation budget is only a cap."""
    import allocnet_amd as aa
    B, s, c, N, M = 4096, 3, 3, 16, 16
    rng = np.random.default_rng(2)
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    pen = _penalty(aa, M)
    c0 = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)[0]
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(), max_evals=40000,
                         ctx=anet_ctx)
    st = out["status"]
    assert np.isin(st, (aa.lbfgs.LBFGS_STOP, aa.lbfgs.LBFGS_CONVERGENCE)).all(), np.unique(st, return_counts=True)
    assert (out["cost"] < c0).all()
    assert (out["T"] > 0).all() and np.isfinite(out["wps"]).all()
    assert (out["evals"] >= out["iters"]).all() and out["evals"].max() < 40000
    # the returned parameters reproduce the returned cost, and the returned coefficients are their MINCO solution
    c1, _, _, co1 = aa.minco_cost_grad(head, tail, out["wps"], out["T"], s, hpolys=hp, penalty=pen, want_coeffs=True,
                                       ctx=anet_ctx)
    assert np.abs(c1 - out["cost"]).max() <= 1e-9 * np.abs(c1).max()
    assert np.abs(co1 - out["coeffs"]).max() <= 1e-9 * np.abs(co1).max()
    for b in range(0, B, 1024):          # ... also according to the numpy oracle (dense adjoint)
        cb, *_ = _oracle_cost_grad(s, head[b], tail[b], out["wps"][b], out["T"][b], hp[b])
        assert abs(cb - out["cost"][b]) <= 1e-8 * abs(cb), b
    # ... and, 64 strided problems, according to the C restatement (classic banded LU with pivoting).  Its gradient at the
    # returned point, in the optimiser's variables (waypoints, tau): a problem that ended with LBFGS_CONVERGENCE meets the
    # gradient test of lbfgs.hpp:590-597 there; one that ended with LBFGS_STOP (the past / delta test, :604-620) promises
    # no gradient size, but the gradient has come down by orders of magnitude from the start's
    idx = np.arange(0, B, B // 64)
    cc, cgP, cgT = cbind.minco_cost_grad_batch(s, head[idx], tail[idx], out["wps"][idx], out["T"][idx], hp[idx], RHO,
                                               nthreads=4, **KW)
    assert np.abs(cc - out["cost"][idx]).max() <= 1e-8 * np.abs(cc).max()
    _, sgP, sgT = cbind.minco_cost_grad_batch(s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], RHO, nthreads=4, **KW)

    def ginf(gP, gT, TT):
        return np.maximum(np.abs(gP).reshape(len(idx), -1).max(axis=1), np.abs(gT * _dfwd(_bwd(TT))).max(axis=1))
    g_end, g_start = ginf(cgP, cgT, out["T"][idx]), ginf(sgP, sgT, T[idx])
    xinf = np.maximum(np.abs(out["wps"][idx]).reshape(len(idx), -1).max(axis=1), np.abs(_bwd(out["T"][idx])).max(axis=1))
    conv = st[idx] == aa.lbfgs.LBFGS_CONVERGENCE
    assert (g_end[conv] / np.maximum(1.0, xinf[conv]) < 1.01 * aa.lbfgs_parameter_t().g_epsilon).all()
    print("configs[3] |g|_inf end / start: median %.2e max %.2e (%d of %d LBFGS_CONVERGENCE)"
          % (np.median(g_end / g_start), (g_end / g_start).max(), int(conv.sum()), len(idx)))
    assert np.median(g_end / g_start) < 1e-2 and (g_end < g_start).all()
    # the spread of the optimised durations this configuration ends with (recorded: the reduced system is accurate to
    # 1e-8 up to a spread of 100; beyond 50 the returned coefficients are re-solved with pivoting)
    spread = out["T"].max(axis=1) / out["T"].min(axis=1)
    print("configs[3] final duration spread: median %.2f max %.2f, re-solved %d" % (np.median(spread), spread.max(),
                                                                                  int(out["wide_spread"].sum())))
    assert np.array_equal(out["wide_spread"], spread > 50.0)
    assert spread.max() < 100.0


@pytest.mark.parametrize("max_iterations", [2, 6])
def test_config3_lbfgs_counters_match_the_restatement(anet_ctx, max_iterations):
    """Same batch, fixed iteration budgets: (status, iterations, evaluations) of a strided sample equal the C
    restatement of lbfgs_optimize driving the numpy objective, iterates to rounding."""
    import allocnet_amd as aa
    B, s, c, N, M = 4096, 3, 3, 16, 16
    rng = np.random.default_rng(2)
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    pen = _penalty(aa, M)
    nw = 3 * (N - 1)
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen,
                         param=aa.lbfgs_parameter_t(max_iterations=max_iterations), max_evals=400, ctx=anet_ctx)
    tol = 1e-9 if max_iterations <= 2 else 1e-7
    for b in range(0, B, 512):
        def fun(x, b=b):
            w = x[:nw].reshape(N - 1, 3)
            tau = x[nw:]
            f, gP, gT, _ = _oracle_cost_grad(s, head[b], tail[b], w, _fwd(tau), hp[b])
            return f, np.r_[gP.T.reshape(-1), gT * _dfwd(tau)]
        x0 = np.r_[wps[b].reshape(-1), _bwd(T[b])]
        ret, xo, fo, it, ev = cbind.lbfgs_optimize(x0, fun, cbind.lbfgs_default_param(max_iterations=max_iterations))
        assert (out["status"][b], out["iters"][b], out["evals"][b]) == (ret, it, ev), (b, ret, it, ev)
        assert abs(out["cost"][b] - fo) <= tol * abs(fo), b
        xg = np.r_[out["wps"][b].reshape(-1), _bwd(out["T"][b])]
        assert np.abs(xg - xo).max() <= tol * max(1.0, np.abs(xo).max()), b


@pytest.mark.parametrize("max_iterations", [3, 12])
def test_config3_lbfgs_counters_match_the_c_driver_on_a_wide_sample(anet_ctx, max_iterations):
    """The same comparison on 256 strided problems with the C objective (oracle/minco_costgrad.c: classic banded LU + adjoint)
    under the C restatement of lbfgs_optimize: (status, iterations, evaluations) equal for at least 99 % of the sample --
    a different rounding of the objective may flip an Armijo / Wolfe test that sits on its threshold -- and the cost of every
    problem whose counters agree equal to 1e-7."""
    import allocnet_amd as aa
    B, s, c, N, M = 4096, 3, 3, 16, 16
    rng = np.random.default_rng(2)
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    pen = _penalty(aa, M)
    out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen,
                         param=aa.lbfgs_parameter_t(max_iterations=max_iterations), max_evals=800, ctx=anet_ctx)
    idx = np.arange(0, B, 16)
    ref = cbind.lbfgs_minco_batch(s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], RHO, nthreads=4,
                                  param=cbind.lbfgs_default_param(max_iterations=max_iterations), **KW)
    same = (out["status"][idx] == ref["status"]) & (out["iters"][idx] == ref["iters"]) & (out["evals"][idx] == ref["evals"])
    assert same.mean() >= 0.99, (same.mean(), np.nonzero(~same)[0][:10])
    rel = np.abs(out["cost"][idx] - ref["cost"]) / np.abs(ref["cost"])
    assert rel[same].max() <= 1e-7, rel[same].max()
    assert np.abs(out["T"][idx][same] - ref["T"][same]).max() <= 1e-6 * ref["T"].max()


def test_config3_converged_costs_against_the_cpu_restatement(anet_ctx):
    """configs[3] run to each problem's own stop on BOTH sides, 256 strided problems: the one-launch kernel against the C
    restatement of lbfgs_optimize on the C objective (oracle_lbfgs_minco_batch).  ~2500 iterations of a non-convex problem
    amplify the different rounding of the two objectives (reduced block-tridiagonal solve and wave reductions here, classic
    banded LU and left-to-right sums there), so the END points differ -- counters are exact at small budgets (tests above) --
    and what can be asked is statistical: the relative difference of the final costs has a median within +-2e-3, and the
    GPU does not end systematically higher (sign test: at most 62 % of the sample, 3 sigma of a fair coin at n = 256).
    The largest single difference is printed (recorded in DESIGN section 7), not bounded: a problem whose two runs part
    ways early may settle in different local minima."""
    import allocnet_amd as aa
    B, s, c, N, M = 4096, 3, 3, 16, 16
    rng = np.random.default_rng(2)
    head, tail, wps, T, hp = corridor_problem(rng, B, N, c, M)
    pen = _penalty(aa, M)
    idx = np.arange(0, B, 16)
    out = aa.lbfgs_minco(head[idx], tail[idx], wps[idx], T[idx], s, hpolys=hp[idx], penalty=pen, param=aa.lbfgs_parameter_t(),
                         max_evals=40000, ctx=anet_ctx)
    ref = cbind.lbfgs_minco_batch(s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], RHO, nthreads=8,
                                  param=cbind.lbfgs_default_param(), **KW)
    ok = (aa.lbfgs.LBFGS_STOP, aa.lbfgs.LBFGS_CONVERGENCE)
    assert np.isin(out["status"], ok).all() and np.isin(ref["status"], ok).all()
    d = (out["cost"] - ref["cost"]) / ref["cost"]
    worse = float((d > 0).mean())
    print("configs[3] converged, GPU vs CPU restatement, 256 problems: median rel diff %.2e, |max| %.2e, GPU higher in "
          "%.1f %%, evaluations GPU %.0f / CPU %.0f mean" % (np.median(d), np.abs(d).max(), 100 * worse, out["evals"].mean(),
                                                           ref["evals"].mean()))
    assert abs(np.median(d)) <= 2e-3
    assert worse <= 0.62
    # both ends are stationary to the same degree: the evaluation counts have the same scale
    assert 0.8 < out["evals"].mean() / ref["evals"].mean() < 1.25


def test_config3_divergence_of_the_two_runs_is_rounding_not_a_late_defect(anet_ctx):
    """Between the budgets where the counters are exact (2 / 6) or nearly so (3 / 12) and the ~2500 iterations of a run to
    convergence there was no evidence that the end points differ by amplified rounding and not by a defect that only shows late.
    256 strided problems of configs[3] at iteration budgets 25 / 50 / 100 / 200 / 400 on both sides (bench.lbfgs_divergence_profile,
    the figures the bench line carries) WITH A CONTROL: the C restatement against itself from a start point perturbed by 1e-13
    relative -- the size of the difference the two objectives have anyway.  Measured (round 6): identical counters for
    1.00 / 1.00 / 0.90 / 0.07 / 0.08 of the problems device-vs-restatement and 1.00 / 1.00 / 0.75 / 0.14 / 0.08 restatement-vs-
    perturbed-restatement; costs of identical-counter problems apart by at most 1e-10 / 9e-6 / 5e-3 / 1e-2 against 2e-9 / 2e-4 /
    1e-2 / 2e-2: this objective amplifies a rounding-level difference about 10^5-fold per 25 iterations, whoever computes it.
    Asserted: the device-vs-restatement profile is NO WORSE than the control's at any budget (a late defect would fall off
    faster), and up to 25 iterations -- before the amplification reaches the seventh digit -- identical counters come with
    costs equal to 1e-8."""
    import allocnet_amd as aa
    import bench
    B, s, c, N, M = 4096, 3, 3, 16, 16
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(2), B, N, c, M)
    idx = np.linspace(0, B - 1, 256).astype(int)
    prof = bench.lbfgs_divergence_profile(aa, cbind, s, head[idx], tail[idx], wps[idx], T[idx], hp[idx], _penalty(aa, M), 8,
                                          ctx=anet_ctx)
    print("configs[3] divergence profile:", prof)
    dev, ctl = prof["gpu_vs_cpu"], prof["cpu_vs_cpu_perturbed"]
    assert prof["budgets"] == [25, 50, 100, 200, 400]
    assert dev["same"][0] == 1.0 and dev["max_rel_same"][0] <= 1e-8
    for k, mi in enumerate(prof["budgets"]):
        # agreement at least the control's (binomial noise of 256 problems: three sigma ~ 0.1) ...
        assert dev["same"][k] >= ctl["same"][k] - 0.1, (mi, dev, ctl)
        # ... and the costs no further apart than the control's, over all problems (median) and among identical counters (max)
        assert dev["median_rel"][k] <= 10.0 * ctl["median_rel"][k] + 1e-12, (mi, dev, ctl)
        if dev["max_rel_same"][k] is not None and ctl["max_rel_same"][k] is not None:
            assert dev["max_rel_same"][k] <= 30.0 * ctl["max_rel_same"][k] + 1e-9, (mi, dev, ctl)
