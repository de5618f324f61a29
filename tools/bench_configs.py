#!/usr/bin/env python3
"""Secondary timings for BASELINE.json configs 3 and 4 (parity-test configurations, not the bench
line): cost+gradient evaluation with corridor/limit penalties, and full L-BFGS to convergence.
    gpurun -- 'python tools/bench_configs.py > gpurun_out/configs.json'
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def to_bm(torch, a, B, ld, device):
    """(B, ...) host array -> batch-minor (F, ld) device tensor."""
    f = a.reshape(B, -1)
    t = torch.zeros(f.shape[1], ld, device=device, dtype=torch.float64)
    t[:, :B] = torch.from_numpy(np.ascontiguousarray(f.T)).to(device)
    return t


from allocnet_amd.synth import corridor_problem as synth  # noqa: E402  (SURVEY 8(d) config 3/5 generator)


def main():
    import torch
    import allocnet_amd as aa
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    out = {}
    only4 = "--only-config4" in sys.argv        # (under rocprofv3: keep the kernel statistics to that run)
    lockstep = aa.lbfgs.OPT_LOCKSTEP if "--lockstep" in sys.argv else 0
    # ---- config 3: B=4096 x 8-seg min-snap, corridor penalties + time gradients -------------------
    for B in (() if only4 else (4096, 1 << 17)):
        s, c, N, M = 4, 3, 8, 16
        ld = aa.recommended_ld(B) if os.environ.get("ANET_CFG_LD", "rec") == "rec" else (B + 63) // 64 * 64
        rng = np.random.default_rng(1)
        head, tail, wps, T, hp = synth(rng, B, N, c, M)
        pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0,
                              max_acc=6.0, res=20, poly_rows=M)
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
        cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
        torch.cuda.synchronize()
        K = 50 if B <= 4096 else 10
        t0 = time.perf_counter()
        for _ in range(K):
            aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost,
                                   gradP=gP, gradT=gT, ctx=ctx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        out[f"config3_cost_grad_B{B}"] = {"ms_per_eval": dt * 1e3, "evals_per_s": B / dt,
                                           "active_penalty_frac": float((cost[:B].cpu().numpy() > 0).mean())}
    # ---- config 4: B=4096 x 16-seg min-jerk, full L-BFGS to convergence ---------------------------
    B, s, c, N, M = 4096, 3, 3, 16, 16
    ld = aa.recommended_ld(B) if os.environ.get("ANET_CFG_LD", "rec") == "rec" else B
    rng = np.random.default_rng(2)
    head, tail, wps, T, hp = synth(rng, B, N, c, M)
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0,
                          res=20, poly_rows=M)
    prm = aa.lbfgs_parameter_t()            # lbfgs.hpp defaults
    th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    c0 = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)[0][:B].cpu().numpy()
    # warm-up (module load, first-launch costs) on a throw-away copy
    aa.lbfgs_minco_dev(*(to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T)), s, c, N, B, hpolys=thp, penalty=pen,
                       param=prm, max_evals=50, opt=3 | lockstep, ctx=ctx)
    # the evaluation budget is a cap, not the stop: every problem must end with an L-BFGS status of its own
    # (LBFGS_STOP = 1 here); 2147483647 in the histogram = still running when the budget ran out
    for cap in (30000, 3000):
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=cap,
                                 opt=3 | lockstep, ctx=ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if cap == 30000:
            res_full = res
        st = res["status"].cpu().numpy(); it = res["iters"].cpu().numpy(); ev = res["evals"].cpu().numpy()
        cf = res["cost"].cpu().numpy()
        hist = {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}
        key = "config4_lbfgs_B4096_N16_jerk" if cap == 30000 else "config4_capped_at_3000_evaluations"
        out[key] = {
            "seconds": dt, "trajectories_per_s": B / dt, "max_evals": cap, "iters_mean": float(it.mean()),
            "iters_max": int(it.max()), "evals_mean": float(ev.mean()),
            "evals_p50_p90_p99": [float(v) for v in np.percentile(ev, [50, 90, 99])], "evals_max": int(ev.max()),
            "ms_per_evaluation_step": dt * 1e3 / max(1, int(ev.max())), "status_hist": hist,
            "cost_initial_mean": float(c0.mean()), "cost_final_mean": float(cf.mean()),
            "shape": "launch per evaluation (lockstep)" if lockstep else "one launch, one wave per problem",
            "lbfgs_params": "lbfgs_parameter_t defaults (mem 8, g_eps 1e-5, past 3, delta 1e-6)"}
    # the same batch with the durations FIXED (AllocNet's own setting: the network allocates the time, the optimiser does the
    # spatial part): waypoints only, the system factorised once per problem
    for rep in range(2):
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=30000,
                               opt=aa.lbfgs.OPT_WAYPOINTS | lockstep, ctx=ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ev = r["evals"].cpu().numpy(); st = r["status"].cpu().numpy()
    out["config4_waypoints_only_fixed_durations"] = {
        "seconds": dt, "trajectories_per_s": B / dt, "evals_mean": float(ev.mean()), "evals_max": int(ev.max()),
        "status_hist": {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
        "cost_final_mean": float(r["cost"].cpu().numpy().mean())}
    if not lockstep:
        # the re-solve case: the same batch again with last call's evaluation counts as the launch order
        # (anet_lbfgs_minco_ordered_dev), and with the counts of a perturbed copy of the batch (~1 cm, 1 %)
        full = out["config4_lbfgs_B4096_N16_jerk"]
        rng2 = np.random.default_rng(99)
        wp2 = wps + 0.01 * rng2.standard_normal(wps.shape); T2 = T * (1.0 + 0.01 * rng2.uniform(-1, 1, size=T.shape))
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wp2, T2, hp))
        evp = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=30000,
                                 opt=3, ctx=ctx)["evals"]
        for key, counts in (("same_batch_counts", res_full["evals"]), ("perturbed_batch_counts", evp)):
            order = aa.launch_order_from_counts(counts)
            th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=30000,
                                   opt=3, launch_order=order, ctx=ctx)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            full[f"seconds_with_launch_order_from_{key}"] = dt
            full[f"identical_results_{key}"] = bool(torch.equal(r["evals"], res_full["evals"]) and
                                                    torch.equal(r["cost"], res_full["cost"]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
