#!/usr/bin/env python3
"""Randomised soak of the interior-point QP kernel against the C port of the structured interior point (oracle/qp_ipm_port.c,
itself pinned to the dense oracle on the reference-assembled fixtures): orders, piece counts, corridor rows, samples per piece,
duration scales from infeasibly short to slack.   gpurun -- 'python tests/soak/soak_qp.py 60'"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import allocnet_amd as aa
from oracle import cbind
from allocnet_amd.synth import corridor_problem


def run(n_cases, seed=777, ctx=None, verbose=True, batches=(1, 7, 64)):
    """Returns (problems solved by both, worst relative objective difference, solved by the port only, solved by the GPU only,
    problems).  (The port gives up on some badly scaled problems -- optimal cost 1e7 and more, durations close to infeasibly
    short -- that the kernel and the dense oracle both solve: tests/soak/qp_disagree.py prints them with the dense verdict.)"""
    ctx = ctx or aa.Context(0)
    rng = np.random.default_rng(seed)
    worst, compared, port_only, gpu_only, total = 0.0, 0, 0, 0, 0
    for case in range(n_cases):
        s = int(rng.choice([3, 4]))
        N = int(rng.choice([1, 2, 3, 5, 8, 11, 16] if s == 3 else [1, 2, 3, 5, 8]))
        M = int(rng.choice([6, 8, 12, 16]))
        res = int(rng.choice([3, 8, 20]))
        B = int(rng.choice(list(batches)))
        head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, M)
        T = T * float(rng.choice([0.3, 0.7, 1.5, 4.0]))
        vmax, amax = float(rng.uniform(2.0, 6.0)), float(rng.uniform(3.0, 9.0))
        g = aa.qp_solve(s, head, tail, hp, T, res=res, max_vel=vmax, max_acc=amax, ctx=ctx)
        state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
        p = cbind.qp_ipm_batch(s, state, T, hp, res=res, vmax=vmax, amax=amax, tol=1e-9, want_coeffs=False, nthreads=4)
        gs, ps = g["status"] == 1, p["status"] >= 1
        both = gs & (p["status"] == 1)  # (status 2: the port stalled above its tolerance -- solved, but to 1e-7 only)
        total += B
        port_only += int((ps & ~gs).sum())
        gpu_only += int((gs & ~ps).sum())
        if both.any():
            rel = np.abs(g["obj"][both] - p["obj"][both]) / np.maximum(1.0, np.abs(p["obj"][both]))
            worst = max(worst, float(rel.max()))
            compared += int(both.sum())
            assert rel.max() <= 2e-5, f"case {case}: s={s} N={N} M={M} res={res} B={B}: objective rel diff {rel.max():.2e}"
        if verbose and case % 10 == 0:
            print(f"case {case}: s={s} N={N} M={M} res={res} B={B} solved gpu {gs.sum()} port {ps.sum()}", flush=True)
    return compared, worst, port_only, gpu_only, total


if __name__ == "__main__":
    t0 = time.time()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batches = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 7, 64)   # e.g. 1600: the two-launch form
    c, w, po, go, t = run(n, batches=batches)
    print(f"{n} cases, {t} problems in {time.time() - t0:.0f} s: {c} solved by both, worst relative objective difference {w:.2e}, "
          f"{po} solved by the port only, {go} by the kernel only")
