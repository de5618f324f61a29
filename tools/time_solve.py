#!/usr/bin/env python3
"""Device time of the coefficient solve for arbitrary shapes (HIP events, 50 launches):
    gpurun -- 'python tools/time_solve.py 4,3,5,262144 3,3,12,262144'      # order,boundary count,pieces,batch"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import allocnet_amd as aa
    from allocnet_amd.synth import random_problem
    from tools.bench_configs import to_bm
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    for arg in sys.argv[1:]:
        s, c, N, B = (int(v) for v in arg.split(","))
        ld = aa.recommended_ld(B)
        rng = np.random.default_rng(0)
        n0 = min(B, 4096)
        head, tail, wps, T = random_problem(rng, n0, N, c)
        rep = (B + n0 - 1) // n0
        th, tt, tw, tT = (to_bm(torch, np.tile(x, (rep,) + (1,) * (x.ndim - 1))[:B], B, ld, dev) for x in (head, tail, wps, T))
        co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64)
        en = torch.empty(ld, device=dev, dtype=torch.float64)
        for _ in range(5):
            aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        nbytes = 8 * (2 * 3 * c + N + 3 * (N - 1) + 3 * 2 * s * N + 1)
        print("s %d c %d N %d B %d: %.4f ms  %.3g traj/s  %.0f GB/s algorithmic" % (s, c, N, B, ms, B / ms * 1e3, B * nbytes / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
