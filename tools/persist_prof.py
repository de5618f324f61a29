#!/usr/bin/env python3
"""Per-phase cycle counts of the one-launch MINCO L-BFGS kernel (lbfgs_minco_persistent.h), problem 0 of a batch.
Needs a library built with the counters compiled in:
    ANET_BUILD_FLAGS=-DANET_PERSIST_PROF python -m allocnet_amd.build --force
    gpurun -- 'python tools/persist_prof.py 2> gpurun_out/persist_prof.txt'
(rebuild without the flag afterwards).  B = 1: latency of a lone wave; B = 4096: with its SIMD shared."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allocnet_amd.synth import corridor_problem  # noqa: E402


def main():
    import allocnet_amd as aa
    ctx = aa.Context(0)
    for (s, N) in ((3, 16), (4, 8)):
        for B in (1, 4096):
            rng = np.random.default_rng(2)
            head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, 16)
            pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0,
                                  max_acc=6.0, res=20, poly_rows=16)
            sys.stderr.write(f"--- s={s} N={N} B={B}\n")
            sys.stderr.flush()
            import time
            for rep in range(2):
                t0 = time.perf_counter()
                out = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp, penalty=pen, param=aa.lbfgs_parameter_t(),
                                     max_evals=40000, want_coeffs=False, ctx=ctx)
                dt = time.perf_counter() - t0
            ev = out["evals"]
            sys.stderr.write(f"evals of problem 0: {ev[0]}, status {out['status'][0]}; host call {dt * 1e3:.2f} ms, "
                             f"max evals {ev.max()}, mean {ev.mean():.0f} -> {dt * 1e6 / ev.max():.2f} us per evaluation of "
                             f"the slowest problem\n")


if __name__ == "__main__":
    main()
