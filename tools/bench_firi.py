#!/usr/bin/env python3
"""Timing of the batched FIRI: B corridors x Np obstacle points, the reference's
defaults (4 iterations).  Prints one JSON object.
    gpurun -- 'python tools/bench_firi.py > gpurun_out/firi.json'
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import allocnet_amd as aa
    from allocnet_amd.synth import firi_scene as make_case, firi_pack as pack
    ctx = aa.Context(0)
    out = {}
    for B, Np in ((5, 1000), (256, 1000), (2048, 500)):
        rng = np.random.default_rng(7)
        cases = [make_case(rng, Np) for _ in range(B)]
        bd, pc, npts, a, b = pack(cases)
        aa.firi(bd[:2], pc[:2], a[:2], b[:2], n_points=npts[:2], max_rows=96, ctx=ctx)       # warm-up
        t0 = time.perf_counter()
        res = aa.firi(bd, pc, a, b, n_points=npts, max_rows=96, ctx=ctx)
        dt = time.perf_counter() - t0
        out[f"firi_B{B}_Np{Np}"] = {"seconds": dt, "corridors_per_s": B / dt, "ok_frac": float((res["ok"] >= 1).mean()),
                                     "rows_mean": float(res["n_rows"].mean()), "rows_max": int(res["n_rows"].max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
