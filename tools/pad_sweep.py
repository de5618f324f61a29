#!/usr/bin/env python3
"""Row-stride (ld) padding sweep for k_minco_solve: fresh allocations per trial so that physical
placement is re-rolled; reports the spread, not a single lucky number."""
import ctypes, os, sys, statistics, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import allocnet_amd as aa
from bench import synth_batch_minor

ap = argparse.ArgumentParser()
ap.add_argument("--pads", type=int, nargs="*", default=[0, 64, 192, 576, 1088, 2112, 4160, 8256, 10240, 16448, 32768, 32832, 65600, 131136])
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--trials", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = aa.Context(0)
B, N, s, c = a.batch, 8, 4, 3
abytes = 1920
keep = []
for pad in a.pads:
    res = []
    for tr in range(a.trials):
        keep.append(torch.empty((tr + 1) * 123457 + pad, device=dev, dtype=torch.float64))   # perturb placement
        ld = (B + 63) // 64 * 64 + pad
        head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, 0, dev)
        co = torch.empty(N * 3 * 8, ld, device=dev, dtype=torch.float64)
        en = torch.empty(ld, device=dev, dtype=torch.float64)
        ts = []
        for rep in range(14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
            e1.record(); torch.cuda.synchronize()
            if rep >= 4:
                ts.append(e0.elapsed_time(e1))
        res.append(B * abytes / statistics.median(ts) / 1e6 / 80.0)
        del head, tail, wps, T, co, en
    print(f"pad {pad:7d}: " + " ".join(f"{r:5.1f}" for r in res) + f"   median {statistics.median(res):5.1f}%  min {min(res):5.1f}%")
