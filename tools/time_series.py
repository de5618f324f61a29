#!/usr/bin/env python3
"""Kernel-time series of k_minco_solve over several seconds (DVFS / power-state drift)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, time
import allocnet_amd as aa
from bench import synth_batch_minor
dev = torch.device("cuda", 0); ctx = aa.Context(0)
B, N, s, c = 1 << 20, 8, 4, 3
for pad in [int(x) for x in sys.argv[1:]] or [0]:
    ld = B + pad
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, 0, dev)
    co = torch.empty(N * 3 * 8, ld, device=dev, dtype=torch.float64); en = torch.empty(ld, device=dev, dtype=torch.float64)
    out = []
    t_start = time.perf_counter()
    for blk in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
        e1.record(); torch.cuda.synchronize()
        out.append(B * 1920 * 50 / e0.elapsed_time(e1) / 1e6 / 80.0)
    print(f"pad {pad}: total {time.perf_counter()-t_start:.1f}s  " + " ".join(f"{x:.0f}" for x in out))
