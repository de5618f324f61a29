#!/usr/bin/env python3
"""Step time of anet_minco_solve_dev versus batch size (launch + kernel, stream time per step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import allocnet_amd as aa
from bench import synth_batch_minor
dev = torch.device("cuda", 0); ctx = aa.Context(0)
N, s, c = 8, 4, 3
for B in [64, 256, 1024, 4096, 8192, 16384, 16385, 32768, 65536, 262144]:
    ld = aa.recommended_ld(B)
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, 0, dev)
    co = torch.empty(N * 3 * 8, ld, device=dev, dtype=torch.float64); en = torch.empty(ld, device=dev, dtype=torch.float64)
    for _ in range(20):
        aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
    torch.cuda.synchronize()
    K = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    print(f"B={B:7d}  {us:8.2f} us/step  {B / us:8.1f} M traj/s")
