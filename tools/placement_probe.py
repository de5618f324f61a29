#!/usr/bin/env python3
"""Does the throughput of the headline solve depend on WHICH output buffer it writes (same process, same
inputs, same kernel)?  Allocates several 2 GB coefficient buffers and times the kernel on each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import allocnet_amd as aa
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    B, N, s, c = 1 << 20, 8, 4, 3
    ld = aa.recommended_ld(B)
    g = torch.Generator(device=dev); g.manual_seed(0)
    head = torch.randn(3 * c, ld, device=dev, dtype=torch.float64, generator=g)
    tail = torch.randn(3 * c, ld, device=dev, dtype=torch.float64, generator=g)
    wps = torch.randn(3 * (N - 1), ld, device=dev, dtype=torch.float64, generator=g)
    T = torch.rand(N, ld, device=dev, dtype=torch.float64, generator=g) * 1.5 + 0.5
    bufs = []
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64)
        en = torch.empty(ld, device=dev, dtype=torch.float64)
        bufs.append((co, en))
    res = []
    for rep in range(2):
        for k, (co, en) in enumerate(bufs):
            for _ in range(3):
                aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            st = torch.cuda.ExternalStream(ctx.stream_handle) if hasattr(ctx, "stream_handle") else None
            import time
            t0 = time.perf_counter()
            for _ in range(20):
                aa.minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=co, energy=en, ctx=ctx)
            ctx.synchronize() if hasattr(ctx, "synchronize") else torch.cuda.synchronize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            res.append((rep, k, co.data_ptr(), dt * 1e3, B * 1920 / dt / 8e12))
    for r in res:
        print("rep %d buf %d ptr %#x  %.4f ms  frac %.3f" % r)


if __name__ == "__main__":
    main()
