#!/usr/bin/env python3
"""Within-process interleaved A/B of k_minco_solve between two builds of the library.
    python tools/ab_solve.py libA.so libB.so [--batch 1048576] [--pieces 8] [--order 4] [--bc 3]
Prints median / min kernel ms of each and the ratio (DVFS makes single runs differ by +-3%)."""
import argparse, ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch_minor


def load(path):
    L = ctypes.CDLL(path)
    L.anet_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.anet_minco_solve_dev.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int64] + [ctypes.c_void_p] * 7
    h = ctypes.c_void_p()
    assert L.anet_create(0, ctypes.byref(h)) == 0
    return L, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--pieces", type=int, default=8)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--bc", type=int, default=3)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--pad", type=int, nargs="*", default=[0], help="extra row stride (ld = batch + pad), several = sweep")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, N, s, c = a.batch, a.pieces, a.order, a.bc
    for pad in a.pad:
        run(a, dev, B, N, s, c, (B + 63) // 64 * 64 + pad)


def run(a, dev, B, N, s, c, ld):
    print("ld =", ld, "(batch + %d)" % (ld - B))
    head, tail, wps, T = synth_batch_minor(torch, B, ld, N, c, 0, dev)
    co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64)
    en = torch.empty(ld, device=dev, dtype=torch.float64)
    libs = [load(os.path.abspath(p)) for p in a.libs]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    times = [[] for _ in libs]
    for rep in range(a.reps + 5):
        for i, (L, h) in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.anet_minco_solve_dev(h, s, c, N, B, ld, p(head), p(tail), p(wps), p(T), p(co), p(en), st)
            e1.record()
            assert rc == 0
            torch.cuda.synchronize()
            if rep >= 5:
                times[i].append(e0.elapsed_time(e1))
    abytes = 8 * (2 * 3 * c + N + 3 * (N - 1) + 3 * 2 * s * N + 1)
    for pth, t in zip(a.libs, times):
        med = statistics.median(t)
        print(f"{os.path.basename(pth):40s} median {med:.4f} ms  min {min(t):.4f} ms  "
              f"{B * abytes / med / 1e6:.0f} GB/s ({B * abytes / med / 1e6 / 80:.1f}% of 8 TB/s)")
    if len(times) == 2:
        print("ratio B/A (median):", statistics.median(times[1]) / statistics.median(times[0]))


if __name__ == "__main__":
    main()
