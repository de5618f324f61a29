#!/usr/bin/env python3
"""One-off stalls in back-to-back loops of small kernels: BASELINE configs[2] (4096 x 8-segment snap cost + gradient, three
launches of ~11 / 37 / 15 us) evaluated in repetitions of 50, each repetition between its own HIP events, with the time since
the process started -- what a single event pair around one 200-evaluation loop cannot tell from kernel time.
    gpurun -- 'python tools/stall_probe.py; python tools/stall_probe.py'"""
import os, sys, time
T0 = time.perf_counter()
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from tools.bench_configs import synth, to_bm
dev = torch.device("cuda", 0); ctx = aa.Context(0)
s, c, N, M = (int(sys.argv[3]) if len(sys.argv) > 3 else 4), 3, 8, 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SOLVE_ONLY = len(sys.argv) > 2 and sys.argv[2] == "solve"     # one launch per call instead of three
ld = aa.recommended_ld(B)
head, tail, wps, T, hp = synth(np.random.default_rng(1), B, N, c, M)
pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=M)
th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
co = torch.empty(N * 3 * 2 * s, ld, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
t_first = time.perf_counter() - T0
K, R = 50, 80
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(R)]
marks = []
for r in range(R):
    marks.append(time.perf_counter() - T0)
    evs[r][0].record()
    for _ in range(K):
        if SOLVE_ONLY:
            aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=co, energy=cost, ctx=ctx)
        else:
            aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)
    evs[r][1].record()
torch.cuda.synchronize()
us = [a.elapsed_time(b) / K * 1e3 for a, b in evs]
med = sorted(us)[len(us) // 2]
print("B", B, "launches per call", 1 if SOLVE_ONLY else 3, "| first GPU work %.2f s after process start; %d repetitions of %d evaluations: median %.1f us, min %.1f, max %.1f" %
      (t_first, R, K, med, min(us), max(us)))
slow = [(r, marks[r], us[r]) for r in range(R) if us[r] > 1.3 * med]
print("repetitions slower than 1.3 x median (index, host time of enqueue since process start [s], us per evaluation):",
      [(r, round(m, 3), round(u, 1)) for r, m, u in slow])
print("all:", " ".join("%.0f" % u for u in us))
