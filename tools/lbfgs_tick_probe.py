#!/usr/bin/env python3
"""Which part of one L-BFGS tick costs what, as a function of batch / history length / piece count.
Run under rocprofv3 --kernel-trace and read the per-kernel means per launch shape:
    gpurun -- 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --output-format csv --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tick -o t \
               -- python $GRAFT_REPO_ROOT/tools/lbfgs_tick_probe.py; python $GRAFT_REPO_ROOT/tools/lbfgs_tick_probe.py --summarize $GRAFT_REPO_ROOT/gpurun_out/tick/t_kernel_trace.csv'
"""
import csv
import os
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(path):
    d = defaultdict(list)
    for row in csv.DictReader(open(path)):
        n = row["Kernel_Name"]
        if "anet" not in n:
            continue
        d[(n[:60], row["Grid_Size_X"], row["Grid_Size_Y"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in sorted(d.items()):
        v = np.array(v) / 1e3
        print("%-62s grid %8s x %3s  n=%5d  mean %7.1f us  p10 %7.1f  p90 %7.1f" % (k + (len(v), v.mean(), np.percentile(v, 10), np.percentile(v, 90))))


def main():
    import torch
    import allocnet_amd as aa
    from tools.bench_configs import synth, to_bm
    dev = torch.device("cuda", 0)
    ctx = aa.Context(0)
    shapes = [(4096, 16, 8), (1024, 16, 8), (256, 16, 8), (4096, 16, 1), (4096, 4, 8), (16384, 16, 8)]
    if os.environ.get("TICK_FIRST_ONLY"):  # counter passes: the config-4 shape only
        shapes = shapes[:1]
    for (B, N, mem) in shapes:
        s, c, M = 3, 3, 16
        ld = aa.recommended_ld(B)
        rng = np.random.default_rng(2)
        head, tail, wps, T, hp = synth(rng, B, N, c, M)
        pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0,
                              max_acc=6.0, res=20, poly_rows=M)
        th, tt, tw, tT, thp = (to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
        prm = aa.lbfgs_parameter_t(mem_size=mem)
        aa.lbfgs_minco_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, param=prm, max_evals=300, ctx=ctx)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        main()
