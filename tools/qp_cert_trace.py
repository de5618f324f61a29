#!/usr/bin/env python3
"""What the Farkas second opinion of k_qp_ipm sees when the window heuristic suspects a problem (library built with
-DANET_IPM_CERT_TRACE: one line per check):  gpurun -- 'bash tools/ab_build.sh "-DANET_IPM_CERT_TRACE" python tools/qp_cert_trace.py 4,8,16,4096'"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
ctx = aa.Context(0); dev = torch.device("cuda", 0)
for arg in sys.argv[1:] or ["4,8,16,4096"]:
    s, N, M, B = (int(v) for v in arg.split(","))
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    state = np.stack([head, tail], axis=1)[..., :3]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r = aa.qp_solve_dev(s, t(state), t(T * 1.5), t(hp), ctx=ctx)
    torch.cuda.synchronize()
    st = r["status"].cpu().numpy(); it = r["iters"].cpu().numpy()
    print("shape", arg, "status hist", dict(zip(*np.unique(st, return_counts=True))), "iters of -3:", np.sort(it[st == -3])[-12:],
          "iters max", it.max(), flush=True)
