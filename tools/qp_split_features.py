#!/usr/bin/env python3
"""What a problem's state after K Newton steps says about the steps it still needs: runs the two-launch interior point
(ANET_IPM_SPLIT_STEPS=K from the environment), reads the parked scalars back from the workspace and stores them with the final
step counts.   gpurun -- 'ANET_IPM_SPLIT_STEPS=6 python tools/qp_split_features.py'  ->  gpurun_out/qp_split_features_K.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem

K = int(os.environ.get("ANET_IPM_SPLIT_STEPS", "6"))
ctx = aa.Context(0)
dev = torch.device("cuda:0")
out = {}
for key, s, N, B in (("snap8", 4, 8, 4096), ("jerk5", 3, 5, 4096), ("snap5", 4, 5, 4096)):
    M, res = 16, 20
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, 3, M)
    state = np.ascontiguousarray(np.stack([head, tail], axis=1)[..., :3])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r = aa.qp_solve_dev(s, t(state), t(T * 1.5), t(hp), ctx=ctx)
    torch.cuda.synchronize()
    ny = 3 * s * (N + 1)
    m_adm = 3 * (6 + s * (N - 1)) + N * res * (M + 12)
    w = r["_work"].cpu().numpy()
    cont = w[2 * m_adm * B + 2 * B: 2 * m_adm * B + 2 * B + (ny + 16) * B].reshape(B, ny + 16)[:, ny:]
    out[key + "_feat"] = cont
    out[key + "_iters"] = r["iters"].cpu().numpy()
    out[key + "_status"] = r["status"].cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/qp_split_features_{K}.npz", **out)
print("saved", K)
