"""World-size-2 gloo test of the shard / all-gather host logic.  The per-rank solve
is done by the CPU oracle here (no GPU in this container); on the GPU box bench.py runs the same
partition + collective with the HIP solve and the nccl (RCCL) backend."""
import os
import socket

import numpy as np
import pytest

from allocnet_amd.distributed import shard_bounds, max_shard


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 64, 1000, 32768, 4097):
        for world in (1, 2, 3, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = shard_bounds(total, world, r)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == total
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(total, world) or total == 0
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, ret):
    import torch
    import torch.distributed as dist
    from oracle import cbind
    from tests.util import random_problem
    from allocnet_amd.distributed import shard_bounds, allgather_costs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)                 # every rank generates the same global problem set
        head, tail, wps, T = random_problem(rng, total, 8, 3)
        lo, hi = shard_bounds(total, world, rank)
        _, e_local = cbind.minco_solve_batch(4, head[lo:hi], tail[lo:hi], wps[lo:hi], T[lo:hi], want_coeffs=False)
        gathered = allgather_costs(torch.from_numpy(e_local), total)
        if rank == 0:
            _, e_all = cbind.minco_solve_batch(4, head, tail, wps, T, want_coeffs=False)
            ret["err"] = float(np.abs(gathered.numpy() - e_all).max())
            ret["n"] = int(gathered.numel())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])       # even and ragged shards
def test_allgather_costs_gloo_world2(total):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, ret), nprocs=2, join=True)
    assert ret["n"] == total
    assert ret["err"] == 0.0
