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


def _status_worker(rank, world, port, total, ret):
    import torch
    import torch.distributed as dist
    from allocnet_amd.distributed import shard_bounds, allgather_costs_status
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        cost_all = torch.rand(total, dtype=torch.float64, generator=g) * 1e3
        status_all = torch.randint(-1030, 3, (total,), dtype=torch.int32, generator=g)      # lbfgs.hpp's return codes are negative too
        lo, hi = shard_bounds(total, world, rank)
        c, st = allgather_costs_status(cost_all[lo:hi].clone(), status_all[lo:hi].clone(), total)
        ret[rank] = bool(torch.equal(c, cost_all) and torch.equal(st, status_all) and st.dtype == torch.int32)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37, 5])       # even, ragged (odd shard length: the int32 half-slot), tiny
def test_allgather_costs_and_status_in_one_buffer_gloo_world2(total):
    """status + cost fused into ONE gather buffer per rank -- one collective, both arrays back bit for bit in
    global trajectory order on every rank."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_status_worker, args=(2, _free_port(), total, ret), nprocs=2, join=True)
    assert ret[0] is True and ret[1] is True


def _overlap_worker(rank, world, port, steps, every, ret):
    """bench.py's step(): slot j = i & 1 is overwritten by the (fake) solve of step i only after the gather issued from it at
    step i - 2 has completed; every gather must deliver the costs of ITS step from every rank."""
    import torch
    import torch.distributed as dist
    from allocnet_amd.distributed import OverlappedCostGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 257
        og = OverlappedCostGather(n, world, "cpu", alloc=n + 63, every=every)
        bad, seen = 0, 0
        pending = {}                                   # slot -> step whose gather is in flight
        for i in range(steps):
            j = og.acquire(i)
            if j in pending:                           # acquire() waited: that gather is complete and must hold step pending[j]
                st = pending.pop(j)
                for r in range(world):
                    exp = torch.arange(n, dtype=torch.float64) + 1000.0 * st + 1e6 * r
                    bad += int(not torch.equal(og.recv[j][r * n:(r + 1) * n], exp))
                seen += 1
            og.send[j][:n] = torch.arange(n, dtype=torch.float64) + 1000.0 * i + 1e6 * rank      # the "solve" of step i
            if og.submit(i) is not None:
                pending[j] = i
        og.drain()
        for j, st in pending.items():
            for r in range(world):
                exp = torch.arange(n, dtype=torch.float64) + 1000.0 * st + 1e6 * r
                bad += int(not torch.equal(og.recv[j][r * n:(r + 1) * n], exp))
            seen += 1
        if rank == 0:
            ret["bad"], ret["seen"], ret["issued"] = bad, seen, og.issued
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("every", [1, 3])
def test_overlapped_cost_gather_double_buffer_order_gloo_world2(every):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    steps = 13
    mp.spawn(_overlap_worker, args=(2, _free_port(), steps, every, ret), nprocs=2, join=True)
    assert ret["bad"] == 0
    assert ret["issued"] == steps // every and ret["seen"] == ret["issued"]


def test_overlapped_cost_gather_disabled_is_a_plain_double_buffer():
    from allocnet_amd.distributed import OverlappedCostGather
    og = OverlappedCostGather(8, 1, "cpu", enabled=False)
    assert [og.acquire(i) for i in range(4)] == [0, 1, 0, 1]
    assert og.submit(0) is None and og.issued == 0 and og.recv is None
    og.drain()


def _retime_worker(rank, world, port, ret):
    """bench.time_steps with two ranks: the runtime stall falls into rank 1's pass only -- BOTH ranks must time the steps again
    (the decision is an all-reduce), and the time reported is the maximum over the ranks of the second pass."""
    import time
    import types
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Ev:
            def __init__(self, enable_timing=True):
                self.t = None

            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3
        fake = types.SimpleNamespace(cuda=types.SimpleNamespace(Event=Ev), tensor=torch.tensor, float64=torch.float64)
        calls = {"n": 0}

        def step(i, ev):
            calls["n"] += 1
            if rank == 1 and calls["n"] == 4:
                time.sleep(0.08)
            ev[0].record()
            time.sleep(0.0005 * (1 + rank))
            ev[1].record()
        elapsed, ev, retimed = bench.time_steps(fake, dist, True, "cpu", 10, step, dist.barrier)
        ret[rank] = (elapsed, retimed, calls["n"])
    finally:
        dist.destroy_process_group()


def test_time_steps_retimes_on_every_rank_gloo_world2():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_retime_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    (e0, r0, n0), (e1, r1, n1) = ret[0], ret[1]
    assert r0 and r1 and n0 == 20 and n1 == 20          # both ranks ran the steps twice
    assert e0 == e1 and 0.009 <= e0 < 0.06              # the maximum over the ranks of the second, clean pass
