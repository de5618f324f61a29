"""Multi-GPU host logic: one process per GPU, independent trajectories sharded contiguously across
ranks, no collective inside a solve; the only exchange is the all-gather of the per-trajectory costs
(north star), issued through torch.distributed ("nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  SURVEY.md 8(e)."""


def shard_bounds(total, world, rank):
    """Contiguous split of `total` trajectories over `world` ranks, remainder to the low ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(total, world):
    return -(-int(total) // int(world))


def allgather_costs(local_costs, total, group=None, out=None):
    """All-gather the per-trajectory costs of every rank's shard into global trajectory order.
    local_costs: 1-D tensor with this rank's shard (device tensor for nccl, CPU tensor for gloo).
    Uneven shards are padded to the largest one so a single all_gather_into_tensor suffices
    (8 B per trajectory: latency-bound, one collective per step)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total, world, rank)
    if local_costs.numel() != hi - lo:
        raise ValueError(f"rank {rank} holds {local_costs.numel()} costs, expected {hi - lo}")
    m = max_shard(total, world)
    send = local_costs
    if hi - lo < m:
        send = torch.zeros(m, dtype=local_costs.dtype, device=local_costs.device)
        send[:hi - lo] = local_costs
    gathered = torch.empty(world * m, dtype=local_costs.dtype, device=local_costs.device)
    dist.all_gather_into_tensor(gathered, send.contiguous(), group=group)
    if total == world * m:
        if out is not None:
            out.copy_(gathered)
            return out
        return gathered
    res = out if out is not None else torch.empty(total, dtype=local_costs.dtype, device=local_costs.device)
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        res[a:b] = gathered[r * m:r * m + (b - a)]
    return res


def pack_cost_status(costs, status, m):
    """One send buffer of a rank for allgather_costs_status: m float64 costs followed by m int32 status words in the bytes of
    (m + 1) // 2 more float64 slots ("fuse status + cost into one buffer" -- the collective is latency-bound, so one
    all-gather of 12 B per trajectory costs what one of 8 B does and saves the second one whole)."""
    import torch
    n = costs.numel()
    buf = torch.zeros(m + (m + 1) // 2, dtype=torch.float64, device=costs.device)
    buf[:n] = costs
    buf[m:].view(torch.int32)[:n] = status.to(torch.int32)
    return buf


def allgather_costs_status(local_costs, local_status, total, group=None):
    """allgather_costs for a solver that also returns a status per trajectory (the L-BFGS: lbfgs_optimize's return code;
    the QP: OSQP's status): ONE all_gather_into_tensor of the packed buffers, unpacked into global trajectory order.
    -> (costs float64 [total], status int32 [total])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total, world, rank)
    if local_costs.numel() != hi - lo or local_status.numel() != hi - lo:
        raise ValueError(f"rank {rank} holds {local_costs.numel()} costs / {local_status.numel()} status words, expected {hi - lo}")
    m = max_shard(total, world)
    w = m + (m + 1) // 2
    send = pack_cost_status(local_costs, local_status, m)
    gathered = torch.empty(world * w, dtype=torch.float64, device=local_costs.device)
    dist.all_gather_into_tensor(gathered, send, group=group)
    costs = torch.empty(total, dtype=torch.float64, device=local_costs.device)
    status = torch.empty(total, dtype=torch.int32, device=local_costs.device)
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        costs[a:b] = gathered[r * w:r * w + (b - a)]
        status[a:b] = gathered[r * w + m:(r + 1) * w].view(torch.int32)[:b - a]
    return costs, status


class NativeComm:
    """RCCL communicator owned by an allocnet_amd Context (anet_comm_*): the all-gather of costs for
    hosts that do not run torch.distributed.  The 128-byte unique id made by rank 0 must reach the other
    ranks through any side channel."""

    def __init__(self, ctx, nranks, rank, unique_id=None):
        import ctypes
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        if unique_id is None:
            if rank != 0:
                raise ValueError("only rank 0 may create the unique id")
            buf = (ctypes.c_ubyte * 128)()
            ctx.check(ctx.lib.anet_comm_unique_id(ctx.handle, buf))
            unique_id = bytes(buf)
        self.unique_id = unique_id
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(unique_id)
        ctx.check(ctx.lib.anet_comm_init(ctx.handle, self.nranks, self.rank, buf))

    def allgather_costs(self, send, recv, count, stream=None):
        """send/recv: torch CUDA float64 tensors (count and nranks*count elements)."""
        import ctypes
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(send.device).cuda_stream
        self.ctx.check(self.ctx.lib.anet_comm_allgather_costs_dev(
            self.ctx.handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), int(count),
            ctypes.c_void_p(stream)))

    def close(self):
        self.ctx.check(self.ctx.lib.anet_comm_destroy(self.ctx.handle))


class OverlappedCostGather:
    """The all-gather of step k runs on the collective's stream while step k + 1 computes: two send / receive slots, and a slot
    is handed back to the solve only after the collective that last read it has completed (`work.wait()` orders the current
    stream behind it; with gloo it blocks the host).  bench.py's `step()` and the world-size-2 gloo test drive this very class.

        j = g.acquire(i)          # slot of step i: waits for the gather issued from it two steps ago
        ... write this step's costs into g.send[j][:count] ...
        g.submit(i)               # issues the gather of slot j (only every `every`-th step)
        g.drain()                 # waits for what is in flight

    `every` > 1 skips the collective on the other steps (bench.py --allgather-every k: lets a lost scaling factor be attributed)."""

    def __init__(self, count, world, device, dtype=None, alloc=None, every=1, group=None, enabled=True):
        import torch
        dtype = dtype or torch.float64
        self.count, self.world, self.every, self.group, self.enabled = int(count), int(world), max(1, int(every)), group, enabled
        self.send = [torch.zeros(int(alloc or count), device=device, dtype=dtype) for _ in range(2)]
        self.recv = [torch.empty(self.world * self.count, device=device, dtype=dtype) for _ in range(2)] if enabled else None
        self.works = [None, None]
        self.issued = 0
        self.last = None            # slot of the last gather issued

    def acquire(self, i):
        j = i & 1
        if self.works[j] is not None:
            self.works[j].wait()
            self.works[j] = None
        return j

    def submit(self, i):
        if not self.enabled or (i + 1) % self.every:
            return None
        import torch.distributed as dist
        j = i & 1
        self.works[j] = dist.all_gather_into_tensor(self.recv[j], self.send[j][:self.count], group=self.group, async_op=True)
        self.issued += 1
        self.last = j
        return self.works[j]

    def drain(self):
        for j in range(2):
            if self.works[j] is not None:
                self.works[j].wait()
                self.works[j] = None
