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
        return gathered
    res = out if out is not None else torch.empty(total, dtype=local_costs.dtype, device=local_costs.device)
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        res[a:b] = gathered[r * m:r * m + (b - a)]
    return res
