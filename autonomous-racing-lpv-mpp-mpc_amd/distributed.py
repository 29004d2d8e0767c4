"""Sharding helpers for the embarrassingly parallel batch.

Instances are independent cold-start QPs, so a batch is split contiguously over ranks and no collective is
needed on the data path; ``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) only
carries the barrier and a few scalars of statistics."""
from __future__ import annotations


def shard_range(total, rank, world):
    """Contiguous [start, stop) of ``total`` instances owned by ``rank`` (sizes differ by at most one)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_stats(elapsed_s, sums, device=None, group=None):
    """MAX of the elapsed time and SUM of a list of counters over all ranks; returns (max_elapsed, summed list).
    With a single process (no initialised process group) the inputs are returned unchanged."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), [float(v) for v in sums]
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    s = torch.tensor([float(v) for v in sums], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), [float(v) for v in s.tolist()]
