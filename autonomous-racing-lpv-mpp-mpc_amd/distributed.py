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


def gather_per_rank(values, device=None, group=None):
    """Every rank's list of scalars, on every rank, in rank order: [[v0, v1, ...] of rank 0, ... of rank 1, ...] (one small
    all-gather after the timed region: lets the printed line tell per-rank straggler luck -- region time, largest iteration
    count -- from a real scaling loss).  Single process: [values]."""
    import torch
    import torch.distributed as dist
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()):
        return [vals]
    world = dist.get_world_size(group)
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = torch.empty((world, len(vals)), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group) if t.is_cuda else dist.all_gather(list(out.unbind(0)), t, group=group)
    return [[float(v) for v in row] for row in out.cpu().tolist()]


def gather_results(u0, status, iters, total, device=None, group=None):
    """The one collective of the path: every rank contributes the first input ``u0`` [n, 2], the
    status and the iteration count of its contiguous shard; every rank gets the global arrays back in instance order
    (``total`` rows).  One ``all_gather`` of n_max x 4 float64 words per rank (shards differ by at most one row: the short
    ones are padded, the padding is dropped).  Single process: the inputs are returned as numpy arrays."""
    import numpy as np
    import torch
    import torch.distributed as dist
    u0 = np.asarray(u0, dtype=np.float64).reshape(-1, 2)
    status = np.asarray(status).reshape(-1); iters = np.asarray(iters).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()):
        return u0, status.astype(np.int32), iters.astype(np.int32)
    world = dist.get_world_size(group)
    n_max = -(-int(total) // world)
    pack = torch.zeros((n_max, 4), dtype=torch.float64)
    n = u0.shape[0]
    pack[:n, :2] = torch.from_numpy(u0); pack[:n, 2] = torch.from_numpy(status.astype(np.float64)); pack[:n, 3] = torch.from_numpy(iters.astype(np.float64))
    if device is not None:
        pack = pack.to(device)
    out = torch.empty((world, n_max, 4), dtype=torch.float64, device=pack.device)
    dist.all_gather_into_tensor(out, pack, group=group) if pack.is_cuda else dist.all_gather(list(out.unbind(0)), pack, group=group)
    out = out.cpu().numpy()
    rows = [out[r, : shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)]
    g = np.concatenate(rows, axis=0)
    return g[:, :2].copy(), g[:, 2].astype(np.int32), g[:, 3].astype(np.int32)
