"""CPU, 2 processes, gloo: the N > 1 path of bench.py -- contiguous sharding without a data-path collective,
MAX-over-ranks timing and SUM of the per-rank counters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpvmpc.distributed import shard_range, reduce_stats
    from lpvmpc import workloads
    total = 50
    w = workloads.controller_batch(total, 20, seed=7)         # every rank builds the same global batch ...
    a, b = shard_range(total, rank, world)                    # ... and owns a contiguous slice of it
    local_iters = float((a + b) * (b - a))                    # stand-in counter that depends on the slice
    dist.barrier()
    elapsed, sums = reduce_stats(0.1 * (rank + 1), [local_iters, float(b - a)])
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([a, b, elapsed, sums[0], sums[1], w["x0"][a:b].sum()]))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_and_reduction(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("r%d.npy" % i)) for i in range(world)]
    assert (r[0][0], r[0][1], r[1][0], r[1][1]) == (0, 25, 25, 50)
    for x in r:
        assert x[2] == pytest.approx(0.2)                     # MAX over ranks of the elapsed time
        assert x[3] == pytest.approx(25 * 25 + 75 * 25)       # SUM of the per-rank counters
        assert x[4] == 50
    from lpvmpc import workloads
    full = workloads.controller_batch(50, 20, seed=7)["x0"]
    assert r[0][5] + r[1][5] == pytest.approx(full.sum())     # the two slices tile the batch
