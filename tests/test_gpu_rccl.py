"""GPU: the collectives of the multi-GPU bench path on the RCCL backend (torch.distributed "nccl"), with the one rank a one-GPU box
has: process-group creation on the device, the barrier, the max / sum all-reduce of the timing statistics and the all-gather of
(u0, status, iters) that closes a run.  The N > 1 logic (sharding, padding, ordering) is covered on CPU with
gloo (tests/test_multi_rank.py); this makes sure the same helpers run on the device backend the driver's scaling run uses."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_collectives():
    import torch
    import torch.distributed as dist
    from lpvmpc.distributed import gather_results, reduce_stats, shard_range
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        dist.barrier()
        t, sums = reduce_stats(0.125, [3.0, 4.5], device=dev)
        assert t == 0.125 and sums == [3.0, 4.5]
        rng = np.random.default_rng(0)
        u0 = rng.normal(size=(37, 2)); st = rng.integers(-3, 3, 37).astype(np.int32); it = rng.integers(25, 4000, 37).astype(np.int32)
        g_u0, g_st, g_it = gather_results(u0, st, it, 37, device=dev)
        assert np.array_equal(g_u0, u0) and np.array_equal(g_st, st) and np.array_equal(g_it, it)
        assert shard_range(37, 0, 1) == (0, 37)
    finally:
        dist.destroy_process_group()
