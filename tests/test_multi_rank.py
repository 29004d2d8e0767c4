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


def _bench_line(cmd, env=None):
    import json, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_launcher_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` (no torchrun): bench.py's own launcher starts the ranks; --dry-run swaps the device work
    for stand-in results and runs the real sharding + barrier + reductions + the final all-gather over gloo."""
    import sys
    for wl, total in (("cfg2", 2048), ("cfg4", 65536), ("cfg5", 8192)):
        out = _bench_line([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--workload", wl])
        assert out["n_gpus"] == 2 and out["dry_run"] is True
        assert out["config"]["global_instances"] == total and out["config"]["gather_ok"] is True
        assert out["config"]["shard_rank0"] == [0, total // 2]
        assert out["config"]["max_elapsed_s"] == pytest.approx(2e-3)


@pytest.mark.timeout(600)
def test_bench_launcher_at_the_target_size_of_eight_ranks():
    """north_star's split is over the 8 GPUs of a node: the launcher path (`python bench.py --gpus 8`: a subprocess.Popen parent, never an
    exec of a process that touched a GPU), shard_range and the closing all-gather at WORLD 8 on gloo -- configs[1] (8 x 1024), configs[3]
    (65 536 = 8 x 8192) and configs[4] (8192 vehicles = 8 x 1024): the eight shards are contiguous, equal and tile the global batch."""
    import sys
    for wl, total in (("cfg2", 8 * 1024), ("cfg4", 65536), ("cfg5", 8192)):
        out = _bench_line([sys.executable, "bench.py", "--gpus", "8", "--dry-run", "--workload", wl])
        c = out["config"]
        assert out["n_gpus"] == 8 and out["dry_run"] is True and c["rccl_world"] == 8 and c["gather_ok"] is True
        assert c["global_instances"] == total
        assert c["shards"] == [[r * total // 8, (r + 1) * total // 8] for r in range(8)]
        # round 6: the line carries every rank's own region time and largest iteration count (rank order), so that a scaling curve can
        # tell one rank's 4000-iteration straggler from a scaling loss (stand-in values here: region r + 1 ms, iterations 25 .. 75)
        assert c["per_rank_region_ms"] == [float(r + 1) for r in range(8)] and len(c["per_rank_max_iters"]) == 8 and set(c["per_rank_max_iters"]) <= {25, 50, 75}
        assert c["max_elapsed_s"] == pytest.approx(8e-3)          # MAX over the eight ranks


def test_shard_range_tiles_every_batch_size_over_eight_ranks():
    """lpvmpc.distributed.shard_range at world 8 for ragged totals: contiguous, ordered, sizes differ by at most one, nothing lost."""
    from lpvmpc.distributed import shard_range
    for total in (0, 1, 7, 8, 9, 1000, 8191, 8192, 65535, 65536):
        cuts = [shard_range(total, r, 8) for r in range(8)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(7))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total


@pytest.mark.timeout(300)
def test_bench_strong_scaling_cuts_the_steps_over_the_ranks():
    """--scaling strong: ONE sequence of steps is split contiguously over the ranks (the line says so); the default stays weak."""
    import sys
    out = _bench_line([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--scaling", "strong", "--steps", "20", "--warmup", "4"])
    assert out["scaling"] == "strong" and out["config"]["steps_rank0"] == [0, 10] and out["config"]["rccl_world"] == 2
    out = _bench_line([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "20", "--warmup", "4"])
    assert out["scaling"] == "weak" and out["config"]["steps_rank0"] == [0, 20]


@pytest.mark.timeout(300)
def test_bench_under_the_drivers_launcher():
    """The driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    import sys
    out = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--dry-run"])
    assert out["n_gpus"] == 2 and out["config"]["gather_ok"] is True


def test_bench_refuses_a_world_size_mismatch():
    import subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
