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


# ---- the real bench.py with one rank per visible GPU (two where the box has them: the driver's scaling run has eight) ----------
def _bench(args, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _ranks():
    import torch
    return min(2, torch.cuda.device_count())


@pytest.mark.timeout(900)
def test_bench_ranks_reproduce_single_gpu_results(tmp_path):
    """bench.py --gpus n (n = 2 where two devices are visible, else 1): n_gpus in the line, and the (u0, status, iters) that the
    closing all-gather returns equal what ONE device computes for each rank's batch (status and iteration counts exactly, the
    first input exactly for the instances the straggler deferral never parked, to 1e-6 for those the tail kernel finished)."""
    from lpvmpc import workloads
    n = _ranks()
    f = str(tmp_path / "gather.npz")
    out = _bench(["--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--dump-results", f])
    assert out["n_gpus"] == n and out["steps"] == 5 and out["value"] > 0 and out["scaling"] == "weak"
    d = np.load(f)
    B = int(d["batch"])
    assert int(d["world"]) == n and d["u0"].shape == (n * B, 2)
    for r in range(n):
        w = workloads.controller_batch(B, N=20, seed=int(d["seeds"][r]))
        eng = workloads.make_solver(w)
        ref = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]); eng.close()
        sl = slice(r * B, (r + 1) * B)
        assert np.array_equal(d["status"][sl], ref["status"]) and np.array_equal(d["iters"][sl], ref["iters"]), r
        early = ref["iters"] <= 100
        assert np.array_equal(d["u0"][sl][early], ref["uPred"][early, 0, :], equal_nan=True)
        late = ~early & np.isfinite(ref["uPred"][:, 0, 0])
        assert np.all(np.abs(d["u0"][sl][late] - ref["uPred"][late, 0, :]) <= 1e-6)


@pytest.mark.timeout(900)
def test_bench_cfg4_shards_one_global_batch(tmp_path):
    """--workload cfg4 --batch 4096: ONE global batch (4096 per rank) cut with shard_range; the gathered results, in global instance
    order, are bit-identical to a single-device solve of the whole batch (no deferral on this path)."""
    from lpvmpc import workloads
    n = _ranks()
    f = str(tmp_path / "gather4.npz")
    out = _bench(["--gpus", str(n), "--workload", "cfg4", "--batch", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dump-results", f])
    half = 2048 * n
    assert out["n_gpus"] == n and out["config"]["global_instances"] == 4096 * n and out["config"]["shard_rank0"] == [0, half // n]
    d = np.load(f)
    assert int(d["half"]) == half and int(d["world"]) == n
    for kind, make in (("ctrl", workloads.controller_batch), ("plan", workloads.planner_batch)):
        w = make(half, N=20, seed=2)
        eng = workloads.make_solver(w)
        ref = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], w["max_ey"], w["cf_new"], w["lap"]); eng.close()
        assert np.array_equal(d[kind + "_status"], ref["status"]) and np.array_equal(d[kind + "_iters"], ref["iters"]), kind
        assert np.array_equal(d[kind + "_u0"], ref["uPred"][:, 0, :], equal_nan=True), kind
