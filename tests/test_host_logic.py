"""CPU: host-side pieces of the product package (track table, workload generators, sharding helpers)."""
import numpy as np
import pytest

from tests._golden import load


@pytest.mark.parametrize("shape", ["oval", "L_shape", "3110", "Euge_Track"])
def test_product_map_matches_reference_tables(shape):
    import lpvmpc
    g = load("tracks")
    mp = lpvmpc.Map(shape, 0.2)
    assert np.max(np.abs(mp.PointAndTangent - g[shape + "_table"])) <= 1e-12
    assert abs(mp.TrackLength - float(g[shape + "_length"])) <= 1e-12
    assert abs(mp.halfWidth - float(g[shape + "_halfWidth"])) <= 1e-12
    from lpvmpc.workloads import curvature_at
    assert np.array_equal(curvature_at(g[shape + "_s"], mp.PointAndTangent), g[shape + "_curv"])


def test_unknown_track_is_rejected():
    import lpvmpc
    with pytest.raises(ValueError):
        lpvmpc.Map("figure8")


def test_workloads_are_seeded_and_in_domain():
    from lpvmpc import workloads
    a = workloads.controller_batch(64, 20, seed=0); b = workloads.controller_batch(64, 20, seed=0)
    assert all(np.array_equal(a[k], b[k]) for k in ("x0", "u_prev", "vel_ref", "curv_s", "u_old"))
    assert a["x0"].shape == (64, 6) and a["u_prev"].shape == (64, 20, 2) and a["vel_ref"].shape == (64, 21)
    assert np.all((a["x0"][:, 0] >= 0.8) & (a["x0"][:, 0] <= 3.0)) and np.all((a["x0"][:, 4] >= 0) & (a["x0"][:, 4] <= 13.0))
    p = workloads.planner_batch(32, 30, seed=1)
    assert p["curv_s"].shape == (32, 31) and np.all(np.abs(p["x0"][:, 3]) <= 0.19) and np.all(p["max_ey"] == 0.2)
    assert not np.array_equal(workloads.controller_batch(64, 20, seed=1)["x0"], a["x0"])


def test_shard_range_partitions_the_batch():
    from lpvmpc.distributed import shard_range
    for total, world in ((65536, 8), (1000, 3), (7, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 2)
