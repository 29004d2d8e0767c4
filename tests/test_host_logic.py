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


# ---------------------------------------------------------------------------------------------------------
# planner -> controller hand-off operators: host-side C++ of the product (no device needed) against scipy,
# which is what the reference calls (PMAIN:112,257-280)
# ---------------------------------------------------------------------------------------------------------
def test_handoff_default_filter_is_the_references_elliptic_design():
    from scipy import signal
    from lpvmpc import _ffi
    c = _ffi.default_handoff_config()
    b, a = signal.ellip(4, 0.01, 120, 0.125)
    assert (c.interp_dt, c.padlen, c.order) == (0.033, 50, 4)
    assert np.max(np.abs(np.array(c.b[:5]) - b)) <= 1e-16 and np.max(np.abs(np.array(c.a[:5]) - a)) <= 1e-15


@pytest.mark.parametrize("N,dt", [(40, 0.05), (43, 0.05), (64, 0.033), (36, 0.05)])
def test_handoff_operators_match_scipy(N, dt):
    from scipy import signal
    from scipy.interpolate import interp1d
    from lpvmpc import api
    from oracle import handoff_ref as H
    W, FW = api.handoff_operators(N, dt)
    M = H.n_resampled(N, dt)
    assert W.shape == FW.shape == (M, N)
    b, a = H.ellip_coefficients()
    t50 = np.linspace(0, N * dt, N); t33 = np.linspace(0, N * dt, M)
    rng = np.random.default_rng(N)
    for _ in range(8):
        x = np.cumsum(rng.normal(0, 1, N))
        r = interp1d(t50, x, kind="cubic")(t33)
        assert np.max(np.abs(W @ x - r)) <= 1e-12 * max(1.0, np.max(np.abs(x)))
        assert np.max(np.abs(FW @ x - signal.filtfilt(b, a, r, padlen=50))) <= 1e-11 * max(1.0, np.max(np.abs(x)))
    dc = (np.sum(b) / np.sum(a)) ** 2                      # even-order elliptic design: -0.01 dB at DC, applied twice
    assert np.max(np.abs(W.sum(axis=1) - 1)) <= 1e-13 and np.max(np.abs(FW.sum(axis=1) - dc)) <= 1e-11


def test_handoff_refuses_what_scipy_refuses():
    import lpvmpc
    from lpvmpc import api
    with pytest.raises(lpvmpc.LpvMpcError) as e:            # N = 30 at 20 Hz -> 45 samples <= padlen 50 (filtfilt raises ValueError)
        api.handoff_operators(30, 0.05)
    assert "padlen" in str(e.value)


def test_body_frame_errors_host_helper():
    import lpvmpc
    g = load("handoff")
    out = np.array([lpvmpc.body_frame_errors(*r, 1.0 / 30) for r in g["bfe_in"]])
    assert np.max(np.abs(out - g["bfe_out"])) <= 1e-12


def test_ros_entry_points_exist_and_seeds_match_the_oracle():
    """The node shims import without ROS (rospy is only imported inside the *_main functions) and their seed
    trajectories equal the oracle's restatement of the two predicted_vectors_generation functions."""
    import os
    from lpvmpc import ros_nodes
    from oracle import lpv_ref as L
    ls = np.array([1.1, 0.02, -0.1, 0.03, 4.2, -0.05])
    xx, uu = ros_nodes.controller_seed(ls); xr, ur = L.ctrl_seed_vectors(ls)
    assert np.array_equal(xx, xr) and np.array_equal(uu, ur)
    x0 = np.array([1.2, 0.01, 0.05, 0.02, -0.03])
    xx, uu = ros_nodes.planner_seed(40, x0, 0.2, 0.05); xr, ur = L.plan_seed_vectors(40, x0, 0.2, 0.05)
    assert np.max(np.abs(xx - xr)) <= 1e-15 and np.array_equal(uu, ur)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("controllerMain.py", "plannerMain.py"):
        assert os.access(os.path.join(root, "ros", f), os.X_OK)
    assert callable(ros_nodes.controller_main) and callable(ros_nodes.planner_main)


def test_bench_algorithmic_bytes_follow_survey_8d():
    """bytes per ADMM iteration of SURVEY 8(d): 28 656 B (controller N = 20) and 36 096 B (planner N = 30) in float64."""
    import bench
    it = np.array([50, 75, 25])
    total, per_iter = bench.algorithmic_bytes(it)
    assert per_iter == 28656
    F, n_z = 21 * (36 + 64), 166
    per_solve = 8 * (2 * F + 20 * (36 + 12) + (6 + 40 + 42 + 2) + (n_z + 4))
    assert total == 150 * 28656 + 3 * per_solve
    _, per_iter_p = bench.algorithmic_bytes(it, N=30, nx=5, m_rows=31 * 5 + 31 * 5 + 30 * 2)
    assert per_iter_p == 36096
    assert bench.usable_cores() >= 1


def test_shard_range_property():
    from hypothesis import given, settings, strategies as st
    from lpvmpc.distributed import shard_range

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 100000), st.integers(1, 64))
    def prop(total, world):
        cuts = [shard_range(total, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    prop()
