"""GPU: row f1 of SURVEY 8f -- plant model, track transforms and a closed-loop fleet on the device, against vectors
produced by the reference's own Simulator / Map / controller classes (tests/golden/plant_and_transforms.npz)."""
import numpy as np
import pytest

from tests._golden import load

pytestmark = pytest.mark.gpu

PATH_TUNING = (np.diag([100.0, 1.0, 1.0, 20.0, 0.0, 900.0]), 0.25 * np.eye(2), 37.5 * np.array([1.3, 1.0]))


def engine(shape):
    import lpvmpc
    Q, R, dR = PATH_TUNING
    return lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=lpvmpc.Map(shape, 0.2).PointAndTangent)


@pytest.mark.parametrize("shape", ["oval", "L_shape"])
def test_track_transforms(shape):
    g = load("plant_and_transforms")
    eng = engine(shape)
    glob = eng.global_position(np.column_stack([g[shape + "_s"], g[shape + "_ey"]]))
    assert np.max(np.abs(glob - g[shape + "_glob"])) <= 1e-12
    loc = eng.local_position(g[shape + "_pts"], float(g[shape + "_hw"]), float(g[shape + "_slack"]))
    assert np.array_equal(loc[:, 3], g[shape + "_loc"][:, 3])
    assert np.max(np.abs(loc - g[shape + "_loc"])) <= 1e-11
    eng.close()


def test_plant_model():
    g = load("plant_and_transforms")
    eng = engine("oval")
    st = g["sim_init"][None].copy()
    worst = 0.0
    for u, ref in zip(g["sim_u"], g["sim_states"]):
        st = eng.plant_step(st, u[None], n_sub=1)
        worst = max(worst, float(np.max(np.abs(st[0] - ref))))
    assert worst <= 1e-11
    # n_sub steps in one call == n_sub calls
    a = eng.plant_step(g["sim_init"][None], g["sim_u"][:1], n_sub=7)
    b = g["sim_init"][None].copy()
    for _ in range(7):
        b = eng.plant_step(b, g["sim_u"][:1], n_sub=1)
    assert np.max(np.abs(a - b)) <= 1e-14
    eng.close()


def test_closed_loop_fleet_matches_reference_trace():
    """40 control ticks of controller + plant + map entirely on the device (B = 3 identical vehicles), against the
    trace of the reference's classes driven in the same synchronous schedule (7 plant steps per tick)."""
    import lpvmpc
    g = load("plant_and_transforms")
    eng = engine("oval")
    mp = lpvmpc.Map("oval", 0.2)
    plant0 = np.tile(np.array([0.01, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (3, 1))
    eng.cl_init(plant0, mp.halfWidth, mp.slack, q9_swap=True, n_sub=7)
    for tick in range(40):
        before = eng.cl_read()["plant"]
        assert np.max(np.abs(before - g["cl_plant"][tick])) <= 2e-6, tick
        eng.cl_tick(1)
        o = eng.cl_read()
        assert np.all(o["status"] == 1)
        assert np.all(o["iters"] == g["cl_iter"][tick]), (tick, o["iters"], g["cl_iter"][tick])
        assert np.max(np.abs(o["local"] - g["cl_local"][tick])) <= 2e-6, tick
        assert np.max(np.abs(o["cmd"] - g["cl_cmd"][tick])) <= 2e-6, tick
        assert np.max(np.abs(o["plant"][0] - o["plant"][2])) == 0.0            # identical vehicles stay identical
    eng.close()


def test_fleet_of_different_vehicles_runs():
    import lpvmpc
    eng = engine("oval")
    mp = lpvmpc.Map("oval", 0.2)
    rng = np.random.default_rng(3)
    B = 512
    s0 = rng.uniform(0.05, 12.5, B); ey0 = rng.normal(0, 0.03, B)
    xyth = eng.global_position(np.column_stack([s0, ey0]))
    plant0 = np.column_stack([xyth[:, 0], xyth[:, 1], rng.uniform(0.8, 1.2, B), np.zeros(B), np.zeros(B), np.zeros(B), xyth[:, 2], np.zeros(B)])
    eng.cl_init(plant0, mp.halfWidth, mp.slack, q9_swap=False, n_sub=7)
    eng.cl_tick(60)
    o = eng.cl_read()
    assert np.mean(np.isin(o["status"], (1, 2))) > 0.98
    # the fleet advanced along the track (about 2 s at ~1 m/s) and stayed on it
    ds = (o["local"][:, 4] - s0) % mp.TrackLength
    assert np.all(o["local"][:, 4] < 9999) and np.median(ds) > 1.0
    assert np.median(np.abs(o["local"][:, 5])) < 0.1
    eng.close()


def test_batch_calls_are_refused_while_a_fleet_runs():
    """The fleet's receding-horizon state lives in the handle's workspace between ticks: a stand-alone batch call on the same
    handle would overwrite (or reallocate) it silently.  It is refused with LPVMPC_E_ARG until lpvmpc_cl_release; the fleet
    is unharmed by the attempt, and a cascade protects its planner handle the same way."""
    import lpvmpc
    from lpvmpc import workloads
    from tests._golden import load
    g = load("plant_and_transforms")
    w = workloads.controller_batch(4, N=20, seed=9)
    mp = lpvmpc.Map("oval", 0.2)
    Q, R, dR = workloads.CTRL_TUNINGS["path"]
    eng = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=mp.PointAndTangent)
    ref = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=mp.PointAndTangent)
    plant0 = np.tile(np.asarray(g["cl_plant"][0], float).reshape(1, 8), (4, 1))          # the fixture's start state
    for e in (eng, ref):
        e.cl_init(plant0, half_width=mp.halfWidth, slack=mp.slack)
        e.cl_tick(3)
    with pytest.raises(lpvmpc.LpvMpcError) as err:
        eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    assert err.value.code == lpvmpc._ffi.E_ARG and "fleet" in str(err.value)
    with pytest.raises(lpvmpc.LpvMpcError):
        eng.local_position(np.zeros((2000, 3)), mp.halfWidth, mp.slack)          # would have reallocated the workspace
    for e in (eng, ref):
        e.cl_tick(3)
    a, b = eng.cl_read(), ref.cl_read()
    for k in ("plant", "cmd", "iters", "status"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    eng.cl_release()
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    assert np.all(out["status"] == 1)
    eng.close(); ref.close()
