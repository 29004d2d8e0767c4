"""GPU: row f2 of SURVEY 8f -- planner -> controller hand-off and the planner + controller + plant cascade on the device,
through the C ABI, against vectors made with the reference's own classes (tests/golden/handoff.npz, cascade.npz) and
against the oracle cascade on a small fleet of different vehicles."""
import numpy as np
import pytest

from tests._golden import load

pytestmark = pytest.mark.gpu


def planner(N=40):
    import lpvmpc
    from lpvmpc import workloads as W
    mp = lpvmpc.Map("L_shape", 0.2)
    eng = lpvmpc.BatchedSolver("planner", N, 0.05, W.PLAN_Q, W.PLAN_R, W.PLAN_dR, L_cf=W.PLAN_L, track=mp.PointAndTangent)
    return eng, mp


def controller_tt(mp):
    import lpvmpc
    from lpvmpc import workloads as W
    Q, R, dR = W.CTRL_TUNINGS["race"]
    return lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=mp.PointAndTangent)


def test_handoff_matches_reference_vectors():
    """All 60 planner ticks of the fixture as one batch: carried state, planner-rate signals and the five My_Planning arrays."""
    g = load("handoff")
    eng, _ = planner()
    assert eng.handoff_setup() == 61
    o = eng.handoff(g["plan_xPred"], g["plan_SS_in"], g["plan_pose_in"], want_sig=True)
    assert np.max(np.abs(o["SS"] - g["plan_SS_out"])) <= 1e-12
    assert np.max(np.abs(o["pose"] - g["plan_pose_out"])) <= 1e-12
    assert np.max(np.abs(o["sig"] - g["plan_sig"])) <= 1e-12
    assert np.max(np.abs(o["refs"] - g["plan_refs"])) <= 1e-11
    eng.close()


def test_handoff_other_horizon_vs_oracle_and_bad_use():
    import lpvmpc
    from oracle import handoff_ref as H
    eng, mp = planner(43)                                  # the commented-out launch block uses N = 43 (MAIN_LAUNCH.launch:24)
    with pytest.raises(lpvmpc.LpvMpcError):
        eng.handoff(np.zeros((1, 44, 5)), np.zeros((1, 44)), np.zeros((1, 3)))            # setup not called
    M = eng.handoff_setup()
    assert M == H.n_resampled(43, 0.05)
    rng = np.random.default_rng(3)
    B = 16
    x = np.stack([rng.uniform(1, 3, (B, 44)), rng.normal(0, 0.05, (B, 44)), rng.normal(0, 0.5, (B, 44)),
                  rng.normal(0, 0.08, (B, 44)), rng.normal(0, 0.1, (B, 44))], axis=2)
    s0 = rng.uniform(0, 2.5 * mp.TrackLength, B)
    SS = s0[:, None] + np.arange(44)[None, :] * 0.07
    pose = np.array([H.get_global_position(mp.PointAndTangent, s, 0.0) for s in SS[:, 0]], float)
    o = eng.handoff(x, SS, pose, want_sig=True)
    for b in range(B):
        SSr, last, xp, yp, yaw, vel, curv = H.planner_pose_refs(mp.PointAndTangent, x[b], SS[b], tuple(pose[b]), 0.05)
        assert np.max(np.abs(o["SS"][b] - SSr)) <= 1e-11 and np.max(np.abs(o["pose"][b] - np.array(last))) <= 1e-11
        assert np.max(np.abs(o["sig"][b] - np.array([xp, yp, yaw, vel, curv]))) <= 1e-11
        assert np.max(np.abs(o["refs"][b] - H.resample_refs(xp, yp, yaw, vel, curv, 0.05))) <= 1e-10
    eng.close()
    short, _ = planner(30)
    with pytest.raises(lpvmpc.LpvMpcError) as e:            # 45 resampled points <= padlen: scipy (and the reference's node) refuse
        short.handoff_setup()
    assert "padlen" in str(e.value)
    short.close()


def fleet_start(c, seed, B, spread=0.015):
    """B vehicles around the fixture's start state.  The planner box holds on x_0 too (quirk Q7): vx0 >= min_vel = 0.9."""
    rng = np.random.default_rng(seed)
    plant0 = np.tile(c["plant0"], (B, 1))
    plant0[:, 1] += rng.normal(0, spread, B); plant0[:, 2] += rng.uniform(-0.05, 0.3, B); plant0[:, 6] += rng.normal(0, spread, B)
    return plant0


def run_trace(prefetch, K):
    c = load("cascade")
    plan, mp = planner()
    plan.handoff_setup()
    ctrl = controller_tt(mp)
    ctrl.set_option("cascade_prefetch", prefetch)
    ctrl.cascade_init(plan, c["plant0"][None], c["cmd0"][None], c["uPred0"][None], lap0=int(c["lap0"]), half_width=mp.halfWidth,
                      slack=mp.slack, plan_max_ey=0.2, q9_swap=True)
    out = []
    for k in range(K):
        before = ctrl.cascade_read()
        ctrl.cascade_tick(1)
        after = ctrl.cascade_read()
        out.append((before, after))
    ctrl.close(); plan.close()
    return c, out


def test_cascade_matches_reference_trace():
    """60 controller ticks / 40 planner ticks of one vehicle against the reference's classes in the same schedule.

    The planner is an open-loop recursion of QPs solved to OSQP's eps = 1e-3.  As long as both sides stop at the same
    termination check the trajectories agree to round-off (strict phase: 1e-5 on states, 1e-4 on commands / references).
    Once a planner QP sits so close to the threshold that the two float64 implementations stop one check (25
    iterations) apart, the two runs carry solutions that differ at the eps level; from there only a loose tolerance is
    meaningful.  The strict phase has to cover the observed 51 controller ticks (34 planner ticks)."""
    c, out = run_trace(0, 60)
    strict = dict(plant=0.0, local=0.0, cmd=0.0, refs=0.0); loose = dict(strict)
    split = None
    for k, (before, after) in enumerate(out):
        j = int(c["ctrl_plan_ticks"][k]) - 1
        assert after["ticks"][0] == k + 1 and after["ticks"][1] == j + 1
        assert after["status"][0] == c["ctrl_status"][k] and after["plan_status"][0] == c["plan_status"][j], (k, j)
        assert after["lap"][0] == c["ctrl_lap"][k]
        if split is None and after["plan_iters"][0] != c["plan_iters"][j]:
            split = (k, j)
        w = strict if split is None else loose
        if split is None:
            assert after["iters"][0] == c["ctrl_iters"][k], k
        w["plant"] = max(w["plant"], float(np.max(np.abs(before["plant"][0] - c["ctrl_plant"][k]))))
        w["local"] = max(w["local"], float(np.max(np.abs(after["local"][0] - c["ctrl_local"][k]))))
        w["cmd"] = max(w["cmd"], float(np.max(np.abs(after["cmd"][0] - c["ctrl_cmd"][k]))))
        w["refs"] = max(w["refs"], float(np.max(np.abs(after["refs"][0] - c["plan_refs"][j]))))
    print("cascade trace: strict phase until (controller tick, planner tick) =", split, strict, "after:", loose)
    # observed: the first planner QP that stops one check apart belongs to controller tick 51 (planner tick 34); a strict
    # phase shorter than that is a regression of the solver arithmetic, not round-off
    assert split is None or (split[0] >= 51 and split[1] >= 34), split
    assert strict["plant"] <= 1e-5 and strict["local"] <= 1e-5 and strict["cmd"] <= 1e-4 and strict["refs"] <= 1e-4
    assert max(loose.values()) <= 2e-2


def test_cascade_prefetch_does_not_change_results():
    _, a = run_trace(0, 24)
    _, b = run_trace(1, 24)
    for (_, x), (_, y) in zip(a, b):
        for key in ("plant", "local", "cmd", "iters", "status", "lap"):
            assert np.array_equal(x[key], y[key]), key


def test_cascade_fleet_vs_oracle():
    """8 different vehicles for 18 controller ticks (12 planner ticks) against the oracle cascade."""
    from oracle import cascade_ref as CR
    from lpvmpc import workloads as W
    c = load("cascade")
    B = 8
    plant0 = fleet_start(c, 11, B)
    cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))
    plan, mp = planner()
    plan.handoff_setup()
    ctrl = controller_tt(mp)
    ctrl.cascade_init(plan, plant0, cmd0, uPred0, lap0=1, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, q9_swap=True)
    ref = CR.CascadeRef(mp.PointAndTangent, W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), plant0, cmd0, uPred0,
                        half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, nthreads=4)
    same_iters = 0
    for k in range(18):
        ctrl.cascade_tick(1); ref.tick()
        o = ctrl.cascade_read(full=False)
        assert np.max(np.abs(o["plant"] - ref.plant)) <= 1e-5, k
        assert np.max(np.abs(o["local"] - ref.local)) <= 1e-5, k
        assert np.max(np.abs(o["cmd"] - ref.cmd)) <= 1e-4, k
        assert np.array_equal(o["status"], ref.ctrl["status"])
        same_iters += int(np.sum(o["iters"] == ref.ctrl["iters"]))
    assert same_iters >= 0.95 * 18 * B
    ctrl.close(); plan.close()


def test_sub_fleets_side_by_side_equal_the_whole_fleet():
    """bench.py's cfg5 leg cuts the fleet into sub-fleets with their own engine pairs, all enqueued before any is read:
    vehicles are independent, so the results have to be bit-identical to one fleet holding all of them."""
    c = load("cascade")
    B, K = 48, 15
    plant0 = fleet_start(c, 5, B)
    cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))

    def run(cuts):
        fleets = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            plan, mp = planner()
            plan.handoff_setup()
            ctrl = controller_tt(mp)
            ctrl.cascade_init(plan, plant0[a:b], cmd0[a:b], uPred0[a:b], half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
            fleets.append((ctrl, plan))
        for ctrl, _ in fleets:
            ctrl.cascade_tick(K)
        outs = [ctrl.cascade_read() for ctrl, _ in fleets]
        for ctrl, plan in fleets:
            ctrl.close(); plan.close()
        return {k: np.concatenate([o[k] for o in outs]) for k in ("plant", "local", "cmd", "iters", "status", "lap", "plan_iters", "plan_status")}

    whole, parts = run([0, B]), run([0, 7, 30, B])
    for k in whole:
        assert np.array_equal(whole[k], parts[k], equal_nan=whole[k].dtype.kind == "f"), k


def test_infeasible_planner_instance_stays_contained():
    """With this start the planner QP of vehicle 0 (and later of a few others) turns primal infeasible (at vx ~ 1.1 m/s the
    forward-Euler lateral dynamics of the planner model are unstable at dt = 0.05, the 40-step prediction blows up).
    The reference's node would publish NaN references from then on; here such a vehicle carries NaN, every other
    vehicle is untouched and the engine keeps ticking."""
    c = load("cascade")
    B = 8
    plant0 = fleet_start(c, 16, B)
    cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))
    plan, mp = planner()
    plan.handoff_setup()
    ctrl = controller_tt(mp)
    ctrl.cascade_init(plan, plant0, cmd0, uPred0, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
    ctrl.cascade_tick(30)
    o = ctrl.cascade_read()
    bad = ~np.all(np.isfinite(o["plant"]), axis=1)
    assert bad[0] and not np.all(bad)
    assert np.all(o["status"][~bad] == 1) and np.all(o["status"][bad] == -10)       # NaN data: UNSOLVED, no iterations spent
    assert np.all(o["iters"][bad] == 0)
    # the healthy vehicles alone: identical results
    ctrl2 = controller_tt(mp)
    plan2, _ = planner(); plan2.handoff_setup()
    ctrl2.cascade_init(plan2, plant0[~bad], cmd0[~bad], uPred0[~bad], half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
    ctrl2.cascade_tick(30)
    o2 = ctrl2.cascade_read()
    assert np.array_equal(o["plant"][~bad], o2["plant"]) and np.array_equal(o["cmd"][~bad], o2["cmd"])
    for e in (ctrl, plan, ctrl2, plan2):
        e.close()


def test_cascade_bad_arguments():
    import lpvmpc
    c = load("cascade")
    plan, mp = planner()
    ctrl = controller_tt(mp)
    args = (c["plant0"][None], c["cmd0"][None], c["uPred0"][None])
    with pytest.raises(lpvmpc.LpvMpcError):                 # hand-off operators missing
        ctrl.cascade_init(plan, *args)
    plan.handoff_setup()
    with pytest.raises(lpvmpc.LpvMpcError):                 # lap 0 is the path-tracking phase (lpvmpc_cl_*)
        ctrl.cascade_init(plan, *args, lap0=0)
    with pytest.raises(lpvmpc.LpvMpcError):                 # roles swapped
        plan.cascade_init(ctrl, args[0], args[1], np.zeros((1, 40, 2)))
    with pytest.raises(lpvmpc.LpvMpcError):
        ctrl.cascade_tick(1)                                # not initialised
    # a controller handle with steering delay reads u_old as [B][2 + delay]; the cascade's measurement kernel writes [B][2]:
    # refused, as lpvmpc_cl_init refuses it (the reference runs steeringDelay = 0, CMAIN:49)
    from lpvmpc import workloads as W_
    Qr, Rr, dRr = W_.CTRL_TUNINGS["race"]
    delayed = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Qr, Rr, dRr, track=mp.PointAndTangent, steering_delay=1)
    with pytest.raises(lpvmpc.LpvMpcError) as e:
        delayed.cascade_init(plan, *args)
    assert "steeringDelay" in str(e.value)
    delayed.close()
    # destroy order: the planner handle may go first (its destroy ends the cascade that points at it); the controller handle
    # is usable and destroyable afterwards
    ctrl.cascade_init(plan, *args, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
    ctrl.cascade_tick(2)
    plan.close()
    with pytest.raises(lpvmpc.LpvMpcError):
        ctrl.cascade_tick(1)                                # the cascade went with its planner
    ctrl.close()


def test_cascade_attrition_matches_the_oracle_cascade():
    """configs[4] is a survival experiment: planner QPs of the reference's open-loop recursion turn primal infeasible and
    the vehicle is lost.  32 vehicles of the bench's start distribution (seed 3) on the device and in
    the oracle cascade on the host.  For the first 25 controller ticks (17 open-loop planner ticks) the two float64
    implementations carry the same trajectories: the SAME vehicles are lost at the SAME ticks.  After that the eps-level
    differences of the planner's un-polished iterates have been amplified by the recursion (plant states differ by 1e-2 after
    40 ticks), individual verdicts can move by some ticks or to a neighbouring vehicle, and what has to agree is the attrition:
    the number of survivors differs by at most two of 32 at every later tick.  The losses belong to the reference's planner
    formulation, not to this engine (tests/diagnostics/cascade_attrition_cpu.py prints the oracle's side alone)."""
    from oracle import cascade_ref as CR
    from lpvmpc import workloads as W
    c = load("cascade")
    B, K, STRICT = 32, 60, 25
    plant0 = fleet_start(c, 3, B, spread=0.01)
    cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))
    plan, mp = planner()
    plan.handoff_setup()
    ctrl = controller_tt(mp)
    ctrl.cascade_init(plan, plant0, cmd0, uPred0, lap0=1, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, q9_swap=True)
    ref = CR.CascadeRef(mp.PointAndTangent, W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), plant0, cmd0, uPred0,
                        half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, nthreads=16)
    dead_dev = np.full(B, -1); dead_ref = np.full(B, -1)
    for k in range(K):
        ctrl.cascade_tick(1); ref.tick()
        o = ctrl.cascade_read(full=False)
        a_dev = np.all(np.isfinite(o["plant"]), axis=1); a_ref = np.all(np.isfinite(ref.plant), axis=1)
        dead_dev[(dead_dev < 0) & ~a_dev] = k; dead_ref[(dead_ref < 0) & ~a_ref] = k
        if k < STRICT:
            assert np.array_equal(a_dev, a_ref), (k, np.nonzero(a_dev != a_ref)[0])
            both = a_dev & a_ref
            assert np.median(np.max(np.abs(o["plant"][both] - ref.plant[both]), axis=1)) <= 1e-3, k      # the typical vehicle; one about to be lost has already drifted
        else:
            assert abs(int(a_dev.sum()) - int(a_ref.sum())) <= 2, (k, int(a_dev.sum()), int(a_ref.sum()))
    early = (dead_ref >= 0) & (dead_ref < STRICT)
    assert np.array_equal(dead_dev[early], dead_ref[early]) and int(early.sum()) >= 1          # the sample does contain early losses
    assert int(np.sum(dead_dev < 0)) >= B // 2
    print("lost on the device (vehicle: controller tick):", {int(b): int(dead_dev[b]) for b in np.nonzero(dead_dev >= 0)[0]},
          "in the oracle:", {int(b): int(dead_ref[b]) for b in np.nonzero(dead_ref >= 0)[0]})
    ctrl.close(); plan.close()


def test_full_size_cfg5_fleet_properties():
    """configs[4] at one GPU's full size: 8192 vehicles, 300 controller ticks (10 s of driving, one lap for the survivors).
    Size-independent properties of the cascade: a vehicle is alive exactly as long as its plant state is finite; a lost vehicle
    stays lost, reports LPVMPC_UNSOLVED and no iterations; live vehicles report valid OSQP statuses, stay on the track and
    inside the speed limits; lap counters never decrease; the fleet is not wiped out and not loss-free (the losses of the
    reference's planner recursion are part of the workload)."""
    from lpvmpc import workloads as W
    c = load("cascade")
    B = 8192
    plant0 = fleet_start(c, 3, B, spread=0.01)
    plan, mp = planner()
    plan.handoff_setup()
    ctrl = controller_tt(mp)
    ctrl.cascade_init(plan, plant0, np.tile(c["cmd0"], (B, 1)), np.tile(c["uPred0"], (B, 1, 1)), lap0=1, half_width=mp.halfWidth,
                      slack=mp.slack, plan_max_ey=0.2, q9_swap=True)
    valid = {1, 2, -2, -3, 3, -4, 4, -10}
    alive_prev = np.ones(B, bool); lap_prev = np.full(B, -(1 << 30))
    fractions = []
    for block in range(6):
        ctrl.cascade_tick(50)
        o = ctrl.cascade_read(full=False)
        alive = np.all(np.isfinite(o["plant"]), axis=1)
        assert not np.any(alive & ~alive_prev)                                   # nobody comes back
        assert set(np.unique(o["status"]).tolist()) <= valid and set(np.unique(o["plan_status"]).tolist()) <= valid
        dead = ~alive
        assert np.all(o["status"][dead] == -10) and np.all(o["iters"][dead] == 0)                      # no iterations on NaN data
        assert np.all(np.isin(o["status"][alive], (1, 2, -2)))                                         # a solution was applied
        assert np.all(o["iters"][alive] >= 25) and np.all(o["iters"][alive] % 25 == 0)
        assert np.all(o["lap"][alive] >= lap_prev[alive])
        assert np.all(np.isfinite(o["cmd"][alive])) and np.all(np.abs(o["cmd"][alive, 0]) <= 0.249 + 1e-6)
        vx = o["plant"][alive, 2]
        assert np.all(vx > 0.3) and np.all(vx < 5.5)
        assert np.all(np.abs(o["local"][alive, 5]) <= mp.halfWidth + mp.slack + 0.1)                   # |ey|: on the track
        alive_prev, lap_prev = alive, np.where(alive, o["lap"], lap_prev)
        fractions.append(float(alive.mean()))
    assert o["ticks"][0] == 300
    assert 0.5 <= fractions[-1] <= 0.99, fractions
    assert int(o["lap"][alive_prev].max()) >= 2                                   # survivors have finished the first racing lap
    print("alive fraction after 50 .. 300 ticks:", fractions, "laps of survivors: min %d max %d" % (o["lap"][alive_prev].min(), o["lap"][alive_prev].max()))
    ctrl.close(); plan.close()
