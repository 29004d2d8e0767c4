"""CPU: the hand-off oracle (oracle/handoff_ref.py) against vectors made with the reference's own Map / Curvature
functions and scipy (tests/golden/handoff.npz)."""
import numpy as np
import pytest

from oracle import handoff_ref as H
from tests._golden import load


def test_ellip_coefficients_match_fixture():
    g = load("handoff")
    b, a = H.ellip_coefficients()
    assert np.max(np.abs(b - g["ellip_b"])) <= 1e-15 and np.max(np.abs(a - g["ellip_a"])) <= 1e-15


def test_pose_and_resampling():
    g = load("handoff")
    tab, dt = g["table"], float(g["dt"])
    for t in range(g["plan_xPred"].shape[0]):
        SS, last, xp, yp, yaw, vel, curv = H.planner_pose_refs(tab, g["plan_xPred"][t], g["plan_SS_in"][t], tuple(g["plan_pose_in"][t]), dt)
        assert np.max(np.abs(SS - g["plan_SS_out"][t])) <= 1e-12
        assert np.max(np.abs(np.array(last) - g["plan_pose_out"][t])) <= 1e-12
        sig = np.array([xp, yp, yaw, vel, curv])
        assert np.max(np.abs(sig - g["plan_sig"][t])) <= 1e-12
        refs = H.resample_refs(xp, yp, yaw, vel, curv, dt)
        assert refs.shape == (5, H.n_resampled(int(g["N"]), dt)) == (5, 61)
        assert np.max(np.abs(refs - g["plan_refs"][t])) <= 1e-12


def test_short_horizons_raise_like_scipy():
    # N = 30 at 20 Hz resamples to 45 points <= padlen 50: scipy's filtfilt refuses (so does the reference's node)
    N = 30
    z = np.linspace(0, 1, N)
    with pytest.raises(ValueError):
        H.resample_refs(z, z, z, z, z, 0.05)


def test_body_frame_errors():
    g = load("handoff")
    out = np.array([H.body_frame_errors(*r, 1.0 / 30) for r in g["bfe_in"]])
    assert np.max(np.abs(out - g["bfe_out"])) <= 1e-12


def test_tracking_glue_reproduces_cascade_measurements():
    """Feeding the fixture's plant states and planner messages through TrackingGlue gives the fixture's LocalState and
    reference windows (including the every-other-tick latch)."""
    c = load("cascade")
    tab = c["table"]
    glue = H.TrackingGlue(20, 1.0 / 30, tab[-1, 3] + tab[-1, 4], lap=int(c["lap0"]))
    for k in range(c["ctrl_plant"].shape[0]):
        refs = c["plan_refs"][int(c["ctrl_plan_ticks"][k]) - 1]
        Lc, vel, curv = glue.measure(c["ctrl_plant"][k], refs)
        assert np.max(np.abs(Lc - c["ctrl_local"][k])) <= 1e-12
        assert np.array_equal(vel, c["ctrl_vel_ref"][k]) and np.array_equal(curv, c["ctrl_curv_ref"][k])
        assert glue.lap == c["ctrl_lap"][k] and abs(glue.SS - c["ctrl_SS"][k]) <= 1e-12


def test_cascade_oracle_reproduces_reference_trace():
    """The assembled oracle cascade (C tick functions + hand-off + plant) against the trace made with the reference's
    classes: 24 controller ticks / 16 planner ticks.  Only solver round-off separates the two (same OSQP restatement
    on both sides, different assembly code)."""
    from oracle import cascade_ref as CR, lpv_ref as L
    from lpvmpc import workloads as W
    c = load("cascade")
    cas = CR.CascadeRef(c["table"], W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), c["plant0"][None], c["cmd0"][None],
                        c["uPred0"][None], lap0=int(c["lap0"]), half_width=L.TrackMap("L_shape", 0.2).halfWidth,
                        slack=L.TrackMap("L_shape", 0.2).slack, plan_max_ey=0.2)
    for k in range(24):
        assert np.max(np.abs(cas.plant[0] - c["ctrl_plant"][k])) <= 1e-7, k
        cas.tick()
        assert np.max(np.abs(cas.local[0] - c["ctrl_local"][k])) <= 1e-7, k
        assert np.max(np.abs(cas.cmd[0] - c["ctrl_cmd"][k])) <= 1e-6, k
        assert cas.ctrl["iters"][0] == c["ctrl_iters"][k] and cas.plan_ticks == c["ctrl_plan_ticks"][k]
        j = cas.plan_ticks - 1
        assert cas.plan["iters"][0] == c["plan_iters"][j], (k, j)
        assert np.max(np.abs(cas.refs[0] - c["plan_refs"][j])) <= 1e-6, (k, j)
