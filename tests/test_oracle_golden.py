"""CPU: the numpy oracle (oracle/lpv_ref.py) against the golden vectors captured from the
reference's own Python (tests/golden/make_golden.py).  Exact pins, tolerance 1e-12."""
import numpy as np
import pytest

from oracle import lpv_ref as L
from tests._golden import cases, load

TOL = 1e-12
P = dict(L.DEFAULT_PARAMS)


def close(a, b, tol=TOL):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], b[~fin], equal_nan=True)
    scale = max(1.0, float(np.max(np.abs(b[fin]), initial=0.0)))
    assert float(np.max(np.abs(a[fin] - b[fin]), initial=0.0)) <= tol * scale


@pytest.mark.parametrize("shape", ["oval", "L_shape", "3110", "Euge_Track"])
def test_track_table_and_curvature(shape):
    g = load("tracks")
    mp = L.TrackMap(shape, 0.2)
    close(mp.PointAndTangent, g[shape + "_table"])
    close(mp.TrackLength, g[shape + "_length"])
    close(mp.halfWidth, g[shape + "_halfWidth"])
    got = np.array([L.curvature(s, mp.PointAndTangent) for s in g[shape + "_s"]])
    assert np.array_equal(got, g[shape + "_curv"])


def test_curvature_rejects_out_of_track():
    mp = L.TrackMap("oval", 0.2)
    with pytest.raises(ValueError):
        L.curvature(-0.1, mp.PointAndTangent)


@pytest.mark.parametrize("name,shape", [("ctrl_n10_cfg1", "oval"), ("ctrl_n20_oval", "oval")])
def test_controller_lpv_and_qp(name, shape):
    mp = L.TrackMap(shape, 0.2)
    for c in cases(name):
        N = int(c["N"]); dt = float(c["dt"])
        S, A, B = L.ctrl_lpv_prediction(P, dt, N, mp.PointAndTangent, c["x0"], c["u_prev"], c["vel_ref"],
                                        c["curv_ref"], float(c["cf_new"]), int(c["lap"]))
        close(S, c["states"]); close(A, c["A"]); close(B, c["B"])
        qp = L.ctrl_build_qp(c["Q"], c["R"], c["dR"], N, A, B, c["x0"], c["old_u"], c["vel_ref"], P["max_vel"])
        close(qp.P, c["P"]); close(qp.q, c["q"]); close(qp.A, c["Aqp"]); close(qp.l, c["l"]); close(qp.u, c["u"])
        xP, uP, lin = L.unpack_solution(c["x_orc"], 6, 2, N)
        close(xP, c["xPred"]); close(uP, c["uPred"]); close(lin, c["LinPoints"])


@pytest.mark.parametrize("name", ["plan_n30_lshape", "plan_n40_lshape"])
def test_planner_lpv_and_qp(name):
    mp = L.TrackMap("L_shape", 0.2)
    for c in cases(name):
        N = int(c["N"]); dt = float(c["dt"])
        S, A, B = L.plan_lpv_prediction(P, dt, N, mp.PointAndTangent, c["x0"], c["SS"], c["u_prev"])
        close(S, c["states"]); close(A, c["A"]); close(B, c["B"])
        qp = L.plan_build_qp(c["Q"], c["R"], c["dR"], c["L_cf"], N, A, B, c["x0"], [0.0, 0.0],
                             float(c["max_ey"]), P["max_vel"], P["min_vel"])
        close(qp.P, c["P"]); close(qp.q, c["q"]); close(qp.A, c["Aqp"]); close(qp.l, c["l"]); close(qp.u, c["u"])
        if np.all(np.isfinite(c["x_orc"])):
            xP, uP, lin = L.unpack_solution(c["x_orc"], 5, 2, N)
            close(xP, c["xPred"]); close(uP, c["uPred"]); close(lin, c["LinPoints"])


def test_seed_mode_linearisation():
    g = load("seed_mode")
    oval = L.TrackMap("oval", 0.2); lsh = L.TrackMap("L_shape", 0.2)
    xx, uu = L.ctrl_seed_vectors(g["ctrl_ls"])
    close(xx, g["ctrl_xx"]); close(uu, g["ctrl_uu"])
    A, B = L.ctrl_estimate_abc(P, 1 / 30.0, 20, oval.PointAndTangent, xx, uu)
    close(A, g["ctrl_A"]); close(B, g["ctrl_B"])
    Q = np.diag([100.0, 1.0, 1.0, 20.0, 0.0, 900.0]); R = 0.25 * np.eye(2); dR = 37.5 * np.array([1.3, 1.0])
    qp = L.ctrl_build_qp(Q, R, dR, 20, A, B, g["ctrl_ls"], [0.0, 0.0], np.ones(20), P["max_vel"])
    close(qp.P, g["ctrl_P"]); close(qp.q, g["ctrl_q"]); close(qp.A, g["ctrl_Aqp"])
    close(qp.l, g["ctrl_l"]); close(qp.u, g["ctrl_u"])
    pxx, puu = L.plan_seed_vectors(30, g["plan_x0"], 0.2, 0.05)
    close(pxx, g["plan_xx"]); close(puu, g["plan_uu"])
    A, B = L.plan_estimate_abc(P, 0.05, 30, lsh.PointAndTangent, pxx, puu)
    close(A, g["plan_A"]); close(B, g["plan_B"])
