"""Seeded synthetic inputs for the BASELINE.json configurations.

Pure numpy, host side; used by bench.py, __graft_entry__.smoke() and the parity tests so that all of
them see identical batches.  Weights are the reference's hard-coded tunings (controllerMain.py:139-150,
plannerMain.py:96-99).
"""
from __future__ import annotations

import numpy as np

from .track import Map

CTRL_TUNINGS = {
    # controllerMain.py:139-142 (path tracking) and :146-150 (racing)
    "path": (np.diag([100.0, 1.0, 1.0, 20.0, 0.0, 900.0]), 0.25 * np.eye(2), 37.5 * np.array([1.3, 1.0])),
    "race": (np.diag([400.0, 1.0, 1.0, 20.0, 0.0, 1100.0]), 0.0 * np.eye(2), np.array([100.0, 45.0])),
}
# plannerMain.py:96-99
PLAN_Q = -np.diag([-0.000000000000088, -9.703658572659423, -0.5, 0.000000000213635, -0.153591566469547])
PLAN_L = -np.array([1.00702414775175, 0.187661946033823, -0.0, 0.0, -0.0329493219494661])
PLAN_R = np.diag([0.8, 0.0])
PLAN_dR = np.array([6.0, 6.0])


def curvature_at(s, table):
    """Vectorised piecewise-constant curvature lookup (same rule as Utilities/utilities.py:31-50)."""
    s = np.asarray(s, dtype=np.float64)
    L = table[-1, 3] + table[-1, 4]
    s = np.where(s > L, s - L * np.ceil(s / L - 1), s)
    out = np.zeros_like(s)
    for row in table:
        out = np.where((s >= row[3]) & (s < row[3] + row[4]), row[5], out)
    return out


def controller_batch(B, N=20, seed=0, shape="oval"):
    """cfg 2: racing tuning, random x0 along the track; vel_ref = vx constant, curv_ref = map curvature
    at s0, u_prev constant over the horizon.  Returns a dict of arrays ready for BatchedSolver.solve."""
    rng = np.random.default_rng(seed)
    mp = Map(shape, 0.2)
    s = rng.uniform(0.0, mp.TrackLength, B)
    vx = rng.uniform(0.8, 3.0, B)
    x0 = np.stack([vx, rng.normal(0, 0.05, B), rng.normal(0, 0.3, B), rng.normal(0, 0.1, B), s,
                   rng.normal(0, 0.1, B)], axis=1)
    u0 = np.stack([rng.normal(0, 0.05, B), rng.normal(0.2, 0.3, B)], axis=1)
    u_prev = np.repeat(u0[:, None, :], N, axis=1)
    vel_ref = np.repeat(vx[:, None], N + 1, axis=1)
    curv = np.repeat(curvature_at(s, mp.PointAndTangent)[:, None], N, axis=1)
    Q, R, dR = CTRL_TUNINGS["race"]
    return dict(kind="controller", N=N, dt=1.0 / 30.0, Q=Q, R=R, dR=dR, L_cf=None, track=mp.PointAndTangent,
                x0=x0, u_prev=u_prev, vel_ref=vel_ref, curv_s=curv, u_old=u0.copy(), max_ey=None, cf_new=60.0, lap=1)


def planner_batch(B, N=30, seed=1, shape="L_shape"):
    """cfg 3: planner weights, dt = 0.05, x0 inside the state box, SS = s0 + k vx dt."""
    rng = np.random.default_rng(seed)
    mp = Map(shape, 0.2)
    dt = 0.05
    vx = rng.uniform(1.0, 4.0, B)
    x0 = np.stack([vx, np.clip(rng.normal(0, 0.03, B), -0.9, 0.9), np.clip(rng.normal(0, 0.2, B), -1.9, 1.9),
                   np.clip(rng.normal(0, 0.05, B), -0.19, 0.19), np.clip(rng.normal(0, 0.05, B), -0.79, 0.79)], axis=1)
    s0 = rng.uniform(0.0, 19.2, B)
    SS = s0[:, None] + np.arange(N + 1)[None, :] * vx[:, None] * dt
    u0 = np.stack([rng.normal(0, 0.03, B), rng.normal(0.2, 0.2, B)], axis=1)
    u_prev = np.repeat(u0[:, None, :], N, axis=1)
    return dict(kind="planner", N=N, dt=dt, Q=PLAN_Q, R=PLAN_R, dR=PLAN_dR, L_cf=PLAN_L, track=mp.PointAndTangent,
                x0=x0, u_prev=u_prev, vel_ref=None, curv_s=SS, u_old=np.zeros((B, 2)), max_ey=np.full(B, 0.2),
                cf_new=60.0, lap=1)


def shard_batch(w, lo, hi):
    """Instances [lo, hi) of a workload dict (the per-instance arrays are sliced, everything else is shared)."""
    import numpy as np
    n = w["x0"].shape[0]
    return {k: (v[lo:hi] if isinstance(v, np.ndarray) and k != "track" and v.ndim >= 1 and v.shape[0] == n else v) for k, v in w.items()}


def make_solver(w, device=0, **settings):
    from .api import BatchedSolver
    return BatchedSolver(w["kind"], w["N"], w["dt"], w["Q"], w["R"], w["dR"], L_cf=w["L_cf"], track=w["track"],
                         device=device, **settings)
