"""Oracle (TEST INFRASTRUCTURE): float64 numpy restatement of the reference's
LPV evaluation, horizon roll-out and dense QP assembly.

Every function cites the reference lines it follows (paths relative to
``/root/reference/workspace/src/barc/src``):

  CTRL  = ControllerObject/PathFollowingLPVMPC.py
  PLAN  = PlannerObject/LPV_MPC_Planner.py
  UTIL  = Utilities/utilities.py
  TRACK = Utilities/trackInitialization.py

Pinned: ``tests/test_oracle_golden.py`` compares every function here with the
golden vectors in ``tests/golden/*.npz``, which were produced by importing the
reference's own Python (``tests/golden/make_golden.py``) -- agreement <= 1e-12.

This module is the checker.  It is never imported by the product package.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------
# vehicle / launch parameters (MAIN_LAUNCH.launch:5-11,40-44)
# --------------------------------------------------------------------------
DEFAULT_PARAMS = dict(lf=0.125, lr=0.125, m=1.98, Iz=0.03, Cf=60.0, Cr=60.0,
                      mu=0.05, max_vel=5.0, min_vel=0.9)


# --------------------------------------------------------------------------
# track table  (TRACK:13-202) and curvature lookup (UTIL:31-50)
# --------------------------------------------------------------------------
def _wrap(a):                       # TRACK:408-416
    if a < -math.pi:
        return 2 * math.pi + a
    if a > math.pi:
        return a - 2 * math.pi
    return a


def _sgn(a):                        # TRACK:419-425  (sign(0) = +1)
    return 1 if a >= 0 else -1


def track_spec(shape):
    """(length, signed radius) rows and (halfWidth override, slack). TRACK:28-81."""
    pi = np.pi
    if shape == "3110":
        k = 0.03
        spec = [[60 * k, 0], [80 * k, 80 * k * 2 / pi], [20 * k, 0],
                [80 * k, 80 * k * 2 / pi], [40 * k, -40 * k * 10 / pi],
                [60 * k, 60 * k * 5 / pi], [40 * k, -40 * k * 10 / pi],
                [80 * k, 80 * k * 2 / pi], [20 * k, 0],
                [80 * k, 80 * k * 2 / pi], [80 * k, 0]]
        return np.array(spec, dtype=float), 0.6, 0.15
    if shape == "oval":
        spec = [[1.0, 0], [4.5, 4.5 / pi], [2.0, 0], [4.5, 4.5 / pi], [1.0, 0]]
        return np.array(spec, dtype=float), None, 0.15
    if shape == "L_shape":
        lc = 4.5
        spec = [[1.0, 0], [lc, lc / pi], [lc / 2, -lc / pi], [lc, lc / pi],
                [lc / pi * 2, 0], [lc / 2, lc / pi]]
        return np.array(spec, dtype=float), None, 0.45
    if shape == "Euge_Track":
        k = 0.03
        c = 30 * k * 2 / pi
        spec = [[30 * k, c], [20 * k, -0], [30 * k, -c], [30 * k, c],
                [30 * k, c], [130 * k, 0], [30 * k, c], [10 * k, -0],
                [30 * k, c], [55 * k, -0], [30 * k, -c], [10 * k, -0],
                [30 * k, c]]
        return np.array(spec, dtype=float), 0.4, 0.15
    raise ValueError("unknown track shape %r" % (shape,))


class TrackMap:
    """PointAndTangent table = rows [x, y, psi, cum_s, seg_len, curvature].

    TRACK:13-202.  ``halfwidth_param`` is the ROS parameter
    /TrajectoryPlanner/halfWidth; the reference adds 0.1 (TRACK:20).
    """

    def __init__(self, shape="oval", halfwidth_param=0.2):
        spec, hw_fixed, slack = track_spec(shape)
        self.shape = shape
        self.slack = slack
        self.halfWidth = hw_fixed if hw_fixed is not None else halfwidth_param + 0.1
        nseg = spec.shape[0]
        T = np.zeros((nseg + 1, 6))
        for i in range(nseg):
            seglen, rad = spec[i]
            if i == 0:
                x0 = y0 = ang = 0.0
                s0 = 0.0
            else:
                x0, y0, ang = T[i - 1, 0], T[i - 1, 1], T[i - 1, 2]
                s0 = T[i - 1, 3] + T[i - 1, 4]
            if rad == 0.0:                                   # TRACK:90-113
                x = x0 + seglen * np.cos(ang)
                y = y0 + seglen * np.sin(ang)
                T[i] = [x, y, ang, s0, seglen, 0.0]
            else:                                            # TRACK:114-167
                d = 1 if rad >= 0 else -1
                cx = x0 + np.abs(rad) * np.cos(ang + d * np.pi / 2)
                cy = y0 + np.abs(rad) * np.sin(ang + d * np.pi / 2)
                span = seglen / np.abs(rad)
                psi = _wrap(ang + span * np.sign(rad))
                an = _wrap(d * np.pi / 2 + ang)
                a0 = -(np.pi - np.abs(an)) * _sgn(an)
                x = cx + np.abs(rad) * np.cos(a0 + d * span)
                y = cy + np.abs(rad) * np.sin(a0 + d * span)
                T[i] = [x, y, psi, s0, seglen, 1 / rad]
        # closing segment back to the origin (TRACK:186-199)
        xs, ys = T[-2, 0], T[-2, 1]
        T[-1] = [0.0, 0.0, 0.0, T[-2, 3] + T[-2, 4],
                 np.sqrt((0 - xs) ** 2 + (0 - ys) ** 2), 0.0]
        self.PointAndTangent = T
        self.TrackLength = T[-1, 3] + T[-1, 4]


def curvature(s, table):
    """Piecewise-constant curvature at abscissa ``s`` (UTIL:31-50).

    Like the reference this fails when no segment contains ``s`` (s < 0, or
    s exactly at the end of a zero-length closing segment)."""
    L = table[-1, 3] + table[-1, 4]
    s = float(s)
    while s > L:
        s = s - L
    hit = np.nonzero((s >= table[:, 3]) & (s < table[:, 3] + table[:, 4]))[0]
    if hit.size != 1:
        raise ValueError("curvature(): abscissa %r is in %d segments" % (s, hit.size))
    return table[int(hit[0]), 5]


# --------------------------------------------------------------------------
# controller: A(rho), B(rho)   (CTRL:203-246 and CTRL:760-803)
# --------------------------------------------------------------------------
def ctrl_AB(p, dt, vx, vy, epsi, ey, cur, delta, Cf=None, Cr=None):
    """Discrete (forward-Euler) 6x6 A and 6x2 B of the controller model.
    States [vx vy wz epsi s ey], inputs [delta a]."""
    lf, lr, m, Iz, mu = p["lf"], p["lr"], p["m"], p["Iz"], p["mu"]
    Cf = p["Cf"] if Cf is None else Cf
    Cr = p["Cr"] if Cr is None else Cr
    sd, cd = np.sin(delta), np.cos(delta)
    se, ce = np.sin(epsi), np.cos(epsi)
    den = 1 - ey * cur
    Ac = np.zeros((6, 6))
    Ac[0, 0] = -mu
    Ac[0, 1] = (sd * Cf) / (m * vx)
    Ac[0, 2] = (sd * Cf * lf) / (m * vx) + vy
    Ac[1, 1] = -(Cr + Cf * cd) / (m * vx)
    Ac[1, 2] = -(lf * Cf * cd - lr * Cr) / (m * vx) - vx
    Ac[2, 1] = -(lf * Cf * cd - lr * Cr) / (Iz * vx)
    Ac[2, 2] = -(lf * lf * Cf * cd + lr * lr * Cr) / (Iz * vx)
    Ac[3, 0] = (1 / den) * (-ce * cur)
    Ac[3, 1] = (1 / den) * (+se * cur)
    Ac[3, 2] = 1.0
    Ac[4, 0] = ce / den
    Ac[4, 1] = se / den
    Ac[5, 0] = se
    Ac[5, 1] = ce
    Bc = np.zeros((6, 2))
    Bc[0, 0] = -(sd * Cf) / m
    Bc[0, 1] = 1.0
    Bc[1, 0] = (cd * Cf) / m
    Bc[2, 0] = (lf * Cf * cd) / Iz
    return np.eye(6) + dt * Ac, dt * Bc


def ctrl_lpv_prediction(p, dt, N, table, x, u, vel_ref, curv_ref, Cf_new, lap):
    """CTRL:166-258.  Returns STATES_vec (N,6), A (N,6,6), B (N,6,2).

    Scheduling: vy, epsi, s, ey from the rolled-out state; vx from
    ``vel_ref[i]`` (quirk Q5, CTRL:200); curvature from the map on lap 0, else
    ``curv_ref[i]`` (CTRL:193-198); Cf = Cr = Cf_new (CTRL:172-173)."""
    st = np.asarray(x, dtype=float).reshape(6).copy()
    u = np.asarray(u, dtype=float)
    S = np.zeros((N, 6))
    A = np.zeros((N, 6, 6))
    B = np.zeros((N, 6, 2))
    for i in range(N):
        vy, epsi, s, ey = st[1], st[3], st[4], st[5]
        cur = curvature(s, table) if lap == 0 else float(curv_ref[i])
        A[i], B[i] = ctrl_AB(p, dt, float(vel_ref[i]), vy, epsi, ey, cur,
                             float(u[i, 0]), Cf=Cf_new, Cr=Cf_new)
        st = A[i] @ st + B[i] @ u[i, :]
        S[i] = st
    return S, A, B


def ctrl_estimate_abc(p, dt, N, table, xlast, upred):
    """CTRL:732-809 (seed-mode linearisation along a given trajectory)."""
    xlast = np.asarray(xlast, dtype=float)
    upred = np.asarray(upred, dtype=float)
    A = np.zeros((N, 6, 6))
    B = np.zeros((N, 6, 2))
    for i in range(N):
        cur = curvature(xlast[i, 4], table)
        A[i], B[i] = ctrl_AB(p, dt, xlast[i, 0], xlast[i, 1], xlast[i, 3],
                             xlast[i, 5], cur, upred[i, 0])
    return A, B


# --------------------------------------------------------------------------
# planner: A(rho), B(rho)   (PLAN:275-308 and PLAN:551-585)
# --------------------------------------------------------------------------
def plan_AB(p, dt, vx, vy, ey, epsi, cur, delta):
    """5x5 / 5x2 planner model, states [vx vy wz ey epsi]."""
    lf, lr, m, Iz, mu = p["lf"], p["lr"], p["m"], p["Iz"], p["mu"]
    Cf, Cr = p["Cf"], p["Cr"]
    sd, cd = np.sin(delta), np.cos(delta)
    A1 = 1 / (1 - ey * cur)
    A2 = np.sin(epsi)
    Ac = np.zeros((5, 5))
    Ac[0, 0] = -mu
    Ac[0, 1] = (sd * Cf) / (m * vx)
    Ac[0, 2] = (sd * Cf * lf) / (m * vx) + vy
    Ac[1, 1] = -(Cr + Cf * cd) / (m * vx)
    Ac[1, 2] = -(lf * Cf * cd - lr * Cr) / (m * vx) - vx
    Ac[2, 1] = -(lf * Cf * cd - lr * Cr) / (Iz * vx)
    Ac[2, 2] = -(lf * lf * Cf * cd + lr * lr * Cr) / (Iz * vx)
    Ac[3, 1] = 1.0
    Ac[3, 4] = vx
    Ac[4, 0] = -A1 * cur
    Ac[4, 1] = A1 * A2 * cur
    Ac[4, 2] = 1.0
    Bc = np.zeros((5, 2))
    Bc[0, 0] = -(sd * Cf) / m
    Bc[0, 1] = 1.0
    Bc[1, 0] = (cd * Cf) / m
    Bc[2, 0] = (lf * Cf * cd) / Iz
    return np.eye(5) + dt * Ac, dt * Bc


def plan_lpv_prediction(p, dt, N, table, x, SS, u):
    """PLAN:242-320.  All scheduling variables from the rolled-out state,
    curvature from the map at SS[i]."""
    st = np.asarray(x, dtype=float).reshape(5).copy()
    u = np.asarray(u, dtype=float)
    S = np.zeros((N, 5))
    A = np.zeros((N, 5, 5))
    B = np.zeros((N, 5, 2))
    for i in range(N):
        cur = curvature(SS[i], table)
        A[i], B[i] = plan_AB(p, dt, st[0], st[1], st[3], st[4], cur, float(u[i, 0]))
        st = A[i] @ st + B[i] @ u[i, :]
        S[i] = st
    return S, A, B


def plan_estimate_abc(p, dt, N, table, xx, uu):
    """PLAN:519-591.  ``xx`` columns [vx vy wz ey epsi s]; ``uu[i]`` = delta."""
    xx = np.asarray(xx, dtype=float)
    uu = np.asarray(uu, dtype=float).reshape(-1)
    A = np.zeros((N, 5, 5))
    B = np.zeros((N, 5, 2))
    for i in range(N):
        cur = curvature(xx[i, 5], table)
        A[i], B[i] = plan_AB(p, dt, xx[i, 0], xx[i, 1], xx[i, 3], xx[i, 4], cur, uu[i])
    return A, B


# --------------------------------------------------------------------------
# QP assembly (dense, float64), in the exact row/column order handed to OSQP
# --------------------------------------------------------------------------
def slew_hessian(R, dR, N):
    """Input block Mu of the cost (CTRL:401-425 / PLAN:148-158): block
    tridiagonal, diagonal R+2diag(dR) (last block R+diag(dR)), off-diagonals
    -diag(dR)."""
    nu = R.shape[0]
    Mu = np.zeros((nu * N, nu * N))
    for k in range(N):
        blk = R + (2.0 if k < N - 1 else 1.0) * np.diag(dR)
        Mu[k * nu:(k + 1) * nu, k * nu:(k + 1) * nu] = blk
        if k + 1 < N:
            for j in range(nu):
                Mu[k * nu + j, (k + 1) * nu + j] = -dR[j]
                Mu[(k + 1) * nu + j, k * nu + j] = -dR[j]
    return Mu


def eq_constraints(A, B, nx, nu, N):
    """G z = E x0 (+L, L == 0).  CTRL:477-529 / PLAN:434-486 (delay = 0)."""
    nz = (N + 1) * nx + N * nu
    G = np.zeros(((N + 1) * nx, nz))
    G[:, :(N + 1) * nx] = np.eye((N + 1) * nx)
    for k in range(N):
        r = slice((k + 1) * nx, (k + 2) * nx)
        G[r, k * nx:(k + 1) * nx] = -A[k]
        G[r, (N + 1) * nx + k * nu:(N + 1) * nx + (k + 1) * nu] = -B[k]
    E = np.zeros(((N + 1) * nx, nx))
    E[:nx] = np.eye(nx)
    return G, E


@dataclass
class QP:
    P: np.ndarray
    q: np.ndarray
    A: np.ndarray
    l: np.ndarray
    u: np.ndarray
    meta: dict = field(default_factory=dict)


def ctrl_build_qp(Q, R, dR, N, A, B, x0, u_old, vel_ref, max_vel, steer_hist=()):
    """QP of PathFollowingLPV_MPC.solve (CTRL:89-162 with a4/a5/a6/a8 of
    SURVEY section 8): inequalities first, then equalities (CTRL:303-308).

    ``u_old`` = [OldSteering[0], OldAccelera[0]] (CTRL:395).  The vel_ref
    tracking point of stage N is vel_ref[-1] (CTRL:438).  ``steer_hist`` =
    OldSteering[1:] (length steeringDelay): one equality row per entry, appended
    after the dynamics rows, pins u_i[0] = OldSteering[i+1] (CTRL:518-527)."""
    nx, nu = 6, 2
    Q = np.asarray(Q, float); R = np.asarray(R, float); dR = np.asarray(dR, float)
    vel_ref = np.asarray(vel_ref, float).reshape(-1)
    nz = (N + 1) * nx + N * nu
    M0 = np.zeros((nz, nz))
    for k in range(N + 1):
        M0[k * nx:(k + 1) * nx, k * nx:(k + 1) * nx] = Q
    M0[(N + 1) * nx:, (N + 1) * nx:] = slew_hessian(R, dR, N)
    xtrack = np.zeros(nz)
    for k in range(N):
        xtrack[k * nx] = vel_ref[k]
    xtrack[N * nx] = vel_ref[-1]
    q = -2.0 * (xtrack @ M0)
    q[(N + 1) * nx:(N + 1) * nx + nu] = -2.0 * (np.asarray(u_old, float) * dR)
    P = 2.0 * M0
    # inequality rows (CTRL:329-378): 2N state rows then 4N input rows
    F = np.zeros((6 * N, nz))
    b = np.zeros(6 * N)
    for k in range(N):
        F[2 * k, k * nx] = -1.0;      b[2 * k] = -0.01
        F[2 * k + 1, k * nx] = 1.0;   b[2 * k + 1] = max_vel
        c = (N + 1) * nx + k * nu
        r = 2 * N + 4 * k
        F[r, c] = 1.0;        b[r] = 0.249
        F[r + 1, c] = -1.0;   b[r + 1] = 0.249
        F[r + 2, c + 1] = 1.0;  b[r + 2] = 4.0
        F[r + 3, c + 1] = -1.0; b[r + 3] = 1.0
    G, E = eq_constraints(A, B, nx, nu, N)
    beq = E @ np.asarray(x0, float).reshape(nx)          # quirk Q1: Eu*uOld dropped
    steer_hist = np.asarray(steer_hist, float).reshape(-1)
    if steer_hist.size:                                  # CTRL:518-527: Gdelay rows, L = OldSteering[i+1]
        Gd = np.zeros((steer_hist.size, nz))
        for i in range(steer_hist.size):
            Gd[i, (N + 1) * nx + i * nu] = 1.0
        G = np.vstack([G, Gd]); beq = np.concatenate([beq, steer_hist])
    Aqp = np.vstack([F, G])
    l = np.concatenate([-np.inf * np.ones(6 * N), beq])
    u = np.concatenate([b, beq])
    return QP(P, q, Aqp, l, u, dict(kind="controller", N=N, nx=nx, nu=nu))


def plan_build_qp(Q, R, dR, L_cf, N, A, B, x0, u_old, max_ey, max_vel, min_vel):
    """QP of LPV_MPC_Planner.solve (PLAN:86-236): equalities first, then the
    identity box on every variable including x_0 (PLAN:173-181,200-202)."""
    nx, nu = 5, 2
    Q = np.asarray(Q, float); R = np.asarray(R, float); dR = np.asarray(dR, float)
    L_cf = np.asarray(L_cf, float)
    nz = (N + 1) * nx + N * nu
    M0 = np.zeros((nz, nz))
    for k in range(N + 1):
        M0[k * nx:(k + 1) * nx, k * nx:(k + 1) * nx] = Q
    M0[(N + 1) * nx:, (N + 1) * nx:] = slew_hessian(R, dR, N)
    q = np.concatenate([np.tile(L_cf, N + 1), np.zeros(N * nu)])
    q[(N + 1) * nx:(N + 1) * nx + nu] = -2.0 * (np.asarray(u_old, float) * dR)
    P = 2.0 * M0
    G, E = eq_constraints(A, B, nx, nu, N)
    beq = E @ np.asarray(x0, float).reshape(nx)
    umin = np.array([-0.249, -0.7]); umax = np.array([0.249, 2.0])
    xmin = np.array([min_vel, -1, -2, -max_ey, -0.8])
    xmax = np.array([max_vel, 1, 2, max_ey, 0.8])
    lo = np.concatenate([np.tile(xmin, N + 1), np.tile(umin, N)])
    hi = np.concatenate([np.tile(xmax, N + 1), np.tile(umax, N)])
    Aqp = np.vstack([G, np.eye(nz)])
    return QP(P, q, Aqp, np.concatenate([beq, lo]), np.concatenate([beq, hi]),
              dict(kind="planner", N=N, nx=nx, nu=nu))


def unpack_solution(z, nx, nu, N):
    """xPred (N+1,nx), uPred (N,nu), LinPoints (N+1,nx)  (CTRL:157-162)."""
    z = np.asarray(z, float)
    xPred = z[:(N + 1) * nx].reshape(N + 1, nx).copy()
    uPred = z[(N + 1) * nx:(N + 1) * nx + N * nu].reshape(N, nu).copy()
    Lin = np.vstack([xPred[1:], xPred[-1:]])
    return xPred, uPred, Lin


# --------------------------------------------------------------------------
# caller-side seed trajectories (needed to reproduce the first ticks)
# --------------------------------------------------------------------------
def ctrl_seed_vectors(local_state):
    """controllerMain.py:510-553 (20 fixed rows)."""
    dvx = [0.05, 0.2, 0.4, 0.6, 0.7, 0.8] + [0.9] * 14
    ds = [0, 0.01, 0.02, 0.04, 0.07, 0.1, 0.14, 0.18, 0.23, 0.55, 0.66, 0.77,
          0.89, 1.00, 1.19, 1.39, 1.59, 1.79, 1.89, 1.999]
    ls = np.asarray(local_state, float)
    xx = np.array([[ls[0] + dvx[i], ls[1], ls[2], 0.0001, ls[4] + ds[i], 0.0001]
                   for i in range(20)])
    acc = [0.0, 0.3, 0.5, 0.7, 0.8, 0.9, 0.9, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4,
           0.30, 0.22, 0.18, 0.14, 0.1, 0.1, 0.1]
    uu = np.array([[0.0, a] for a in acc])
    return xx, uu


def plan_seed_vectors(Hp, x0, accel_rate, dt):
    """plannerMain.py:465-505.  Returns xx (Hp+1,6) = [vx vy wz ey epsi s] and
    uu (Hp,) zeros (passed 1-D: numpy >= 1.24 rejects the (Hp,1) form)."""
    x0 = np.asarray(x0, float)
    Vx = np.zeros(Hp + 1); S = np.zeros(Hp + 1)
    Vx[0] = x0[0]
    acc = 0.1 + accel_rate * np.arange(Hp)
    for i in range(Hp):
        Vx[i + 1] = Vx[i] + acc[i] * dt
        S[i + 1] = S[i] + ((Vx[i] * np.cos(x0[4]) - x0[1] * np.sin(x0[4])) / (1 - x0[3] * 0)) * dt
    xx = np.column_stack([Vx, np.full(Hp + 1, x0[1]), np.full(Hp + 1, x0[2]),
                          np.full(Hp + 1, x0[3]), np.full(Hp + 1, x0[4]), S])
    return xx, np.zeros(Hp)
