"""Oracle (TEST INFRASTRUCTURE): numpy restatement of the caller-side pieces needed for closed-loop runs
(SURVEY.md section 8f, row f1): the plant model and the track coordinate transforms.

  simulator_f           vehicleSimulator.py:164-199   (Simulator.f: bicycle model, linear tyres Fy = 60 alpha)
  get_global_position   Utilities/trackInitialization.py:205-262
  get_local_position    Utilities/trackInitialization.py:283-383  (+ computeAngle :393-411)
Pinned by tests/golden/plant_and_transforms.npz (generated from the reference's own classes).
"""
from __future__ import annotations

import math

import numpy as np

SIM_PARAMS = dict(lf=0.125, lr=0.125, m=1.98, Iz=0.03, mu=0.05, dt=0.005)     # MAIN_LAUNCH.launch:5-11,60,72


def simulator_f(st, u, p=SIM_PARAMS):
    """One Euler step of Simulator.f.  st = [x y vx vy ax ay yaw psiDot], u = [a, delta]."""
    x, y, vx, vy, ax, ay, yaw, w = st
    lf, lr, m, Iz, mu, dt = p["lf"], p["lr"], p["m"], p["Iz"], p["mu"], p["dt"]
    aF = aR = 0.0
    if abs(vx) > 0.2:
        aF = u[1] - np.arctan((vy + lf * w) / abs(vx))
        aR = np.arctan((-vy + lr * w) / abs(vx))
    FyF, FyR = 60 * aF, 60 * aR
    nx_ = x + dt * (np.cos(yaw) * vx - np.sin(yaw) * vy)
    ny_ = y + dt * (np.sin(yaw) * vx + np.cos(yaw) * vy)
    nvx = vx + dt * (ax + w * vy)
    nvy = vy + dt * (ay - w * vx)
    nax = u[0] - mu * vx - FyF / m * np.sin(u[1])
    nay = 1.0 / m * (FyF * np.cos(u[1]) + FyR)
    nyaw = yaw + dt * w
    nw = w + dt * (1.0 / Iz * (lf * FyF * np.cos(u[1]) - lr * FyR))
    return np.array([nx_, ny_, abs(nvx), nvy, nax, nay, nyaw, nw])


def _wrap(a):
    if a < -np.pi:
        return 2 * np.pi + a
    if a > np.pi:
        return a - 2 * np.pi
    return a


def _sgn(a):
    return 1 if a >= 0 else -1


def _angle(p1, o, p2):
    v1 = np.array(p1, float) - np.array(o, float); v2 = np.array(p2, float) - np.array(o, float)
    return np.arctan2(v1[0] * v2[1] - v1[1] * v2[0], v1[0] * v2[0] + v1[1] * v2[1])


def get_global_position(table, s, ey):
    """(s, ey) -> (x, y, theta).  Fails like the reference when no segment contains s."""
    L = table[-1, 3] + table[-1, 4]
    while s > L:
        s = s - L
    hit = np.nonzero((s >= table[:, 3]) & (s < table[:, 3] + table[:, 4]))[0]
    if hit.size != 1:
        raise ValueError("abscissa outside the table")
    i = int(hit[0])
    if table[i, 5] == 0.0:
        xf, yf, xs, ys, psi = table[i, 0], table[i, 1], table[i - 1, 0], table[i - 1, 1], table[i, 2]
        dL = table[i, 4]; rL = s - table[i, 3]
        x = (1 - rL / dL) * xs + rL / dL * xf + ey * np.cos(psi + np.pi / 2)
        y = (1 - rL / dL) * ys + rL / dL * yf + ey * np.sin(psi + np.pi / 2)
        return x, y, psi
    r = 1 / table[i, 5]; ang = table[i - 1, 2]
    d = 1 if r >= 0 else -1
    cx = table[i - 1, 0] + np.abs(r) * np.cos(ang + d * np.pi / 2)
    cy = table[i - 1, 1] + np.abs(r) * np.sin(ang + d * np.pi / 2)
    span = (s - table[i, 3]) / (np.pi * np.abs(r)) * np.pi
    an = _wrap(d * np.pi / 2 + ang)
    a0 = -(np.pi - np.abs(an)) * _sgn(an)
    x = cx + (np.abs(r) - d * ey) * np.cos(a0 + d * span)
    y = cy + (np.abs(r) - d * ey) * np.sin(a0 + d * span)
    return x, y, ang + d * span


def get_local_position(table, half_width, slack, x, y, psi):
    """(x, y, psi) -> (s, ey, epsi, inside) with the reference's 10000 sentinels when off the track."""
    done = 0
    s = ey = epsi = 0.0
    for i in range(table.shape[0]):
        if done:
            break
        xf, yf, xs, ys = table[i, 0], table[i, 1], table[i - 1, 0], table[i - 1, 1]
        if table[i, 5] == 0.0:
            pu = np.unwrap([table[i - 1, 2], psi])[1]
            epsi = pu - table[i - 1, 2]
            if np.hypot(xs - x, ys - y) == 0:
                s, ey, done = table[i, 3], 0, 1
            elif np.hypot(xf - x, yf - y) == 0:
                s, ey, done = table[i, 3] + table[i, 4], 0, 1
            elif abs(_angle([x, y], [xs, ys], [xf, yf])) <= np.pi / 2 and abs(_angle([x, y], [xf, yf], [xs, ys])) <= np.pi / 2:
                v1 = np.array([x, y]) - np.array([xs, ys])
                a = _angle([xf, yf], [xs, ys], [x, y])
                s = np.linalg.norm(v1) * np.cos(a) + table[i, 3]
                ey = np.linalg.norm(v1) * np.sin(a)
                if abs(ey) <= half_width + slack:
                    done = 1
        else:
            r = 1 / table[i, 5]
            d = 1 if r >= 0 else -1
            ang = table[i - 1, 2]
            cx = xs + np.abs(r) * np.cos(ang + d * np.pi / 2)
            cy = ys + np.abs(r) * np.sin(ang + d * np.pi / 2)
            if np.hypot(xs - x, ys - y) == 0:
                ey = 0; epsi = np.unwrap([ang, psi])[1] - ang; s = table[i, 3]; done = 1
            elif np.hypot(xf - x, yf - y) == 0:
                s = table[i, 3] + table[i, 4]; ey = 0
                epsi = np.unwrap([table[i, 2], psi])[1] - table[i, 2]; done = 1
            else:
                arc1 = table[i, 4] * table[i, 5]
                arc2 = _angle([xs, ys], [cx, cy], [x, y])
                if np.sign(arc1) == np.sign(arc2) and abs(arc1) >= abs(arc2):
                    v = np.array([x, y]) - np.array([cx, cy])
                    s = abs(arc2) * abs(r) + table[i, 3]
                    ey = -np.sign(d) * (np.linalg.norm(v) - abs(r))
                    epsi = np.unwrap([ang + arc2, psi])[1] - (ang + arc2)
                    if abs(ey) <= half_width + slack:
                        done = 1
    if not done:
        return 10000, 10000, 10000, 0
    return s, ey, epsi, 1
