"""Track table for callers without the reference's ROS environment.

Mirror of ``Map.__init__`` (reference Utilities/trackInitialization.py:13-202): a track is a list of
(length, signed radius) segments turned into rows ``[x, y, psi, cum_s, seg_len, curvature]``
(``PointAndTangent``).  Only the table feeds the solve path (curvature lookup on the device,
csrc/lpvmpc_device.hpp ``track_curvature`` == Utilities/utilities.py:31-50); the coordinate
transforms (getGlobalPosition / getLocalPosition) are caller-side and out of this package's scope.

Any object exposing ``.PointAndTangent`` and ``.halfWidth`` -- e.g. the reference's own ``Map`` -- can be
passed to the drop-in classes instead.
"""
from __future__ import annotations

import math

import numpy as np

_PI = np.pi

TRACK_SPECS = {
    # shape: (segments as (length, radius; 0 = straight), fixed halfWidth or None, slack)   TRACK:28-81
    "oval": ([(1.0, 0), (4.5, 4.5 / _PI), (2.0, 0), (4.5, 4.5 / _PI), (1.0, 0)], None, 0.15),
    "L_shape": ([(1.0, 0), (4.5, 4.5 / _PI), (4.5 / 2, -4.5 / _PI), (4.5, 4.5 / _PI), (4.5 / _PI * 2, 0),
                 (4.5 / 2, 4.5 / _PI)], None, 0.45),
    "3110": ([(60 * 0.03, 0), (80 * 0.03, 80 * 0.03 * 2 / _PI), (20 * 0.03, 0), (80 * 0.03, 80 * 0.03 * 2 / _PI),
              (40 * 0.03, -40 * 0.03 * 10 / _PI), (60 * 0.03, 60 * 0.03 * 5 / _PI),
              (40 * 0.03, -40 * 0.03 * 10 / _PI), (80 * 0.03, 80 * 0.03 * 2 / _PI), (20 * 0.03, 0),
              (80 * 0.03, 80 * 0.03 * 2 / _PI), (80 * 0.03, 0)], 0.6, 0.15),
    "Euge_Track": ([(30 * 0.03, 30 * 0.03 * 2 / _PI), (20 * 0.03, 0), (30 * 0.03, -30 * 0.03 * 2 / _PI),
                    (30 * 0.03, 30 * 0.03 * 2 / _PI), (30 * 0.03, 30 * 0.03 * 2 / _PI), (130 * 0.03, 0),
                    (30 * 0.03, 30 * 0.03 * 2 / _PI), (10 * 0.03, 0), (30 * 0.03, 30 * 0.03 * 2 / _PI),
                    (55 * 0.03, 0), (30 * 0.03, -30 * 0.03 * 2 / _PI), (10 * 0.03, 0),
                    (30 * 0.03, 30 * 0.03 * 2 / _PI)], 0.4, 0.15),
}


def _wrap_pi(a):
    if a < -math.pi:
        return a + 2 * math.pi
    if a > math.pi:
        return a - 2 * math.pi
    return a


def build_table(segments):
    """Rows [x y psi cum_s len kappa] for each segment plus the closing row back to the origin."""
    n = len(segments)
    tab = np.zeros((n + 1, 6))
    px = py = heading = 0.0
    s_acc = 0.0
    for i, (length, radius) in enumerate(segments):
        if i > 0:
            px, py, heading = tab[i - 1, 0], tab[i - 1, 1], tab[i - 1, 2]
            s_acc = tab[i - 1, 3] + tab[i - 1, 4]
        if radius == 0:
            tab[i] = (px + length * np.cos(heading), py + length * np.sin(heading), heading, s_acc, length, 0.0)
            continue
        turn = 1 if radius >= 0 else -1
        rad = np.abs(radius)
        cx = px + rad * np.cos(heading + turn * _PI / 2)
        cy = py + rad * np.sin(heading + turn * _PI / 2)
        sweep = length / rad
        normal = _wrap_pi(turn * _PI / 2 + heading)
        start = -(_PI - np.abs(normal)) * (1 if normal >= 0 else -1)
        tab[i] = (cx + rad * np.cos(start + turn * sweep), cy + rad * np.sin(start + turn * sweep),
                  _wrap_pi(heading + sweep * np.sign(radius)), s_acc, length, 1 / radius)
    gap = np.sqrt((0 - tab[-2, 0]) ** 2 + (0 - tab[-2, 1]) ** 2)
    tab[-1] = (0.0, 0.0, 0.0, tab[-2, 3] + tab[-2, 4], gap, 0.0)
    return tab


class Map:
    """``Map(shape, halfWidth_param)``: ``.PointAndTangent``, ``.TrackLength``, ``.halfWidth``, ``.slack``.

    ``halfWidth_param`` plays the role of the ROS parameter /TrajectoryPlanner/halfWidth (the reference
    adds 0.1 to it, TRACK:20)."""

    def __init__(self, shape="oval", halfWidth_param=0.2):
        if shape not in TRACK_SPECS:
            raise ValueError("unknown track shape %r (have %s)" % (shape, sorted(TRACK_SPECS)))
        segments, fixed_hw, slack = TRACK_SPECS[shape]
        self.shape = shape
        self.slack = slack
        self.halfWidth = fixed_hw if fixed_hw is not None else halfWidth_param + 0.1
        self.PointAndTangent = build_table(segments)
        self.TrackLength = self.PointAndTangent[-1, 3] + self.PointAndTangent[-1, 4]
        self._eng = None

    # -- coordinate transforms of the reference's Map, evaluated by the device kernels (one point per call) ----------
    def _engine(self):
        if self._eng is None:
            from .api import BatchedSolver                  # raises LpvMpcError without a HIP device: no CPU fallback
            self._eng = BatchedSolver("controller", 8, 1.0 / 30.0, np.eye(6), np.eye(2), np.ones(2), track=self.PointAndTangent)
        return self._eng

    def getGlobalPosition(self, s, ey):
        """TRACK:205-262: (s, ey) -> (x, y, theta)."""
        x, y, th = self._engine().global_position(np.array([[float(s), float(ey)]]))[0]
        return float(x), float(y), float(th)

    def getLocalPosition(self, x, y, psi):
        """TRACK:283-383: (x, y, psi) -> (s, ey, epsi, insideTrack); 10000 sentinels when the point is off the track."""
        s, ey, epsi, inside = self._engine().local_position(np.array([[float(x), float(y), float(psi)]]), self.halfWidth, self.slack)[0]
        return float(s), float(ey), float(epsi), int(inside)
