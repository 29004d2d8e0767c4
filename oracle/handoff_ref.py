"""Oracle (TEST INFRASTRUCTURE): numpy / scipy restatement of the planner -> controller reference hand-off
 and of the trajectory-tracking branch of the controller's main loop.

  planner_pose_refs     plannerMain.py:201-224   s integration, centre-line pose, XY / yaw reconstruction,
                                                 vel = vx, curv = wz / vx
  resample_refs         plannerMain.py:112,257-280   cubic interp1d 50 ms -> 33 ms, elliptic filtfilt(padlen=50) on curvature
  body_frame_errors     controllerMain.py:495-506
  TrackingGlue          controllerMain.py:198-283  (LapNumber >= 1: yaw unwinding, reference latch with the
                                                 `index` toggle, dead-reckoned s, racing lap counter)

scipy is the reference's own dependency for this step (interp1d, signal.ellip, signal.filtfilt), so the oracle
calls it directly; the product builds its own interpolation / filter operators (csrc/handoff.hip) and is compared
against this file.  Pinned by tests/golden/handoff.npz (generated with the reference's Map / Curvature functions).
"""
from __future__ import annotations

import numpy as np
from scipy import signal
from scipy.interpolate import interp1d

from .lpv_ref import curvature
from .plant_ref import get_global_position, _wrap

INTERP_DT = 0.033                                   # PMAIN:257
PADLEN = 50                                         # PMAIN:280


def ellip_coefficients():
    """PMAIN:112  b_filter, a_filter = signal.ellip(4, 0.01, 120, 0.125)"""
    return signal.ellip(4, 0.01, 120, 0.125)


def n_resampled(N, dt, interp_dt=INTERP_DT):
    """PMAIN:259 (np.around(N*dt/interp_dt); modern numpy needs the int())."""
    return int(np.around(N * dt / interp_dt))


def planner_pose_refs(table, xPred, SS, last_pose, dt):
    """PMAIN:201-224.  xPred (N+1,5) = [vx vy wz ey epsi]; SS (N+1,) carried between ticks (modified copy
    returned); last_pose = (Xlast, Ylast, Thetalast).  Returns (SS, last_pose, xp, yp, yaw, vel, curv)."""
    xPred = np.asarray(xPred, float)
    N = xPred.shape[0] - 1
    SS = np.array(SS, float)
    Xref = np.zeros(N + 1); Yref = np.zeros(N + 1); Th = np.zeros(N + 1)
    Xref[0], Yref[0], Th[0] = last_pose
    for j in range(N):
        cv = curvature(SS[j], table)
        SS[j + 1] = SS[j] + ((xPred[j, 0] * np.cos(xPred[j, 4]) - xPred[j, 1] * np.sin(xPred[j, 4]))
                             / (1 - xPred[j, 3] * cv)) * dt
        Xref[j + 1], Yref[j + 1], Th[j + 1] = get_global_position(table, SS[j + 1], 0.0)
    SS[0] = SS[1]
    last = (Xref[1], Yref[1], Th[1])
    yaw = Th[:N] + xPred[:N, 4]
    xp = Xref[:N] - xPred[:N, 3] * np.sin(yaw)
    yp = Yref[:N] + xPred[:N, 3] * np.cos(yaw)
    vel = xPred[:N, 0].copy()
    curv = xPred[:N, 2] / xPred[:N, 0]
    return SS, last, xp, yp, yaw, vel, curv


def resample_refs(xp, yp, yaw, vel, curv, dt, interp_dt=INTERP_DT, padlen=PADLEN, ba=None):
    """PMAIN:257-280 + :303-307: the five My_Planning arrays (x_d, y_d, psi_d, vx_d, curv_d), each of length
    n_resampled(N, dt).  Raises like scipy does when the resampled length is <= padlen (N < 34 at the
    reference's rates)."""
    N = len(xp)
    b, a = ba if ba is not None else ellip_coefficients()
    t50 = np.linspace(0, N * dt, num=N, endpoint=True)
    t33 = np.linspace(0, N * dt, num=n_resampled(N, dt, interp_dt), endpoint=True)
    out = [interp1d(t50, np.asarray(v, float), kind="cubic")(t33) for v in (xp, yp, yaw, vel, curv)]
    out[4] = signal.filtfilt(b, a, out[4], padlen=padlen)
    return np.array(out)


def body_frame_errors(x, y, psi, xd, yd, psid, s0, vx, vy, curv, dt):
    """CMAIN:495-506 -> (s, ex, ey, epsi)."""
    ex = (x - xd) * np.cos(psid) + (y - yd) * np.sin(psid)
    ey = -(x - xd) * np.sin(psid) + (y - yd) * np.cos(psid)
    epsi = _wrap(psi - psid)
    s = s0 + ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - ey * curv)) * dt
    return s, ex, ey, epsi


class TrackingGlue:
    """Per-vehicle caller state of the LapNumber >= 1 branch of controllerMain.py (one object per vehicle).

    measure(plant_state, refs) -> (LocalState(6), vel_ref(N), curv_ref(N)) follows CMAIN:176-182,198-248,268-283:
      * GlobalState[5] = wrap(psi - 2 pi LapNumber)
      * the reference vectors are re-read from the latest My_Planning message only when `index` == 0, and `index`
        toggles 0 -> 1 -> 0, so every second tick re-uses the previous tick's windows (max_window = 0)
      * s is dead-reckoned from SS with Body_Frame_Errors, SS <- s
      * racing lap counter: |X| < 0.1 and s >= 0.9 TrackLength  =>  LapNumber += 1, SS = 0
    """

    def __init__(self, N, dt, track_length, lap=1, SS=0.0):
        self.N, self.dt, self.L = int(N), float(dt), float(track_length)
        self.lap, self.SS, self.index = int(lap), float(SS), 0
        self.win = None

    def measure(self, plant, refs):
        N = self.N
        x, y, vx, vy, _, _, yaw, w = [float(v) for v in plant]
        G = np.array([vx, vy, w, x, y, yaw])
        Lc = G.copy()
        if Lc[0] < 0.01:
            Lc[0] = 0.01
        G[5] = _wrap(G[5] - 2 * np.pi * self.lap)
        if self.index <= 0:
            if self.index == 0:
                self.win = np.array(refs, float)[:, :N].copy()
            self.index += 1
        else:
            self.index = 0
        x_ref, y_ref, yaw_ref, vel_ref, curv_ref = self.win
        s, _ex, ey, epsi = body_frame_errors(G[3], G[4], G[5], x_ref[0], y_ref[0], yaw_ref[0], self.SS,
                                             Lc[0], Lc[1], curv_ref[0], self.dt)
        Lc[4], Lc[5], Lc[3] = s, ey, epsi
        self.SS = s
        if self.lap >= 1 and abs(G[3]) < 0.1 and Lc[4] >= self.L - self.L / 10:
            self.lap += 1
            self.SS = 0.0
        return Lc, vel_ref.copy(), curv_ref.copy()
