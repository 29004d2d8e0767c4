"""Oracle (TEST INFRASTRUCTURE): the planner + controller + plant cascade of BASELINE.json configs[4] for a fleet,
assembled from the other oracle pieces (C tick functions of oracle/lpv_ref.c + oracle/osqp_ref.c, the hand-off of
oracle/handoff_ref.py, the plant of oracle/plant_ref.py).  Same synchronous schedule as the fixture
tests/golden/cascade.npz (generated with the reference's classes): controller tick k is preceded by planner ticks
0 .. floor(2k/3); the plant takes n_sub[k % 3] steps of 5 ms after each controller tick.

Follows plannerMain.py:126-224,257-308 and controllerMain.py:176-298,337-386 (LapNumber >= 1).
"""
from __future__ import annotations

import numpy as np

from . import handoff_ref as H, lpv_ref as L, osqp_ref, plant_ref as PR


class CascadeRef:
    def __init__(self, track, ctrl_tuning, plan_weights, plant0, cmd0, uPred0, lap0=1, half_width=0.3, slack=0.15, plan_max_ey=0.2, q9_swap=True,
                 Nc=20, dtc=1.0 / 30, Np=40, dtp=0.05, n_sub=(7, 7, 6), nthreads=1):
        self.track = np.asarray(track, float)
        self.TL = float(self.track[-1, 3] + self.track[-1, 4])
        self.Qc, self.Rc, self.dRc = ctrl_tuning
        self.Qp, self.Rp, self.dRp, self.Lp = plan_weights
        self.plant = np.array(plant0, float).reshape(-1, 8)
        self.B = self.plant.shape[0]
        self.cmd = np.array(cmd0, float).reshape(self.B, 2)
        self.uPred = np.array(uPred0, float).reshape(self.B, Nc, 2)
        self.hw, self.slack, self.max_ey, self.q9 = half_width, slack, plan_max_ey, q9_swap
        self.Nc, self.dtc, self.Np, self.dtp, self.n_sub, self.nthreads = Nc, dtc, Np, dtp, tuple(n_sub), nthreads
        self.glue = [H.TrackingGlue(Nc, dtc, self.TL, lap=lap0) for _ in range(self.B)]
        self.k = 0
        self.plan_ticks = 0
        self.SS = np.zeros((self.B, Np + 1)); self.pose = np.zeros((self.B, 3))
        self.pxPred = self.puPred = self.refs = None
        self.local = np.zeros((self.B, 6))
        self.ctrl = self.plan = None

    # ---- planner node ----
    def planner_tick(self):
        B, Np, dtp = self.B, self.Np, self.dtp
        if self.plan_ticks == 0:
            xP = np.empty((B, Np + 1, 5)); uP = np.empty((B, Np, 2)); it = np.empty(B, np.int32); stt = np.empty(B, np.int32)
            for b in range(B):
                p = self.plant[b]
                s, ey, epsi, _ = PR.get_local_position(self.track, self.hw, self.slack, p[0], p[1], p[6])
                x0 = np.array([p[2], p[3], p[7], epsi, ey] if self.q9 else [p[2], p[3], p[7], ey, epsi], float)
                xx, uu = L.plan_seed_vectors(Np, x0, 0.2, dtp)
                A, Bm = L.plan_estimate_abc(L.DEFAULT_PARAMS, dtp, Np, self.track, xx, uu)
                qp = L.plan_build_qp(self.Qp, self.Rp, self.dRp, self.Lp, Np, A, Bm, x0, np.zeros(2), self.max_ey,
                                     L.DEFAULT_PARAMS["max_vel"], L.DEFAULT_PARAMS["min_vel"])
                r = osqp_ref.solve_qp(qp.P, qp.q, qp.A, qp.l, qp.u)
                xP[b], uP[b], _lin = L.unpack_solution(r.x, 5, 2, Np)
                it[b], stt[b] = r.info.iter, r.info.status_val
            self.plan = dict(xPred=xP, uPred=uP, iters=it, status=stt)
        else:
            w = dict(N=Np, dt=dtp, Q=self.Qp, R=self.Rp, dR=self.dRp, L_cf=self.Lp, track=self.track, x0=self.pxPred[:, 1, :].copy(),
                     u_prev=self.puPred, curv_s=self.SS, u_old=np.zeros((B, 2)), max_ey=np.full(B, self.max_ey))
            self.plan = osqp_ref.plan_tick_batch(w, nthreads=self.nthreads)
        self.pxPred, self.puPred = self.plan["xPred"], self.plan["uPred"]
        refs = []
        M = H.n_resampled(Np, dtp)
        for b in range(B):
            try:
                SS, last, xp, yp, yaw, vel, curv = H.planner_pose_refs(self.track, self.pxPred[b], self.SS[b], tuple(self.pose[b]), dtp)
                self.SS[b] = SS; self.pose[b] = last
                refs.append(H.resample_refs(xp, yp, yaw, vel, curv, dtp))
            except ValueError:
                # NaN plan (an infeasible planner QP): the reference's node would publish NaN from here on; this vehicle is
                # lost, the others go on (what the device engine does, tests/test_gpu_cascade.py)
                self.SS[b] = np.nan; self.pose[b] = np.nan
                refs.append(np.full((5, M), np.nan))
        self.refs = np.array(refs)
        self.plan_ticks += 1

    # ---- controller node + plant ----
    def tick(self):
        B, Nc = self.B, self.Nc
        while self.plan_ticks < (2 * self.k) // 3 + 1:
            self.planner_tick()
        vel = np.empty((B, Nc + 1)); curv = np.empty((B, Nc))
        for b in range(B):
            self.local[b], v, c = self.glue[b].measure(self.plant[b], self.refs[b])
            vel[b, :Nc] = v; vel[b, Nc] = v[-1]; curv[b] = c
        w = dict(N=Nc, dt=self.dtc, Q=self.Qc, R=self.Rc, dR=self.dRc, track=self.track, x0=self.local.copy(), u_prev=self.uPred,
                 vel_ref=vel, curv_s=curv, u_old=self.cmd.copy(), cf_new=60.0, lap=1)
        self.ctrl = osqp_ref.ctrl_tick_batch(w, nthreads=self.nthreads)
        self.uPred = self.ctrl["uPred"]
        self.cmd = self.uPred[:, 0, :].copy()
        for b in range(B):
            st = self.plant[b]
            for _ in range(self.n_sub[self.k % 3]):
                st = PR.simulator_f(st, [self.cmd[b, 1], self.cmd[b, 0]])
            self.plant[b] = st
        self.k += 1

    @property
    def lap(self):
        return np.array([g.lap for g in self.glue])
