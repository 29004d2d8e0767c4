"""ROS node shims: the loop bodies of the reference's two nodes written against the
drop-in classes, so that ``roslaunch barc MAIN_LAUNCH.launch`` keeps working with ``ros/controllerMain.py`` and
``ros/plannerMain.py`` of this repository in place of the reference's scripts.

  ControllerNode.step   controllerMain.py:176-437   (measurement, lap logic, command publication, LPV-MPC tick,
                                                     open-loop prediction and Racing_Info messages)
  PlannerNode.step      plannerMain.py:126-308      (initial state, LPV-MPP tick, hand-off, My_Planning message)

The two classes hold the loop state and take / return plain Python values, so they can be driven without ROS (the GPU
tests do that against tests/golden/cascade.npz); ``controller_main`` / ``planner_main`` wire them to rospy with the
reference's topic names and message types (ecu: ECU, OL_predictions: prediction, Racing_Info, My_Planning, pos_info).
Logging to .dat files, plotting and the offline .mat references of the reference scripts are not reproduced.
"""
from __future__ import annotations

import numpy as np

from .api import LPV_MPC_Planner, PathFollowingLPV_MPC, PlannerHandoff, body_frame_errors
from .workloads import CTRL_TUNINGS, PLAN_L, PLAN_Q, PLAN_R, PLAN_dR

# predicted_vectors_generation of controllerMain.py:510-553: 20 fixed rows on top of the local state, steering seeds zero
_SEED_DVX = (0.05, 0.2, 0.4, 0.6, 0.7, 0.8) + (0.9,) * 14
_SEED_DS = (0, 0.01, 0.02, 0.04, 0.07, 0.1, 0.14, 0.18, 0.23, 0.55, 0.66, 0.77, 0.89, 1.00, 1.19, 1.39, 1.59, 1.79, 1.89, 1.999)
_SEED_ACC = (0.0, 0.3, 0.5, 0.7, 0.8, 0.9, 0.9, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.30, 0.22, 0.18, 0.14, 0.1, 0.1, 0.1)


def controller_seed(local_state):
    ls = np.asarray(local_state, float)
    xx = np.array([[ls[0] + dv, ls[1], ls[2], 0.0001, ls[4] + ds, 0.0001] for dv, ds in zip(_SEED_DVX, _SEED_DS)])
    uu = np.array([[0.0, a] for a in _SEED_ACC])
    return xx, uu


def planner_seed(N, x0, accel_rate, dt):
    """predicted_vectors_generation of plannerMain.py:465-505 -> xx (N+1, 6) = [vx vy wz ey epsi s], uu (N,) zeros."""
    x0 = np.asarray(x0, float)
    vx = np.zeros(N + 1); s = np.zeros(N + 1)
    vx[0] = x0[0]
    for i in range(N):
        vx[i + 1] = vx[i] + (0.1 + accel_rate * i) * dt
        s[i + 1] = s[i] + ((vx[i] * np.cos(x0[4]) - x0[1] * np.sin(x0[4])) / (1 - x0[3] * 0)) * dt
    xx = np.column_stack([vx] + [np.full(N + 1, x0[k]) for k in (1, 2, 3, 4)] + [s])
    return xx, np.zeros(N)


def _wrap(a):
    return 2 * np.pi + a if a < -np.pi else (a - 2 * np.pi if a > np.pi else a)


class ControllerNode(object):
    """Loop state and one iteration of the controller node."""

    def __init__(self, track_map, N, dt=1.0 / 30.0, number_of_laps=10, params=None, device=0):
        Qp, Rp, dRp = CTRL_TUNINGS["path"]; Qr, Rr, dRr = CTRL_TUNINGS["race"]         # CMAIN:142-150
        self.map, self.N, self.dt = track_map, int(N), float(dt)
        self.Controller = PathFollowingLPV_MPC(Qp, Rp, dRp, N, 1, dt, track_map, "OSQP", 0, 0, params=params, device=device)
        self.Controller_TT = PathFollowingLPV_MPC(Qr, Rr, dRr, N, 1, dt, track_map, "OSQP", 0, 0, params=params, device=device)
        self.first_it, self.LapNumber, self.HalfTrack, self.SS, self.index = 1, 0, 0, 0.0, 0
        self.NumberOfLaps, self.run = int(number_of_laps), True
        self.servo = self.motor = 0.0
        self.Cf_new = 60
        self.vel_ref = self.curv_ref = self.x_ref = self.y_ref = self.yaw_ref = None
        self._win = None
        self.lap_events = []

    def step(self, current_state, planning=None):
        """current_state = [vx vy psiDot x y psi] (pos_info); planning = the five My_Planning arrays (x_d, y_d, psi_d,
        vx_d, curv_d) or None on lap 0.  Returns dict(published=(servo, motor) sent at the top of this iteration,
        cmd=(servo, motor) computed by it, LocalState, prediction=(s, ey, epsi), LapNumber, run)."""
        N, mp = self.N, self.map
        G = np.array(current_state, float); Lc = G.copy()
        if Lc[0] < 0.01:
            Lc[0] = 0.01
        if self.LapNumber == 0:
            Lc[4], Lc[3], Lc[5], _inside = mp.getLocalPosition(G[3], G[4], G[5])             # CMAIN:188 (quirk Q9)
            if Lc[4] >= 3 * mp.TrackLength / 4:
                self.HalfTrack = 1
        else:
            G[5] = _wrap(G[5] - 2 * np.pi * self.LapNumber)
            if self.index <= 0:                                                              # max_window = 0
                if self.index == 0:
                    self._win = [np.asarray(a, float)[0:N].copy() for a in planning]
                self.x_ref, self.y_ref, self.yaw_ref, self.vel_ref, self.curv_ref = [a[self.index:self.index + N] for a in self._win]
                self.index += 1
            else:
                self.index = 0
            Lc[4], _ex, Lc[5], Lc[3] = body_frame_errors(G[3], G[4], G[5], self.x_ref[0], self.y_ref[0], self.yaw_ref[0], self.SS,
                                                         Lc[0], Lc[1], self.curv_ref[0], self.dt)
            self.SS = Lc[4]
        # end of lap (CMAIN:254-283)
        if self.HalfTrack == 1 and Lc[4] <= mp.TrackLength / 4:
            self.HalfTrack = 0; self.LapNumber += 1; self.SS = 0
            self.lap_events.append(self.LapNumber)
        elif self.LapNumber >= 1 and abs(G[3]) < 0.1 and Lc[4] >= mp.TrackLength - mp.TrackLength / 10:
            self.LapNumber += 1; self.SS = 0
            self.lap_events.append(self.LapNumber)
            if self.LapNumber > self.NumberOfLaps:
                self.run = False
        # last command enters the slew-rate term (CMAIN:286-298); it is published before the solve (CMAIN:301)
        c = self.Controller if self.LapNumber == 0 else self.Controller_TT
        c.OldSteering.append(self.servo); c.OldAccelera.append(self.motor)
        c.OldSteering.pop(0); c.OldAccelera.pop(0)
        published = (self.servo, self.motor)
        C0, CT = self.Controller, self.Controller_TT
        if self.first_it < 10:                                                               # CMAIN:310-320
            xx, uu = controller_seed(Lc)
            C0.solve(Lc[0:6], xx, uu, False, np.ones(N), 0, 0, 0, self.first_it)
            self.first_it += 1
            active = C0
        elif self.LapNumber == 0:                                                            # CMAIN:325-336
            S, A_L, B_L, C_L = C0.LPVPrediction(Lc[0:6], C0.uPred, np.ones(N + 1), np.zeros(N), self.Cf_new, 0)
            C0.solve(S[0, :], S, C0.uPred, False, np.ones(N + 1), A_L, B_L, C_L, self.first_it)
            CT.uPred = C0.uPred
            active = C0
        else:                                                                                # CMAIN:361-363
            vel_ref = self.vel_ref if self.vel_ref is not None else np.ones(N + 1)
            curv_ref = self.curv_ref if self.curv_ref is not None else np.zeros(N)
            S, A_L, B_L, C_L = CT.LPVPrediction(Lc[0:6], CT.uPred, vel_ref, curv_ref, self.Cf_new, self.LapNumber)
            CT.solve(Lc[0:6], 0.0, CT.uPred, False, vel_ref, A_L, B_L, C_L, self.first_it)
            active = CT
        # CMAIN:381-386 reads the delays of `Controller` in both branches; they are 0
        u = C0.uPred if self.LapNumber == 0 else CT.uPred
        self.servo, self.motor = float(u[0, 0]), float(u[0, 1])
        pred = C0 if self.LapNumber < 1 else CT
        return dict(published=published, cmd=(self.servo, self.motor), LocalState=Lc,
                    prediction=(pred.xPred[:, 4].copy(), pred.xPred[:, 5].copy(), pred.xPred[:, 3].copy()),
                    LapNumber=self.LapNumber, run=self.run, status=active.status_val, iters=active.iters)


class PlannerNode(object):
    """Loop state and one iteration of the trajectory-planner node."""

    def __init__(self, track_map, N, dt, half_width, params=None, device=0, testing=False):
        self.map, self.N, self.dt, self.HW = track_map, int(N), float(dt), float(half_width)
        self.testing = bool(testing)         # PLANNER_TEST.launch (Testing = 1): no estimator, start from [1, 0, 0, 0, 0] (PMAIN:59-69, 144-146)
        self.Planner = LPV_MPC_Planner(PLAN_Q, PLAN_R, PLAN_dR, PLAN_L, N, dt, track_map, "OSQP", params=params, device=device)   # PMAIN:98-106
        self.handoff = PlannerHandoff(self.Planner)
        self.first_it = 1

    def step(self, current_state=None):
        """current_state = [vx vy psiDot x y psi] (pos_info; only used on the first iteration, PMAIN:137-141; ignored in
        testing mode, where the first state is [1, 0, 0] at the pose (0, 0, 0) the node's arrays start from, PMAIN:145-146).
        Returns the five My_Planning arrays."""
        P = self.Planner
        if self.first_it == 1:
            G = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0]) if self.testing else np.asarray(current_state, float)
            x0 = np.array([G[0], G[1], G[2], 0.0, 0.0])
            _s, x0[4], x0[3], _inside = self.map.getLocalPosition(G[3], G[4], G[5])           # PMAIN:141 / 146 (quirk Q9)
            xx, uu = planner_seed(self.N, x0, 0.2, self.dt)
            P.solve(x0, xx, uu, 0, 0, 0, self.first_it, self.HW)
            self.first_it += 1
        else:
            _S, A_L, B_L, C_L = P.LPVPrediction(P.xPred[1, :], self.handoff.SS, P.uPred)
            P.solve(P.xPred[1, :], 0, 0, A_L, B_L, C_L, self.first_it, self.HW)
        P.OldSteering.append(P.uPred[0, 0]); P.OldAccelera.append(P.uPred[0, 1])             # PMAIN:186-187 (quirk Q3)
        return self.handoff.update()


# ---------------------------------------------------------------------------------------------------------
# rospy wiring (needs a ROS box with the barc message package; not exercised by the test-suite)
# ---------------------------------------------------------------------------------------------------------
def _ros_map():
    """The reference's Map when its package is importable (it reads trackShape from rospy), else this package's."""
    import rospy
    try:
        from trackInitialization import Map as RefMap          # reference Utilities on PYTHONPATH
        return RefMap()
    except ImportError:
        from .track import Map
        return Map(rospy.get_param("trackShape"), rospy.get_param("/TrajectoryPlanner/halfWidth"))


class _Latest(object):
    def __init__(self, value):
        self.value = value

    def set(self, value):
        self.value = value


def controller_main():
    import rospy
    from barc.msg import ECU, My_Planning, Racing_Info, pos_info, prediction
    rospy.init_node("LPV-MPC")
    pub_cmd = rospy.Publisher("ecu", ECU, queue_size=1)
    pub_pred = rospy.Publisher("OL_predictions", prediction, queue_size=1)
    pub_info = rospy.Publisher("Racing_Info", Racing_Info, queue_size=1)
    N = rospy.get_param("/control/N")
    Np = rospy.get_param("/TrajectoryPlanner/N")
    state = _Latest([0.0] * 6)
    plan = _Latest([np.zeros(Np), np.zeros(Np), np.zeros(Np), np.ones(Np), np.zeros(Np)])     # PlanningData defaults (DS:52-56)
    rospy.Subscriber("pos_info", pos_info, lambda m: state.set([m.v_x, m.v_y, m.psiDot, m.x, m.y, m.psi]))
    rospy.Subscriber("My_Planning", My_Planning, lambda m: plan.set([m.x_d, m.y_d, m.psi_d, m.vx_d, m.curv_d]))
    loop_rate = 30.0
    rate = rospy.Rate(loop_rate)
    node = ControllerNode(_ros_map(), N, 1.0 / loop_rate)
    rospy.sleep(1)
    cmd, pred, info = ECU(), prediction(), Racing_Info()
    while not rospy.is_shutdown() and node.run:
        servo, motor = node.servo, node.motor
        cmd.servo, cmd.motor = servo, motor
        pub_cmd.publish(cmd)                                   # the command computed by the previous iteration (CMAIN:301)
        out = node.step(state.value, plan.value)
        pred.s, pred.ey, pred.epsi = out["prediction"]; pred.ex = []
        pub_pred.publish(pred)
        info.LapNumber = out["LapNumber"]
        pub_info.publish(info)
        rate.sleep()


def planner_main():
    import rospy
    from barc.msg import My_Planning, Racing_Info, pos_info
    rospy.init_node("Trajectory_Planner")
    pub = rospy.Publisher("My_Planning", My_Planning, queue_size=1)
    HW = rospy.get_param("/TrajectoryPlanner/halfWidth")
    loop_rate = rospy.get_param("/TrajectoryPlanner/Frecuency")
    N = rospy.get_param("/TrajectoryPlanner/N")
    testing = int(rospy.get_param("/TrajectoryPlanner/Testing", 0)) == 1      # PLANNER_TEST.launch: planner alone, open loop (PMAIN:59-69)
    state = _Latest([0.0] * 6)
    lap = _Latest(0)
    if not testing:
        rospy.Subscriber("pos_info", pos_info, lambda m: state.set([m.v_x, m.v_y, m.psiDot, m.x, m.y, m.psi]))
    rospy.Subscriber("Racing_Info", Racing_Info, lambda m: lap.set(m.LapNumber))
    rate = rospy.Rate(loop_rate)
    node = PlannerNode(_ros_map(), N, 1.0 / loop_rate, HW, testing=testing)
    refs = My_Planning()
    while not rospy.is_shutdown():
        if lap.value >= 1 or testing:                          # PMAIN:129
            refs.x_d, refs.y_d, refs.psi_d, refs.vx_d, refs.curv_d = [list(a) for a in node.step(state.value)]
            pub.publish(refs)
        rate.sleep()
