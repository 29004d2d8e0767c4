#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE's own
Python (read from /root/reference at generation time only).

How: the four hot-path modules are copied to a temp dir, converted with lib2to3
(only `print` statements change), and imported with stub `rospy` (parameter dict
with the MAIN_LAUNCH.launch values), `cvxopt` (import-only) and `osqp` modules.
The `osqp` stub records the exact (P, q, A, l, u, settings) the reference hands to
OSQP and -- because the real wheel is absent -- returns the solution of the oracle
restatement (oracle/osqp_ref.c) so that multi-tick recursions can run.

Captured (all float64):
  * LPV evaluation + roll-out outputs and the assembled QP      -> exact pins (<= 1e-12)
  * oracle solutions with their KKT-certified optimum            -> solution pins
Nothing of the reference's source text is stored: only inputs and numeric outputs.

Usage:  python tests/golden/make_golden.py [--out DIR] [handoff | delay]   (needs /root/reference; NOT run on the GPU box)

Every file written is also entered in MANIFEST.json (its keys, with shape and dtype): tests/test_oracle_golden.py fails when a
committed fixture's key set differs from what this generator writes.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = HERE        # --out DIR: write there instead (to compare a fresh set with the committed one)
sys.path.insert(0, ROOT)
REF = "/root/reference/workspace/src/barc/src"

from oracle import osqp_ref, kkt_cert, lpv_ref  # noqa: E402

PARAMS = {"lf": 0.125, "lr": 0.125, "m": 1.98, "Iz": 0.03, "Cf": 60.0, "Cr": 60.0, "mu": 0.05,
          "/TrajectoryPlanner/max_vel": 5.0, "/TrajectoryPlanner/min_vel": 0.9,
          "/TrajectoryPlanner/halfWidth": 0.2, "trackShape": "oval"}

CAPTURE = []      # list of dicts, one per OSQP().setup() call


def install_stubs():
    rospy = types.ModuleType("rospy")
    rospy.get_param = lambda k, *a: PARAMS[k]
    sys.modules["rospy"] = rospy

    cvx = types.ModuleType("cvxopt")
    cvx.spmatrix = cvx.matrix = lambda *a, **k: None
    solvers = types.ModuleType("cvxopt.solvers")
    solvers.options = {}
    solvers.qp = lambda *a, **k: None
    cvx.solvers = solvers
    sys.modules["cvxopt"] = cvx
    sys.modules["cvxopt.solvers"] = solvers

    osqp = types.ModuleType("osqp")

    class OSQP(object):
        _C = {"OSQP_SOLVED": 1, "OSQP_SOLVED_INACCURATE": 2, "OSQP_MAX_ITER_REACHED": -2}

        def setup(self, P=None, q=None, A=None, l=None, u=None, **settings):
            self.P = np.asarray(P.todense()); self.q = np.array(q, float)
            self.A = np.asarray(A.todense()); self.l = np.array(l, float); self.u = np.array(u, float)
            self.settings = settings
            CAPTURE.append(dict(P=self.P, q=self.q, A=self.A, l=self.l, u=self.u, settings=dict(settings)))

        def warm_start(self, **k):
            raise AssertionError("reference never passes initvals")

        def constant(self, name):
            return self._C[name]

        def solve(self):
            r = osqp_ref.solve_qp(self.P, self.q, self.A, self.l, self.u)
            CAPTURE[-1]["x"] = r.x.copy(); CAPTURE[-1]["y"] = r.y.copy()
            CAPTURE[-1]["iter"] = r.info.iter; CAPTURE[-1]["status_val"] = r.info.status_val
            CAPTURE[-1]["status_polish"] = r.info.status_polish
            return types.SimpleNamespace(x=r.x, y=r.y, info=types.SimpleNamespace(
                status_val=r.info.status_val, status=r.info.status, iter=r.info.iter))

    osqp.OSQP = OSQP
    sys.modules["osqp"] = osqp

    # ROS message / tf packages imported by vehicleSimulator.py: attribute access yields a dummy class
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (object,), {"__init__": lambda self, *a, **k: None})
    for name in ("geometry_msgs", "geometry_msgs.msg", "barc", "barc.msg", "sensor_msgs", "sensor_msgs.msg",
                 "marvelmind_nav", "marvelmind_nav.msg", "tf"):
        sys.modules[name] = _Any(name)
    rospy.Rate = lambda *a, **k: None
    rospy.Publisher = lambda *a, **k: None
    rospy.Subscriber = lambda *a, **k: None
    PARAMS.update({"simulator/c_f": 0.8, "simulator/B": 6.0, "simulator/C": 1.6, "simulator/mu": 0.05,
                   "simulator/init_vx": 0.2, "simulator/dt": 0.005})


def import_reference():
    tmp = tempfile.mkdtemp(prefix="refimport_")
    for rel in ("ControllerObject/PathFollowingLPVMPC.py", "PlannerObject/LPV_MPC_Planner.py",
                "Utilities/utilities.py", "Utilities/trackInitialization.py", "vehicleSimulator.py"):
        shutil.copy(os.path.join(REF, rel), tmp)
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", tmp], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # vehicleSimulator.py mixes tabs and spaces (legal in Python 2, tab stops every 8 columns)
    vs = os.path.join(tmp, "vehicleSimulator.py")
    with open(vs) as fh:
        lines = [ln.expandtabs(8) for ln in fh]
    with open(vs, "w") as fh:
        fh.writelines(lines)
    sys.path.insert(0, tmp)
    install_stubs()
    import PathFollowingLPVMPC as CTRL
    import LPV_MPC_Planner as PLAN
    import trackInitialization as TRACK
    import utilities as UTIL
    return CTRL, PLAN, TRACK, UTIL, tmp


def make_map(TRACK, shape, hw=0.2):
    PARAMS["trackShape"] = shape
    PARAMS["/TrajectoryPlanner/halfWidth"] = hw
    return TRACK.Map()


TUNINGS = {
    "path": (np.diag([100.0, 1.0, 1.0, 20.0, 0.0, 900.0]), 0.25 * np.eye(2), 37.5 * np.array([1.3, 1.0])),
    "race": (np.diag([400.0, 1.0, 1.0, 20.0, 0.0, 1100.0]), 0.0 * np.eye(2), np.array([100.0, 45.0])),
}
PLAN_Q = -np.diag([-0.000000000000088, -9.703658572659423, -0.5, 0.000000000213635, -0.153591566469547])
PLAN_L = -np.array([1.00702414775175, 0.187661946033823, -0.0, 0.0, -0.0329493219494661])
PLAN_R = np.diag([0.8, 0.0])
PLAN_dR = np.array([6.0, 6.0])


def stackL(lst):
    return np.stack([np.asarray(a, float) for a in lst])


def certified_optimum(c):
    l = np.where(c["l"] < -1e29, -np.inf, c["l"]); u = np.where(c["u"] > 1e29, np.inf, c["u"])
    if not np.all(np.isfinite(c["x"])):
        return np.full_like(c["x"], np.nan), np.full_like(c["y"], np.nan), np.array([np.nan] * 3)
    try:
        xs, ys, _ = kkt_cert.active_set_optimum(c["P"], c["q"], c["A"], l, u, c["x"], c["y"])
    except RuntimeError:
        return np.full_like(c["x"], np.nan), np.full_like(c["y"], np.nan), np.array([np.nan] * 3)
    return xs, ys, np.array(kkt_cert.kkt_residuals(c["P"], c["q"], c["A"], l, u, xs, ys))


def ctrl_case(CTRL, mp, tuning, N, dt, x0, u_prev, vel_ref, curv_ref, lap, old_u, cf_new=60.0, steer_hist=()):
    """steer_hist = OldSteering[1:] (length steeringDelay): the pinned steering commands of CTRL:518-527."""
    Q, R, dR = TUNINGS[tuning]
    c = CTRL.PathFollowingLPV_MPC(Q, R, dR, N, 1, dt, mp, "OSQP", len(steer_hist), 0)
    c.OldSteering = [float(old_u[0])] + [float(v) for v in steer_hist]; c.OldAccelera = [float(old_u[1])]
    S, A_L, B_L, C_L = c.LPVPrediction(x0, u_prev, vel_ref, curv_ref, cf_new, lap)
    CAPTURE.clear()
    c.solve(x0, 0.0, u_prev, False, vel_ref, A_L, B_L, C_L, 10)
    cap = CAPTURE[-1]
    xs, ys, cert = certified_optimum(cap)
    return dict(x0=x0, u_prev=u_prev, vel_ref=np.asarray(vel_ref, float), curv_ref=np.asarray(curv_ref, float),
                lap=lap, old_u=np.asarray(old_u, float), cf_new=cf_new, Q=Q, R=R, dR=dR, N=N, dt=dt,
                steer_hist=np.asarray(steer_hist, float), states=S, A=stackL(A_L), B=stackL(B_L), P=cap["P"], q=cap["q"], Aqp=cap["A"], l=cap["l"], u=cap["u"],
                x_orc=cap["x"], y_orc=cap["y"], iter_orc=cap["iter"], status_orc=cap["status_val"],
                polish_orc=cap["status_polish"], x_star=xs, y_star=ys, cert=cert,
                xPred=np.array(c.xPred), uPred=np.array(c.uPred), LinPoints=np.array(c.LinPoints))


def plan_case(PLAN, mp, N, dt, x0, SS, u_prev, max_ey):
    p = PLAN.LPV_MPC_Planner(PLAN_Q, PLAN_R, PLAN_dR, PLAN_L, N, dt, mp, "OSQP")
    S, A_L, B_L, C_L = p.LPVPrediction(x0, SS, u_prev)
    CAPTURE.clear()
    p.solve(x0, 0, 0, A_L, B_L, C_L, 2, max_ey)
    cap = CAPTURE[-1]
    xs, ys, cert = certified_optimum(cap)
    return dict(x0=x0, SS=np.asarray(SS, float), u_prev=u_prev, max_ey=max_ey, N=N, dt=dt,
                Q=PLAN_Q, R=PLAN_R, dR=PLAN_dR, L_cf=PLAN_L,
                states=S, A=stackL(A_L), B=stackL(B_L), P=cap["P"], q=cap["q"], Aqp=cap["A"], l=cap["l"], u=cap["u"],
                x_orc=cap["x"], y_orc=cap["y"], iter_orc=cap["iter"], status_orc=cap["status_val"],
                polish_orc=cap["status_polish"], x_star=xs, y_star=ys, cert=cert,
                xPred=np.array(p.xPred), uPred=np.array(p.uPred), LinPoints=np.array(p.LinPoints))


MANIFEST = {}


def save_npz(filename, arrays):
    """Write one fixture and record its key set (name -> [shape, dtype]) for MANIFEST.json."""
    np.savez_compressed(os.path.join(OUT, filename), **arrays)
    MANIFEST[filename] = {k: [list(np.asarray(v).shape), str(np.asarray(v).dtype)] for k, v in sorted(arrays.items())}


def write_manifest():
    """Merge this run's entries into MANIFEST.json (a partial regeneration keeps the other files' entries)."""
    import json
    path = os.path.join(OUT, "MANIFEST.json")
    old = {}
    if os.path.exists(path):
        old = json.load(open(path))
    elif os.path.exists(os.path.join(HERE, "MANIFEST.json")):
        old = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    old.update(MANIFEST)
    json.dump(old, open(path, "w"), indent=0, sort_keys=True)


def save_cases(name, cases):
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%02d_%s" % (i, k)] = np.asarray(v)
    flat["n_cases"] = np.array(len(cases))
    save_npz(name + ".npz", flat)
    print("wrote %s.npz (%d cases)" % (name, len(cases)))


def gen_handoff(CTRL, PLAN, TRACK, UTIL):
    """planner -> controller hand-off (PMAIN:201-224,257-280) and the planner + controller
    cascade (CMAIN:198-283,337-363,381-386).  The glue of the two main() loops is restated here (it cannot be
    imported); every function it calls is the reference's own (Map, Curvature, the two MPC classes, Simulator) or
    scipy's, which is what the reference itself calls."""
    import vehicleSimulator as SIM
    from scipy import signal
    from scipy.interpolate import interp1d
    lsh = make_map(TRACK, "L_shape")
    table = np.array(lsh.PointAndTangent)
    Np, dtp, HW = 40, 0.05, 0.2                           # MAIN_LAUNCH.launch:38-47 (N = 40, 20 Hz, halfWidth 0.2)
    b_f, a_f = signal.ellip(4, 0.01, 120, 0.125)          # PMAIN:112
    n33 = int(np.around(Np * dtp / 0.033))

    class PlannerNode(object):                            # state of plannerMain.main()
        def __init__(self):
            self.P = PLAN.LPV_MPC_Planner(PLAN_Q, PLAN_R, PLAN_dR, PLAN_L, Np, dtp, lsh, "OSQP")
            self.first = 1
            self.SS = np.zeros(Np + 1)
            self.last = [0.0, 0.0, 0.0]
            self.log = []

        def tick(self, local_state):
            P = self.P
            if self.first == 1:
                x0 = np.array(local_state, float)
                xx, uu = lpv_ref.plan_seed_vectors(Np, x0, 0.2, dtp)
                P.solve(x0, xx, uu, 0, 0, 0, self.first, HW)
                self.first += 1
            else:
                S, A_L, B_L, C_L = P.LPVPrediction(P.xPred[1, :], self.SS, P.uPred)
                P.solve(P.xPred[1, :], 0, 0, A_L, B_L, C_L, self.first, HW)
            P.OldSteering.append(P.uPred[0, 0]); P.OldAccelera.append(P.uPred[0, 1])
            rec = dict(xPred=np.array(P.xPred), uPred=np.array(P.uPred), SS_in=self.SS.copy(), pose_in=np.array(self.last),
                       iters=CAPTURE[-1]["iter"], status=CAPTURE[-1]["status_val"])
            SS = self.SS
            Xref = np.zeros(Np + 1); Yref = np.zeros(Np + 1); Th = np.zeros(Np + 1)
            Xref[0], Yref[0], Th[0] = self.last
            for j in range(Np):
                cv = UTIL.Curvature(SS[j], lsh.PointAndTangent)
                SS[j + 1] = (SS[j] + ((P.xPred[j, 0] * np.cos(P.xPred[j, 4]) - P.xPred[j, 1] * np.sin(P.xPred[j, 4]))
                                      / (1 - P.xPred[j, 3] * cv)) * dtp)
                Xref[j + 1], Yref[j + 1], Th[j + 1] = lsh.getGlobalPosition(SS[j + 1], 0.0)
            SS[0] = SS[1]
            self.last = [Xref[1], Yref[1], Th[1]]
            yaw = np.zeros(Np); xp = np.zeros(Np); yp = np.zeros(Np)
            for i in range(Np):
                yaw[i] = Th[i] + P.xPred[i, 4]
                xp[i] = Xref[i] - P.xPred[i, 3] * np.sin(yaw[i])
                yp[i] = Yref[i] + P.xPred[i, 3] * np.cos(yaw[i])
            vel = np.array(P.xPred[0:Np, 0]); curv = np.array(P.xPred[0:Np, 2] / P.xPred[0:Np, 0])
            t50 = np.linspace(0, Np * dtp, num=Np, endpoint=True)
            t33 = np.linspace(0, Np * dtp, num=n33, endpoint=True)
            refs = [interp1d(t50, v, kind="cubic")(t33) for v in (xp, yp, yaw, vel, curv)]
            refs[4] = signal.filtfilt(b_f, a_f, refs[4], padlen=50)
            rec.update(SS_out=SS.copy(), pose_out=np.array(self.last), sig=np.array([xp, yp, yaw, vel, curv]), refs=np.array(refs))
            self.log.append(rec)
            return np.array(refs)

    # ---- (1) planner alone, open loop for 60 ticks (3 s: into and through the first corner) ----
    node = PlannerNode()
    for tick in range(60):
        node.tick([1.0, 0.0, 0.0, 0.03, -0.02])
    keys = ("xPred", "uPred", "SS_in", "pose_in", "SS_out", "pose_out", "sig", "refs", "iters", "status")
    ho = {"plan_" + k: np.array([r[k] for r in node.log]) for k in keys}
    ho.update(ellip_b=np.array(b_f), ellip_a=np.array(a_f), table=table, N=np.array(Np), dt=np.array(dtp))
    # Body_Frame_Errors samples are cheap to pin exactly as well (CMAIN:495-506, with TRACK.wrap)
    rng = np.random.default_rng(11)
    bfe_in = np.column_stack([rng.uniform(-3, 3, 64), rng.uniform(-3, 3, 64), rng.uniform(-4, 4, 64), rng.uniform(-3, 3, 64),
                              rng.uniform(-3, 3, 64), rng.uniform(-4, 4, 64), rng.uniform(0, 19, 64), rng.uniform(0.5, 4, 64),
                              rng.normal(0, 0.2, 64), rng.uniform(-2, 2, 64)])
    bfe_out = []
    for r in bfe_in:
        x, y, psi, xd, yd, psid, s0, vx, vy, cv = [float(v) for v in r]
        ex = (x - xd) * np.cos(psid) + (y - yd) * np.sin(psid)
        ey = -(x - xd) * np.sin(psid) + (y - yd) * np.cos(psid)
        epsi = TRACK.wrap(psi - psid)
        s = s0 + ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - ey * cv)) * (1.0 / 30)
        bfe_out.append([s, ex, ey, epsi])
    ho.update(bfe_in=bfe_in, bfe_out=np.array(bfe_out))
    save_npz("handoff.npz", ho)
    print("wrote handoff.npz")

    # ---- (2) cascade: lap-0 approach to the start line, lap event, then planner + Controller_TT + plant ----
    # schedule: controller tick k is preceded by planner ticks 0 .. floor(2k/3) (20 Hz against
    # 30 Hz), the plant takes 7, 7, 6 steps of 5 ms after controller ticks 3j, 3j+1, 3j+2 (100 ms per 3 ticks);
    # measurement = ground truth; the command is applied right after the solve, as in the f1 fixture.
    N, dt = 20, 1.0 / 30
    Qp, Rp, dRp = TUNINGS["path"]; Qr, Rr, dRr = TUNINGS["race"]
    C0 = CTRL.PathFollowingLPV_MPC(Qp, Rp, dRp, N, 1, dt, lsh, "OSQP", 0, 0)
    CT = CTRL.PathFollowingLPV_MPC(Qr, Rr, dRr, N, 1, dt, lsh, "OSQP", 0, 0)
    sim = SIM.Simulator()
    sim.x, sim.y, sim.yaw, sim.vx = -0.55, 0.02, 0.01, 1.0
    cmd = [0.0, 0.0]; first_it = 1; LapNumber = 0; HalfTrack = 1; SS = 0.0; index = 0
    L = lsh.TrackLength
    pre = dict(plant=[], local=[], cmd=[], lap=[])
    for tick in range(60):                                  # lap 0 until the reference's lap event fires
        G_ = np.array([sim.vx, sim.vy, sim.psiDot, sim.x, sim.y, sim.yaw]); Lc = G_.copy()
        if Lc[0] < 0.01:
            Lc[0] = 0.01
        Lc[4], Lc[3], Lc[5], inside = lsh.getLocalPosition(G_[3], G_[4], G_[5])
        if HalfTrack == 1 and Lc[4] <= L / 4:              # CMAIN:254-259
            HalfTrack = 0; LapNumber += 1; SS = 0
        Cx = C0 if LapNumber == 0 else CT                   # CMAIN:289-298
        Cx.OldSteering.append(cmd[0]); Cx.OldAccelera.append(cmd[1]); Cx.OldSteering.pop(0); Cx.OldAccelera.pop(0)
        if first_it < 10:
            xx, uu = lpv_ref.ctrl_seed_vectors(Lc)
            C0.solve(Lc[0:6], xx, uu, False, np.ones(N), 0, 0, 0, first_it); first_it += 1
        elif LapNumber == 0:
            S, A_L, B_L, C_L = C0.LPVPrediction(Lc[0:6], C0.uPred, np.ones(N + 1), np.zeros(N), 60.0, 0)
            C0.solve(S[0, :], S, C0.uPred, False, np.ones(N + 1), A_L, B_L, C_L, first_it)
            CT.uPred = C0.uPred                              # CMAIN:336
        else:
            # the iteration on which the lap event fires already runs the trajectory-tracking branch, with the lap-0
            # references still in place: vel_ref = ones(N+1), curv_ref = zeros(N)  (CMAIN:326-327,361-363)
            S, A_L, B_L, C_L = CT.LPVPrediction(Lc[0:6], CT.uPred, np.ones(N + 1), np.zeros(N), 60, LapNumber)
            CT.solve(Lc[0:6], 0.0, CT.uPred, False, np.ones(N + 1), A_L, B_L, C_L, first_it)
        Cx = C0 if LapNumber == 0 else CT
        cmd = [float(Cx.uPred[0, 0]), float(Cx.uPred[0, 1])]
        pre["plant"].append([sim.x, sim.y, sim.vx, sim.vy, sim.ax, sim.ay, sim.yaw, sim.psiDot]); pre["local"].append(Lc.copy())
        pre["cmd"].append(list(cmd)); pre["lap"].append(LapNumber)
        for _ in range(7):
            sim.f([cmd[1], cmd[0]])
        if LapNumber == 1:
            break
    assert LapNumber == 1 and len(pre["lap"]) > 10, (LapNumber, len(pre["lap"]))
    # the fleet fixture starts after the event iteration: the planner node sees LapNumber >= 1 from then on (PMAIN:129)
    cas = dict(plant0=np.array([sim.x, sim.y, sim.vx, sim.vy, sim.ax, sim.ay, sim.yaw, sim.psiDot]), cmd0=np.array(cmd),
               uPred0=np.array(CT.uPred), lap0=np.array(LapNumber), pre_ticks=np.array(len(pre["lap"])),
               pre_plant=np.array(pre["plant"]), pre_local=np.array(pre["local"]), pre_cmd=np.array(pre["cmd"]), pre_lap=np.array(pre["lap"]))
    node = PlannerNode()
    refs = None; plan_done = 0
    K = 60
    tr = {k: [] for k in ("plant", "local", "vel_ref", "curv_ref", "cmd", "iters", "status", "lap", "SS", "plan_ticks", "uPred", "xPred")}
    nsub = (7, 7, 6)
    for k in range(K):
        G_ = np.array([sim.vx, sim.vy, sim.psiDot, sim.x, sim.y, sim.yaw]); Lc = G_.copy()
        while plan_done < (2 * k) // 3 + 1:                 # planner node, own thread of time (PMAIN:137-141, Testing == 0)
            pl = np.array([G_[0], G_[1], G_[2], 0.0, 0.0])
            _s, pl[4], pl[3], _in = lsh.getLocalPosition(G_[3], G_[4], G_[5])
            refs = node.tick(pl); plan_done += 1
        if Lc[0] < 0.01:
            Lc[0] = 0.01
        G_[5] = G_[5] - 2 * np.pi * LapNumber               # CMAIN:200-202
        G_[5] = TRACK.wrap(G_[5])
        if index <= 0:                                      # CMAIN:219-235, max_window = 0
            if index == 0:
                Xv = refs[0][0:N]; Yv = refs[1][0:N]; Pv = refs[2][0:N]; Vv = refs[3][0:N]; Cv = refs[4][0:N]
            x_ref = Xv[index:index + N]; y_ref = Yv[index:index + N]; yaw_ref = Pv[index:index + N]
            vel_ref = Vv[index:index + N]; curv_ref = Cv[index:index + N]
            index += 1
        else:
            index = 0
        psid = yaw_ref[0]
        ex = (G_[3] - x_ref[0]) * np.cos(psid) + (G_[4] - y_ref[0]) * np.sin(psid)     # Body_Frame_Errors, CMAIN:495-506
        ey = -(G_[3] - x_ref[0]) * np.sin(psid) + (G_[4] - y_ref[0]) * np.cos(psid)
        epsi = TRACK.wrap(G_[5] - psid)
        s = SS + ((Lc[0] * np.cos(epsi) - Lc[1] * np.sin(epsi)) / (1 - ey * curv_ref[0])) * dt
        Lc[4], Lc[5], Lc[3] = s, ey, epsi
        SS = Lc[4]
        if LapNumber >= 1 and abs(G_[3]) < 0.1 and Lc[4] >= (L - L / 10):             # CMAIN:268-272
            LapNumber += 1; SS = 0
        CT.OldSteering.append(cmd[0]); CT.OldAccelera.append(cmd[1]); CT.OldSteering.pop(0); CT.OldAccelera.pop(0)
        S, A_L, B_L, C_L = CT.LPVPrediction(Lc[0:6], CT.uPred, vel_ref, curv_ref, 60, LapNumber)
        CT.solve(Lc[0:6], 0.0, CT.uPred, False, vel_ref, A_L, B_L, C_L, first_it)
        cmd = [float(CT.uPred[0, 0]), float(CT.uPred[0, 1])]
        tr["plant"].append([sim.x, sim.y, sim.vx, sim.vy, sim.ax, sim.ay, sim.yaw, sim.psiDot]); tr["local"].append(Lc.copy())
        tr["vel_ref"].append(np.array(vel_ref)); tr["curv_ref"].append(np.array(curv_ref)); tr["cmd"].append(list(cmd))
        tr["iters"].append(CAPTURE[-1]["iter"]); tr["status"].append(CAPTURE[-1]["status_val"]); tr["lap"].append(LapNumber)
        tr["SS"].append(SS); tr["plan_ticks"].append(plan_done); tr["uPred"].append(np.array(CT.uPred)); tr["xPred"].append(np.array(CT.xPred))
        for _ in range(nsub[k % 3]):
            sim.f([cmd[1], cmd[0]])
    cas.update({"ctrl_" + k: np.array(v) for k, v in tr.items()})
    cas.update({"plan_" + k: np.array([r[k] for r in node.log]) for k in ("xPred", "uPred", "refs", "iters", "status", "SS_out")})
    cas.update(table=table)
    save_npz("cascade.npz", cas)
    print("wrote cascade.npz")


def main():
    global OUT
    warnings.simplefilter("ignore")
    if "--out" in sys.argv:
        i = sys.argv.index("--out"); OUT = os.path.abspath(sys.argv[i + 1]); del sys.argv[i:i + 2]
        os.makedirs(OUT, exist_ok=True)
    CTRL, PLAN, TRACK, UTIL, tmp = import_reference()
    try:
        try:
            generate(CTRL, PLAN, TRACK, UTIL)
        finally:
            write_manifest()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def generate(CTRL, PLAN, TRACK, UTIL):
    if True:
        if "handoff" in sys.argv[1:]:                       # regenerate only handoff.npz / cascade.npz
            gen_handoff(CTRL, PLAN, TRACK, UTIL)
            return
        if "delay" in sys.argv[1:]:                         # only ctrl_n20_delay.npz (steeringDelay = 1, 2, 3; CTRL:518-527)
            oval = make_map(TRACK, "oval")
            rng = np.random.default_rng(5)
            N = 20; dt = 1.0 / 30.0
            cases = []
            for i in range(8):
                delay = (1, 2, 2, 3)[i % 4]
                s = rng.uniform(0, 13); vx = rng.uniform(0.8, 3.0)
                x0 = np.array([vx, rng.normal(0, 0.05), rng.normal(0, 0.3), rng.normal(0, 0.1), s, rng.normal(0, 0.1)])
                u_prev = np.tile([rng.normal(0, 0.05), rng.normal(0.2, 0.3)], (N, 1))
                hist = rng.normal(0, 0.08, delay) if i < 6 else np.array([0.249, -0.249, 0.1][:delay])   # last two: pinned at the box
                curv = np.full(N, UTIL.Curvature(s, oval.PointAndTangent))
                cases.append(ctrl_case(CTRL, oval, "race" if i % 2 == 0 else "path", N, dt, x0, u_prev, np.full(N + 1, vx), curv,
                                       0 if i % 4 == 3 else 1, u_prev[0], steer_hist=hist))
            save_cases("ctrl_n20_delay", cases)
            return
        # ---------------- track tables + curvature samples ----------------
        tracks = {}
        for shape in ("oval", "L_shape", "3110", "Euge_Track"):
            mp = make_map(TRACK, shape)
            tab = np.array(mp.PointAndTangent)
            ss = np.linspace(0.0, 2.2 * mp.TrackLength, 97)[:-1] + 1e-3
            tracks[shape + "_table"] = tab
            tracks[shape + "_length"] = np.array(mp.TrackLength)
            tracks[shape + "_halfWidth"] = np.array(mp.halfWidth)
            tracks[shape + "_s"] = ss
            tracks[shape + "_curv"] = np.array([UTIL.Curvature(s, mp.PointAndTangent) for s in ss])
        save_npz("tracks.npz", tracks)
        print("wrote tracks.npz")

        oval = make_map(TRACK, "oval")
        # ---------------- cfg 1: single controller solve, N=10, oval, fixed x0 ----------------
        N = 10; dt = 1.0 / 30.0
        x0 = np.array([1.0, 0.0, 0.0, 0.0, 0.5, 0.0])
        _, uu = lpv_ref.ctrl_seed_vectors(x0)
        cases = [ctrl_case(CTRL, oval, "path", N, dt, x0, uu[:N].copy(), np.ones(N + 1), np.zeros(N), 0, [0.0, 0.0])]
        save_cases("ctrl_n10_cfg1", cases)

        # ---------------- 16x controller N=20 oval, random x0, both tunings ----------------
        rng = np.random.default_rng(0)
        N = 20
        cases = []
        for i in range(16):
            tuning = "race" if i % 2 == 0 else "path"
            s = rng.uniform(0, 13); vx = rng.uniform(0.8, 3.0)
            scale = 1.0 if i < 10 else 3.0           # later cases push steering / accel bounds
            x0 = np.array([vx, rng.normal(0, 0.05) * scale, rng.normal(0, 0.3) * scale,
                           rng.normal(0, 0.1) * scale, s, rng.normal(0, 0.1) * scale])
            u_prev = np.tile([rng.normal(0, 0.05), rng.normal(0.2, 0.3)], (N, 1))
            lap = 0 if i % 4 == 3 else 1
            vref = np.full(N + 1, vx if i < 12 else vx + 1.5)
            curv = np.full(N, UTIL.Curvature(s, oval.PointAndTangent))
            cases.append(ctrl_case(CTRL, oval, tuning, N, dt, x0, u_prev, vref, curv, lap, u_prev[0]))
        save_cases("ctrl_n20_oval", cases)

        # ---------------- 16x planner N=30 L_shape ----------------
        lsh = make_map(TRACK, "L_shape")
        rng = np.random.default_rng(1)
        N = 30; dtp = 0.05
        cases = []
        for i in range(16):
            vx = rng.uniform(1, 4)
            x0 = np.array([vx, np.clip(rng.normal(0, 0.03), -0.9, 0.9), np.clip(rng.normal(0, 0.2), -1.9, 1.9),
                           np.clip(rng.normal(0, 0.05), -0.19, 0.19), np.clip(rng.normal(0, 0.05), -0.79, 0.79)])
            s0 = rng.uniform(0, 19.2)
            SS = s0 + np.arange(N + 1) * vx * dtp
            u_prev = np.tile([rng.normal(0, 0.03), rng.normal(0.2, 0.2)], (N, 1))
            cases.append(plan_case(PLAN, lsh, N, dtp, x0, SS, u_prev, 0.2))
        save_cases("plan_n30_lshape", cases)

        # ---------------- 4x planner N=40 (launch default) ----------------
        N = 40
        cases = []
        for i in range(4):
            vx = rng.uniform(1, 3)
            x0 = np.array([vx, 0.0, 0.0, rng.normal(0, 0.03), rng.normal(0, 0.03)])
            SS = rng.uniform(0, 10) + np.arange(N + 1) * vx * dtp
            u_prev = np.tile([0.0, 0.2], (N, 1))
            cases.append(plan_case(PLAN, lsh, N, dtp, x0, SS, u_prev, 0.2))
        save_cases("plan_n40_lshape", cases)

        # ---------------- seed-mode linearisations (a3 / a11) ----------------
        N = 20
        ls = np.array([1.2, 0.01, 0.05, 0.02, 3.0, -0.03])
        xx, uu = lpv_ref.ctrl_seed_vectors(ls)
        Q, R, dR = TUNINGS["path"]
        c = CTRL.PathFollowingLPV_MPC(Q, R, dR, N, 1, dt, oval, "OSQP", 0, 0)
        CAPTURE.clear()
        c.solve(ls, xx, uu, False, np.ones(N), 0, 0, 0, 1)
        cap = CAPTURE[-1]
        seed = dict(ctrl_ls=ls, ctrl_xx=xx, ctrl_uu=uu, ctrl_A=stackL(c.A), ctrl_B=stackL(c.B),
                    ctrl_P=cap["P"], ctrl_q=cap["q"], ctrl_Aqp=cap["A"], ctrl_l=cap["l"], ctrl_u=cap["u"],
                    ctrl_xPred=np.array(c.xPred), ctrl_uPred=np.array(c.uPred))
        Np = 30
        px0 = np.array([1.0, 0.0, 0.0, 0.0, 0.0])
        pxx, puu = lpv_ref.plan_seed_vectors(Np, px0, 0.2, dtp)
        p = PLAN.LPV_MPC_Planner(PLAN_Q, PLAN_R, PLAN_dR, PLAN_L, Np, dtp, lsh, "OSQP")
        CAPTURE.clear()
        p.solve(px0, pxx, puu, 0, 0, 0, 1, 0.2)
        cap = CAPTURE[-1]
        seed.update(plan_x0=px0, plan_xx=pxx, plan_uu=puu, plan_A=stackL(p.A), plan_B=stackL(p.B),
                    plan_P=cap["P"], plan_q=cap["q"], plan_Aqp=cap["A"], plan_l=cap["l"], plan_u=cap["u"],
                    plan_xPred=np.array(p.xPred), plan_uPred=np.array(p.uPred))
        save_npz("seed_mode.npz", seed)
        print("wrote seed_mode.npz")

        # ---------------- closed-loop recursions (20 ticks) ----------------
        # controller, lap-0 call pattern of controllerMain.py:310-331 (plant = the LPV prediction itself)
        N = 20
        Q, R, dR = TUNINGS["path"]
        c = CTRL.PathFollowingLPV_MPC(Q, R, dR, N, 1, dt, oval, "OSQP", 0, 0)
        st = np.array([1.0, 0.0, 0.0, 0.0, 0.3, 0.0])
        trace_x, trace_u, trace_x0 = [], [], []
        first_it = 1
        cmd = np.zeros(2)
        for tick in range(20):
            c.OldSteering.append(float(cmd[0])); c.OldAccelera.append(float(cmd[1]))
            c.OldSteering.pop(0); c.OldAccelera.pop(0)
            trace_x0.append(st.copy())
            if first_it < 10:
                xx, uu = lpv_ref.ctrl_seed_vectors(st)
                c.solve(st, xx, uu, False, np.ones(N), 0, 0, 0, first_it)
                first_it += 1
            else:
                S, A_L, B_L, C_L = c.LPVPrediction(st, c.uPred, np.ones(N + 1), np.zeros(N), 60.0, 0)
                c.solve(S[0, :], S, c.uPred, False, np.ones(N + 1), A_L, B_L, C_L, first_it)
            cmd = np.array(c.uPred[0, :])
            trace_x.append(np.array(c.xPred)); trace_u.append(np.array(c.uPred))
            st = np.array(c.xPred[1, :])          # ideal plant: next state = predicted
        cl = dict(ctrl_x0=np.array(trace_x0), ctrl_xPred=np.array(trace_x), ctrl_uPred=np.array(trace_u))
        # planner, open-loop test mode of plannerMain.py:152-176 (s integration PMAIN:201-216)
        Np = 30
        p = PLAN.LPV_MPC_Planner(PLAN_Q, PLAN_R, PLAN_dR, PLAN_L, Np, dtp, lsh, "OSQP")
        px0 = np.array([1.0, 0.0, 0.0, 0.0, 0.0])
        SS = np.zeros(Np + 1)
        first = 1
        tx, tu, tss = [], [], []
        for tick in range(20):
            if first == 1:
                pxx, puu = lpv_ref.plan_seed_vectors(Np, px0, 0.2, dtp)
                p.solve(px0, pxx, puu, 0, 0, 0, first, 0.2)
                first += 1
            else:
                S, A_L, B_L, C_L = p.LPVPrediction(p.xPred[1, :], SS, p.uPred)
                p.solve(p.xPred[1, :], 0, 0, A_L, B_L, C_L, first, 0.2)
            p.OldSteering.append(p.uPred[0, 0]); p.OldAccelera.append(p.uPred[0, 1])
            for j in range(Np):
                cv = UTIL.Curvature(SS[j], lsh.PointAndTangent)
                SS[j + 1] = SS[j] + ((p.xPred[j, 0] * np.cos(p.xPred[j, 4]) - p.xPred[j, 1] * np.sin(p.xPred[j, 4]))
                                     / (1 - p.xPred[j, 3] * cv)) * dtp
            SS[0] = SS[1]
            tx.append(np.array(p.xPred)); tu.append(np.array(p.uPred)); tss.append(SS.copy())
        cl.update(plan_xPred=np.array(tx), plan_uPred=np.array(tu), plan_SS=np.array(tss))
        save_npz("closed_loop.npz", cl)
        print("wrote closed_loop.npz")

        # ---------------- plant model + coordinate transforms ----------------
        import vehicleSimulator as SIM
        pt = {}
        rng = np.random.default_rng(5)
        for shape in ("oval", "L_shape"):
            mp = make_map(TRACK, shape)
            ss = rng.uniform(0.001, mp.TrackLength * 0.999, 200); eys = rng.uniform(-0.28, 0.28, 200)
            glob = np.array([[float(v) for v in mp.getGlobalPosition(s_, e_)] for s_, e_ in zip(ss, eys)])
            psis = glob[:, 2] + rng.normal(0, 0.3, 200)
            pts = np.column_stack([glob[:, 0], glob[:, 1], psis])
            pts = np.vstack([pts, np.column_stack([rng.uniform(-4, 4, 40), rng.uniform(-2, 5, 40), rng.uniform(-3, 3, 40)])])
            loc = np.array([[float(v) for v in mp.getLocalPosition(*p_)] for p_ in pts])
            pt.update({shape + "_s": ss, shape + "_ey": eys, shape + "_glob": glob, shape + "_pts": pts, shape + "_loc": loc,
                       shape + "_hw": np.array(mp.halfWidth), shape + "_slack": np.array(mp.slack)})
        sim = SIM.Simulator()
        useq = np.column_stack([rng.uniform(-1.0, 2.0, 300), rng.uniform(-0.25, 0.25, 300)])     # [a, delta]
        sts = []
        for u_ in useq:
            sim.f([float(u_[0]), float(u_[1])])
            sts.append([sim.x, sim.y, sim.vx, sim.vy, sim.ax, sim.ay, sim.yaw, sim.psiDot])
        pt.update(sim_u=useq, sim_states=np.array(sts), sim_init=np.array([0.01, 0.0, 0.2, 0.0, 0.0, 0.0, 0.0, 0.0]))

        # ---------------- closed loop: reference controller + reference plant + reference map ----------------
        # synchronous schedule (documented in DESIGN.md): per 30 Hz control tick the plant takes 7 steps of 5 ms;
        # measurement = ground truth; lap-0 path-tracking branch of controllerMain.py:179-190,289-331,381-386.
        PARAMS["trackShape"] = "oval"
        oval2 = make_map(TRACK, "oval")
        N = 20
        Q, R, dR = TUNINGS["path"]
        c = CTRL.PathFollowingLPV_MPC(Q, R, dR, N, 1, dt, oval2, "OSQP", 0, 0)
        sim = SIM.Simulator()
        sim.vx = 1.0
        first_it = 1
        cmd = [0.0, 0.0]                                    # [servo, motor]
        tr_plant, tr_local, tr_cmd, tr_iter = [], [], [], []
        for tick in range(40):
            G_ = np.array([sim.vx, sim.vy, sim.psiDot, sim.x, sim.y, sim.yaw])
            Lc = G_.copy()
            if Lc[0] < 0.01:
                Lc[0] = 0.01
            Lc[4], Lc[3], Lc[5], inside = oval2.getLocalPosition(G_[3], G_[4], G_[5])     # quirk Q9 (CMAIN:188)
            c.OldSteering.append(cmd[0]); c.OldAccelera.append(cmd[1])
            c.OldSteering.pop(0); c.OldAccelera.pop(0)
            if first_it < 10:
                xx, uu = lpv_ref.ctrl_seed_vectors(Lc)
                c.solve(Lc[0:6], xx, uu, False, np.ones(N), 0, 0, 0, first_it)
                first_it += 1
            else:
                S, A_L, B_L, C_L = c.LPVPrediction(Lc[0:6], c.uPred, np.ones(N + 1), np.zeros(N), 60.0, 0)
                c.solve(S[0, :], S, c.uPred, False, np.ones(N + 1), A_L, B_L, C_L, first_it)
            cmd = [float(c.uPred[0, 0]), float(c.uPred[0, 1])]
            tr_plant.append([sim.x, sim.y, sim.vx, sim.vy, sim.ax, sim.ay, sim.yaw, sim.psiDot])
            tr_local.append(Lc.copy()); tr_cmd.append(list(cmd)); tr_iter.append(CAPTURE[-1]["iter"])
            for _ in range(7):
                sim.f([cmd[1], cmd[0]])                     # u = [motor, servo]  (vehicleSimulator.py:330)
        pt.update(cl_plant=np.array(tr_plant), cl_local=np.array(tr_local), cl_cmd=np.array(tr_cmd), cl_iter=np.array(tr_iter))
        save_npz("plant_and_transforms.npz", pt)
        print("wrote plant_and_transforms.npz")
        gen_handoff(CTRL, PLAN, TRACK, UTIL)


if __name__ == "__main__":
    main()
