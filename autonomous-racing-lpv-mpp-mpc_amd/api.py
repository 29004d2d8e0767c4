"""Host-side mirror of the reference's operator interface for the solve path.

* ``PathFollowingLPV_MPC`` / ``LPV_MPC_Planner`` -- drop-ins with the reference's constructor, ``solve``
  and ``LPVPrediction`` signatures and result attributes (reference ControllerObject/PathFollowingLPVMPC.py:30-258,
  PlannerObject/LPV_MPC_Planner.py:29-320), each backed by a batch-of-one call into liblpvmpc.so.
* ``BatchedSolver`` -- the explicit batched interface (thousands of independent instances per launch).

All numerics run in the HIP library; this file only marshals numpy arrays.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import datetime

import numpy as np

from . import _ffi
from ._ffi import KIND_CONTROLLER, KIND_PLANNER, LpvMpcError, STATUS_TEXT, f64, ptr

# MAIN_LAUNCH.launch:5-11,40-41 -- used when neither rospy nor an explicit ``params`` dict is available
DEFAULT_PARAMS = dict(lf=0.125, lr=0.125, m=1.98, Iz=0.03, Cf=60.0, Cr=60.0, mu=0.05, max_vel=5.0, min_vel=0.9)
_ROS_NAMES = dict(lf="lf", lr="lr", m="m", Iz="Iz", Cf="Cf", Cr="Cr", mu="mu",
                  max_vel="/TrajectoryPlanner/max_vel", min_vel="/TrajectoryPlanner/min_vel")


def _vehicle_params(params, need_min_vel):
    """Parameters from an explicit dict, else from the ROS parameter server (CTRL:38-48 / PLAN:70-82),
    else the launch-file defaults."""
    out = dict(DEFAULT_PARAMS)
    if params is not None:
        out.update(params)
        return out
    try:
        import rospy  # noqa: F401  (only present on a ROS box)
    except ImportError:
        return out
    for key, ros_name in _ROS_NAMES.items():
        if key == "min_vel" and not need_min_vel:
            continue
        out[key] = rospy.get_param(ros_name)
    return out


class BatchedSolver:
    """Batched LPV-MPC (``kind="controller"``) / LPV-MPP (``kind="planner"``) solver on one MI355X.

    Parameters follow the reference constructors; ``track`` is a PointAndTangent table (or None when only
    caller-supplied curvature / LPV matrices are used); ``settings`` overrides OSQP settings and the
    controller limits (see ``_ffi.SETTING_FIELDS``)."""

    def __init__(self, kind, N, dt, Q, R, dR, L_cf=None, track=None, params=None, device=0, **settings):
        self.kind = {"controller": KIND_CONTROLLER, "planner": KIND_PLANNER}.get(kind, kind)
        if self.kind not in (KIND_CONTROLLER, KIND_PLANNER):
            raise ValueError("kind must be 'controller' or 'planner'")
        self.nx = 6 if self.kind == KIND_CONTROLLER else 5
        self.nu = 2
        self.N = int(N)
        lib = _ffi.load()
        cfg = _ffi.default_config(self.kind)
        cfg.N = self.N
        cfg.device = int(device)
        cfg.dt = float(dt)
        p = dict(DEFAULT_PARAMS)
        if params:
            p.update(params)
        for k in ("lf", "lr", "m", "Iz", "Cf", "Cr", "mu", "max_vel", "min_vel"):
            setattr(cfg, k, float(p[k]))
        Q = f64(Q, (self.nx, self.nx), "Q")
        R = f64(R, (2, 2), "R")
        dR = f64(dR, (2,), "dR")
        for i in range(36):
            cfg.Q[i] = 0.0
        for i, v in enumerate(Q.reshape(-1)):
            cfg.Q[i] = v
        for i, v in enumerate(R.reshape(-1)):
            cfg.R[i] = v
        cfg.dR[0], cfg.dR[1] = dR
        for i in range(6):
            cfg.L_cf[i] = 0.0
        if self.kind == KIND_PLANNER:
            if L_cf is None:
                raise ValueError("the planner needs L_cf")
            for i, v in enumerate(f64(L_cf, (5,), "L_cf")):
                cfg.L_cf[i] = v
        for k, v in settings.items():
            if k not in _ffi.SETTING_FIELDS:
                raise TypeError("unknown setting %r" % k)
            setattr(cfg, k, v)
        if track is not None:
            tab = f64(track, name="track")
            if tab.ndim != 2 or tab.shape[1] != 6 or tab.shape[0] > _ffi.MAX_TRACK_ROWS:
                raise ValueError("track table must be (rows<=%d, 6)" % _ffi.MAX_TRACK_ROWS)
            cfg.track_rows = tab.shape[0]
            for i, v in enumerate(tab.reshape(-1)):
                cfg.track[i] = v
        self.cfg = cfg
        self._lib = lib
        self._h = lib.lpvmpc_create(C.byref(cfg))
        if not self._h:
            msg = lib.lpvmpc_last_error(None)
            raise LpvMpcError(lib.lpvmpc_last_error_code() or _ffi.E_ARG, msg.decode() if msg else "lpvmpc_create failed")
        self._ho_M = 0                       # samples per My_Planning array once handoff_setup() has run
        self._cas = self._cas_planner = None

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.lpvmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _ffi.check(self._h, rc)

    def reserve(self, B):
        self._chk(self._lib.lpvmpc_reserve(self._h, int(B)))

    def set_option(self, name, value):
        self._chk(self._lib.lpvmpc_set_option(self._h, name.encode(), int(value)))

    def set_timing(self, on=True):
        self._chk(self._lib.lpvmpc_set_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        return float(self._lib.lpvmpc_last_kernel_ms(self._h))

    def join(self, stream=0):
        """Order ``stream`` behind the outstanding resume launches of the straggler deferral (option "defer_after"): the
        outputs of the deferred calls are complete for work enqueued on ``stream`` afterwards."""
        self._chk(self._lib.lpvmpc_join(self._h, C.c_void_p(int(stream))))

    def defer_stats(self):
        """(parked, refused) of the straggler deferral since the handle's first deferred call (lpvmpc_defer_stats: waits for the
        stream of the last deferred call)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.lpvmpc_defer_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def resume_time_stats(self):
        tot = C.c_double(0.0); n = C.c_int32(0)
        self._chk(self._lib.lpvmpc_resume_time_stats(self._h, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    def kernel_time_stats(self):
        """(total_ms, launches) of the solve kernel since set_timing(True)."""
        tot, n = C.c_double(0.0), C.c_int32(0)
        self._chk(self._lib.lpvmpc_kernel_time_stats(self._h, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    # -- host-array entry points ------------------------------------------------------------------
    def lpv(self, x0, u_prev, vel_ref=None, curv_s=None, cf_new=60.0, lap=1):
        """Batched LPVPrediction.  Returns (states [B,N,nx], A [B,N,nx,nx], Bm [B,N,nx,2])."""
        N, nx = self.N, self.nx
        x0 = f64(x0).reshape(-1, nx)
        B = x0.shape[0]
        u_prev = f64(u_prev, (B, N, 2), "u_prev")
        ctrl = self.kind == KIND_CONTROLLER
        vel_ref = f64(vel_ref, (B, N + 1), "vel_ref") if ctrl else None
        if curv_s is not None:
            curv_s = f64(curv_s, (B, N) if ctrl else (B, N + 1), "curv_ref" if ctrl else "SS")
        states = np.empty((B, N, nx)); A = np.empty((B, N, nx, nx)); Bm = np.empty((B, N, nx, 2))
        self._chk(self._lib.lpvmpc_lpv_batch(self._h, B, ptr(x0), ptr(u_prev), ptr(vel_ref), ptr(curv_s),
                                             float(cf_new), int(lap), ptr(states), ptr(A), ptr(Bm)))
        return states, A, Bm

    def estimate_abc(self, xlast, delta):
        """Batched _EstimateABC.  xlast [B,N,6], delta [B,N] -> (A, Bm)."""
        N, nx = self.N, self.nx
        xlast = f64(xlast)
        B = xlast.shape[0]
        xlast = f64(xlast, (B, N, 6), "xlast")
        delta = f64(delta, (B, N), "delta")
        A = np.empty((B, N, nx, nx)); Bm = np.empty((B, N, nx, 2))
        self._chk(self._lib.lpvmpc_estimate_abc_batch(self._h, B, ptr(xlast), ptr(delta), ptr(A), ptr(Bm)))
        return A, Bm

    def _outputs(self, B):
        N, nx = self.N, self.nx
        return dict(xPred=np.empty((B, N + 1, nx)), uPred=np.empty((B, N, 2)), status=np.empty(B, np.int32),
                    iters=np.empty(B, np.int32), resid=np.empty((B, 4)), polish=np.empty(B, np.int32))

    def solve_AB(self, x0, A, Bm, vel_ref=None, u_old=None, max_ey=None):
        """QP build + ADMM solve with caller-supplied LPV matrices.  Returns a dict of arrays
        xPred [B,N+1,nx], uPred [B,N,2], status, iters, resid [B,4]=(pri,dua,obj,rho), polish."""
        N, nx = self.N, self.nx
        x0 = f64(x0).reshape(-1, nx)
        B = x0.shape[0]
        A = f64(A, (B, N, nx, nx), "A"); Bm = f64(Bm, (B, N, nx, 2), "B")
        ctrl = self.kind == KIND_CONTROLLER
        vel_ref = f64(vel_ref, (B, N + 1), "vel_ref") if ctrl else None
        u_old = None if u_old is None else f64(u_old, (B, 2 + self.cfg.steering_delay), "u_old")
        max_ey = None if ctrl else f64(np.broadcast_to(np.asarray(max_ey, float), (B,)), (B,), "max_ey")
        o = self._outputs(B)
        self._chk(self._lib.lpvmpc_solve_batch_AB(self._h, B, ptr(x0), ptr(A), ptr(Bm), ptr(vel_ref), ptr(u_old),
                                                  ptr(max_ey), ptr(o["xPred"]), ptr(o["uPred"]), ptr(o["status"]),
                                                  ptr(o["iters"]), ptr(o["resid"]), ptr(o["polish"])))
        return o

    def solve(self, x0, u_prev, vel_ref=None, curv_s=None, u_old=None, max_ey=None, cf_new=60.0, lap=1):
        """Fused tick: LPV evaluation + roll-out, QP build, ADMM solve (host arrays in/out)."""
        N, nx = self.N, self.nx
        x0 = f64(x0).reshape(-1, nx)
        B = x0.shape[0]
        u_prev = f64(u_prev, (B, N, 2), "u_prev")
        ctrl = self.kind == KIND_CONTROLLER
        vel_ref = f64(vel_ref, (B, N + 1), "vel_ref") if ctrl else None
        if curv_s is not None:
            curv_s = f64(curv_s, (B, N) if ctrl else (B, N + 1), "curv_ref" if ctrl else "SS")
        u_old = None if u_old is None else f64(u_old, (B, 2 + self.cfg.steering_delay), "u_old")
        max_ey = None if ctrl else f64(np.broadcast_to(np.asarray(max_ey, float), (B,)), (B,), "max_ey")
        o = self._outputs(B)
        self._chk(self._lib.lpvmpc_solve_batch(self._h, B, ptr(x0), ptr(u_prev), ptr(vel_ref), ptr(curv_s), ptr(u_old),
                                               ptr(max_ey), float(cf_new), int(lap), ptr(o["xPred"]), ptr(o["uPred"]),
                                               ptr(o["status"]), ptr(o["iters"]), ptr(o["resid"]), ptr(o["polish"])))
        return o

    # -- caller-side helpers and the closed-loop fleet ------------------------------------
    def local_position(self, xy_psi, half_width, slack):
        """Batched Map.getLocalPosition: [B,3] -> [B,4] = (s, ey, epsi, inside)."""
        a = f64(xy_psi).reshape(-1, 3); out = np.empty((a.shape[0], 4))
        self._chk(self._lib.lpvmpc_local_position_batch(self._h, a.shape[0], ptr(a), float(half_width), float(slack), ptr(out)))
        return out

    def global_position(self, s_ey):
        """Batched Map.getGlobalPosition: [B,2] -> [B,3] = (x, y, theta)."""
        a = f64(s_ey).reshape(-1, 2); out = np.empty((a.shape[0], 3))
        self._chk(self._lib.lpvmpc_global_position_batch(self._h, a.shape[0], ptr(a), ptr(out)))
        return out

    def plant_step(self, state, u, n_sub=1, dt_sim=0.005, mu_sim=0.05):
        """n_sub steps of Simulator.f on state [B,8] under u [B,2] = (a, delta); returns the new state."""
        st = f64(state).reshape(-1, 8).copy(); u = f64(u, (st.shape[0], 2), "u")
        self._chk(self._lib.lpvmpc_plant_step_batch(self._h, st.shape[0], ptr(st), ptr(u), int(n_sub), float(dt_sim), float(mu_sim)))
        return st

    def cl_init(self, plant0, half_width, slack, q9_swap=True, n_sub=7, dt_sim=0.005, mu_sim=0.05):
        p0 = f64(plant0).reshape(-1, 8)
        self._cl_B = p0.shape[0]
        self._chk(self._lib.lpvmpc_cl_init(self._h, self._cl_B, ptr(p0), float(half_width), float(slack), 1 if q9_swap else 0,
                                           int(n_sub), float(dt_sim), float(mu_sim)))

    def cl_tick(self, n_ticks=1):
        self._chk(self._lib.lpvmpc_cl_tick(self._h, int(n_ticks)))

    def cl_read(self):
        B = self._cl_B
        o = dict(plant=np.empty((B, 8)), local=np.empty((B, 6)), cmd=np.empty((B, 2)), iters=np.empty(B, np.int32),
                 status=np.empty(B, np.int32))
        self._chk(self._lib.lpvmpc_cl_read(self._h, ptr(o["plant"]), ptr(o["local"]), ptr(o["cmd"]), ptr(o["iters"]), ptr(o["status"])))
        return o

    # -- planner -> controller hand-off ------------------------------------------------
    def handoff_setup(self, cfg=None):
        """Build the resampling / filtering operators for this PLANNER handle; returns M (samples per My_Planning array)."""
        cfg = cfg if cfg is not None else _ffi.default_handoff_config()
        M = self._lib.lpvmpc_handoff_setup(self._h, C.byref(cfg))
        self._chk(min(M, 0))
        self._ho_M = M
        return M

    def handoff(self, xPred, SS, pose, want_sig=False):
        """PMAIN:201-224,257-308 for a batch.  Returns dict(SS, pose, refs[, sig]); SS / pose are the carried state."""
        N = self.N
        x = f64(xPred).reshape(-1, N + 1, 5)
        B = x.shape[0]
        SS = f64(SS, (B, N + 1), "SS").copy(); pose = f64(pose, (B, 3), "pose").copy()
        refs = np.empty((B, 5, self._ho_M)); sig = np.empty((B, 5, N)) if want_sig else None
        self._chk(self._lib.lpvmpc_handoff_batch(self._h, B, ptr(x), ptr(SS), ptr(pose), ptr(sig), ptr(refs)))
        out = dict(SS=SS, pose=pose, refs=refs)
        if want_sig:
            out["sig"] = sig
        return out

    # -- planner + controller + plant cascade (configs[4]) ------------------------------------------------
    def cl_release(self):
        """End the closed-loop fleet / cascade of this engine: batch calls are accepted again (they are refused while a fleet
        runs, because the fleet's state lives in the engine's workspace)."""
        self._chk(self._lib.lpvmpc_cl_release(self._h))
        self._cas = None
        self._cas_planner = None

    def cascade_init(self, planner, plant0, cmd0, uPred0, lap0=1, half_width=0.3, slack=0.15, plan_max_ey=0.2, q9_swap=True,
                     n_sub=(7, 7, 6), dt_sim=0.005, mu_sim=0.05):
        p0 = f64(plant0).reshape(-1, 8)
        B = p0.shape[0]
        c0 = f64(cmd0, (B, 2), "cmd0"); u0 = f64(uPred0, (B, self.N, 2), "uPred0")
        ns = np.ascontiguousarray(n_sub, np.int32)
        if ns.shape != (3,):
            raise ValueError("n_sub must have 3 entries")
        self._chk(self._lib.lpvmpc_cascade_init(self._h, planner._h, B, ptr(p0), ptr(c0), ptr(u0), int(lap0), float(half_width),
                                                float(slack), float(plan_max_ey), 1 if q9_swap else 0, ptr(ns), float(dt_sim), float(mu_sim)))
        self._cas = (B, planner.N, planner._ho_M)
        self._cas_planner = planner                      # keep the planner handle alive as long as the cascade

    def cascade_tick(self, n_ticks=1):
        self._chk(self._lib.lpvmpc_cascade_tick(self._h, int(n_ticks)))

    def cascade_read(self, full=True):
        if self._cas is None:
            raise LpvMpcError(_ffi.E_ARG, "cascade_read: call cascade_init first")
        B, Np, M = self._cas
        o = dict(plant=np.empty((B, 8)), local=np.empty((B, 6)), cmd=np.empty((B, 2)), iters=np.empty(B, np.int32),
                 status=np.empty(B, np.int32), lap=np.empty(B, np.int32), lap_tick=np.empty(B, np.int32),
                 plan_iters=np.empty(B, np.int32), plan_status=np.empty(B, np.int32), ticks=np.empty(2, np.int32))
        if full:
            o.update(refs=np.empty((B, 5, M)), plan_xPred=np.empty((B, Np + 1, 5)))
        self._chk(self._lib.lpvmpc_cascade_read(self._h, ptr(o["plant"]), ptr(o["local"]), ptr(o["cmd"]), ptr(o["iters"]), ptr(o["status"]),
                                                ptr(o["lap"]), ptr(o["lap_tick"]), ptr(o.get("refs")), ptr(o.get("plan_xPred")),
                                                ptr(o["plan_iters"]), ptr(o["plan_status"]), ptr(o["ticks"])))
        return o

    def cascade_alive_ticks(self):
        """Controller ticks each vehicle of the cascade has entered with a finite plant state ([B] int32)."""
        if self._cas is None:
            raise LpvMpcError(_ffi.E_ARG, "cascade_alive_ticks: call cascade_init first")
        out = np.empty(self._cas[0], np.int32)
        self._chk(self._lib.lpvmpc_cascade_alive_ticks(self._h, ptr(out)))
        return out

    # -- device-pointer entry point (torch tensors or raw integers) -----------------------------------
    def solve_dev(self, B, x0, u_prev, vel_ref, curv_s, u_old, max_ey, xPred, uPred, status=None, iters=None,
                  resid=None, polish=None, cf_new=60.0, lap=1, stream=0):
        """Enqueue the fused tick on ``stream`` with DEVICE pointers (ints or objects with ``data_ptr()``)."""
        def dp(t):
            if t is None:
                return None
            return C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        self._chk(self._lib.lpvmpc_solve_batch_dev(self._h, int(B), dp(x0), dp(u_prev), dp(vel_ref), dp(curv_s),
                                                   dp(u_old), dp(max_ey), float(cf_new), int(lap), dp(xPred),
                                                   dp(uPred), dp(status), dp(iters), dp(resid), dp(polish),
                                                   C.c_void_p(int(stream))))


def handoff_operators(N, dt, cfg=None):
    """Host-only: the hand-off operators for an N-sample planner horizon at period dt (no device needed).
    Returns (W, FW), each (M, N): refs = W @ signal for x, y, yaw, vx and FW @ curvature (PMAIN:257-280)."""
    lib = _ffi.load()
    cfg = cfg if cfg is not None else _ffi.default_handoff_config()
    M = lib.lpvmpc_handoff_length(int(N), float(dt), C.byref(cfg))
    _ffi.check(None, min(M, 0))
    W = np.empty((M, int(N))); FW = np.empty((M, int(N)))
    _ffi.check(None, min(lib.lpvmpc_handoff_operators(int(N), float(dt), C.byref(cfg), ptr(W), ptr(FW)), 0))
    return W, FW


# =====================================================================================================
# drop-in classes
# =====================================================================================================
def _stack_list(L, shape, name):
    a = np.stack([np.asarray(m, dtype=np.float64) for m in L])
    if a.shape != shape:
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, shape))
    return a


class _DropInBase(object):
    # ---- the assembled QP, as the reference leaves it on the object (CTRL:79,108-110; PLAN:108) -----------------------------
    # The device path never forms these matrices; they are built on the host (qp_matrices.py) from the data of the last
    # solve() the first time one of them is read, and dropped by the next solve().
    _qp_cache = None
    _qp_inputs = None

    def _qp(self):
        if self._qp_cache is None:
            if self._qp_inputs is None:
                raise AttributeError("the QP matrices exist after the first solve()")
            self._qp_cache = self._assemble_qp(**self._qp_inputs)
        return self._qp_cache

    def qp_matrices(self):
        """(P, q, A, l, u) of the last solve in the form the reference hands to OSQP (CTRL:303-308: inequality rows first;
        PLAN:200-202: equality rows first), dense float64."""
        m = self._qp()
        return m["P"], m["q"], m["A"], m["l"], m["u"]

    E = property(lambda self: self._qp()["E"])
    L = property(lambda self: self._qp()["L"])
    Eu = property(lambda self: self._qp()["Eu"])
    q = property(lambda self: self._qp()["q"])

    def _finish(self, out, start):
        nx, N = self.n_states, self.N
        self.status_val = int(out["status"][0])
        self.status = STATUS_TEXT.get(self.status_val, "?")
        self.iters = int(out["iters"][0])
        self.status_polish = int(out["polish"][0])
        # reference: SOLVED / SOLVED_INACCURATE / MAX_ITER_REACHED count as feasible (CTRL:322-324, PLAN:214-216)
        self.feasible = 1 if self.status_val in (1, 2, -2) else 0
        if self.status_val != 1 and self.verbose_status:
            print("OSQP exited with status '%s'" % self.status)
        if self.feasible == 0:
            print("QUIT...")
        xPred = out["xPred"][0].copy()
        self.xPred = xPred
        self.uPred = out["uPred"][0].copy()
        self.LinPoints = np.concatenate((xPred[1:, :], xPred[-1:, :]), axis=0)
        self.solverTime = datetime.datetime.now() - start


class PathFollowingLPV_MPC(_DropInBase):
    """Drop-in for the reference class of the same name (CTRL:30-258).

    Same constructor arguments; ``params`` (vehicle parameters dict) and ``device`` are optional extras for
    use without ROS.  ``Solver`` is accepted for signature compatibility: the solve always runs the HIP
    ADMM path (the reference's mains only ever pass "OSQP", controllerMain.py:142,150).
    ``steeringDelay`` > 0 adds the pinned-steering equality rows of CTRL:518-527 (``OldSteering`` then has 1 + delay
    entries, CTRL:71); ``velocityDelay`` is stored and, as in the reference, never used."""

    def __init__(self, Q, R, dR, N, vt, dt, map, Solver="OSQP", steeringDelay=0, velocityDelay=0,
                 params=None, device=0, **settings):
        p = _vehicle_params(params, need_min_vel=False)
        self.lf, self.lr, self.m, self.I = p["lf"], p["lr"], p["m"], p["Iz"]
        self.Cf, self.Cr, self.mu, self.g = p["Cf"], p["Cr"], p["mu"], 9.81
        self.max_vel = p["max_vel"]
        self.A, self.B, self.C = [], [], []
        self.N = int(N)
        self.n = self.n_states = np.asarray(Q).shape[0]
        self.d = np.asarray(R).shape[0]
        self.vt = vt
        self.Q, self.R, self.dR = np.asarray(Q, float), np.asarray(R, float), np.asarray(dR, float)
        self.LinPoints = np.zeros((self.N + 2, self.n))
        self.dt = dt
        self.map = map
        self.halfWidth = map.halfWidth
        self.first_it = 1
        self.steeringDelay = steeringDelay
        self.velocityDelay = velocityDelay
        self.OldSteering = [0.0] * int(1 + steeringDelay)
        self.OldAccelera = [0.0] * int(1)
        self.OldPredicted = [0.0] * int(1 + steeringDelay + N)
        self.Solver = Solver
        self.verbose_status = True
        self._eng = BatchedSolver("controller", N, dt, self.Q, self.R, self.dR, track=map.PointAndTangent,
                                  params=p, device=device, steering_delay=int(steeringDelay), **settings)
        # one vehicle per handle: the latency form of the N = 20 kernel (four wavefronts per instance; lpvmpc.h, kernel_variant 9 --
        # other horizons and steeringDelay > 0 take their default kernels under it)
        self._eng.set_option("kernel_variant", 9)

    # CTRL:108-110 leaves G, E, L, Eu, M, q on the object; F, b exist from the constructor on (CTRL:79)
    G = property(lambda self: self._qp()["G"])
    M = property(lambda self: self._qp()["M"])

    def _bounds(self):
        c = self._eng.cfg
        return dict(vx_min=c.ctrl_vx_min, delta_max=c.ctrl_delta_max, a_max=c.ctrl_a_max, a_min_abs=c.ctrl_a_min_abs)

    @property
    def F(self):
        return self._Fb()[0]

    @property
    def b(self):
        return self._Fb()[1]

    def _Fb(self):
        if getattr(self, "_Fb_cache", None) is None:
            from . import qp_matrices
            self._Fb_cache = qp_matrices.controller_inequalities(self.N, self.n, self.d, self.max_vel, **self._bounds())
        return self._Fb_cache

    def _assemble_qp(self, **kw):
        from . import qp_matrices
        return qp_matrices.controller_qp(self.Q, self.R, self.dR, self.N, self.A, self.B, self.C, max_vel=self.max_vel,
                                         bounds=self._bounds(), **kw)

    def solve(self, x0, Last_xPredicted, uPred, NN_LPV_MPC, vel_ref, A_L, B_L, C_L, first_it):
        """CTRL:89-162.  Results in .xPred (N+1,6), .uPred (N,2), .LinPoints; returns None."""
        start = datetime.datetime.now()
        N = self.N
        uPred = np.asarray(uPred, dtype=np.float64)
        if (NN_LPV_MPC == False) and (first_it < 10):                    # noqa: E712  (CTRL:99-100)
            xl = np.asarray(Last_xPredicted, dtype=np.float64)[:N, :6]
            A, Bm = self._eng.estimate_abc(xl[None], uPred[:N, 0][None])
            A, Bm = A[0], Bm[0]
        else:
            A = _stack_list(A_L, (N, 6, 6), "A_L")
            Bm = _stack_list(B_L, (N, 6, 2), "B_L")
        self.A = [A[i] for i in range(N)]
        self.B = [Bm[i] for i in range(N)]
        self.C = [np.zeros((6, 1)) for _ in range(N)]
        vr = np.asarray(vel_ref, dtype=np.float64).reshape(-1)
        vfull = np.concatenate((vr[:N], vr[-1:]))                         # CTRL:434-438: stage N tracks vel_ref[-1]
        d = int(self.steeringDelay)
        # CTRL:395 (uOld) followed by the pinned commands OldSteering[1 .. delay] of CTRL:523
        u_old = np.array([[self.OldSteering[0], self.OldAccelera[0]] + [float(v) for v in self.OldSteering[1:1 + d]]], dtype=np.float64)
        self._qp_cache = None
        self._qp_inputs = dict(x0=np.array(x0, dtype=np.float64).reshape(6), u_old=u_old[0, :2].copy(), vel_ref=vfull.copy(),
                               steer_hist=tuple(u_old[0, 2:]))
        self.linearizationTime = datetime.datetime.now() - start
        start = datetime.datetime.now()
        out = self._eng.solve_AB(np.asarray(x0, float).reshape(1, 6), A[None], Bm[None], vfull[None], u_old)
        self._finish(out, start)

    def LPVPrediction(self, x, u, vel_ref, curv_ref, Cf_new, LapNumber):
        """CTRL:166-258.  Returns (STATES_vec (N,6), Atv, Btv, Ctv) with Atv/Btv/Ctv lists of N arrays."""
        N = self.N
        vr = np.asarray(vel_ref, dtype=np.float64).reshape(-1)
        vfull = np.concatenate((vr[:N], vr[-1:]))
        curv = None
        if LapNumber != 0:
            curv = np.asarray(curv_ref, dtype=np.float64).reshape(-1)[:N][None]
        S, A, Bm = self._eng.lpv(np.asarray(x, float).reshape(1, 6), np.asarray(u, float)[:N, :2][None], vfull[None],
                                 curv, cf_new=float(Cf_new), lap=int(LapNumber))
        return (S[0], [A[0, i] for i in range(N)], [Bm[0, i] for i in range(N)],
                [np.zeros((6, 1)) for _ in range(N)])


class LPV_MPC_Planner(_DropInBase):
    """Drop-in for the reference LPV-MPP planner class (PLAN:29-320)."""

    def __init__(self, Q, R, dR, L_cf, N, dt, map, Solver="OSQP", params=None, device=0, **settings):
        p = _vehicle_params(params, need_min_vel=True)
        self.A, self.B, self.C = [], [], []
        self.N = int(N)
        self.nx = self.n_states = np.asarray(Q).shape[0]
        self.nu = np.asarray(R).shape[0]
        self.Q = np.asarray(Q, float); self.QN = self.Q
        self.R, self.dR, self.L_cf = np.asarray(R, float), np.asarray(dR, float), np.asarray(L_cf, float)
        self.LinPoints = np.zeros((self.N + 2, self.nx))
        self.dt = dt
        self.map = map
        self.halfWidth = map.halfWidth
        self.first_it = 1
        self.Solver = Solver
        self.steeringDelay = 0
        self.OldSteering = [0.0]
        self.OldAccelera = [0.0]
        self.lf, self.lr, self.m, self.I = p["lf"], p["lr"], p["m"], p["Iz"]
        self.Cf, self.Cr, self.mu, self.g, self.epss = p["Cf"], p["Cr"], p["mu"], 9.81, 0.00000001
        self.max_vel, self.min_vel = p["max_vel"], p["min_vel"]
        self.verbose_status = False                                       # PLAN:212-213: status print commented out
        self._eng = BatchedSolver("planner", N, dt, self.Q, self.R, self.dR, L_cf=self.L_cf,
                                  track=map.PointAndTangent, params=p, device=device, **settings)
        # one vehicle per handle: the latency form where a horizon has one of its own (N = 20; N = 30 / 40 take four wavefronts for a
        # lone instance by default) -- lpvmpc.h, kernel_variant 9
        self._eng.set_option("kernel_variant", 9)

    Aeq = property(lambda self: self._qp()["Aeq"])                      # PLAN:108

    def _assemble_qp(self, **kw):
        from . import qp_matrices
        c = self._eng.cfg
        xlo, xhi = np.array(c.plan_xmin[:]), np.array(c.plan_xmax[:])
        xlo[0], xhi[0] = self.min_vel, self.max_vel                       # PLAN:176-177
        xlo[3], xhi[3] = -kw["max_ey"], kw["max_ey"]
        return qp_matrices.planner_qp(self.Q, self.R, self.dR, self.L_cf, self.N, self.A, self.B, self.C, min_vel=self.min_vel,
                                      max_vel=self.max_vel, xbox=(xlo, xhi), ubox=(np.array(c.plan_umin[:]), np.array(c.plan_umax[:])), **kw)

    def solve(self, x0, Last_xPredicted, uPred, A_LPV, B_LPV, C_LPV, first_it, max_ey):
        """PLAN:86-236."""
        start = datetime.datetime.now()
        N = self.N
        if first_it < 2:                                                  # PLAN:99-100
            xl = np.asarray(Last_xPredicted, dtype=np.float64)[:N, :6]
            delta = np.asarray(uPred, dtype=np.float64).reshape(-1)[:N] if np.ndim(uPred) < 2 or np.shape(uPred)[1] == 1 \
                else np.asarray(uPred, dtype=np.float64)[:N, 0]
            A, Bm = self._eng.estimate_abc(xl[None], delta[None])
            A, Bm = A[0], Bm[0]
        else:
            A = _stack_list(A_LPV, (N, 5, 5), "A_LPV")
            Bm = _stack_list(B_LPV, (N, 5, 2), "B_LPV")
        self.A = [A[i] for i in range(N)]
        self.B = [Bm[i] for i in range(N)]
        self.C = [np.zeros((5, 1)) for _ in range(N)]
        u_old = np.array([[self.OldSteering[0], self.OldAccelera[0]]], dtype=np.float64)   # PLAN:114 (quirk Q3)
        self._qp_cache = None
        self._qp_inputs = dict(x0=np.array(x0, dtype=np.float64).reshape(5), u_old=u_old[0].copy(), max_ey=float(max_ey))
        out = self._eng.solve_AB(np.asarray(x0, float).reshape(1, 5), A[None], Bm[None], None, u_old,
                                 max_ey=float(max_ey))
        self._finish(out, start)

    def LPVPrediction(self, x, SS, u):
        """PLAN:242-320."""
        N = self.N
        S, A, Bm = self._eng.lpv(np.asarray(x, float).reshape(1, 5), np.asarray(u, float)[:N, :2][None], None,
                                 np.asarray(SS, float).reshape(-1)[:N + 1][None])
        return (S[0], [A[0, i] for i in range(N)], [Bm[0, i] for i in range(N)],
                [np.zeros((5, 1)) for _ in range(N)])


# =====================================================================================================
# caller-side helpers of the two nodes, single vehicle
# =====================================================================================================
class PlannerHandoff(object):
    """The post-processing block of plannerMain.py for one vehicle: ``refs = PlannerHandoff(Planner).update()`` replaces
    PMAIN:112 (filter design), :189-224 (s integration, pose reconstruction) and :257-280 (resampling, filtering).

    State carried between ticks, as in the node: ``SS`` (N+1,), ``Xlast / Ylast / Thetalast``.  After ``update()``:
    ``xp, yp, yaw, vel, curv`` (N,) at the planner's rate and ``x_d, y_d, psi_d, vx_d, curv_d`` (M,) -- the five arrays
    of the My_Planning message (PMAIN:303-307).  Computed on the device through lpvmpc_handoff_batch."""

    def __init__(self, planner, handoff_config=None):
        self._eng = planner._eng
        self.N = planner.N
        self.M = self._eng.handoff_setup(handoff_config)
        self.SS = np.zeros(self.N + 1)
        self.Xlast = self.Ylast = self.Thetalast = 0.0
        self._planner = planner

    def update(self, xPred=None):
        xPred = self._planner.xPred if xPred is None else xPred
        o = self._eng.handoff(np.asarray(xPred, float)[None], self.SS[None], np.array([[self.Xlast, self.Ylast, self.Thetalast]]),
                              want_sig=True)
        self.SS = o["SS"][0]
        self.Xlast, self.Ylast, self.Thetalast = (float(v) for v in o["pose"][0])
        self.xp, self.yp, self.yaw, self.vel, self.curv = (o["sig"][0, i] for i in range(5))
        self.x_d, self.y_d, self.psi_d, self.vx_d, self.curv_d = (o["refs"][0, i] for i in range(5))
        return o["refs"][0]


def body_frame_errors(x, y, psi, xd, yd, psid, s0, vx, vy, curv, dt):
    """Body_Frame_Errors of controllerMain.py:495-506 -> (s, ex, ey, epsi); host arithmetic, a dozen flops."""
    ex = (x - xd) * np.cos(psid) + (y - yd) * np.sin(psid)
    ey = -(x - xd) * np.sin(psid) + (y - yd) * np.cos(psid)
    d = psi - psid
    epsi = 2 * np.pi + d if d < -np.pi else (d - 2 * np.pi if d > np.pi else d)            # TRACK:413-421 wrap()
    s = s0 + ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - ey * curv)) * dt
    return s, ex, ey, epsi
