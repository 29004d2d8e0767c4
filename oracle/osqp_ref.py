"""Oracle (TEST INFRASTRUCTURE): ctypes front end of ``osqp_ref.c``.

Builds ``oracle/_build/liboracle.so`` with gcc on first use (or through
``oracle/Makefile``).  See the header of ``osqp_ref.c`` for what is restated
and why parity with the real OSQP binary is *unpinned*.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_SRCS = ["osqp_ref.c", "lpv_ref.c"]

STATUS = {1: "solved", 2: "solved inaccurate", 3: "primal infeasible inaccurate",
          4: "dual infeasible inaccurate", -2: "maximum iterations reached",
          -3: "primal infeasible", -4: "dual infeasible", -7: "problem non convex",
          -10: "unsolved"}


class Settings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel",
                                           "eps_prim_inf", "eps_dual_inf", "delta",
                                           "adaptive_rho_tolerance")] + \
               [(k, C.c_int) for k in ("max_iter", "check_termination", "scaling", "adaptive_rho",
                                        "adaptive_rho_interval", "polish", "polish_refine_iter",
                                        "scaled_termination")]


class Info(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("iter", "status_val", "status_polish", "rho_updates")] + \
               [(k, C.c_double) for k in ("obj_val", "pri_res", "dua_res", "rho_estimate", "rho_final")]


def build(force=False):
    srcs = [os.path.join(_HERE, s) for s in _SRCS if os.path.exists(os.path.join(_HERE, s))]
    if not force and os.path.exists(_LIB) and all(
            os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", _LIB] + srcs + ["-lm"]
    subprocess.run(cmd, check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.osqp_ref_default_settings.argtypes = [C.POINTER(Settings)]
        _lib.osqp_ref_solve.restype = C.c_int
    return _lib


def default_settings(**over):
    s = Settings()
    lib().osqp_ref_default_settings(C.byref(s))
    for k, v in over.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


def kkt_ordering(P, A):
    """Fill-reducing ordering of the (n+m) KKT unknowns (stand-in for OSQP's AMD;
    the ordering only changes round-off)."""
    n = P.shape[0]
    m = A.shape[0]
    Pb = (abs(sp.csr_matrix(P)) > 0).astype(np.int8)
    Ab = (abs(sp.csr_matrix(A)) > 0).astype(np.int8)
    K = sp.bmat([[Pb + Pb.T + sp.eye(n, dtype=np.int8), Ab.T],
                 [Ab, sp.eye(m, dtype=np.int8)]], format="csr")
    return np.ascontiguousarray(reverse_cuthill_mckee(K, symmetric_mode=True), dtype=np.int32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def solve_qp(P, q, A, l, u, perm="rcm", **settings):
    """Solve min 1/2 x'Px + q'x s.t. l <= Ax <= u with the OSQP restatement.

    ``P`` (n,n) symmetric, ``A`` (m,n): dense arrays or scipy sparse.  Returns a
    namespace ``x, y, info`` where ``info`` carries iter / status_val / status /
    status_polish / obj_val / pri_res / dua_res / rho_updates / rho_estimate."""
    P = sp.csc_matrix(P)
    A = sp.csc_matrix(A)
    n, m = P.shape[0], A.shape[0]
    Pu = sp.triu(P, format="csc")
    Pu.sort_indices(); A.sort_indices()
    Pu.eliminate_zeros(); A.eliminate_zeros()
    q = np.ascontiguousarray(q, dtype=np.float64)
    l = np.ascontiguousarray(l, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    if isinstance(perm, str):
        perm = kkt_ordering(P, A) if perm == "rcm" else None
    s = settings.pop("settings", None) or default_settings(**settings)
    x = np.empty(n); y = np.empty(m)
    info = Info()
    Pp = Pu.indptr.astype(np.int32); Pi = Pu.indices.astype(np.int32); Px = Pu.data.astype(np.float64)
    Ap = A.indptr.astype(np.int32); Ai = A.indices.astype(np.int32); Ax = A.data.astype(np.float64)
    rc = lib().osqp_ref_solve(
        C.c_int(n), C.c_int(m), _ptr(Pp, C.c_int), _ptr(Pi, C.c_int), _ptr(Px, C.c_double), _ptr(q, C.c_double),
        _ptr(Ap, C.c_int), _ptr(Ai, C.c_int), _ptr(Ax, C.c_double), _ptr(l, C.c_double), _ptr(u, C.c_double),
        _ptr(perm, C.c_int) if perm is not None else None, C.byref(s), _ptr(x, C.c_double), _ptr(y, C.c_double),
        C.byref(info))
    if rc != 0:
        raise RuntimeError("osqp_ref_solve failed (KKT factorisation), rc=%d" % rc)
    return SimpleNamespace(
        x=x, y=y,
        info=SimpleNamespace(iter=info.iter, status_val=info.status_val,
                             status=STATUS.get(info.status_val, "?"), status_polish=info.status_polish,
                             obj_val=info.obj_val, pri_res=info.pri_res, dua_res=info.dua_res,
                             rho_updates=info.rho_updates, rho_estimate=info.rho_estimate,
                             rho_final=info.rho_final))


def ctrl_tick_batch(w, nthreads=1, params=None, warm=None, shift=True):
    """Whole controller tick (LPV roll-out + QP assembly + OSQP restatement) for a batch, in C
    (oracle/lpv_ref.c).  ``w`` is a workload dict (keys N, dt, Q, R, dR, track, x0, u_prev, vel_ref,
    curv_s, u_old, cf_new, lap).  Returns dict(xPred, uPred, status, iters, z, y); ``warm`` = a previous result
    dict whose (z, y) warm-start this tick (shifted by one stage when ``shift``)."""
    from .lpv_ref import DEFAULT_PARAMS
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    pv = np.array([p["lf"], p["lr"], p["m"], p["Iz"], p["Cf"], p["Cr"], p["mu"], p["max_vel"]], dtype=np.float64)
    N = int(w["N"])
    x0 = np.ascontiguousarray(w["x0"], np.float64); B = x0.shape[0]
    arrs = dict(Q=np.ascontiguousarray(w["Q"], np.float64), R=np.ascontiguousarray(w["R"], np.float64),
                dR=np.ascontiguousarray(w["dR"], np.float64), track=np.ascontiguousarray(w["track"], np.float64),
                u_prev=np.ascontiguousarray(w["u_prev"], np.float64), vel_ref=np.ascontiguousarray(w["vel_ref"], np.float64),
                curv=np.ascontiguousarray(w["curv_s"] if w["curv_s"] is not None else np.zeros((B, N)), np.float64),
                u_old=np.ascontiguousarray(w["u_old"], np.float64))
    xPred = np.empty((B, N + 1, 6)); uPred = np.empty((B, N, 2))
    status = np.empty(B, np.int32); iters = np.empty(B, np.int32)
    nz = (N + 1) * 6 + N * 2; m = 6 * N + (N + 1) * 6
    z = np.empty((B, nz)); y = np.empty((B, m))
    zw = yw = None
    if warm is not None:
        zw = np.ascontiguousarray(warm["z"], np.float64); yw = np.ascontiguousarray(warm["y"], np.float64)
    d = C.c_double
    f = lib().oracle_ctrl_tick_batch
    f.restype = C.c_int
    f(C.c_int(B), C.c_int(N), d(float(w["dt"])), _ptr(pv, d), _ptr(arrs["Q"], d), _ptr(arrs["R"], d), _ptr(arrs["dR"], d),
      _ptr(arrs["track"], d), C.c_int(arrs["track"].shape[0]), _ptr(x0, d), _ptr(arrs["u_prev"], d), _ptr(arrs["vel_ref"], d),
      _ptr(arrs["curv"], d), _ptr(arrs["u_old"], d), d(float(w["cf_new"])), C.c_int(int(w["lap"])),
      _ptr(xPred, d), _ptr(uPred, d), _ptr(status, C.c_int), _ptr(iters, C.c_int), C.c_int(int(nthreads)),
      _ptr(zw, d) if zw is not None else None, _ptr(yw, d) if yw is not None else None, C.c_int(1 if shift else 0),
      _ptr(z, d), _ptr(y, d))
    return dict(xPred=xPred, uPred=uPred, status=status, iters=iters, z=z, y=y)


def plan_tick_batch(w, nthreads=1, params=None):
    """Whole planner tick for a batch in C (oracle/lpv_ref.c); ``w`` as produced by workloads.planner_batch."""
    from .lpv_ref import DEFAULT_PARAMS
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    pv = np.array([p["lf"], p["lr"], p["m"], p["Iz"], p["Cf"], p["Cr"], p["mu"], p["max_vel"], p["min_vel"]], dtype=np.float64)
    N = int(w["N"])
    c = lambda a: np.ascontiguousarray(a, np.float64)
    x0 = c(w["x0"]); B = x0.shape[0]
    Q, R, dR, Lcf, track = c(w["Q"]), c(w["R"]), c(w["dR"]), c(w["L_cf"]), c(w["track"])
    u_prev, SS, u_old, mey = c(w["u_prev"]), c(w["curv_s"]), c(w["u_old"]), c(np.broadcast_to(w["max_ey"], (B,)))
    xPred = np.empty((B, N + 1, 5)); uPred = np.empty((B, N, 2))
    status = np.empty(B, np.int32); iters = np.empty(B, np.int32)
    d = C.c_double
    f = lib().oracle_plan_tick_batch
    f.restype = C.c_int
    f(C.c_int(B), C.c_int(N), d(float(w["dt"])), _ptr(pv, d), _ptr(Q, d), _ptr(R, d), _ptr(dR, d), _ptr(Lcf, d),
      _ptr(track, d), C.c_int(track.shape[0]), _ptr(x0, d), _ptr(u_prev, d), _ptr(SS, d), _ptr(u_old, d), _ptr(mey, d),
      _ptr(xPred, d), _ptr(uPred, d), _ptr(status, C.c_int), _ptr(iters, C.c_int), C.c_int(int(nthreads)))
    return dict(xPred=xPred, uPred=uPred, status=status, iters=iters)
