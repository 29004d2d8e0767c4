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
