"""ctypes binding of liblpvmpc.so (C ABI in include/lpvmpc.h).

There is no CPU fallback: if the shared library is missing this module raises at load time, and if no
HIP device is usable ``lpvmpc_create`` fails with LPVMPC_E_NODEVICE which surfaces as ``LpvMpcError``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblpvmpc.so")

KIND_CONTROLLER, KIND_PLANNER = 0, 1
MAX_TRACK_ROWS, MAX_N = 16, 52
E_ARG, E_NODEVICE, E_HIP, E_NOMEM = -1, -2, -3, -4

STATUS_TEXT = {1: "solved", 2: "solved inaccurate", 3: "primal infeasible inaccurate",
               4: "dual infeasible inaccurate", -2: "maximum iterations reached",
               -3: "primal infeasible", -4: "dual infeasible", -7: "problem non convex", -10: "unsolved"}

_d, _i = C.c_double, C.c_int32


class Config(C.Structure):
    """Mirror of ``struct lpvmpc_config`` (include/lpvmpc.h) -- keep field order identical."""
    _fields_ = [
        ("kind", _i), ("N", _i), ("device", _i), ("steering_delay", _i),
        ("dt", _d),
        ("lf", _d), ("lr", _d), ("m", _d), ("Iz", _d), ("Cf", _d), ("Cr", _d), ("mu", _d),
        ("max_vel", _d), ("min_vel", _d),
        ("Q", _d * 36), ("R", _d * 4), ("dR", _d * 2), ("L_cf", _d * 6),
        ("ctrl_vx_min", _d), ("ctrl_delta_max", _d), ("ctrl_a_max", _d), ("ctrl_a_min_abs", _d),
        ("plan_xmin", _d * 5), ("plan_xmax", _d * 5), ("plan_umin", _d * 2), ("plan_umax", _d * 2),
        ("rho", _d), ("sigma", _d), ("alpha", _d), ("eps_abs", _d), ("eps_rel", _d),
        ("eps_prim_inf", _d), ("eps_dual_inf", _d), ("polish_delta", _d), ("adaptive_rho_tolerance", _d),
        ("max_iter", _i), ("check_termination", _i), ("scaling", _i), ("adaptive_rho", _i),
        ("adaptive_rho_interval", _i), ("polish", _i), ("polish_refine_iter", _i), ("reserved1", _i),
        ("track_rows", _i), ("reserved2", _i),
        ("track", _d * (MAX_TRACK_ROWS * 6)),
    ]


SETTING_FIELDS = ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "polish_delta",
                  "adaptive_rho_tolerance", "max_iter", "check_termination", "scaling", "adaptive_rho",
                  "adaptive_rho_interval", "polish", "polish_refine_iter",
                  "ctrl_vx_min", "ctrl_delta_max", "ctrl_a_max", "ctrl_a_min_abs", "steering_delay")


class LpvMpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lpvmpc error %d: %s" % (code, msg))
        self.code = code


MAX_FILTER_ORDER = 8


class HandoffConfig(C.Structure):
    """Mirror of ``struct lpvmpc_handoff_config`` (include/lpvmpc.h)."""
    _fields_ = [("interp_dt", _d), ("padlen", _i), ("order", _i), ("b", _d * (MAX_FILTER_ORDER + 1)), ("a", _d * (MAX_FILTER_ORDER + 1))]


EXPORTS = ("lpvmpc_version", "lpvmpc_default_config", "lpvmpc_create", "lpvmpc_destroy", "lpvmpc_last_error", "lpvmpc_last_error_code",
           "lpvmpc_reserve", "lpvmpc_lpv_batch", "lpvmpc_estimate_abc_batch", "lpvmpc_solve_batch_AB",
           "lpvmpc_solve_batch", "lpvmpc_solve_batch_dev", "lpvmpc_last_kernel_ms", "lpvmpc_set_timing",
           "lpvmpc_kernel_time_stats", "lpvmpc_set_option",
           "lpvmpc_local_position_batch", "lpvmpc_global_position_batch", "lpvmpc_plant_step_batch",
           "lpvmpc_cl_init", "lpvmpc_cl_tick", "lpvmpc_cl_read", "lpvmpc_cl_release", "lpvmpc_join", "lpvmpc_resume_time_stats", "lpvmpc_defer_stats",
           "lpvmpc_handoff_default_config", "lpvmpc_handoff_length", "lpvmpc_handoff_operators", "lpvmpc_handoff_setup",
           "lpvmpc_handoff_batch", "lpvmpc_cascade_init", "lpvmpc_cascade_tick", "lpvmpc_cascade_read", "lpvmpc_cascade_alive_ticks")

_lib = None


def _bind_to_torchs_hip_runtime():
    """A PyTorch-ROCm wheel ships its own HIP / HSA runtime (torch/lib/libamdhip64.so, same SONAME as the system's).  Two HIP
    runtimes cannot both own the GPU in one process: with the system runtime loaded first (by liblpvmpc.so), torch's later
    initialisation reports "No HIP GPUs are available"; with torch's loaded first, liblpvmpc.so binds to it and streams and device
    pointers are shared.  So, where such a wheel is installed and torch is not imported yet, its runtime is loaded here (the
    shared object only -- torch itself is not imported).  LPVMPC_SYSTEM_HIP=1 keeps the system runtime (processes that never use torch)."""
    if "torch" in sys.modules or os.environ.get("LPVMPC_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for d in (spec.submodule_search_locations if spec else []):
            cand = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
    except Exception:           # no torch, or an unusual layout: the system runtime it is
        pass


def load():
    """Load liblpvmpc.so; in a process that has (or may later import) a PyTorch-ROCm wheel it binds to that wheel's HIP runtime
    (see _bind_to_torchs_hip_runtime), so the order of `import torch` and the first lpvmpc call does not matter."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found -- build it with `python __graft_entry__.py` (or `make -C %s`); "
                          "there is no CPU fallback" % (LIB_PATH, os.path.join(_HERE, "csrc")))
    _bind_to_torchs_hip_runtime()
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if "torch" in sys.modules else C.RTLD_LOCAL)
    P = C.POINTER
    vp = C.c_void_p
    lib.lpvmpc_version.restype = C.c_int
    lib.lpvmpc_default_config.argtypes = [_i, P(Config)]
    lib.lpvmpc_default_config.restype = None
    lib.lpvmpc_create.argtypes = [P(Config)]
    lib.lpvmpc_create.restype = vp
    lib.lpvmpc_destroy.argtypes = [vp]
    lib.lpvmpc_destroy.restype = None
    lib.lpvmpc_last_error.argtypes = [vp]
    lib.lpvmpc_last_error.restype = C.c_char_p
    lib.lpvmpc_last_error_code.argtypes = []
    lib.lpvmpc_last_error_code.restype = C.c_int
    lib.lpvmpc_reserve.argtypes = [vp, _i]
    lib.lpvmpc_lpv_batch.argtypes = [vp, _i, vp, vp, vp, vp, _d, _i, vp, vp, vp]
    lib.lpvmpc_estimate_abc_batch.argtypes = [vp, _i, vp, vp, vp, vp]
    lib.lpvmpc_solve_batch_AB.argtypes = [vp, _i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lpvmpc_solve_batch.argtypes = [vp, _i, vp, vp, vp, vp, vp, vp, _d, _i, vp, vp, vp, vp, vp, vp]
    lib.lpvmpc_solve_batch_dev.argtypes = [vp, _i, vp, vp, vp, vp, vp, vp, _d, _i, vp, vp, vp, vp, vp, vp, vp]
    lib.lpvmpc_last_kernel_ms.argtypes = [vp]
    lib.lpvmpc_last_kernel_ms.restype = _d
    lib.lpvmpc_set_timing.argtypes = [vp, _i]
    lib.lpvmpc_set_option.argtypes = [vp, C.c_char_p, _i]
    lib.lpvmpc_set_option.restype = C.c_int
    lib.lpvmpc_kernel_time_stats.argtypes = [vp, P(_d), P(_i)]
    lib.lpvmpc_kernel_time_stats.restype = C.c_int
    lib.lpvmpc_local_position_batch.argtypes = [vp, _i, vp, _d, _d, vp]
    lib.lpvmpc_global_position_batch.argtypes = [vp, _i, vp, vp]
    lib.lpvmpc_plant_step_batch.argtypes = [vp, _i, vp, vp, _i, _d, _d]
    lib.lpvmpc_cl_init.argtypes = [vp, _i, vp, _d, _d, _i, _i, _d, _d]
    lib.lpvmpc_cl_tick.argtypes = [vp, _i]
    lib.lpvmpc_cl_read.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.lpvmpc_cl_release.argtypes = [vp]
    lib.lpvmpc_join.argtypes = [vp, vp]
    lib.lpvmpc_resume_time_stats.argtypes = [vp, P(_d), P(_i)]
    try:        # (tools/ab_equal.py / ab_lib.sh load older builds of the library through this module: round 5's has no such export)
        lib.lpvmpc_defer_stats.argtypes = [vp, P(C.c_int64), P(C.c_int64)]
        lib.lpvmpc_defer_stats.restype = C.c_int
    except AttributeError:
        pass
    lib.lpvmpc_handoff_default_config.argtypes = [P(HandoffConfig)]
    lib.lpvmpc_handoff_default_config.restype = None
    lib.lpvmpc_handoff_length.argtypes = [_i, _d, P(HandoffConfig)]
    lib.lpvmpc_handoff_operators.argtypes = [_i, _d, P(HandoffConfig), vp, vp]
    lib.lpvmpc_handoff_setup.argtypes = [vp, P(HandoffConfig)]
    lib.lpvmpc_handoff_batch.argtypes = [vp, _i, vp, vp, vp, vp, vp]
    lib.lpvmpc_cascade_init.argtypes = [vp, vp, _i, vp, vp, vp, _i, _d, _d, _d, _i, vp, _d, _d]
    lib.lpvmpc_cascade_tick.argtypes = [vp, _i]
    lib.lpvmpc_cascade_read.argtypes = [vp] + [vp] * 12
    lib.lpvmpc_cascade_alive_ticks.argtypes = [vp, vp]
    lib.lpvmpc_cascade_alive_ticks.restype = C.c_int
    for name in ("lpvmpc_handoff_length", "lpvmpc_handoff_operators", "lpvmpc_handoff_setup", "lpvmpc_handoff_batch",
                 "lpvmpc_cascade_init", "lpvmpc_cascade_tick", "lpvmpc_cascade_read"):
        getattr(lib, name).restype = C.c_int
    for name in ("lpvmpc_local_position_batch", "lpvmpc_global_position_batch", "lpvmpc_plant_step_batch",
                 "lpvmpc_cl_init", "lpvmpc_cl_tick", "lpvmpc_cl_read", "lpvmpc_cl_release", "lpvmpc_join", "lpvmpc_resume_time_stats"):
        getattr(lib, name).restype = C.c_int
    for name in ("lpvmpc_reserve", "lpvmpc_lpv_batch", "lpvmpc_estimate_abc_batch", "lpvmpc_solve_batch_AB",
                 "lpvmpc_solve_batch", "lpvmpc_solve_batch_dev", "lpvmpc_set_timing"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def default_config(kind):
    cfg = Config()
    load().lpvmpc_default_config(kind, C.byref(cfg))
    return cfg


def default_handoff_config():
    cfg = HandoffConfig()
    load().lpvmpc_handoff_default_config(C.byref(cfg))
    return cfg


def f64(a, shape=None, name="array"):
    """Contiguous float64 copy/view with an optional exact shape check."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(handle, rc):
    if rc != 0:
        msg = load().lpvmpc_last_error(handle)
        raise LpvMpcError(rc, msg.decode() if msg else "?")
