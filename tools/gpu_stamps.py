#!/usr/bin/env python3
"""Per-phase shader cycles per ADMM iteration from the diagnostic stamps build (liblpvmpc_stamps.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "liblpvmpc_stamps.so")
from lpvmpc import workloads
for B in (256, 1024):
    w = workloads.controller_batch(B, N=20, seed=0)
    eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=200)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    r = out["resid"]
    print("B=%d cycles/iter: build_rhs %.0f | kkt fwd+pivot %.0f | kkt bwd %.0f | update %.0f | total %.0f (median over instances)" % (
        B, *np.median(r, axis=0), np.median(r.sum(1))))
    eng.close()
