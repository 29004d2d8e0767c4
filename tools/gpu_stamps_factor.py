#!/usr/bin/env python3
"""Shader cycles spent in the phases of the KKT factorisation (diagnostic build: make -C csrc stamps STAMPS=2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("LPVMPC_STAMPS_LIB", "liblpvmpc_stamps.so"))
from lpvmpc import workloads
for B in (256, 1024):
    w = workloads.controller_batch(B, N=20, seed=0)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    r = out["resid"]
    print("B=%d cycles per solve in factor(): kd_block %.0f | schur_step %.0f | chol_inverse %.0f | publish_and_invert %.0f | sum %.0f (median; 10 chain steps x ~3 factorisations)" % (
        B, *np.median(r, axis=0), np.median(r.sum(1))))
    eng.close()
