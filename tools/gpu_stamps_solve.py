#!/usr/bin/env python3
"""Shader cycles of wavefront 0 in the parts of a whole default solve (diagnostic build: make -C csrc stamps STAMPS=5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("LPVMPC_STAMPS_LIB", "liblpvmpc_stamps.so"))
from lpvmpc import workloads
names = ["set-up", "equilibration", "first factorisation", "ADMM iterations", "checks + rho updates", "polish factorisation", "polish (rest)", "output"]
for B in (256, 1024, 4096):
    w = workloads.controller_batch(B, N=20, seed=0)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    r = out["xPred"].reshape(B, -1)[:, :8]
    tot = r.sum(1)
    print("B=%d: mean cycles per solve %.0f (mean iterations %.1f)" % (B, tot.mean(), out["iters"].mean()))
    for i, n in enumerate(names):
        print("    %-24s %9.0f  %5.1f %%" % (n, r[:, i].mean(), 100 * r[:, i].mean() / tot.mean()))
    eng.close()
