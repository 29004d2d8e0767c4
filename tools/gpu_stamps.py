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

# planner N = 30, two wavefronts: DPP sweeps (kernel_variant 5) against the MFMA sweeps (kernel_variant 4)
for variant in (5, 4):
    w = workloads.planner_batch(512, N=30, seed=1)
    eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=200)
    eng.set_option("kernel_variant", variant)
    out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    r = out["resid"]
    print("planner N=30 variant %d B=512 cycles/iter: build_rhs %.0f | kkt fwd+pivot %.0f | kkt bwd %.0f | update %.0f | total %.0f" % (
        variant, *np.median(r, axis=0), np.median(r.sum(1))))
    eng.close()

# planner N = 40, four wavefronts per instance: the six barrier-to-barrier segments of iterate4
for B in (256, 512):
    w = workloads.planner_batch(B, N=40, seed=1)
    eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=200)
    out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    r = out["xPred"].reshape(B, -1)[:, :6]
    print("planner N=40 four wavefronts B=%d cycles/iter: rhs set 1 %.0f | outer forward + rhs set 2 %.0f | inner forward %.0f | inner backward %.0f | outer backward + update set 2 %.0f | update set 1 %.0f | total %.0f" % (
        B, *np.median(r, axis=0), np.median(r.sum(1))))
    eng.close()
    print("   outer top wavefront's forward (STAMPS=3 builds): barrier B0 to the first step %.0f | the ten steps %.0f" % tuple(np.median(out["xPred"].reshape(B, -1)[:, 6:8], axis=0)))
    r2 = out["xPred"].reshape(B, -1)[:, 8:16]
    print("   inner top wavefront: rhs set 2 + prologue %.0f | wait B1 %.0f | forward %.0f | wait B2 %.0f | backward %.0f | wait B3 %.0f | update set 2 %.0f | wait B4 %.0f" % tuple(np.median(r2, axis=0)))
