#!/usr/bin/env python3
"""Phase timing of the solve kernel by differencing runs with different settings (B instances, controller N=20)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads

def run(B, reps=5, kind="controller", **settings):
    w = workloads.controller_batch(B, N=20, seed=0) if kind == "controller" else workloads.planner_batch(B, N=30, seed=1)
    eng = workloads.make_solver(w, **settings); eng.set_timing(True)
    ms = []
    for _ in range(reps):
        out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], w["max_ey"], w["cf_new"], w["lap"])
        ms.append(eng.last_kernel_ms())
    eng.close()
    return min(ms), out

for B in (256, 1024):
    base = dict(adaptive_rho=0, polish=0, check_termination=0)
    t1, _ = run(B, max_iter=1, **base)
    t101, _ = run(B, max_iter=101, **base)
    t201, _ = run(B, max_iter=201, **base)
    t1ns, _ = run(B, max_iter=1, scaling=0, **base)
    print("B=%d: 1 iter %.3f ms | 101 iters %.3f ms | 201 iters %.3f ms -> per-iteration %.2f us (B instances, %d resident rounds) | no-scaling 1 iter %.3f ms -> scaling %.3f ms"
          % (B, t1, t101, t201, (t201 - t101) / 100 * 1e3, 1, t1ns, t1 - t1ns), flush=True)
    tp, o = run(B, max_iter=50, adaptive_rho=0, polish=1, check_termination=50, eps_abs=1e3, eps_rel=1e3)   # terminates at iter 50 as "solved" -> polish
    tn, _ = run(B, max_iter=50, adaptive_rho=0, polish=0, check_termination=50, eps_abs=1e3, eps_rel=1e3)
    print("   polish cost %.3f ms (status %s)" % (tp - tn, np.unique(o["status"])), flush=True)
    tr, _ = run(B, max_iter=100, adaptive_rho=1, adaptive_rho_interval=25, adaptive_rho_tolerance=1.0000001, polish=0, check_termination=0)
    tq, _ = run(B, max_iter=100, adaptive_rho=0, polish=0, check_termination=0)
    print("   4 forced refactorisations + residual evaluations cost %.3f ms -> %.3f ms each" % (tr - tq, (tr - tq) / 4), flush=True)
    tc, _ = run(B, max_iter=100, adaptive_rho=0, polish=0, check_termination=25, eps_abs=1e-12, eps_rel=1e-12)
    print("   4 termination checks cost %.3f ms -> %.3f ms each" % (tc - tq, (tc - tq) / 4), flush=True)
