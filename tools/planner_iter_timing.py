#!/usr/bin/env python3
"""Cost of one ADMM iteration of the planner kernels (fixed iteration counts, no checks / rho updates / polish), per kernel variant.
NPLAN=40|30, B=512 (one residency at two instances per CU), VARIANTS=6,0,3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads

N = int(os.environ.get("NPLAN", "40"))
variants = [int(v) for v in os.environ.get("VARIANTS", "6,0,3" if N == 40 else "4,7,0,5").split(",")]
for B in [int(b) for b in os.environ.get("B", "512,256").split(",")]:
    w = workloads.planner_batch(B, N=N, seed=1)
    for v in variants:
        t = {}
        for it in (101, 301):
            eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=it); eng.set_timing(True)
            eng.set_option("kernel_variant", v)
            ms = []
            for _ in range(4):
                eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
                ms.append(eng.last_kernel_ms())
            t[it] = min(ms)
            eng.close()
        print("N = %d B = %d variant %d: %.3f us per iteration (%.3f ms at 101, %.3f ms at 301 iterations)" % (N, B, v, (t[301] - t[101]) / 200 * 1e3, t[101], t[301]), flush=True)
