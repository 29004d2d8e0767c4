#!/usr/bin/env python3
"""Workload of a counter pass over the planner kernels: B instances (default 512: two per CU) that never terminate, run for a fixed
number of ADMM iterations (no checks, rho updates or polish), one launch per kernel variant.
  rocprofv3 --pmc ... --kernel-trace --output-format csv -d <dir> -o c -- python3 tools/planner_pmc.py <N> <variant,variant,...> [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpvmpc import workloads

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "6,0").split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
w = workloads.planner_batch(B, N=N, seed=1)
for v in variants:
    for it in (100, 400):       # the difference of the two launches is 300 iterations of every instance
        eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=it)
        eng.set_option("kernel_variant", v)
        o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        print("variant", v, "iterations", np.unique(o["iters"]), flush=True)
        eng.close()
