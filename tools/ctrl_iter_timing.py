#!/usr/bin/env python3
"""Cost of one ADMM iteration of the headline kernel in the PRODUCTION build (no stamps): fixed iteration counts, no checks / rho updates /
polish; the difference of two launches that differ by 200 iterations, for a batch that fills the chip once (B = 1024: four instances per
CU), a quarter of it (B = 256: one per CU) and four times over (B = 4096)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads

for B, variant in ((256, 0), (1024, 0), (4096, 0), (1, 0), (1, 9), (256, 9)):       # (variant 9: the four-wavefront latency form)
    w = workloads.controller_batch(B, N=20, seed=0)
    t = {}
    for it in (101, 301):
        eng = workloads.make_solver(w, adaptive_rho=0, polish=0, check_termination=0, max_iter=it); eng.set_timing(True)
        eng.set_option("kernel_variant", variant)
        ms = []
        for _ in range(5):
            eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
            ms.append(eng.last_kernel_ms())
        t[it] = min(ms)
        eng.close()
    us = (t[301] - t[101]) / 200 * 1e3 / max(1, B // 1024)
    print("controller N = 20, B = %d%s: %.3f us per iteration and residency (%.3f ms at 101, %.3f ms at 301 iterations)"
          % (B, ", four wavefronts per instance (kernel_variant 9)" if variant == 9 else "", us, t[101], t[301]), flush=True)
