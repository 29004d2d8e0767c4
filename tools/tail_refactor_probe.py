#!/usr/bin/env python3
"""How much of a straggler's time in the whole-CU tail kernel goes to re-factorisations (rho updates) and the K^-1 builds behind them:
the default workload with the straggler deferral, stamps build STAMPS=2 (make -C csrc stamps STAMPS=2); prints the instances that ran
beyond 1000 iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "liblpvmpc_stamps.so")
from lpvmpc import workloads
for seed in range(1, 9):
    w = workloads.controller_batch(1024, N=20, seed=seed)
    eng = workloads.make_solver(w)
    eng.set_option("defer_pool", 1024); eng.set_option("defer_after", 100)
    o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    it = np.array(o["iters"]); r = np.array(o["resid"])
    for i in np.where(it > 1000)[0]:
        print("seed %d instance %4d: %4d iterations, status %2d | tail kernel: factorisation %.0f cycles, K^-1 build %.0f cycles (stamp 6: %.0f, 7: %.0f)" % (seed, i, it[i], o["status"][i], *r[i]))
    eng.close()
