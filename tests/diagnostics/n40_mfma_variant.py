#!/usr/bin/env python3
# (test infrastructure: uses the oracle as a checker)
"""Planner N = 40: DPP sweeps (default) against the MFMA sweeps (default; kernel_variant 3 = DPP) on a GPU box -- agreement and speed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads
from oracle import osqp_ref as O

w = workloads.planner_batch(2048, N=int(os.environ.get("NPLAN", "40")), seed=1)
ref = O.plan_tick_batch(w, nthreads=16)
outs = {}
for v in (3, 0, 3, 0):          # twice, alternating: the first timing of a process comes out slower (first-use set-up)
    eng = workloads.make_solver(w); eng.set_option("kernel_variant", v)
    o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    t0 = time.perf_counter()
    for _ in range(3):
        o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    outs[v] = o
    print("variant %d: %.1f ms per 2048-instance call" % (v, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    eng.close()
sane = ref["status"] != -10
for v in (3, 0):
    o = outs[v]
    fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & np.isfinite(o["uPred"]).all(axis=(1, 2)) & sane
    d = np.abs(o["uPred"][fin] - ref["uPred"][fin]).max(axis=(1, 2))
    print("variant %d vs oracle: status equal %d / %d, iters equal %d / %d, max |du| %.2e, > 1e-6: %d" % (
        v, np.sum(o["status"][sane] == ref["status"][sane]), sane.sum(), np.sum(o["iters"][sane] == ref["iters"][sane]), sane.sum(), d.max(), np.sum(d > 1e-6)))
a, b = outs[3], outs[0]
fin = np.isfinite(a["uPred"]).all(axis=(1, 2)) & np.isfinite(b["uPred"]).all(axis=(1, 2))
print("DPP (variant 3) vs default: status equal %d, iters equal %d of %d; max |du| %.2e" % (np.sum(a["status"] == b["status"]), np.sum(a["iters"] == b["iters"]), len(a["status"]),
                                                                          np.abs(a["uPred"][fin] - b["uPred"][fin]).max()))
