#!/usr/bin/env python3
"""The four-wavefront controller kernel (kernel_variant 9) against the default two-wavefront one: outputs and time per solve at
batch sizes from one instance to a full chip.  tools/c4_probe.py [lib.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import _ffi
if len(sys.argv) > 1: _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
from lpvmpc import workloads

def run(w, variant, reps):
    eng = workloads.make_solver(w)
    eng.set_option("kernel_variant", variant)
    args = (w["x0"], w["u_prev"], w.get("vel_ref"), w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]) if w.get("max_ey") is None else \
           (w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    o = eng.solve(*args)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.solve(*args); ts.append(time.perf_counter() - t0)
    eng.close()
    return o, float(np.median(ts)) * 1e3

for B in (1, 16, 64, 256, 1024, 4096):
    w = workloads.controller_batch(B, N=20, seed=11)
    o2, t2 = run(w, 0, 30)
    o4, t4 = run(w, 9, 30)
    same = all(np.array(o2[k]).tobytes() == np.array(o4[k]).tobytes() for k in ("xPred", "uPred", "status", "iters", "polish"))
    dx = float(np.nanmax(np.abs(np.array(o2["xPred"]) - np.array(o4["xPred"]))))
    print("B=%-5d two wavefronts %.3f ms | four %.3f ms | %s (status equal %s, iterations equal %s, max |dx| %.2e; mean iterations %.1f)"
          % (B, t2, t4, "every word equal" if same else "DIFFERENT", np.array_equal(o2["status"], o4["status"]), np.array_equal(o2["iters"], o4["iters"]), dx, float(np.mean(o2["iters"]))))

for B in (1, 64, 256, 1024):
    w = workloads.planner_batch(B, N=20, seed=12)
    o2, t2 = run(w, 0, 10)
    o4, t4 = run(w, 9, 10)
    same = all(np.array(o2[k]).tobytes() == np.array(o4[k]).tobytes() for k in ("xPred", "uPred", "status", "iters", "polish"))
    fin = np.isfinite(np.array(o2["xPred"])).all(axis=(1, 2)) & np.isfinite(np.array(o4["xPred"])).all(axis=(1, 2))
    dx = float(np.max(np.abs(np.array(o2["xPred"])[fin] - np.array(o4["xPred"])[fin]))) if fin.any() else 0.0
    print("planner N=20 B=%-5d two wavefronts %.3f ms | four %.3f ms | %s (status equal %s, iterations equal %s, max |dx| %.2e; mean iterations %.1f)"
          % (B, t2, t4, "every word equal" if same else "DIFFERENT", np.array_equal(o2["status"], o4["status"]), np.array_equal(o2["iters"], o4["iters"]), dx, float(np.mean(o2["iters"]))))
