#!/usr/bin/env python3
# (test infrastructure: uses the oracle as a checker; lives under tests/ for that reason)
"""Wide parity sweep on a GPU box: every kernel variant on large batches against the C oracle tick, plus run-to-run
reproducibility.  Prints one line per (workload, variant)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads
from oracle import osqp_ref as O

def run(w, variant):
    eng = workloads.make_solver(w)
    eng.set_option("kernel_variant", variant)
    if w["kind"] == "controller":
        f = lambda: eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    else:
        f = lambda: eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    a = f(); b = f()
    eng.close()
    return a, b

cases = [("ctrl N=20", workloads.controller_batch(2048, N=20, seed=0), (0, 1, 2)),
         ("ctrl N=10", workloads.controller_batch(1024, N=10, seed=5), (0, 1)),
         ("ctrl N=8", workloads.controller_batch(1024, N=8, seed=6), (0,)),
         ("plan N=30", workloads.planner_batch(2048, N=30, seed=1), (0, 1, 2)),
         ("plan N=40", workloads.planner_batch(2048, N=40, seed=1), (0, 1)),
         ("plan N=20", workloads.planner_batch(1024, N=20, seed=7), (0,))]
for name, w, variants in cases:
    t = time.time()
    ref = (O.ctrl_tick_batch if w["kind"] == "controller" else O.plan_tick_batch)(w, nthreads=16)
    tor = time.time() - t
    for v in variants:
        a, b = run(w, v)
        rep = np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["status"], b["status"]) and np.array_equal(a["uPred"], b["uPred"], equal_nan=True)
        fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & np.isfinite(a["uPred"]).all(axis=(1, 2))
        d = np.abs(a["uPred"][fin] - ref["uPred"][fin]).max(axis=(1, 2))
        nanmis = int(np.sum(np.isfinite(ref["uPred"]).all(axis=(1, 2)) != np.isfinite(a["uPred"]).all(axis=(1, 2))))
        bad = np.nonzero(fin)[0][d > 1e-4]
        print("%-10s variant %d: reproducible %s | status equal %d/%d, iters equal %d/%d, finite-mismatch %d | du <=1e-6: %d/%d, max %.2e | worst %s (oracle %.1fs)"
              % (name, v, rep, np.sum(a["status"] == ref["status"]), len(ref["status"]), np.sum(a["iters"] == ref["iters"]), len(ref["iters"]), nanmis,
                 np.sum(d <= 1e-6), fin.sum(), d.max() if d.size else 0.0, list(zip(bad[:4], a["status"][bad[:4]], a["iters"][bad[:4]], ref["iters"][bad[:4]])), tor), flush=True)
