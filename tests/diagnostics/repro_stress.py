#!/usr/bin/env python3
"""Run-to-run reproducibility stress on a GPU box: large batches, several repetitions, all outputs compared bitwise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads

def runs(w, reps, variant=0, defer=0):
    eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant)
    if defer:                          # straggler deferral: the host-array call joins with the whole-CU tail kernel
        eng.set_option("defer_after", defer)
    out = []
    for _ in range(reps):
        if w["kind"] == "controller":
            o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
        else:
            o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        out.append(o)
    eng.close()
    return out

for name, w, reps in (("ctrl N=20 B=8192", workloads.controller_batch(8192, N=20, seed=21), 6),
                      ("ctrl N=20 B=65536", workloads.controller_batch(65536, N=20, seed=22), 3),
                      ("ctrl N=20 B=3000 (ragged)", workloads.controller_batch(3000, N=20, seed=23), 6),
                      ("plan N=30 B=4096", workloads.planner_batch(4096, N=30, seed=24), 4),
                      ("plan N=40 B=4096", workloads.planner_batch(4096, N=40, seed=25), 4),
                      ("ctrl N=8 B=8192", workloads.controller_batch(8192, N=8, seed=26), 6)):
    o = runs(w, reps)
    same = all(np.array_equal(o[0][k], r[k], equal_nan=True) for r in o[1:] for k in ("xPred", "uPred", "status", "iters", "polish", "resid"))
    print("%-28s %d runs bit-identical: %s (statuses %s)" % (name, reps, same, dict(zip(*np.unique(o[0]["status"], return_counts=True)))), flush=True)

# the tail kernel: deferred calls (parked at 100 iterations, finished by the whole-CU kernel at the join), run to run
for name, w, reps in (("ctrl N=20 B=8192, tail kernel", workloads.controller_batch(8192, N=20, seed=21), 6),
                      ("ctrl N=20 B=1024 seed 19, tail kernel", workloads.controller_batch(1024, N=20, seed=19), 8)):
    o = runs(w, reps, defer=100)
    same = all(np.array_equal(o[0][k], r[k], equal_nan=True) for r in o[1:] for k in ("xPred", "uPred", "status", "iters", "polish", "resid"))
    print("%-40s %d runs bit-identical: %s (parked: %d, max iterations %d)" % (name, reps, same, int((o[0]["iters"] > 100).sum()), int(o[0]["iters"].max())), flush=True)
