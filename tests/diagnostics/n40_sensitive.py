#!/usr/bin/env python3
"""Which instances of the N = 40 planner batch of tests/test_gpu_parity.py::test_planner_n40_batch_against_oracle are
threshold sensitive (diagnostic; the test lists them explicitly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpvmpc import workloads
from oracle import osqp_ref as O
w = workloads.planner_batch(512, N=40, seed=1)
ref = O.plan_tick_batch(w, nthreads=8)
for variant in (0, 3):
    eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant)
    out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"]); eng.close()
    sane = ref["status"] != -10
    fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & sane
    d = np.full(512, 0.0); d[fin] = np.abs(out["uPred"][fin] - ref["uPred"][fin]).max(axis=(1, 2))
    print("variant", variant, "oracle gave up on", np.nonzero(~sane)[0].tolist())
    print("  iteration count differs:", [(int(i), int(out["iters"][i]), int(ref["iters"][i])) for i in np.nonzero((out["iters"] != ref["iters"]) & sane)[0]])
    print("  status differs:", [(int(i), int(out["status"][i]), int(ref["status"][i])) for i in np.nonzero((out["status"] != ref["status"]) & sane)[0]])
    print("  |du| > 1e-6:", [(int(i), float("%.2e" % d[i]), int(out["polish"][i]), int(ref["iters"][i])) for i in np.nonzero(d > 1e-6)[0]])
