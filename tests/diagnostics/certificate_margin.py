#!/usr/bin/env python3
# (test infrastructure: uses the oracle; CPU only)
"""Margins of the primal-infeasibility certificate of ONE instance in the CPU oracle, check by check.
usage: certificate_margin.py SHAPE SEED N INDEX      (a planner instance of tests/diagnostics/seed_sweep.py: seed = 200 + sweep seed)
Round 6: names the one instance of the 110 202-instance sweep whose iteration count differs from the oracle's
(Euge_Track, sweep seed 4, planner N = 40, #393: the device's certificate fires at the check of iteration 50, the oracle's at 75)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["OSQP_REF_TRACE"] = "1"
import numpy as np
from lpvmpc import workloads
from oracle import lpv_ref as L, osqp_ref as O

shape, seed, N, j = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = workloads.planner_batch(512, N=N, seed=seed, shape=shape)
p = dict(L.DEFAULT_PARAMS)
S, A, Bm = L.plan_lpv_prediction(p, w["dt"], N, w["track"], w["x0"][j], w["curv_s"][j], w["u_prev"][j])
qp = L.plan_build_qp(w["Q"], w["R"], w["dR"], w["L_cf"], N, A, Bm, w["x0"][j], w["u_old"][j], float(w["max_ey"][j]), p["max_vel"], p["min_vel"])
r = O.solve_qp(qp.P, qp.q, qp.A, qp.l, qp.u)
print("oracle: status %d after %d iterations" % (r.info.status_val, r.info.iter))
