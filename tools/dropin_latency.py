#!/usr/bin/env python3
"""Single-instance latency of the drop-in classes (what one ROS node tick would see)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads

mp = lpvmpc.Map("oval", 0.2)
Q, R, dR = workloads.CTRL_TUNINGS["race"]
c = lpvmpc.PathFollowingLPV_MPC(Q, R, dR, 20, 1, 1 / 30.0, mp, "OSQP", 0, 0)
x = np.array([1.5, 0.02, 0.1, 0.03, 3.0, -0.02]); u = np.tile([0.01, 0.2], (20, 1))
vel = np.full(21, 1.5); curv = np.zeros(20)
tl, ts = [], []
for i in range(200):
    t0 = time.perf_counter(); S, A, B, C = c.LPVPrediction(x, u, vel, curv, 60.0, 1); t1 = time.perf_counter()
    c.solve(x, 0.0, u, False, vel, A, B, C, 10); t2 = time.perf_counter()
    tl.append(t1 - t0); ts.append(t2 - t1)
print("controller N=20, one instance: LPVPrediction p50 %.3f ms, solve p50 %.3f ms (p99 %.3f), iters %d; reference budget 33 ms/tick,"
      " reference Python pre-solver cost alone 2.4 ms (BASELINE.md)" % (np.median(tl[20:]) * 1e3, np.median(ts[20:]) * 1e3, np.percentile(ts[20:], 99) * 1e3, c.iters))
p = lpvmpc.LPV_MPC_Planner(workloads.PLAN_Q, workloads.PLAN_R, workloads.PLAN_dR, workloads.PLAN_L, 30, 0.05, lpvmpc.Map("L_shape", 0.2), "OSQP")
x = np.array([2.0, 0.0, 0.0, 0.01, 0.01]); SS = 1.0 + np.arange(31) * 2.0 * 0.05; u = np.tile([0.0, 0.2], (30, 1))
tl, ts = [], []
for i in range(100):
    t0 = time.perf_counter(); S, A, B, C = p.LPVPrediction(x, SS, u); t1 = time.perf_counter()
    p.solve(x, 0, 0, A, B, C, 2, 0.2); t2 = time.perf_counter()
    tl.append(t1 - t0); ts.append(t2 - t1)
print("planner N=30, one instance: LPVPrediction p50 %.3f ms, solve p50 %.3f ms, iters %d, status %s" % (np.median(tl[10:]) * 1e3, np.median(ts[10:]) * 1e3, p.iters, p.status))
