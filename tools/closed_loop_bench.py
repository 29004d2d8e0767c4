#!/usr/bin/env python3
"""Closed-loop fleet throughput (controller in the lap-0 path-tracking branch + plant + map, all on the device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc

Q, R, dR = np.diag([100.0, 1.0, 1.0, 20.0, 0.0, 900.0]), 0.25 * np.eye(2), 37.5 * np.array([1.3, 1.0])
mp = lpvmpc.Map("oval", 0.2)
for B in (1024, 8192):
    for warm in (0, 2):
        eng = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=mp.PointAndTangent)
        eng.set_option("warm_start", warm)
        rng = np.random.default_rng(3)
        s0 = rng.uniform(0.05, 12.5, B); ey0 = rng.normal(0, 0.03, B)
        xyth = eng.global_position(np.column_stack([s0, ey0]))
        plant0 = np.column_stack([xyth[:, 0], xyth[:, 1], rng.uniform(0.8, 1.2, B), np.zeros(B), np.zeros(B), np.zeros(B), xyth[:, 2], np.zeros(B)])
        eng.cl_init(plant0, mp.halfWidth, mp.slack, q9_swap=False, n_sub=7)
        eng.cl_tick(20); eng.cl_read()                     # seed phase + warm-up
        T = 200
        t = time.perf_counter(); eng.cl_tick(T); o = eng.cl_read(); t = time.perf_counter() - t
        ok = np.mean(np.isin(o["status"], (1, 2)))
        print("B=%5d warm_start=%d: %d ticks in %.3f s -> %.2f ms/tick, %.0f vehicle-ticks/s (real time factor %.1fx at 30 Hz), solved %.3f, last iters mean %.1f max %d"
              % (B, warm, T, t, t / T * 1e3, B * T / t, (T / 30.0) / t, ok, o["iters"].mean(), o["iters"].max()), flush=True)
        eng.close()
