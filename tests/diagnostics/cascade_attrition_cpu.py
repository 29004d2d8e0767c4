#!/usr/bin/env python3
"""CPU-only: attrition of the ORACLE cascade (oracle/cascade_ref.py: the reference's planner / controller recursion with the
oracle's OSQP restatement) on the Monte-Carlo start distribution of bench.py --workload cfg5, and what kills each vehicle.
Prints per block of ticks the alive fraction, and for every loss the planner status and how far the planner's initial state
(= stage 1 of its previous plan, PMAIN:175-176) sits outside the state box.  Usage: cascade_attrition_cpu.py [B] [ticks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import cascade_ref, lpv_ref as L
from lpvmpc import workloads as W
from lpvmpc.track import Map

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
mp = Map("L_shape", 0.2)
rng = np.random.default_rng(3)
plant0 = np.tile(c["plant0"], (B, 1))
plant0[:, 1] += rng.normal(0, 0.01, B); plant0[:, 6] += rng.normal(0, 0.01, B); plant0[:, 2] += rng.uniform(-0.05, 0.3, B)
ref = cascade_ref.CascadeRef(mp.PointAndTangent, W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), plant0,
                             np.tile(c["cmd0"], (B, 1)), np.tile(c["uPred0"], (B, 1, 1)), half_width=mp.halfWidth, slack=mp.slack,
                             plan_max_ey=0.2, nthreads=8)
lo = np.array([L.DEFAULT_PARAMS["min_vel"], -1.0, -2.0, -0.2, -0.8]); hi = np.array([L.DEFAULT_PARAMS["max_vel"], 1.0, 2.0, 0.2, 0.8])
names = ["vx", "vy", "wz", "ey", "epsi"]
dead_at = np.full(B, -1)
prev_x1 = None
t0 = time.perf_counter()
for k in range(T):
    x1_before = None if ref.pxPred is None else ref.pxPred[:, 1, :].copy()
    pt = ref.plan_ticks
    ref.tick()
    if ref.plan_ticks != pt and x1_before is not None:
        st = ref.plan["status"]
        for b in range(B):
            if dead_at[b] < 0 and not np.all(np.isfinite(ref.pxPred[b])):
                dead_at[b] = k
                x0 = x1_before[b]
                viol = np.maximum(lo - x0, x0 - hi)
                j = int(np.argmax(viol))
                print("tick %4d (t = %5.2f s) vehicle %3d lost: planner status %d; x0 = previous plan's stage 1: %s outside its box by %+.2e (vx %.3f)"
                      % (k, k / 30.0, b, st[b], names[j], viol[j], x0[0]), flush=True)
    if (k + 1) % 75 == 0:
        print("t = %5.1f s: alive %.3f   (%.0f s of CPU)" % ((k + 1) / 30.0, np.mean(dead_at < 0), time.perf_counter() - t0), flush=True)
