#!/usr/bin/env python3
"""Survival of the planner + controller + plant cascade against the start-speed distribution (diagnostic for configs[4]):
alive fraction and laps after each block of ticks, for several speed offsets added to the lap-event state of
tests/golden/cascade.npz.  Usage: cascade_survival.py [B] [blocks] [ticks_per_block]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
mp = lpvmpc.Map("L_shape", 0.2)
Qr, Rr, dRr = W.CTRL_TUNINGS["race"]
print("lap-event state: vx = %.3f m/s" % c["plant0"][2])
for lo, hi, spread in ((-0.05, 0.3, 0.01), (0.0, 0.0, 0.0), (0.5, 0.8, 0.01), (1.0, 1.3, 0.01), (1.5, 2.0, 0.01), (1.0, 1.3, 0.03)):
    rng = np.random.default_rng(3)
    plant0 = np.tile(c["plant0"], (B, 1))
    plant0[:, 1] += rng.normal(0, spread, B) if spread else 0.0
    plant0[:, 6] += rng.normal(0, spread, B) if spread else 0.0
    plant0[:, 2] += rng.uniform(lo, hi, B)
    plan = lpvmpc.BatchedSolver("planner", 40, 0.05, W.PLAN_Q, W.PLAN_R, W.PLAN_dR, L_cf=W.PLAN_L, track=mp.PointAndTangent)
    plan.handoff_setup()
    ctrl = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Qr, Rr, dRr, track=mp.PointAndTangent)
    ctrl.cascade_init(plan, plant0, np.tile(c["cmd0"], (B, 1)), np.tile(c["uPred0"], (B, 1, 1)), half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
    line = []
    t0 = time.perf_counter()
    for k in range(blocks):
        ctrl.cascade_tick(per); o = ctrl.cascade_read(full=False)
        alive = np.all(np.isfinite(o["plant"]), axis=1)
        laps = o["lap"][alive] if alive.any() else np.zeros(1, int)
        line.append("t=%4.0fs alive %.3f laps[min %d med %d max %d] vx[%.2f..%.2f]" % ((k + 1) * per / 30.0, alive.mean(), laps.min(), np.median(laps), laps.max(),
                    o["plant"][alive, 2].min() if alive.any() else 0, o["plant"][alive, 2].max() if alive.any() else 0))
    print("dv ~ U(%.2f, %.2f), spread %.2f (%.1f s): " % (lo, hi, spread, time.perf_counter() - t0) + " | ".join(line), flush=True)
    ctrl.close(); plan.close()
