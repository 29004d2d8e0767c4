#!/usr/bin/env python3
"""Run-to-run reproducibility of the fleet engines on a GPU box (lap-0 fleet, cascade with and without prefetch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads as W

def lap0(B, warm):
    Q, R, dR = W.CTRL_TUNINGS["path"]
    mp = lpvmpc.Map("oval", 0.2)
    eng = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Q, R, dR, track=mp.PointAndTangent)
    eng.set_option("warm_start", warm)
    rng = np.random.default_rng(3)
    s0 = rng.uniform(0.05, 12.5, B); ey0 = rng.normal(0, 0.03, B)
    xyth = eng.global_position(np.column_stack([s0, ey0]))
    plant0 = np.column_stack([xyth[:, 0], xyth[:, 1], rng.uniform(0.8, 1.2, B), np.zeros(B), np.zeros(B), np.zeros(B), xyth[:, 2], np.zeros(B)])
    eng.cl_init(plant0, mp.halfWidth, mp.slack, q9_swap=False, n_sub=7)
    eng.cl_tick(60)
    o = eng.cl_read(); eng.close()
    return o

def cascade(B, prefetch):
    c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
    mp = lpvmpc.Map("L_shape", 0.2)
    Qr, Rr, dRr = W.CTRL_TUNINGS["race"]
    rng = np.random.default_rng(3)
    plant0 = np.tile(c["plant0"], (B, 1))
    plant0[:, 1] += rng.normal(0, 0.01, B); plant0[:, 6] += rng.normal(0, 0.01, B); plant0[:, 2] += rng.uniform(-0.05, 0.3, B)
    plan = lpvmpc.BatchedSolver("planner", 40, 0.05, W.PLAN_Q, W.PLAN_R, W.PLAN_dR, L_cf=W.PLAN_L, track=mp.PointAndTangent)
    plan.handoff_setup()
    ctrl = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Qr, Rr, dRr, track=mp.PointAndTangent)
    ctrl.set_option("cascade_prefetch", prefetch)
    ctrl.cascade_init(plan, plant0, np.tile(c["cmd0"], (B, 1)), np.tile(c["uPred0"], (B, 1, 1)), half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
    ctrl.cascade_tick(45)
    o = ctrl.cascade_read(full=False); ctrl.close(); plan.close()
    return o

def same(a, b, keys):
    return all(np.array_equal(a[k], b[k], equal_nan=True) for k in keys)

for B, warm in ((4096, 0), (4096, 2)):
    a, b = lap0(B, warm), lap0(B, warm)
    print("lap-0 fleet B=%d warm=%d: reproducible %s" % (B, warm, same(a, b, ("plant", "cmd", "iters", "status"))), flush=True)
a, b, c = cascade(2048, 1), cascade(2048, 1), cascade(2048, 0)
print("cascade B=2048: prefetch run-to-run %s, prefetch vs none %s, alive %.3f" % (same(a, b, ("plant", "cmd", "iters", "status", "lap")), same(a, c, ("plant", "cmd", "iters", "status", "lap")),
      np.mean(np.all(np.isfinite(a["plant"]), axis=1))), flush=True)
