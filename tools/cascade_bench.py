#!/usr/bin/env python3
"""Planner + controller + plant cascade for a Monte-Carlo fleet (BASELINE.json configs[4] shape, one GPU's share):
throughput in vehicle-ticks/s, real-time factor, how many vehicles are still alive (the reference's planner QP turns
infeasible for some starts -- its forward-Euler model is unstable below ~1.5 m/s at dt = 0.05 -- and those vehicles carry NaN
from then on), laps completed.

Start: the state at the reference's lap event (tests/golden/cascade.npz: plant0 / cmd0 / uPred0) with per-vehicle
perturbations of lateral position, heading and speed (seed 3).
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--fleets", default="1024,8192")
ap.add_argument("--ticks", type=int, default=300)
ap.add_argument("--modes", default="0:0:1,0:0:0,2:1:1", help="ctrl_warm:plan_warm:prefetch triples")
ap.add_argument("--spread", type=float, default=0.01)
args = ap.parse_args()

c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
mp = lpvmpc.Map("L_shape", 0.2)
Qr, Rr, dRr = W.CTRL_TUNINGS["race"]
for B in [int(v) for v in args.fleets.split(",")]:
    rng = np.random.default_rng(3)
    plant0 = np.tile(c["plant0"], (B, 1))
    plant0[:, 1] += rng.normal(0, args.spread, B); plant0[:, 6] += rng.normal(0, args.spread, B); plant0[:, 2] += rng.uniform(-0.05, 0.3, B)
    cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))
    for mode in args.modes.split(","):
        cw, pw, pf = [int(v) for v in mode.split(":")]
        plan = lpvmpc.BatchedSolver("planner", 40, 0.05, W.PLAN_Q, W.PLAN_R, W.PLAN_dR, L_cf=W.PLAN_L, track=mp.PointAndTangent)
        plan.handoff_setup()
        ctrl = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Qr, Rr, dRr, track=mp.PointAndTangent)
        ctrl.set_option("warm_start", cw); plan.set_option("warm_start", pw); ctrl.set_option("cascade_prefetch", pf)
        ctrl.set_timing(True); plan.set_timing(True)
        ctrl.cascade_init(plan, plant0, cmd0, uPred0, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2)
        ctrl.cascade_tick(3); ctrl.cascade_read(full=False)                 # first planner tick (seed) + warm-up
        T = args.ticks
        t = time.perf_counter(); ctrl.cascade_tick(T); o = ctrl.cascade_read(full=False); t = time.perf_counter() - t
        alive = np.all(np.isfinite(o["plant"]), axis=1)
        cms, cn = ctrl.kernel_time_stats(); pms, pn = plan.kernel_time_stats()
        print("B=%5d ctrl_warm=%d plan_warm=%d prefetch=%d: %d ctrl ticks (%d planner ticks) in %.3f s -> %.2f ms/tick, %.0f vehicle-ticks/s, "
              "real-time factor %.2fx; alive %.4f, laps max %d; solve kernels: ctrl %.2f ms avg, planner %.2f ms avg; last iters ctrl %.0f / planner %.0f (alive mean)"
              % (B, cw, pw, pf, T, o["ticks"][1], t, t / T * 1e3, B * T / t, (T / 30.0) / t, alive.mean(), o["lap"][alive].max() if alive.any() else -1,
                 cms / max(cn, 1), pms / max(pn, 1), o["iters"][alive].mean() if alive.any() else 0, o["plan_iters"][alive].mean() if alive.any() else 0), flush=True)
        ctrl.close(); plan.close()
