#!/usr/bin/env python3
# (test infrastructure: uses the oracle as a checker; lives under tests/ for that reason)
"""Seed / track sweep on a GPU box: default kernels against the C oracle tick on many random batches.
Every instance whose uPred differs from the oracle's by more than 1e-6 is listed (round 5): device status, polish flag, iteration
count, |du|, and -- on the instance's own QP, rebuilt on the host -- the objective of both points relative to each other and
the primal residual of the device's point against OSQP's tolerance (tests/_tolerance.py holds the same rules for the suite)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads
from oracle import osqp_ref as O
from tests import _tolerance as T

tot = dict(n=0, st=0, it=0, close=0, fin=0)
cls = {}
worst = 0.0
SHAPES = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("oval", "L_shape", "3110", "Euge_Track")
SEEDS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else tuple(range(6))
# LPVMPC_SWEEP_VARIANT=9: the N = 20 batches only, on the four-wavefront latency form of their kernels (lpvmpc.h, kernel_variant)
VARIANT = int(os.environ.get("LPVMPC_SWEEP_VARIANT", "0"))
CASES = (("controller", 20, 1), ("controller", 20, 0), ("controller", 8, 1), ("planner", 20, 1), ("planner", 30, 1), ("planner", 40, 1))
if VARIANT == 9: CASES = (CASES[0], CASES[1], CASES[3])        # (the N = 20 kernels have the form: controller, both laps, and planner)
for shape in SHAPES:
    for seed in SEEDS:
        for kind, N, lap in CASES:
            B = 1024 if kind == "controller" else 512
            w = workloads.controller_batch(B, N=N, seed=100 + seed, shape=shape) if kind == "controller" else workloads.planner_batch(B, N=N, seed=200 + seed, shape=shape)
            w["lap"] = lap
            eng = workloads.make_solver(w)
            if VARIANT: eng.set_option("kernel_variant", VARIANT)
            if kind == "controller":
                a = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], lap)
                ref = O.ctrl_tick_batch(w, nthreads=16)
            else:
                a = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
                ref = O.plan_tick_batch(w, nthreads=16)
            eng.close()
            # no answer from the oracle: KKT breakdown (-10), or a rolled-out abscissa outside the track table, where the
            # reference's Curvature() raises (the C oracle marks it with NaN, the device reports LPVMPC_UNSOLVED with NaN outputs)
            sane = (ref["status"] != -10) & ~(np.isnan(ref["uPred"]).any(axis=(1, 2)) & (ref["status"] == 1))
            fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & np.isfinite(a["uPred"]).all(axis=(1, 2)) & sane
            d = np.abs(a["uPred"][fin] - ref["uPred"][fin]).max(axis=(1, 2)) if fin.any() else np.zeros(0)
            st = int(np.sum(a["status"][sane] == ref["status"][sane])); it = int(np.sum(a["iters"][sane] == ref["iters"][sane]))
            tot["n"] += int(sane.sum()); tot["st"] += st; tot["it"] += it; tot["close"] += int(np.sum(d <= 1e-6)); tot["fin"] += int(fin.sum())
            worst = max(worst, float(d.max()) if d.size else 0.0)
            # round 6: EVERY instance whose status or iteration count differs from the oracle's, by name
            for j in np.nonzero(sane & ((a["status"] != ref["status"]) | (a["iters"] != ref["iters"])))[0]:
                both = bool(np.isfinite(a["uPred"][j]).all()), bool(np.isfinite(ref["uPred"][j]).all())
                dj = float(np.abs(a["uPred"][j] - ref["uPred"][j]).max()) if all(both) else float("nan")
                rule = "status flip at the cap" if (a["iters"][j] == ref["iters"][j] and {int(a["status"][j]), int(ref["status"][j])} == {2, -2}) else \
                       ("class D (certificate one check apart; the oracle's RCM elimination order stops where the device does)" if T.class_d(w, kind, int(j), a, ref) else "UNEXPLAINED")
                print("MISMATCH %s seed %d %s N=%d lap=%d #%d: status %d/%d iters %d/%d polish %d finite %s/%s |du| %.2e  resid(dev) %s -> %s"
                      % (shape, seed, kind, N, lap, j, a["status"][j], ref["status"][j], a["iters"][j], ref["iters"][j],
                         a["polish"][j], both[0], both[1], dj,
                         np.array2string(np.asarray(a["resid"][j]), precision=3) if "resid" in a else "-", rule), flush=True)
            idx = np.nonzero(fin)[0]
            for j in idx[d > 1e-6]:
                r = T.outlier_report(w, kind, int(j), a, ref)
                key = (r["status"], r["polish"], r["class"])
                cls[key] = cls.get(key, 0) + 1
                print("OUTLIER %s seed %d %s N=%d lap=%d #%d: status %d/%d polish %d iters %d/%d |du| %.2e  obj gap %.2e  pri %.2e (tol %.2e)  -> %s"
                      % (shape, seed, kind, N, lap, j, r["status"], r["status_ref"], r["polish"], r["iters"], r["iters_ref"], r["du"], r["obj_gap"], r["pri"], r["pri_tol"], r["class"]), flush=True)
            if st != sane.sum() or it < 0.995 * sane.sum() or (d.size and np.mean(d <= 1e-6) < 0.98):
                print("ATTENTION %s seed %d %s N=%d lap=%d: status %d/%d iters %d/%d close %d/%d max %.2e" % (shape, seed, kind, N, lap, st, sane.sum(), it, sane.sum(), np.sum(d <= 1e-6), fin.sum(), d.max() if d.size else 0), flush=True)
    print(shape, "done", tot, "worst du %.2e" % worst, "outlier classes (status, polish, class): %s" % cls, flush=True)
