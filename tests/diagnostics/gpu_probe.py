#!/usr/bin/env python3
# (test infrastructure: uses the oracle as a checker; lives under tests/ for that reason)
"""Diagnostic dump for a GPU box: per-case status / iterations / error vs the oracle, and a first timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lpvmpc
from lpvmpc import workloads
from oracle import lpv_ref as L
from tests._golden import cases

def vfull(c):
    N = int(c["N"]); v = np.asarray(c["vel_ref"], float).reshape(-1)
    return np.concatenate([v[:N], v[-1:]])

tab = lpvmpc.Map("oval", 0.2).PointAndTangent
for name in ["ctrl_n10_cfg1", "ctrl_n20_oval"]:
    for i, c in enumerate(cases(name)):
        N = int(c["N"])
        eng = lpvmpc.BatchedSolver("controller", N, float(c["dt"]), c["Q"], c["R"], c["dR"], track=tab)
        S, A, B = eng.lpv(c["x0"][None], c["u_prev"][None], vfull(c)[None], c["curv_ref"][None], cf_new=float(c["cf_new"]), lap=int(c["lap"]))
        eA = np.abs(A[0]-c["A"]).max(); eS = np.abs(S[0]-c["states"]).max()
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], vfull(c)[None], c["old_u"][None])
        xP, uP, _ = L.unpack_solution(c["x_orc"], 6, 2, N)
        print("%s[%02d] lpv errA %.1e errS %.1e | status %d/%d iters %d/%d polish %d/%d errx %.2e erru %.2e resid %s" % (
            name, i, eA, eS, out["status"][0], c["status_orc"], out["iters"][0], c["iter_orc"], out["polish"][0], c["polish_orc"],
            np.abs(out["xPred"][0]-xP).max(), np.abs(out["uPred"][0]-uP).max(), np.array2string(out["resid"][0], precision=3)), flush=True)
        eng.close()
tab = lpvmpc.Map("L_shape", 0.2).PointAndTangent
for name in ["plan_n30_lshape", "plan_n40_lshape"]:
    for i, c in enumerate(cases(name)):
        N = int(c["N"])
        eng = lpvmpc.BatchedSolver("planner", N, float(c["dt"]), c["Q"], c["R"], c["dR"], L_cf=c["L_cf"], track=tab)
        S, A, B = eng.lpv(c["x0"][None], c["u_prev"][None], None, c["SS"][None])
        eA = np.abs(A[0]-c["A"]).max()/max(1,np.abs(c["A"]).max()); eS = np.abs(S[0]-c["states"]).max()/max(1,np.abs(c["states"]).max())
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], None, np.zeros((1, 2)), max_ey=float(c["max_ey"]))
        if np.all(np.isfinite(c["x_orc"])):
            xP, uP, _ = L.unpack_solution(c["x_orc"], 5, 2, N)
            ex = np.abs(out["xPred"][0]-xP).max(); eu = np.abs(out["uPred"][0]-uP).max()
        else:
            ex = eu = float("nan")
        print("%s[%02d] lpv relA %.1e relS %.1e | status %d/%d iters %d/%d polish %d/%d errx %.2e erru %.2e resid %s" % (
            name, i, eA, eS, out["status"][0], c["status_orc"], out["iters"][0], c["iter_orc"], out["polish"][0], c["polish_orc"],
            ex, eu, np.array2string(out["resid"][0], precision=3)), flush=True)
        eng.close()

for B in (64, 1024, 4096):
    w = workloads.controller_batch(B, N=20, seed=0)
    eng = workloads.make_solver(w); eng.set_timing(True)
    for rep in range(3):
        t = time.time()
        out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
        t = time.time() - t
    print("ctrl B=%d: host wall %.2f ms, solve kernel %.3f ms, iters mean %.1f max %d, status %s" % (
        B, t*1e3, eng.last_kernel_ms(), out["iters"].mean(), out["iters"].max(), dict(zip(*np.unique(out["status"], return_counts=True)))), flush=True)
    eng.close()
for B in (64, 1024):
    w = workloads.planner_batch(B, N=30, seed=1)
    eng = workloads.make_solver(w); eng.set_timing(True)
    for rep in range(2):
        t = time.time()
        out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        t = time.time() - t
    print("plan B=%d: host wall %.2f ms, solve kernel %.3f ms, iters mean %.1f max %d, status %s" % (
        B, t*1e3, eng.last_kernel_ms(), out["iters"].mean(), out["iters"].max(), dict(zip(*np.unique(out["status"], return_counts=True)))), flush=True)
    eng.close()
