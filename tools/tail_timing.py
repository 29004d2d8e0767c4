#!/usr/bin/env python3
"""Lone-instance timing of the whole-CU tail kernel (GPU): microseconds per ADMM iteration and per termination check, by
differencing runs with different max_iter / check_termination (termination switched off by tiny tolerances)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lpvmpc import workloads
dev = torch.device("cuda", 0)
EPS_INF = float(os.environ.get("EPS_INF", "1e-30"))     # 1e-30: every infeasibility stage of a check runs (the longest path); 1e30: the shortest
w = workloads.controller_batch(8, N=20, seed=0)
t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a[:1])).to(dev)
ins = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), vel=t(w["vel_ref"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]))
o = dict(xPred=torch.empty((1, 21, 6), dtype=torch.float64, device=dev), uPred=torch.empty((1, 20, 2), dtype=torch.float64, device=dev),
         status=torch.empty(1, dtype=torch.int32, device=dev), iters=torch.empty(1, dtype=torch.int32, device=dev),
         resid=torch.empty((1, 4), dtype=torch.float64, device=dev), polish=torch.empty(1, dtype=torch.int32, device=dev))


def run(mi, chk, tail, **kw):
    e = workloads.make_solver(w, max_iter=mi, adaptive_rho=0, polish=0, check_termination=chk, eps_abs=1e-30, eps_rel=1e-30, eps_prim_inf=EPS_INF, eps_dual_inf=EPS_INF, **kw)
    e.reserve(1); e.set_option("defer_after", 100); e.set_option("defer_budget", 0); e.set_option("defer_tail", tail)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.solve_dev(1, ins["x0"], ins["u_prev"], ins["vel"], ins["curv"], ins["u_old"], None, o["xPred"], o["uPred"], o["status"], o["iters"], o["resid"], o["polish"],
                    cf_new=w["cf_new"], lap=w["lap"], stream=0)
        e.join(0); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    e.close()
    return best * 1e6


for tail in (1, 0):
    r = {}
    for chk in (25, 100):
        r[chk] = (run(4100, chk, tail) - run(2100, chk, tail)) / 2000
    per_check = (r[25] - r[100]) * 100 / 3          # 4 checks per 100 iterations against 1
    print("tail=%d: %.3f us per iteration at check_termination 25, %.3f at 100 -> %.2f us per check, %.3f us per bare iteration"
          % (tail, r[25], r[100], per_check, r[100] - per_check / 100))
