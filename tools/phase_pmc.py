#!/usr/bin/env python3
"""Workload of tools/phase_pmc.sh: 1024 controller instances (seed 0) that never terminate (eps = 1e-30, no adaptive rho, no
polish), run for a fixed number of ADMM iterations -- once by the two-wavefront solve kernel alone (plain launch, max_iter 300)
and once parked at iteration 25 and finished by the whole-CU tail kernel (max_iter 150).  Run under rocprofv3 --pmc with a
library built with -DLPVMPC_PHASE_ONLY=n (python3 tools/phase_pmc.py <library file name>)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lpvmpc import _ffi
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
from lpvmpc import workloads

dev = torch.device("cuda", 0)
B = 1024
w = workloads.controller_batch(B, N=20, seed=0)
t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = [t(w[k]) for k in ("x0", "u_prev", "vel_ref", "curv_s", "u_old")]
o = dict(xPred=torch.zeros((B, 21, 6), dtype=torch.float64, device=dev), uPred=torch.zeros((B, 20, 2), dtype=torch.float64, device=dev),
         status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
kw = dict(adaptive_rho=0, polish=0, check_termination=25, eps_abs=1e-30, eps_rel=1e-30, eps_prim_inf=1e-30, eps_dual_inf=1e-30)
for name, mi, defer in (("main", 300, 0), ("tail", 150, 25)):
    eng = workloads.make_solver(w, max_iter=mi, **kw)
    eng.reserve(B)
    if defer:
        eng.set_option("defer_pool", B); eng.set_option("defer_after", defer); eng.set_option("defer_budget", -1)
    eng.solve_dev(B, ins[0], ins[1], ins[2], ins[3], ins[4], None, o["xPred"], o["uPred"], o["status"], o["iters"], None, None,
                  cf_new=w["cf_new"], lap=w["lap"], stream=0)
    eng.join(0); torch.cuda.synchronize()
    it = o["iters"].cpu().numpy()
    print(name, "iterations", np.unique(it), flush=True)
    eng.close()
