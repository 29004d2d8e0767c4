#!/usr/bin/env python3
"""Time of one ADMM iteration of a lone instance in the three kernels that can continue it after iteration 100: the two-wavefront
solve kernel, the whole-CU tail kernel behind lpvmpc_join, and the ring-drain form of the tail kernel on a lane's reserved CU.
Differences of two runs with termination switched off (max_iter 1100 / 2100)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lpvmpc
from lpvmpc import workloads

dev = torch.device("cuda", 0)
w = workloads.controller_batch(1024, N=20, seed=0)
t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a[:1])).to(dev)
ins = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), vel=t(w["vel_ref"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]))
o = dict(xPred=torch.empty((1, 21, 6), dtype=torch.float64, device=dev), uPred=torch.empty((1, 20, 2), dtype=torch.float64, device=dev),
         status=torch.empty(1, dtype=torch.int32, device=dev), iters=torch.empty(1, dtype=torch.int32, device=dev),
         resid=torch.empty((1, 4), dtype=torch.float64, device=dev), polish=torch.empty(1, dtype=torch.int32, device=dev))
lane = lpvmpc.Lane(device=0, reserved_cus=8, step_streams=1, ring_entries=16)
for name, defer, use_lane in (("solve kernel", 0, False), ("tail kernel (join)", 100, False), ("tail kernel (lane drain)", 100, True)):
    ts = {}
    for mi in (1100, 2100):
        e = workloads.make_solver(w, max_iter=mi, adaptive_rho=0, polish=0, eps_abs=1e-30, eps_rel=1e-30, eps_prim_inf=1e-30, eps_dual_inf=1e-30)
        e.reserve(1)
        e.set_option("defer_after", defer); e.set_option("defer_budget", 0 if not use_lane else -1)
        st = 0
        if use_lane:
            e.attach_lane(lane, promote_after=100); st = lane.step_streams[0]
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.solve_dev(1, ins["x0"], ins["u_prev"], ins["vel"], ins["curv"], ins["u_old"], None, o["xPred"], o["uPred"], o["status"], o["iters"],
                        o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"], stream=st)
            if defer:
                e.join(st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert int(o["iters"].cpu()[0]) == mi, (name, int(o["iters"].cpu()[0]))
        ts[mi] = best
        e.close()
    print("%-28s %.3f us per iteration (1100: %.3f ms, 2100: %.3f ms)" % (name, (ts[2100] - ts[1100]) / 1000 * 1e6, ts[1100] * 1e3, ts[2100] * 1e3), flush=True)
lane.close()
