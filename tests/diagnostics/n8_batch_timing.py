import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, lpvmpc
from lpvmpc import workloads
w = workloads.controller_batch(8192, N=8, seed=26)
eng = workloads.make_solver(w)
o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
t0 = time.perf_counter()
for _ in range(5): o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
print("controller N=8, 8192 instances: %.2f ms per host call, mean iters %.1f" % ((time.perf_counter() - t0) / 5 * 1e3, o["iters"].mean()))
