import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from lpvmpc import workloads
w = workloads.controller_batch(8, N=20, seed=0)
eng = workloads.make_solver(w)
o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
print("lpvmpc first: ok", o["status"][:3])
import torch
try:
    torch.cuda.init(); print("torch after lpvmpc: ok", torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); print(x.sum().item())
except Exception as e:
    print("torch after lpvmpc FAILED:", e)
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::8])
