import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|MHz' | head -8")
from lpvmpc import workloads
from oracle import osqp_ref as O
w = workloads.controller_batch(1024, 20, seed=0)
O.ctrl_tick_batch({k: (v[:64] if isinstance(v, np.ndarray) and v.ndim and v.shape[0]==1024 else v) for k,v in w.items()}, nthreads=8)
for nt in (1, 8, 16, 32, 64, 128, 256):
    t=time.time(); O.ctrl_tick_batch(w, nthreads=nt); t=time.time()-t
    print("threads %3d: %.1f ms -> %.0f solves/s" % (nt, t*1e3, 1024/t), flush=True)
