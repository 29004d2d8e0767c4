import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import lpvmpc
from lpvmpc import workloads
dev = torch.device("cuda", 0)
B, N, S, NBAT = 1024, 20, 4, 32
ws = [workloads.controller_batch(B, N=N, seed=i) for i in range(NBAT)]
w = ws[0]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = [dict(x0=t(wi["x0"]), u_prev=t(wi["u_prev"]), curv=t(wi["curv_s"]), u_old=t(wi["u_old"]), vel_ref=t(wi["vel_ref"])) for wi in ws]
def new_outs(b):
    return dict(xPred=torch.empty(b, N + 1, 6, dtype=torch.float64, device=dev), uPred=torch.empty(b, N, 2, dtype=torch.float64, device=dev),
                status=torch.empty(b, dtype=torch.int32, device=dev), iters=torch.empty(b, dtype=torch.int32, device=dev),
                resid=torch.empty(b, 4, dtype=torch.float64, device=dev), polish=torch.empty(b, dtype=torch.int32, device=dev))
outs = [new_outs(B) for _ in range(NBAT)]
for timing in (False, True):
    engines = [workloads.make_solver(w, device=0) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for e in engines:
        e.set_option("defer_after", 100); e.set_option("defer_budget", 100); e.set_timing(timing)
    def step(j):
        e = j % S; o, d = outs[j % NBAT], ins[j % NBAT]
        engines[e].solve_dev(B, d["x0"], d["u_prev"], d["vel_ref"], d["curv"], d["u_old"], None, o["xPred"], o["uPred"], o["status"], o["iters"], o["resid"], o["polish"],
                             cf_new=w["cf_new"], lap=w["lap"], stream=streams[e].cuda_stream)
    def sync():
        for i, e in enumerate(engines): e.join(streams[i].cuda_stream)
        torch.cuda.synchronize()
    for j in range(20): step(j)
    sync()
    for K in (400, 1200):
        t0 = time.perf_counter()
        for j in range(K): step(j)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        print("timing=%s K=%d: enqueue loop %.1f ms (%.3f ms/step), total %.1f ms (%.3f ms/step) -> %.3f M solves/s" % (timing, K, (t1 - t0) * 1e3, (t1 - t0) / K * 1e3, (t2 - t0) * 1e3, (t2 - t0) / K * 1e3, B * K / (t2 - t0) / 1e6), flush=True)
    for e in engines: e.close()
