"""Diagnostic (GPU): the whole-CU tail kernel against the plain launch, batch by batch -- status / iteration / polish equality,
largest difference of the solutions, and the latency of one synchronous batch with the tail kernel on and off.
Usage: python tests/diagnostics/tail_probe.py [seed ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from lpvmpc import workloads
    seeds = [int(a) for a in sys.argv[1:]] or [0, 3, 16, 19]
    dev = torch.device("cuda", 0)
    B = 1024
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def outs():
        return dict(xPred=torch.full((B, 21, 6), -7.0, dtype=torch.float64, device=dev), uPred=torch.full((B, 20, 2), -7.0, dtype=torch.float64, device=dev),
                    status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev),
                    resid=torch.zeros((B, 4), dtype=torch.float64, device=dev), polish=torch.zeros(B, dtype=torch.int32, device=dev))

    def call(eng, ins, o, w):
        eng.solve_dev(B, ins["x0"], ins["u_prev"], ins["vel"], ins["curv"], ins["u_old"], None, o["xPred"], o["uPred"], o["status"], o["iters"],
                      o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"], stream=0)

    for seed in seeds:
        w = workloads.controller_batch(B, N=20, seed=seed)
        ins = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), vel=t(w["vel_ref"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]))
        plain = workloads.make_solver(w); plain.reserve(B)
        o = outs(); call(plain, ins, o, w); torch.cuda.synchronize()
        ref = {k: v.cpu().numpy() for k, v in o.items()}
        lat = {}
        res = {}
        for tail in (0, 1):
            eng = workloads.make_solver(w); eng.reserve(B)
            eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100); eng.set_option("defer_tail", tail)
            o = outs()
            call(eng, ins, o, w); eng.join(0); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); call(eng, ins, o, w); eng.join(0); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            lat[tail] = min(ts)
            res[tail] = {k: v.cpu().numpy() for k, v in o.items()}
            eng.close()
        t0 = time.perf_counter(); o = outs(); call(plain, ins, o, w); torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) * 1e3
        plain.close()
        g = res[1]
        ne_it = np.nonzero(g["iters"] != ref["iters"])[0]
        ne_st = np.nonzero(g["status"] != ref["status"])[0]
        ne_po = np.nonzero(g["polish"] != ref["polish"])[0]
        ok = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & np.isfinite(g["uPred"]).all(axis=(1, 2))
        du = np.abs(g["uPred"] - ref["uPred"]).max(axis=(1, 2)); dx = np.abs(g["xPred"] - ref["xPred"]).max(axis=(1, 2))
        pol = ref["polish"] == 1
        print("seed %d: max iters %d, parked %d | tail: iters differ %s status differ %s polish differ %s | max|du| polished %.2e unpolished %.2e max|dx| %.2e | "
              "bitwise-equal instances %d/%d | latency ms plain %.2f deferred(no tail) %.2f deferred(tail) %.2f | no-tail path bit-identical %s"
              % (seed, ref["iters"].max(), int((ref["iters"] > 100).sum()), ne_it[:6].tolist(), ne_st[:6].tolist(), ne_po[:6].tolist(),
                 du[ok & pol].max(initial=0.0), du[ok & ~pol].max(initial=0.0), dx[ok].max(initial=0.0),
                 int(np.sum((g["uPred"] == ref["uPred"]).all(axis=(1, 2)))), B, t_plain, lat[0], lat[1],
                 all(np.array_equal(res[0][k], ref[k], equal_nan=True) for k in ref)), flush=True)
        if len(ne_it):
            print("   iters tail/ref:", g["iters"][ne_it][:6], ref["iters"][ne_it][:6])


if __name__ == "__main__":
    main()
