#!/usr/bin/env python3
# (test infrastructure: uses the oracle as a checker; lives under tests/ for that reason)
"""Parity sweep of the whole-CU tail kernel on a GPU box: 4 tracks x 6 seeds x lap 1 / lap 0 x 4096 controller instances = 196 608.
A deferred call (parked at 100 iterations, bounded passes of 100, closing pass = the tail kernel) against the plain launch of the
same batch: status, iteration count, polish flag of every instance; never-parked instances bit for bit, parked ones to round-off;
the instances beyond 1000 iterations against the CPU oracle.  tests/test_gpu_deferral.py::test_tail_parity_sweep_slice is a slice
of this inside the suite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from lpvmpc import workloads
from oracle import osqp_ref as O

tot = dict(n=0, parked=0, st=0, it=0, pol=0, long=0, long_st=0, long_it=0)
worst_pol = worst_it = 0.0
t0 = time.time()
for shape in ("oval", "L_shape", "3110", "Euge_Track"):
    for seed in range(300, 306):
        for lap in (1, 0):
            B = 4096
            w = workloads.controller_batch(B, N=20, seed=seed, shape=shape)
            w["lap"] = lap
            plain = workloads.make_solver(w)
            ref = plain.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], lap); plain.close()
            eng = workloads.make_solver(w)
            eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
            got = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], lap); eng.close()
            parked = ref["iters"] > 100
            tot["n"] += B; tot["parked"] += int(parked.sum())
            tot["st"] += int(np.sum(got["status"] != ref["status"])); tot["it"] += int(np.sum(got["iters"] != ref["iters"])); tot["pol"] += int(np.sum(got["polish"] != ref["polish"]))
            fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & np.isfinite(got["uPred"]).all(axis=(1, 2))
            if not np.array_equal(got["uPred"][~parked], ref["uPred"][~parked], equal_nan=True):
                print("ATTENTION %s seed %d lap %d: a never-parked instance differs" % (shape, seed, lap), flush=True)
            d = np.abs(got["uPred"] - ref["uPred"]).max(axis=(1, 2))
            pol = ref["polish"] == 1
            if (fin & pol & parked).any():
                worst_pol = max(worst_pol, float(d[fin & pol & parked].max()))
            if (fin & ~pol & parked).any():
                worst_it = max(worst_it, float(d[fin & ~pol & parked].max()))
            bad = np.nonzero((got["status"] != ref["status"]) | (got["iters"] != ref["iters"]) | (got["polish"] != ref["polish"]))[0]
            for j in bad[:8]:
                print("DIFF %s seed %d lap %d #%d: status %d/%d iters %d/%d polish %d/%d" % (shape, seed, lap, j, got["status"][j], ref["status"][j], got["iters"][j], ref["iters"][j], got["polish"][j], ref["polish"][j]), flush=True)
            long_ = np.nonzero(ref["iters"] > 1000)[0]
            if len(long_):
                sub = {k: (v[long_] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k != "track" else v) for k, v in w.items()}
                orc = O.ctrl_tick_batch(sub, nthreads=16)
                sane = orc["status"] != -10
                tot["long"] += int(sane.sum())
                tot["long_st"] += int(np.sum(got["status"][long_][sane] != orc["status"][sane])); tot["long_it"] += int(np.sum(got["iters"][long_][sane] != orc["iters"][sane]))
    print("%s done after %d s: %s max |du| parked+polished %.2e, parked+unpolished %.2e" % (shape, time.time() - t0, tot, worst_pol, worst_it), flush=True)
print("RESULT: %d instances, %d finished by the tail kernel; differences from the plain launch: status %d, iterations %d, polish flag %d; %d instances beyond 1000 iterations against the oracle: status %d, iterations %d differ"
      % (tot["n"], tot["parked"], tot["st"], tot["it"], tot["pol"], tot["long"], tot["long_st"], tot["long_it"]))
