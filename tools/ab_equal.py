#!/usr/bin/env python3
"""Bitwise A/B of two builds of the library: tools/ab_equal.py <a.so> <b.so>  (file names inside the package directory).

Every (kind, N) that has a kernel of its own is solved once by each build (a child process per build: the library is loaded once
per process) on the same seeded batches -- plain launches and the deferral with the tail kernel -- and the outputs are compared
word for word.  Prints one line per case; exit code 1 on any difference.  -0.0 against +0.0 counts as equal and is reported."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

CASES = [("controller", 20, 2048, {}), ("controller", 20, 1024, {"defer": 100}), ("controller", 10, 512, {}), ("controller", 8, 512, {}),
         ("controller", 13, 256, {}), ("planner", 20, 1024, {}), ("planner", 20, 512, {"defer": 100}), ("planner", 30, 512, {}),
         ("planner", 40, 512, {}), ("planner", 30, 256, {"variant": 4}), ("planner", 40, 256, {"variant": 6})]

def child(lib, out):
    from lpvmpc import _ffi
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), lib)
    from lpvmpc import workloads
    res = {}
    for i, (kind, N, B, opt) in enumerate(CASES):
        w = (workloads.controller_batch if kind == "controller" else workloads.planner_batch)(B, N=N, seed=3 + i)
        eng = workloads.make_solver(w)
        if "variant" in opt: eng.set_option("kernel_variant", opt["variant"])
        if "defer" in opt:      # (a pool entry for every instance: which instances a smaller pool has no room for -- they stay in the main launch -- depends on timing)
            eng.set_option("defer_pool", 2 * B); eng.set_option("defer_after", opt["defer"])      # (2 B: admission is by age class -- the young class may take three quarters of a pool)
        o = eng.solve(w["x0"], w["u_prev"], w.get("vel_ref") if kind == "controller" else None, w["curv_s"], w["u_old"],
                      None if kind == "controller" else w["max_ey"], *( (w["cf_new"], w["lap"]) if kind == "controller" else ()))
        for k in ("xPred", "uPred", "status", "iters", "polish", "resid"):
            res["%d_%s" % (i, k)] = np.array(o[k])
        eng.close()
    np.savez(out, **res)

def main():
    a, b = sys.argv[1:3]
    with tempfile.TemporaryDirectory() as td:
        outs = []
        for lib in (a, b):
            out = os.path.join(td, "%d_%s.npz" % (len(outs), lib))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, out], check=True)
            outs.append(np.load(out))
        bad = 0
        for i, (kind, N, B, opt) in enumerate(CASES):
            diffs, zeros = [], 0
            for k in ("xPred", "uPred", "status", "iters", "polish", "resid"):
                x, y = outs[0]["%d_%s" % (i, k)], outs[1]["%d_%s" % (i, k)]
                if x.tobytes() != y.tobytes():
                    if np.array_equal(x, y, equal_nan=True): zeros += 1
                    else: diffs.append("%s (%d words, max |d| %.3g)" % (k, int(np.sum(~((x == y) | (np.isnan(x) & np.isnan(y))))), float(np.nanmax(np.abs(x.astype(float) - y.astype(float))))))
            print("%-10s N=%-2d B=%-4d %-16s %s%s" % (kind, N, B, opt or "", "identical" if not diffs else "DIFFERENT: " + ", ".join(diffs),
                                                     " (signed zeros differ in %d arrays)" % zeros if zeros else ""))
            bad += bool(diffs)
        sys.exit(1 if bad else 0)

if __name__ == "__main__":
    if sys.argv[1] == "--child": child(sys.argv[2], sys.argv[3])
    else: main()
