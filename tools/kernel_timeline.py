#!/usr/bin/env python3
"""Timeline of the solve-kernel launches of the last K steps of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
tools/kernel_timeline.py <kernel_trace.csv> [K].  Classes: main (grid 1024 x 128), resume (pool-sized grid of the same kernel),
tail (512-thread kernel over the pool)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def cls(r):
    n = r["Kernel_Name"]
    if "admm_solve" in n:
        g, w = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
        if w == 512:
            return "tail"
        return "main" if g >= 1024 * 128 else ("resume" if g > 128 else "one")
    return "lpv" if "lpv" in n else "other"


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r), r["Queue_Id"]) for r in rows)
mains = [e for e in ev if e[2] == "main"]
t0 = mains[-K][0]
print("  start     end     dur  class  queue   (ms, relative to the first of the last %d main launches)" % K)
busy = {}
for s, e, c, q in ev:
    if s >= t0 - 1000 and c not in ("lpv", "other"):
        print("%7.3f %7.3f %7.3f  %-6s q%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, c, q))
        busy[c] = busy.get(c, 0.0) + (e - s) / 1e6
print("sum of durations by class (ms):", {k: round(v, 3) for k, v in busy.items()})
