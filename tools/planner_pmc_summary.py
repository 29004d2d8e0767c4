#!/usr/bin/env python3
"""Per iteration-instance counters of the planner kernels from a tools/planner_pmc.py pass: tools/planner_pmc_summary.py <dir> <B>"""
import csv, glob, os, sys
d, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 512
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
rows = {}
for r in csv.DictReader(open(f)):
    if "admm_solve_kernel" not in r["Kernel_Name"]: continue
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(rows)
for a, b in zip(keys[0::2], keys[1::2]):      # (100, 400) iterations of one variant
    ca, cb = rows[a], rows[b]
    per = {k: (cb[k] - ca[k]) / (300.0 * B) for k in cb if k in ca}
    print(b[1], " per iteration-instance: " + ", ".join("%s %.0f" % (k.replace("SQ_", ""), v) for k, v in sorted(per.items())))
