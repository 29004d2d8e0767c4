#!/usr/bin/env python3
"""Per-phase table from the passes of tools/phase_pmc.sh: counters of the solve-kernel dispatches, less the `phase0` build (loop
with no phase: set-up, checks), per iteration and instance.  Usage: phase_pmc_summary.py <outdir> [out.txt]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
ITER = {"main": 300, "tail": 125}          # iterations the dispatch runs (tail: 150 - 25 done before parking)
B = 1024
NAMES = {"phase1": "right-hand side", "phase2": "KKT solve (sweeps / dense product)", "phase3": "update", "full": "whole iteration (product build)"}


def load(v):
    """{(class, counter): [value per dispatch, in dispatch order]}; class main = the 128-thread kernel over the whole batch (the first
    such dispatch is the plain launch, the second only runs up to the parking at iteration 25), tail = the 512-thread kernel."""
    out = {}
    for f in glob.glob(os.path.join(src, v, "**", "*counter_collection.csv"), recursive=True):
        recs = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in recs:
            if "admm_solve_kernel" not in r["Kernel_Name"]:
                continue
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)))
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
            cls = "tail" if wg == 512 else ("main" if grid >= B * 128 else None)
            if cls:
                out.setdefault((cls, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return out


rows = {}
for v in ("phase0", "phase1", "phase2", "phase3", "full"):
    d = load(v)
    for (cls, cnt), vals in d.items():
        # plain launch = the first dispatch of the 128-thread kernel; tail = the only 512-thread dispatch
        rows.setdefault(cls, {}).setdefault(v, {})[cnt] = vals[0]
lines = []
for cls in ("main", "tail"):
    if cls not in rows or "phase0" not in rows[cls]:
        continue
    base = rows[cls]["phase0"]
    lines.append("%s kernel (%s), per ADMM iteration and instance, counters of the `phase0` build subtracted:" % (
        "two-wavefront solve" if cls == "main" else "whole-CU tail", "admm_solve_kernel<6, 20, 2, MFMA sweeps>" if cls == "main" else "admm_solve_kernel<6, 20, 8, tail>"))
    lines.append("  %-36s %10s %12s %14s %10s %12s %12s" % ("phase", "LDS instr", "LDS active", "bank conflict", "conf/act", "VALU instr", "wave cycles"))
    for v in ("phase1", "phase2", "phase3", "full"):
        if v not in rows[cls]:
            continue
        c = rows[cls][v]
        n = ITER[cls] * B
        g = lambda k: (c.get(k, 0.0) - base.get(k, 0.0)) / n
        act, conf = 4 * g("SQ_ACTIVE_INST_LDS"), 4 * g("SQ_LDS_BANK_CONFLICT")
        lines.append("  %-36s %10.1f %12.0f %14.0f %10.2f %12.1f %12.0f" % (NAMES[v], g("SQ_INSTS_LDS"), act, conf, conf / act if act else 0.0, g("SQ_INSTS_VALU"), 4 * g("SQ_WAVE_CYCLES")))
    lines.append("  (LDS active / bank conflict / wave cycles in cycles = 4 x the quad-cycle counters, summed over the wavefronts of an instance)")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
