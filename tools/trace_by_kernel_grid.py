#!/usr/bin/env python3
"""A rocprofv3 kernel trace summed by (kernel, grid size, workgroup size): tools/trace_by_kernel_grid.py <t_kernel_trace.csv> <out.csv>
(the planner and cascade legs launch one kernel name at several grid sizes: the stats file of rocprofv3 averages over all of them)."""
import csv, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
    rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Kernel", "Grid_Size", "Workgroup_Size", "Calls", "AverageNs", "TotalDurationNs", "MinNs", "MaxNs"])
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([key[0], key[1], key[2], len(d), round(sum(d) / len(d), 1), sum(d), min(d), max(d)])
