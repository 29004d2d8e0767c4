#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/collect_pmc.sh into profiles/<round>_pmc.json (what bench.py reports as
roofline.traffic / roofline.valu, with the file named as the source) and copy the kernel-trace summaries next to it.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in separate passes, KB units,
FETCH_SIZE doubled on gfx950 for coalesced streaming reads, WRITE_SIZE as reported.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES /
SQ_WAIT_* count quad-cycles (x4 = cycles), summed over all waves of the dispatch.
Usage: pmc_summary.py <collect_pmc outdir> <profiles/rNN prefix>"""
import csv
import glob
import json
import os
import shutil
import sys

N_SIMD = 256 * 4


def rows(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


def solve_rows(d, pattern):
    fs = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    out = []
    for f in fs:
        out += [r for r in rows(f) if "admm_solve_kernel" in r.get("Kernel_Name", "")]
    return out


def per_launch(rs, counter, grid_min):
    """Mean counter value over the full-batch launches (Grid_Size >= grid_min: leaves out the one-instance set-up launch)."""
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and int(r["Grid_Size"]) >= grid_min]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    B = 1024
    grid_min = B * 128
    out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- " + open(os.path.join(src, "command.txt")).read().strip()
                      + " (one pass per counter group: tools/collect_pmc.sh)",
           "kernel": "admm_solve_kernel<6, 20, 2, MFMA sweeps>", "batch": B, "seed": 0}
    f, nf = per_launch(solve_rows(os.path.join(src, "fetch"), "*counter_collection.csv"), "FETCH_SIZE", grid_min)
    w, nw = per_launch(solve_rows(os.path.join(src, "write"), "*counter_collection.csv"), "WRITE_SIZE", grid_min)
    if f is not None and w is not None:
        out.update({"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": nf,
                    "correction": "gfx950: FETCH_SIZE under-reports coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
                    "traffic_bytes_per_launch": 2.0 * f * 1024 + w * 1024})
    sq = {}
    for d in ("sq1", "sq2"):
        rs = solve_rows(os.path.join(src, d), "*counter_collection.csv")
        for name in sorted({r["Counter_Name"] for r in rs}):
            v, n = per_launch(rs, name, grid_min)
            if v is not None:
                sq[name] = v
    # duration of the profiled launches from the same CSVs (timestamps in ns)
    rs = solve_rows(os.path.join(src, "sq1"), "*counter_collection.csv")
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if r["Counter_Name"] == "SQ_WAVES" and int(r["Grid_Size"]) >= grid_min]
    out["sq_per_launch"] = sq
    if dur:
        out["kernel_ns_under_pmc"] = sum(dur) / len(dur)
    if "SQ_INSTS_VALU" in sq and "SQ_ACTIVE_INST_VALU" in sq and dur:
        # issue cycles = cycles in which some wave has a VALU instruction active, summed over the waves (quad-cycles x 4);
        # available = SIMDs x shader cycles of the launch (SQ_BUSY_CYCLES is summed over the 32 SEs x ...: use the clock instead)
        clk = sq.get("GRBM_GUI_ACTIVE")
        cycles = (clk / 8.0) if clk else (out["kernel_ns_under_pmc"] * 2.1)            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        issue = 4.0 * sq["SQ_ACTIVE_INST_VALU"]
        out["valu"] = {"wave_instr_per_launch": sq["SQ_INSTS_VALU"], "issue_cycles": issue,
                       "frac": issue / (N_SIMD * cycles), "launch_cycles": cycles,
                       "definition": "issue_cycles = 4 x SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves); frac = issue_cycles / (1024 SIMDs x launch cycles): "
                                     "launch-average VALU issue utilisation of ONE isolated 1024-instance launch (the launch lasts as long as its slowest instance)"}
        if "SQ_WAVE_CYCLES" in sq:
            out["valu"]["active_share_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_ANY" in sq:
                out["valu"]["wait_share_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
    for tag, dst in (("trace", "_streams1_kernel_stats.csv"), ("trace32", "_bench_kernel_stats.csv")):
        fs = glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True)
        if fs:
            shutil.copy(fs[0], prefix + dst)
    for d, dst in (("fetch", "_pmc_FETCH_SIZE_counter_collection.csv"), ("write", "_pmc_WRITE_SIZE_counter_collection.csv"),
                   ("sq1", "_pmc_SQ1_counter_collection.csv"), ("sq2", "_pmc_SQ2_counter_collection.csv")):
        fs = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
        if fs:
            keep = [r for r in rows(fs[0]) if "admm_solve_kernel" in r.get("Kernel_Name", "") or "lpv_kernel" in r.get("Kernel_Name", "")]
            with open(prefix + dst, "w", newline="") as fh:
                wr = csv.DictWriter(fh, fieldnames=list(keep[0].keys()) if keep else ["empty"])
                wr.writeheader()
                for r in keep:
                    wr.writerow(r)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
