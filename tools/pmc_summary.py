#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/collect_pmc.sh into profiles/<round>_pmc.json (what bench.py reports as
roofline.traffic / roofline.valu / roofline.valu_timed, with the file named as the source) and copy the kernel-trace summaries
next to it.

Launch classes.  With straggler deferral three kinds of launches solve instances: MAIN launches (grid = batch x 128 work-items:
every instance's first 100 iterations), RESUME passes of the same kernel name (grid = pool entries x 128: parked instances,
100 more iterations each) and the whole-CU TAIL kernel (512-thread workgroups: lpvmpc_join runs what is left to completion).
The per-class tables (`*_kernel_classes.csv`: count, average / total / min / max ns per kernel and class) are what the bench
line's kernel_avg_ms (MAIN only) and all_launches_avg_ms (all three) can be recomputed from.

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
B = 1024


def rows(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


def grid_of(r):
    return int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))


def wg_of(r):
    return int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"])


def launch_class(r):
    """main / resume / tail / setup for a solve-kernel dispatch, None for anything else."""
    if "admm_solve_kernel" not in r.get("Kernel_Name", ""):
        return None
    g, wg = grid_of(r), wg_of(r)
    if wg == 512:
        return "tail"
    if g >= B * wg:
        return "main"
    if g <= wg:
        return "setup"                     # the one-instance launches that create the stream queues before the timed region
    return "resume"


def solve_rows(d, pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out += [r for r in rows(f) if "admm_solve_kernel" in r.get("Kernel_Name", "")]
    return out


def per_launch(rs, counter, cls):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and launch_class(r) == cls]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def class_table(trace_csv, dst):
    """count / avg / total / min / max duration per (kernel, class) from a rocprofv3 kernel_trace.csv."""
    acc = {}
    for r in rows(trace_csv):
        name = r["Kernel_Name"]
        cls = launch_class(r) or "-"
        short = name.split("(")[0].replace("void ", "")
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = acc.setdefault((short, cls, grid_of(r) if cls != "-" else 0, wg_of(r) if cls != "-" else 0), [])
        a.append(d)
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Class", "Grid_Size", "Workgroup_Size", "Calls", "AverageNs", "TotalDurationNs", "MinNs", "MaxNs"])
        for (short, cls, g, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([short, cls, g, wg, len(v), "%.1f" % (sum(v) / len(v)), sum(v), min(v), max(v)])
    return {"%s|%s" % (k[0], k[1]): {"calls": len(v), "avg_ns": sum(v) / len(v), "total_ns": sum(v)} for k, v in acc.items() if k[1] != "-"}


def counters(src, dirs, cls):
    sq = {}
    for d in dirs:
        rs = solve_rows(os.path.join(src, d), "*counter_collection.csv")
        for name in sorted({r["Counter_Name"] for r in rs}):
            v, n = per_launch(rs, name, cls)
            if v is not None:
                sq[name] = v
    rs = solve_rows(os.path.join(src, dirs[0]), "*counter_collection.csv")
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if r["Counter_Name"] == "SQ_WAVES" and launch_class(r) == cls]
    return sq, (sum(dur) / len(dur) if dur else None), len(dur)


def valu_block(sq, ns, what):
    if not ("SQ_INSTS_VALU" in sq and "SQ_ACTIVE_INST_VALU" in sq and ns):
        return None
    clk = sq.get("GRBM_GUI_ACTIVE")
    cycles = (clk / 8.0) if clk else ns * 2.1                 # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    issue = 4.0 * sq["SQ_ACTIVE_INST_VALU"]
    out = {"wave_instr_per_launch": sq["SQ_INSTS_VALU"], "issue_cycles": issue, "frac": issue / (N_SIMD * cycles), "launch_cycles": cycles,
           "definition": "issue_cycles = 4 x SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves); frac = issue_cycles / (1024 SIMDs x launch cycles): "
                         "launch-average VALU issue utilisation of " + what}
    if "SQ_WAVE_CYCLES" in sq:
        out["active_share_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in sq:
            out["wait_share_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    return out


def traffic(src, fdir, wdir, cls):
    f, nf = per_launch(solve_rows(os.path.join(src, fdir), "*counter_collection.csv"), "FETCH_SIZE", cls)
    w, nw = per_launch(solve_rows(os.path.join(src, wdir), "*counter_collection.csv"), "WRITE_SIZE", cls)
    if f is None or w is None:
        return {}
    return {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": nf,
            "correction": "gfx950: FETCH_SIZE under-reports coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
            "traffic_bytes_per_launch": 2.0 * f * 1024 + w * 1024}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    cmd_iso = open(os.path.join(src, "command.txt")).read().strip()
    out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- " + cmd_iso + " (one pass per counter group: tools/collect_pmc.sh)",
           "kernel": "admm_solve_kernel<6, 20, 2, MFMA sweeps>", "batch": B, "seed": 0}
    # ---- which build: sha256 of the library the passes ran with (written next to them on the GPU box) and the commit this summary is made at
    import subprocess
    build = {}
    try:
        build["liblpvmpc_sha256"] = open(os.path.join(src, "lib_sha256.txt")).read().strip()
    except OSError:
        pass
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        build["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
        build["git_dirty_sources"] = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "autonomous-racing-lpv-mpp-mpc_amd/csrc", "include"], text=True).strip())
    except (OSError, subprocess.CalledProcessError):
        pass
    out["build"] = build
    # ---- isolated plain launches (no deferral, one stream) --------------------------------------------------------------
    out.update(traffic(src, "fetch", "write", "main"))
    sq, ns, n = counters(src, ("sq1", "sq2"), "main")
    out["sq_per_launch"] = sq
    if ns:
        out["kernel_ns_under_pmc"] = ns
    v = valu_block(sq, ns, "ONE isolated plain 1024-instance launch (it lasts as long as its slowest instance)")
    if v:
        out["valu"] = v
    # ---- the timed configuration (deferral 100 / 100, 4 streams): main launches, resume passes, tail kernel -----------------
    if os.path.exists(os.path.join(src, "command_timed.txt")):
        cmd_t = open(os.path.join(src, "command_timed.txt")).read().strip()
        timed = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- " + cmd_t,
                 "note": "the headline configuration; under --pmc rocprofv3 runs the dispatches one at a time, so every launch is alone on the GPU "
                         "while its counters are read (kernel_ns_under_pmc is that solitary duration, not the duration next to three neighbours)"}
        for cls in ("main", "resume", "tail"):
            sqc, nsc, nc = counters(src, ("t_sq1", "t_sq2"), cls)
            if not nc:
                continue
            blk = {"launches_averaged": nc, "kernel_ns_under_pmc": nsc, "sq_per_launch": sqc}
            blk.update(traffic(src, "t_fetch", "t_write", cls))
            vb = valu_block(sqc, nsc, "a %s launch of the timed configuration, alone on the GPU" % cls)
            if vb:
                blk["valu"] = vb
            timed[cls] = blk
        out["timed"] = timed
    json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
    # ---- kernel traces: stats as rocprofv3 wrote them + the per-class tables ------------------------------------------------
    classes = {}
    for tag, name in (("trace", "_streams1"), ("trace_default", "_bench_default"), ("trace_driver", "_bench_driver")):
        fs = glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True)
        if fs:
            shutil.copy(fs[0], prefix + name + "_kernel_stats.csv")
        ft = glob.glob(os.path.join(src, tag, "**", "*kernel_trace.csv"), recursive=True)
        if ft:
            classes[name[1:]] = class_table(ft[0], prefix + name + "_kernel_classes.csv")
    json.dump(classes, open(prefix + "_kernel_classes.json", "w"), indent=1)
    for d, dst in (("fetch", "_pmc_FETCH_SIZE_counter_collection.csv"), ("write", "_pmc_WRITE_SIZE_counter_collection.csv"),
                   ("sq1", "_pmc_SQ1_counter_collection.csv"), ("sq2", "_pmc_SQ2_counter_collection.csv"),
                   ("t_fetch", "_pmc_timed_FETCH_SIZE_counter_collection.csv"), ("t_write", "_pmc_timed_WRITE_SIZE_counter_collection.csv"),
                   ("t_sq1", "_pmc_timed_SQ1_counter_collection.csv"), ("t_sq2", "_pmc_timed_SQ2_counter_collection.csv")):
        fs = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
        if fs:
            keep = [r for r in rows(fs[0]) if "admm_solve_kernel" in r.get("Kernel_Name", "") or "lpv_kernel" in r.get("Kernel_Name", "")]
            drop = ("Kind", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Correlation_Id", "Kernel_Id")       # keep the files small
            with open(prefix + dst, "w", newline="") as fh:
                names = [k for k in (keep[0].keys() if keep else ["empty"]) if k not in drop]
                wr = csv.DictWriter(fh, fieldnames=names, extrasaction="ignore")
                wr.writeheader()
                for r in keep:
                    wr.writerow(r)
    print(json.dumps({k: v for k, v in out.items() if k != "sq_per_launch"}, indent=1)[:6000])
    print(json.dumps(classes, indent=1)[:4000])


if __name__ == "__main__":
    main()
