#!/usr/bin/env python3
"""Fail the build when a solve kernel spills vector registers (gfx950 assembly kept by the Makefile).

The N = 20 kernels run at two wavefronts per SIMD (256 registers): a spill there is scratch traffic on the one path whose
HBM traffic is supposed to be inputs + outputs only (round 1 carried 6 MB of scratch stores per 1024-instance launch).
Usage: check_kernel_resources.py file.s [name-substring ...] [--allow name-substring=bytes ...]
(default: every kernel of the file; --allow tolerates up to that much scratch per lane in the kernels it names: the planner
N = 30 kernel that has to fit 256 registers to run three instances per CU keeps 17 loop-invariant registers of its
termination-check / re-factorisation code in scratch, none on the per-iteration path -- DESIGN.md section 4)"""
import re
import sys


def kernels(path):
    out, cur = [], None
    for ln in open(path):
        m = re.match(r"\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name" and not v.startswith(("_Z", "k")) and cur is not None:
            continue
        if k == "name":
            cur = None
        # the metadata lists the keys of one kernel alphabetically: .name comes before the counts
        if k == "name" and v.startswith("_Z"):
            cur = {"name": v}
            out.append(cur)
        elif cur is not None and k != "name":
            cur[k] = int(v)
    return out


if __name__ == "__main__":
    path, args = sys.argv[1], sys.argv[2:]
    allow, pats = {}, []
    while args:
        a = args.pop(0)
        if a == "--allow":
            name, _, lim = args.pop(0).partition("=")
            allow[name] = int(lim)
        else:
            pats.append(a)
    bad = 0
    for k in kernels(path):
        if pats and not any(p in k["name"] for p in pats):
            continue
        spills, scratch = k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0)
        limit = max([v for n, v in allow.items() if n in k["name"]] or [0])
        flag = scratch > limit or (spills > 0 and limit == 0)
        print("%s %-70s vgpr %3d  vgpr spills %2d  scratch %3d B  (sgpr spills %d)" % (
            "SPILL" if flag else ("allow" if spills or scratch else "ok   "), k["name"][:70], k.get("vgpr_count", -1), spills, scratch, k.get("sgpr_spill_count", 0)))
        bad += flag
    sys.exit(1 if bad else 0)
