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


def hot_path(path):
    """Per kernel: the code of one ADMM iteration -- everything that can execute between the `; LPVMPC_HOT_BEGIN` and `; LPVMPC_HOT_END`
    markers the solve kernel puts around it -- scratch accesses (VGPR spills), v_readlane / v_writelane on the VGPRs that hold spilled
    SGPRs (the registers v_writelane targets anywhere in the kernel), and the instruction count.  The region is followed along the
    control flow (fall-through and branch targets, until the END marker, the next BEGIN marker or s_endpgm): blocks of the iteration
    that the compiler laid out behind the loop are counted and checked, blocks of other code laid out in between are not (until
    round 5 the region was taken in text order)."""
    out, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(ln)
        if ln.startswith(".Lfunc_end"):
            spill_regs = set(re.findall(r"v_writelane_b32\s+(v\d+)", "".join(body)))
            labels = {}
            for i, b in enumerate(body):
                m = re.match(r"^(\.LBB\d+_\d+):", b)
                if m:
                    labels[m.group(1)] = i
            starts = [i + 1 for i, b in enumerate(body) if "LPVMPC_HOT_BEGIN" in b]
            seen, n_ins, scratch, stores, lanes = set(), 0, 0, 0, 0
            work = list(starts)
            while work:
                i = work.pop()
                while i < len(body) and i not in seen:
                    seen.add(i)
                    b = body[i]
                    if "LPVMPC_HOT_END" in b or "LPVMPC_HOT_BEGIN" in b:
                        break
                    if b.startswith("\t") and not b.strip().startswith((".", ";")):
                        op = b.split()[0]
                        n_ins += 1
                        scratch += op.startswith("scratch_")
                        stores += op.startswith("scratch_store")
                        if op in ("v_readlane_b32", "v_writelane_b32") and set(re.findall(r"\bv\d+\b", b)) & spill_regs:
                            lanes += 1
                        if op == "s_endpgm":
                            break
                        m = re.search(r"(\.LBB\d+_\d+)", b) if op.startswith(("s_cbranch", "s_branch")) else None
                        if m and m.group(1) in labels:
                            work.append(labels[m.group(1)])
                        if op == "s_branch":
                            break
                    i += 1
            out[name] = dict(instructions=n_ins, scratch=scratch, scratch_stores=stores, spill_lane_moves=lanes)
            name = None
    return out


def role_barriers(path):
    """Per kernel and role group: the s_barrier instructions between each `; LPVMPC_ROLE_BEGIN <group> [xN]` / `; LPVMPC_ROLE_END <group>` pair
    (text order; xN: the region is a loop body that runs N times per pass).  Role paths are pieces of one kernel that different wavefronts
    of a workgroup execute INSTEAD of each other (relay4's four parts of a KKT solve; the two sides of the tail kernel's loop), each with
    the workgroup's barriers inside: that is outside HIP's convergence rules and works because s_barrier counts wavefronts -- as long as
    every path passes the same number.  Returns {kernel: {group: [count, ...]}}."""
    out, name, open_ = {}, None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, open_ = m.group(1), {}
            continue
        if name is None:
            continue
        m = re.search(r"LPVMPC_ROLE_(BEGIN|END)\s+(\w+)(?:\s+x(\d+))?", ln)
        if m:
            grp = m.group(2)
            if m.group(1) == "BEGIN":
                open_[grp] = [0, int(m.group(3) or 1)]
            elif grp in open_:
                cnt, mul = open_.pop(grp)
                out.setdefault(name, {}).setdefault(grp, []).append(cnt * mul)
            continue
        if ln.startswith("\t") and ln.split() and ln.split()[0] == "s_barrier":
            for v in open_.values():
                v[0] += 1
    return out


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
    allow, pats, hot_lanes_max, hot_allow, hot_scratch = {}, [], None, {}, {}
    while args:
        a = args.pop(0)
        if a == "--allow":
            name, _, lim = args.pop(0).partition("=")
            allow[name] = int(lim)
        elif a == "--hot-lane-allow":           # name-substring=count: a kernel outside the benchmarked set that may carry more lane moves
            name, _, lim = args.pop(0).partition("=")
            hot_allow[name] = int(lim)
        elif a == "--hot-scratch-allow":        # name-substring=count: scratch RELOADS (never stores) tolerated between the markers (loop-invariant values)
            name, _, lim = args.pop(0).partition("=")
            hot_scratch[name] = int(lim)
        elif a == "--hot-lane-moves":           # largest number of SGPR-spill lane moves tolerated between the hot-path markers
            hot_lanes_max = int(args.pop(0))
        else:
            pats.append(a)
    bad = 0
    hot = hot_path(path)
    for k in kernels(path):
        if pats and not any(p in k["name"] for p in pats):
            continue
        spills, scratch = k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0)
        limit = max([v for n, v in allow.items() if n in k["name"]] or [0])
        flag = scratch > limit or (spills > 0 and limit == 0)
        print("%s %-70s vgpr %3d  vgpr spills %2d  scratch %3d B  (sgpr spills %d)" % (
            "SPILL" if flag else ("allow" if spills or scratch else "ok   "), k["name"][:70], k.get("vgpr_count", -1), spills, scratch, k.get("sgpr_spill_count", 0)))
        bad += flag
        h = hot.get(k["name"])
        if h and h["instructions"]:
            lanes_max = max([v for n, v in hot_allow.items() if n in k["name"]] or [hot_lanes_max if hot_lanes_max is not None else 1 << 30])
            # the allowance covers RELOADS of loop-invariant values only: a scratch store between the markers always fails
            loads = h["scratch"] - h["scratch_stores"]
            hflag = loads > max([v for n, v in hot_scratch.items() if n in k["name"]] or [0]) or h["scratch_stores"] > 0 or h["spill_lane_moves"] > lanes_max
            print("      per-iteration code (reachable between the markers): %d instructions, %d scratch loads, %d scratch stores, %d SGPR-spill lane moves%s"
                  % (h["instructions"], loads, h["scratch_stores"], h["spill_lane_moves"], "  <-- FAIL" if hflag else ""))
            bad += hflag
    # role-divergent barriers: every path of a group must pass the same number (see role_barriers)
    for kname, groups in sorted(role_barriers(path).items()):
        if pats and not any(p in kname for p in pats):
            continue
        for grp, counts in sorted(groups.items()):
            ok = len(set(counts)) == 1 and counts[0] > 0
            print("%s %-70s role group %-10s barriers per path %s%s" % ("ok   " if ok else "ROLES", kname[:70], grp, counts, "" if ok else "  <-- FAIL: the role paths pass different numbers of workgroup barriers"))
            bad += not ok
    sys.exit(1 if bad else 0)
