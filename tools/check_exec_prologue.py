#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -save-temps / -S output) for vector instructions that sit in front of the
`s_or_b64 exec, exec, s[..]` which re-opens the execution mask at the top of a control-flow join block.

Such an instruction runs with the mask of ONE branch only, so a whole-register copy placed there (seen from LLVM's
machine-sink pass under register pressure: `v_mov_b32 v200, v188` in front of the restore, v200 read by other
lanes later) leaves the remaining lanes of its destination stale: a miscompile that showed up as wrong block
reductions in the N = 40 planner kernel.  Only plain VGPR / AGPR copies are reported.  Exit code 1 on a suspect."""
import re
import sys


def dest_regs(instr):
    """VGPR numbers written by a plain copy (first operand)."""
    m = re.match(r"\S+\s+v(\d+),", instr)
    if m:
        return [int(m.group(1))]
    m = re.match(r"\S+\s+v\[(\d+):(\d+)\],", instr)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def writes(instr, reg):
    """True if the instruction's first operand (its destination) covers v<reg>."""
    ops = instr.split(None, 1)[1] if " " in instr else ""
    if instr.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store", "s_")):
        return False
    m = re.match(r"v\[(\d+):(\d+)\]|v(\d+)", ops)
    if not m:
        return False
    return int(m.group(3)) == reg if m.group(3) is not None else int(m.group(1)) <= reg <= int(m.group(2))


def reads(instrs, reg):
    """True if register v<reg> appears as a SOURCE operand of one of the instructions."""
    for t in instrs:
        ops = t.split(None, 1)[1] if " " in t else ""
        srcs = ops.split(",", 1)[1] if "," in ops else ""
        if t.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store")):
            srcs = ops                                   # stores only read
        for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", srcs):
            if m.group(3) is not None:
                if int(m.group(3)) == reg:
                    return True
            elif int(m.group(1)) <= reg <= int(m.group(2)):
                return True
    return False


def scan(path, window=8, lookback=2000):
    suspects = []
    kernel = None
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if re.match(r"^\.LBB\d+_\d+:", ln):
            label = ln.split(":")[0]
            body = []
            j = i + 1
            while j < len(lines) and len(body) < window:
                t = lines[j].strip()
                j += 1
                if not t or t.startswith(";") or t.startswith("."):
                    if re.match(r"^\.LBB", t):
                        break
                    continue
                body.append(t)
            for k, t in enumerate(body):
                if re.match(r"s_or_b64\s+exec,\s*exec,", t):
                    # plain whole-register copies are what the register allocator / machine-sink insert; computations,
                    # stores and v_writelane (exec-independent SGPR spills) in front of the mask restore are normally the
                    # legitimate tail of the branch that falls through into the join block
                    pre = []
                    # a mask-narrowing instruction between the label and the restore opens a NEW guarded region inside this block
                    # (if-then without else: `v_mov v33, default; s_and_saveexec; v_mov v33, v234; s_or_b64 exec`): copies behind
                    # it belong to that region's lanes by construction, the other lanes keep the value set in front of it
                    # (only instructions that NARROW the mask open such a region -- s_or_saveexec / s_xor_saveexec / s_mov exec widen or
                    # restore it -- and a copy behind the narrowing is exempt only if its destination was given a value under the wider
                    # mask earlier in this block: otherwise the lanes outside the region reach the join with a stale register)
                    opened = [i_ for i_, b_ in enumerate(body[:k]) if re.match(r"s_(and|andn2)_saveexec_b64|s_and_b64\s+exec,\s*exec,|s_andn2_b64\s+exec,\s*exec,", b_)]
                    first_open = opened[0] if opened else k
                    # the ELSE entry of a structurised if / else: `s_or_saveexec_b64 sX, sY` (all lanes of the construct) followed by
                    # `s_xor_b64 exec, exec, sX` (the lanes that skipped the THEN side).  A copy behind that pair is the else side's
                    # value of a phi: legitimate when the then side, the straight-line code in front of the label, wrote the same
                    # destination for its own lanes (looked up in the `lookback` instructions in front of the label)
                    else_open = None
                    for i_, b_ in enumerate(body[:k]):
                        m_ = re.match(r"s_or_saveexec_b64\s+(s\[\d+:\d+\])", b_)
                        if m_:
                            for i2, b2 in enumerate(body[i_ + 1:k], i_ + 1):
                                if re.match(r"s_xor_b64\s+exec,\s*exec,\s*" + re.escape(m_.group(1)), b2):
                                    else_open = i2
                                    break
                            break
                    # ... or, in its shorter form, `s_andn2_saveexec_b64 sX, sY` as the block's first instruction (the lanes of the construct
                    # that are not in the then side's mask sY)
                    if else_open is None and body and re.match(r"s_andn2_saveexec_b64\s", body[0]):
                        else_open = 0
                    then_side = []
                    if else_open is not None:
                        j2 = i - 1
                        # (round 6: the THEN side is the ONE basic block in front of the label -- the walk stops at the previous label or
                        # branch.  A window of 2000 instructions made the exemption nearly unconditional in kernels that write almost
                        # every register within it.)
                        while j2 >= 0 and len(then_side) < lookback and not re.match(r"^_Z\w+:", lines[j2]):
                            t2 = lines[j2].strip()
                            j2 -= 1
                            if re.match(r"^\.LBB\d+_\d+:", t2) or re.match(r"s_(cbranch|branch|setpc|endpgm)", t2):
                                break
                            if t2 and not t2.startswith((";", ".")):
                                then_side.append(t2)
                    for idx, b in enumerate(body[:k]):
                        if idx > first_open and re.match(r"(v_mov_b32_e32\s+v\d+,\s*v\d+$|v_mov_b64_e32\s+v\[[\d:]+\],\s*v\[|v_accvgpr_(read|write)_b32)", b):
                            if all(any(writes(early, d) for early in body[:first_open]) for d in dest_regs(b)):
                                continue
                        if not re.match(r"(v_mov_b32_e32\s+v\d+,\s*v\d+$|v_mov_b64_e32\s+v\[[\d:]+\],\s*v\[|v_accvgpr_(read|write)_b32)", b):
                            continue
                        if else_open is not None and idx > else_open and all(any(writes(t2, d) for t2 in then_side) for d in dest_regs(b)):
                            continue
                        # a copy whose destination is consumed again before the mask restore is a temporary of the branch
                        # that falls through into the join (e.g. the halves of a 64-bit address product feeding a store
                        # of that branch), not a value handed to the lanes behind the join
                        if all(reads(later, d) for d in dest_regs(b) for later in [body[idx + 1:k]]):
                            continue
                        pre.append(b)
                    if pre:
                        suspects.append((kernel, label, pre, t))
                    break
                if re.match(r"s_(cbranch|branch|barrier|endpgm)", t):
                    break
        i += 1
    return suspects


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for kernel, label, pre, t in scan(p):
            bad += 1
            print("%s: %s %s: vector instruction(s) before '%s': %s" % (p, kernel, label, t, "; ".join(pre)))
    print("%d suspect join-block prologue(s)" % bad)
    sys.exit(1 if bad else 0)
