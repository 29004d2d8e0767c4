#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 assembly listing (hipcc -save-temps): per basic block and for a chosen
block range (e.g. the ADMM iteration loop).  Usage: asm_mix.py file.s <kernel-name-substring> [first_label last_label]"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.endswith("_dpp") or "dpp" in op: return "dpp"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if re.match(r"v_(fma|mul|add|max|min|rsq|rcp|sqrt|div|trig|ldexp|frexp|cmp|cmpx|cndmask).*_f64", op) or op in ("v_fmac_f64_e32",): return "f64"
    if op.startswith("v_"): return "valu32"
    return "other"


def kernel_body(path, name):
    lines = open(path).read().split("\n")
    out, on = [], False
    for ln in lines:
        if not on and re.match(r"^_Z\w+:", ln) and name in ln:
            on = True
        if on:
            out.append(ln)
            if "s_endpgm" in ln:
                break
    return out


def blocks(body):
    res = OrderedDict()
    cur = "entry"
    res[cur] = []
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            res[cur] = []
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        res[cur].append(t)
    return res


def mix(instrs):
    c = Counter()
    for t in instrs:
        op = t.split()[0]
        cls = classify(op)
        if cls == "valu32" and "dpp" in t: cls = "dpp"
        c[cls] += 1
    return c


if __name__ == "__main__":
    body = kernel_body(sys.argv[1], sys.argv[2])
    bl = blocks(body)
    names = list(bl)
    if len(sys.argv) >= 5:
        i0, i1 = names.index(sys.argv[3]), names.index(sys.argv[4])
        tot = Counter()
        for n in names[i0:i1 + 1]:
            tot += mix(bl[n])
        print("blocks %s .. %s: %d instructions" % (sys.argv[3], sys.argv[4], sum(tot.values())))
        for k, v in tot.most_common():
            print("  %-8s %6d" % (k, v))
    else:
        for n in names:
            c = mix(bl[n])
            if sum(c.values()) >= 40:
                br = [t for t in bl[n] if t.startswith(("s_cbranch", "s_branch"))]
                print("%-12s %5d  %s  %s" % (n, sum(c.values()), dict(c.most_common(8)), " ".join(b.split()[-1] for b in br)))
