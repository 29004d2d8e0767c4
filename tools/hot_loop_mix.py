#!/usr/bin/env python3
"""Per-block instruction mix of the hot region (reachable between HOT markers) of one kernel."""
import re, sys
from collections import Counter
sys.path.insert(0, '/root/repo/tools')
from asm_mix import classify
path, name = sys.argv[1], sys.argv[2]
body, on = [], False
for ln in open(path):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        on = name in m.group(1); body = [] if on else body
        continue
    if on:
        body.append(ln)
        if ln.startswith(".Lfunc_end"): break
labels = {}
for i, b in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", b)
    if m: labels[m.group(1)] = i
starts = [i + 1 for i, b in enumerate(body) if "LPVMPC_HOT_BEGIN" in b]
seen = set(); work = list(starts)
while work:
    i = work.pop()
    while i < len(body) and i not in seen:
        seen.add(i); b = body[i]
        if "LPVMPC_HOT_END" in b or "LPVMPC_HOT_BEGIN" in b: break
        if b.startswith("\t") and not b.strip().startswith((".", ";")):
            op = b.split()[0]
            if op == "s_endpgm": break
            m = re.search(r"(\.LBB\d+_\d+)", b) if op.startswith(("s_cbranch", "s_branch")) else None
            if m and m.group(1) in labels: work.append(labels[m.group(1)])
            if op == "s_branch": break
        i += 1
# group by basic block in text order
cur, blocks = "start", {}
order = []
for i in sorted(seen):
    b = body[i]
    m = re.match(r"^(\.LBB\d+_\d+):", b)
    if m: cur = m.group(1)
    if b.startswith("\t") and not b.strip().startswith((".", ";")):
        if cur not in blocks: blocks[cur] = Counter(); order.append(cur)
        blocks[cur][classify(b.split()[0])] += 1
        if b.split()[0].startswith('ds_'):
            blocks[cur]['lds_' + ('rd' if 'read' in b.split()[0] else 'wr')] += 1
tot = Counter()
for k in order:
    c = blocks[k]; tot += c
    print("%-12s %4d  %s" % (k, sum(v for kk, v in c.items() if not kk.startswith('lds_')), " ".join("%s %d" % kv for kv in sorted(c.items()))))
print("total", sum(v for kk, v in tot.items() if not kk.startswith('lds_')), dict(tot))
