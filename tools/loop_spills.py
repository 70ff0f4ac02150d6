#!/usr/bin/env python3
"""Scalar-register spill traffic per loop of one kernel in a hipcc -S listing (tools/kernel_resources.sh writes one):
instructions, v_readlane (reloads) and v_writelane (spills) per loop header.  The traversal loops run at the
106-SGPR ceiling, so a value more kept live across them shows up here as reloads per trip.
Usage: tools/loop_spills.py listing.s [kernel name substring]"""
import re, sys
from collections import defaultdict
txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "renderKernelILi16ELb0ELi8E"
m = re.search(r"^(\S*" + re.escape(want) + r"\S*): ", txt, re.M)
i = m.start()
lines = [l.strip() for l in txt[i:txt.index("s_endpgm", i)].split("\n")]
cur, per, depth = None, defaultdict(lambda: [0, 0, 0]), {}
for l in lines:
    m = re.match(r"\.LBB\d+_(\d+):\s*;\s*(.*)", l)
    if m:
        h = re.search(r"Header=BB\d+_(\d+) Depth=(\d+)", m.group(2))
        if h:
            cur = int(h.group(1)); depth[cur] = int(h.group(2))
        elif "Parent Loop" in m.group(2) or "Loop Header" in m.group(2):
            cur = int(m.group(1))
        else:
            cur = None
        continue
    if cur is None or not l or l.startswith((";", ".")):
        continue
    op = l.split()[0]
    per[cur][0] += 1
    per[cur][1] += op == "v_readlane_b32"
    per[cur][2] += op == "v_writelane_b32"
tot = [sum(v[k] for v in per.values()) for k in range(3)]
print(f"{m and ''}{want}: {len(lines)} lines; in loops {tot[0]} instructions, {tot[1]} readlane, {tot[2]} writelane")
for h, (n, r, w) in sorted(per.items(), key=lambda x: -x[1][0])[:10]:
    print(f"  loop {h:5d} depth {depth.get(h, '?')}: {n:5d} instructions, {r:4d} readlane, {w:4d} writelane")
