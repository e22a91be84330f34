#!/usr/bin/env python3
"""Where a gfx950 kernel spills: per basic block counts of scratch stores / loads, MFMAs and barriers, with the loop depth.

    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only x.hip -o x.s ;  python tools/spill_report.py x.s <kernel substring>
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") is False and ":" in l)
blk, cnt, order = "entry", {}, []
for i in range(start, len(lines)):
    l = lines[i]
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blk = m.group(1)
        depth = re.search(r"Depth=(\d+)", m.group(2))
        cnt[blk] = [0, 0, 0, 0, int(depth.group(1)) if depth else 0]
        order.append(blk)
    if blk not in cnt:
        cnt[blk] = [0, 0, 0, 0, 0]
        order.append(blk)
    c = cnt[blk]
    c[0] += "scratch_store" in l
    c[1] += "scratch_load" in l
    c[2] += "v_mfma" in l
    c[3] += "s_barrier" in l
tot = [0, 0]
for b in order:
    c = cnt[b]
    if c[4] > 0 and (c[0] or c[1] or c[2]):
        print(f"{b:14s} depth {c[4]} scratch st {c[0]:3d} ld {c[1]:3d} mfma {c[2]:3d} barriers {c[3]}")
        tot[0] += c[0]
        tot[1] += c[1]
print(f"in loops: {tot[0]} scratch stores, {tot[1]} scratch loads (dwordx4 counted once)")
