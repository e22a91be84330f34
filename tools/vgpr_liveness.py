#!/usr/bin/env python3
"""VGPR liveness of one gfx950 kernel from its assembly: live-in count of every basic block that contains MFMAs, and which
registers are live through a block without being touched in it.

    python tools/vgpr_liveness.py x.s <kernel substring> [block label to detail]
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and ":" in l)


def regs(tok):
    """VGPRs v0.. as 0.., AGPRs a0.. as 256.. (one unified file on gfx950)"""
    out = []
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            base = 256 if m.group(1) == "a" else 0
            out += [base + k for k in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append((256 if m.group(4) == "a" else 0) + int(m.group(5)))
    return out


blocks, order, cur = {}, [], "entry"
blocks[cur] = []
order.append(cur)
for i in range(start + 1, len(lines)):
    l = lines[i]
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        order.append(cur)
        continue
    c = l.split(";")[0].rstrip()
    if c.startswith("\t") and not c.startswith("\t."):
        blocks[cur].append(c.strip())

succ, use, defs, nm = {}, {}, {}, {}
for bi, b in enumerate(order):
    u, d, s, n = set(), set(), [], 0
    fall = True
    for ins in blocks[b]:
        parts = ins.split(None, 1)
        op = parts[0]
        ops = parts[1].split(",") if len(parts) > 1 else []
        if op.startswith("s_cbranch"):
            s.append(ops[0].strip())
            continue
        if op == "s_branch":
            s.append(ops[0].strip())
            fall = False
            continue
        if op in ("s_endpgm",):
            fall = False
            continue
        n += op.startswith("v_mfma")
        store = op.startswith("ds_write") or "store" in op or op.startswith("global_atomic") or op.startswith("v_cmp") or \
            op.startswith("global_load_lds") or op.startswith("buffer_store") or op.startswith("ds_bpermute") and False
        srcs = ops if store else ops[1:]
        for o in srcs:
            for r in regs(o):
                if r not in d:
                    u.add(r)
        if not store and ops:
            dst = regs(ops[0])
            if op.startswith("v_mfma") or op.startswith("v_fmac") or op.startswith("v_mac") or "permlane" in op or op.startswith("v_dot2c"):
                for r in dst:      # read-modify-write destinations
                    if r not in d:
                        u.add(r)
            d.update(dst)
    if fall and bi + 1 < len(order):
        s.append(order[bi + 1])
    succ[b], use[b], defs[b], nm[b] = s, u, d, n

live_in = {b: set() for b in order}
live_out = {b: set() for b in order}
changed = True
while changed:
    changed = False
    for b in reversed(order):
        lo = set()
        for t in succ[b]:
            if t in live_in:
                lo |= live_in[t]
        li = use[b] | (lo - defs[b])
        if li != live_in[b] or lo != live_out[b]:
            live_in[b], live_out[b] = li, lo
            changed = True
for b in order:
    if nm[b]:
        through = live_in[b] & live_out[b] - use[b] - defs[b]
        print(f"{b:12s} mfma {nm[b]:3d}  live-in {len(live_in[b]):3d}  live-out {len(live_out[b]):3d}  touched {len(use[b] | defs[b]):3d}  live-through-untouched {len(through):3d}")
if len(sys.argv) > 3:
    b = sys.argv[3]
    through = sorted(live_in[b] & live_out[b] - use[b] - defs[b])
    print("live through", b, "untouched:", through)
    # where is each next used?
    for r in through[:200]:
        for t in order[order.index(b):]:
            if r in use[t]:
                print(f"  v{r}: next used in {t}")
                break
