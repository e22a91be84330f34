#!/usr/bin/env python3
"""Audit of hand-issued LDS reads in hipcc output (cdna_hip_programming.md 5.7 item 1): an `asm volatile` ds_read's VGPR
destination counts as written at ;;#ASMEND, so under register pressure the compiler may copy or spill it before the data has
landed.  For every kernel of the assembly file: each register written by a ds_read inside an ASMSTART/ASMEND block must not be
referenced by any instruction outside such blocks until an ASM block containing s_waitcnt lgkmcnt has named it (the "+v"
operands of the wait statement are not printed, so the rule checked is: no compiler instruction touches a pending register
before the next asm s_waitcnt whose count covers it -- conservatively, before the next asm s_waitcnt at all).

    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only x.hip -o x.s ; python tools/asm_load_audit.py x.s
Exit code 1 when a violation is found."""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


bad = 0
kernel = None
pending = {}          # reg -> line number of the asm read
in_asm = False
for ln, l in enumerate(open(sys.argv[1]), 1):
    c = l.split(";")[0].strip() if not l.strip().startswith(";;#") else l.strip()
    if re.match(r"^_Z\w+:", l):
        kernel, pending = l.split(":")[0], {}
        continue
    if l.strip().startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if l.strip().startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not c or c.startswith("."):
        if re.match(r"^\.LBB", l) and pending:
            # a basic-block boundary with reads in flight: the wait must be in the same block in this code base
            pass
        continue
    parts = c.split(None, 1)
    op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
    if in_asm:
        if op.startswith("ds_read"):
            for r in regs(rest.split(",")[0]):
                pending[r] = ln
        elif op == "s_waitcnt" and "lgkmcnt" in rest:
            n = int(re.search(r"lgkmcnt\((\d+)\)", rest).group(1))
            if n == 0:
                pending = {}
            else:
                # the n most recent read instructions may stay in flight: drop all but the registers of the last n reads
                lines = sorted(set(pending.values()))
                keep = set(lines[-n:]) if n <= len(lines) else set(lines)
                pending = {r: w for r, w in pending.items() if w in keep}
        continue
    touched = [r for r in regs(rest) if r in pending]
    if touched:
        bad += 1
        print(f"{kernel}: line {ln}: `{c}` touches v{touched} while the asm read of line {pending[touched[0]]} is in flight")
print(f"{bad} violation(s)")
sys.exit(1 if bad else 0)
