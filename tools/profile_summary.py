#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats / PMC counters) into the small summaries kept
under profiles/.  Usage on the GPU box (after rocprofv3 ... --output-format csv -d DIR):

    python tools/profile_summary.py stats DIR  > profiles/rNN_kernel_stats.md
    python tools/profile_summary.py pmc DIR    > profiles/rNN_pmc.md   (also rewrites profiles/pmc_traffic.json)
    python tools/profile_summary.py timeline DIR > profiles/rNN_timeline.md   (one step: order, queues, overlap)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mmb::", "")


def stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k[:90]}` | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / tot:.1f} |")
    print(f"\ntotal kernel time {tot / 1e3:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


def timeline(d, which=-2):
    """One step of the kernel trace as a timeline (start offset, duration, queue) + how much of it ran two-deep.
    A step ends with the last `lstm_unpack_dw_kernel` before the first forward recurrence that follows a backward one."""
    which = int(which)      # (index of the step in the trace; a replayed-graph step sits in the middle of a bench.py run)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows))
    cuts, seen_bwd = [0], False
    for i, e in enumerate(ev):
        if e[2].startswith(("lstm_rec_bwd", "lstm_fs_bwd")):
            seen_bwd = True
        if e[2].startswith(("lstm_rec_fwd", "lstm_fs_fwd")) and seen_bwd:
            # the previous step ends with the last weight-gradient unpack kernel before this recurrence
            j = i
            while j > 0 and not ev[j - 1][2].startswith("lstm_unpack_dw"):
                j -= 1
            j = j if j > 0 else i
            cuts.append(j)
            seen_bwd = False
    cuts.append(len(ev))
    steps = [ev[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    st = steps[which] if len(steps) >= abs(which) else steps[-1]
    t0 = st[0][0]
    queues = sorted({e[3] for e in st})
    print(f"step with {len(st)} dispatches on {len(queues)} queue(s); t = start offset from the step's first dispatch\n")
    print("| t us | dur us | queue | kernel | runs beside |")
    print("|---:|---:|---|---|---|")
    points = []
    for i, (a, b, k, q) in enumerate(st):
        beside = sorted({k2[:28] for (a2, b2, k2, q2) in st if q2 != q and a2 < b and b2 > a})
        print(f"| {(a - t0) / 1e3:.1f} | {(b - a) / 1e3:.1f} | {queues.index(q)} | `{k[:70]}` | {', '.join(beside)} |")
        points += [(a, 1), (b, -1)]
    points.sort()
    depth, last, busy1, busy2 = 0, points[0][0], 0, 0
    for t, dlt in points:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += dlt
        last = t
    span = max(e[1] for e in st) - t0
    ksum = sum(b - a for a, b, _, _ in st)
    print(f"\nstep span {span / 1e3:.1f} us; sum of kernel durations {ksum / 1e3:.1f} us; some kernel running {busy1 / 1e3:.1f} us; "
          f"two or more kernels running {busy2 / 1e3:.1f} us")
    side = [e for e in st if queues.index(e[3]) != queues.index(st[0][3])]
    if side:
        print(f"dispatches off the main queue: {len(side)}, {sum(b - a for a, b, _, _ in side) / 1e3:.1f} us of kernel time")


def pmc(d, cfg="cfg2"):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    print("| kernel | " + " | ".join(f"{c} /launch" for c in counters) + " |")
    print("|---|" + "---:|" * len(counters))
    traffic = {}
    for k in sorted(acc):
        vals = []
        for c in counters:
            vals.append(acc[k][c] / max(cnt[k][c], 1) if c in acc[k] else float("nan"))
        print(f"| `{k[:80]}` | " + " | ".join(f"{v:.4g}" for v in vals) + " |")
        if ("FETCH_SIZE" in acc[k] or "WRITE_SIZE" in acc[k]) and re.match(r"(att_|lstm_|gemm_)", k):
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE (KiB) reads exactly half of a wide coalesced stream on gfx950 -> x2;
            # WRITE_SIZE (KiB) is exact for 16-B-per-lane stores.
            t = {}
            if "FETCH_SIZE" in acc[k]:
                t["fetch_bytes_corrected"] = 2.0 * 1024 * acc[k]["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"]
            if "WRITE_SIZE" in acc[k]:
                t["write_bytes"] = 1024 * acc[k]["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"]
            traffic[k] = t
    if traffic:
        print("\nHBM-side traffic per launch (bytes; FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE as is):")
        for k, v in traffic.items():
            print(f"- `{k[:80]}`: " + ", ".join(f"{n} {x:.4g}" for n, x in v.items()))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "profiles", "pmc_traffic.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        # the file is stamped with a hash of the kernel sources it was measured on (bench.py quotes it only when that
        # matches the sources it is running): measurements of other sources are dropped
        sys.path.insert(0, root)
        from mmbidaf_amd.build import source_hash      # the hash compiled into the library (mmb_build_hash)
        stamp = source_hash()
        if old.get("source_hash") != stamp:
            old = {"source_hash": stamp}
        parts = old.setdefault(cfg, {}).setdefault("_parts", {})
        for k, v in traffic.items():
            stem = re.sub(r"<.*$", "", k)            # bench.py looks kernels up by their symbol stem
            parts.setdefault(stem, {}).update(v)
            if "fetch_bytes_corrected" in parts[stem] and "write_bytes" in parts[stem]:
                old[cfg][stem] = parts[stem]["fetch_bytes_corrected"] + parts[stem]["write_bytes"]
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])      # pmc DIR [config name]
