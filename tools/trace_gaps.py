#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of one train step from a rocprofv3 --kernel-trace CSV.

    python tools/trace_gaps.py gpurun_out/prof_x/trace/**/..._kernel_trace.csv [--skip 0.5]

Only the last (1 - skip) fraction of the dispatches is used (clock ramp / warm-up)."""
import csv
import sys
import collections

path = sys.argv[1]
skip = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--skip" else 0.5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev is not None:
        gap[name].append(s - prev)
    prev = e
tot = 0.0
print(f"{'kernel':50s} {'n':>5s} {'mean us':>9s} {'gap before us':>14s}")
for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = gap.get(name, [0])
    print(f"{name:50s} {len(d):5d} {sum(d) / len(d) / 1e3:9.2f} {sum(g) / len(g) / 1e3:14.2f}")
