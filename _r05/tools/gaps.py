#!/usr/bin/env python3
"""Largest idle gaps of the busiest queue inside one training step of a rocprofv3 --kernel-trace CSV (all queues are
shown as busy/idle context).  usage: gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys
from collections import Counter

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Queue_Id"]) for r in rows]
adam = [i for i, e in enumerate(ev) if e[2].startswith("adamw_kernel")]
a0, a1 = adam[-2], adam[-1]
step = ev[a0 + 1:a1 + 1]
t0 = step[0][0]
print("step %.2f ms, %d kernels" % ((step[-1][1] - t0) / 1e6, len(step)))
# union of busy intervals over all queues
iv = sorted((s, e) for s, e, _, _ in step)
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
idle = sum(b[0] - a[1] for a, b in zip(merged[:-1], merged[1:]))
print("GPU idle (no kernel on any queue): %.2f ms" % (idle / 1e6))
gaps = []
for a, b in zip(merged[:-1], merged[1:]):
    g = (b[0] - a[1]) / 1e3
    if g >= ming:
        prev = max((e for e in step if e[1] <= a[1] + 1), key=lambda e: e[1])
        nxt = min((e for e in step if e[0] >= b[0] - 1), key=lambda e: e[0])
        gaps.append((g, (a[1] - t0) / 1e3, prev[2], nxt[2]))
for g, t, p, n in sorted(gaps, reverse=True)[:40]:
    print("%7.1f us idle at %8.1f us   after %-40s before %s" % (g, t, p, n))
print("idle in gaps >= %.0f us: %.2f ms; smaller gaps: %.2f ms" % (ming, sum(g[0] for g in gaps) / 1e3, idle / 1e6 - sum(g[0] for g in gaps) / 1e3))
