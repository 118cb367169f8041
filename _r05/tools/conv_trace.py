#!/usr/bin/env python3
"""Join a rocprofv3 --kernel-trace CSV of tools/conv_bench.py with its group list: true GPU time per (layer, op).
usage: conv_trace.py <kernel_trace.csv> <groups.json>"""
import csv
import json
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
g = json.load(open(sys.argv[2]))
iters, groups = g["iters"], g["groups"]
segs, cur, inside = [], None, False
for r in rows:
    name = r["Kernel_Name"]
    if "flip" in name:
        if inside:
            segs.append(cur); inside = False
        else:
            cur = []; inside = True
        continue
    if inside:
        cur.append((name.split("(")[0][:44], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
table = {}
for label, seg in zip(groups, segs):
    layer, op = label.split("|")
    tot = sum(d for _, d in seg) / iters / 1e3
    names = {}
    for n, d in seg:
        names[n] = names.get(n, 0) + d / iters / 1e3
    table.setdefault(layer, {})[op] = (tot, names)
print("%-22s | %8s %8s | %8s %8s | %8s %8s   (GPU us per call, all kernels of the call)" % ("layer", "fwd1", "fwd2", "dgrad1", "dgrad2", "wgrad1", "wgrad2"))
for layer, ops in table.items():
    print("%-22s | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f" % tuple([layer] + [ops.get(k, (0, 0))[0] for k in ("fwd1", "fwd2", "dgrad1", "dgrad2", "wgrad1", "wgrad2")]))
if len(sys.argv) > 3:
    for layer, ops in table.items():
        for op, (tot, names) in ops.items():
            print(layer, op, {k: round(v, 1) for k, v in names.items()})
