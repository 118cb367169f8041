#!/usr/bin/env python3
"""Kernel timeline of the clustering phase (first bq_* kernel ... sec_mean) of the last step in a rocprofv3
--kernel-trace CSV, per queue.  usage: cluster_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?")) for r in rows]
ends = [i for i, e in enumerate(ev) if e[2].startswith("cp_merge")]
last = ends[-1]
first = next(i for i in range(max(last - 400, 0), last) if (ev[i][2].startswith("bq_") or ev[i][2].startswith("bqg_")))
first = max(first - 12, 0)
t0 = ev[first][0]
prev_end = {}
for e in ev[first:last + 1]:
    q = e[3]
    gap = (e[0] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e[1]
    print("q%-3s %9.1f us  +%8.1f us  gap %7.1f  %s" % (q, (e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, gap, e[2]))
