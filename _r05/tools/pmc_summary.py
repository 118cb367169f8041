#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: launches, mean counter value per launch.
usage: pmc_summary.py <counter_collection.csv> [<more.csv> ...] > summary.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name") or "?"
            name = name.split("(")[0].replace("void ", "")
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value") or 0)
            a = acc[name][c]
            a[0] += 1
            a[1] += v
print("kernel,counter,launches,mean_per_launch,total")
for name in sorted(acc, key=lambda n: -sum(v[1] for v in acc[n].values())):
    for c, (n, tot) in sorted(acc[name].items()):
        print("%s,%s,%d,%.3f,%.3f" % (name, c, n, tot / max(n, 1), tot))
