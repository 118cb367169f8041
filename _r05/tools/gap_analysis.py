#!/usr/bin/env python3
"""GPU busy / idle analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals vs wall span, the largest idle
gaps and which kernel precedes them.  usage: gap_analysis.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in rows)
ev = ev[int(len(ev) * skip):]          # drop warm-up
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
prev = ev[0][2]
gaps = []
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = n
busy += cur_e - cur_s
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f %%)  idle %.2f ms" % (len(ev), span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6))
hist = {}
for g, n in gaps:
    b = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else ">=100us"
    h = hist.setdefault(b, [0, 0]); h[0] += 1; h[1] += g
for b, (c, t) in hist.items():
    print("  gaps %-7s count %6d  total %.2f ms" % (b, c, t / 1e6))
big = {}
for g, n in gaps:
    if g >= 2e4:
        a = big.setdefault(n, [0, 0]); a[0] += 1; a[1] += g
for n, (c, t) in sorted(big.items(), key=lambda x: -x[1][1])[:12]:
    print("  idle after %-50s count %4d total %.2f ms" % (n, c, t / 1e6))
