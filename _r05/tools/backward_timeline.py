#!/usr/bin/env python3
"""Per-stream GPU occupancy of the backward phase of the detector step from a rocprofv3 --kernel-trace CSV:
backward = [end of ce_fwd_kernel's step-local successor ... first AdamW kernel).  usage: backward_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    if "multi_tensor_apply" in n:
        return "multi_tensor_apply(AdamW)"
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0].replace("void ", "").replace("at::native::", "")[:42]


ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
# step boundaries: AdamW kernels
adam = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] or e[2].startswith("adamw_kernel")]
ce = [i for i, e in enumerate(ev) if e[2].startswith("ce_fwd_kernel")]
steps = []
for c in ce:
    nxt = [a for a in adam if a > c]
    if nxt:
        steps.append((c, nxt[0]))
steps = steps[-3:]
for c, a in steps:
    t0, t1 = ev[c][0], ev[a][0]
    seg = [e for e in ev[c:a] if e[0] >= t0]
    perq = defaultdict(list)
    for e in seg:
        perq[e[3]].append(e)
    print("backward window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
    for q, es in perq.items():
        busy = sum(e[1] - e[0] for e in es)
        print("   queue %-4s kernels %4d busy %.2f ms  first %.2f last-end %.2f" % (q, len(es), busy / 1e6, (es[0][0] - t0) / 1e6, (max(e[1] for e in es) - t0) / 1e6))
    # top kernels in the window
    agg = defaultdict(lambda: [0, 0])
    for e in seg:
        agg[e[2]][0] += 1; agg[e[2]][1] += e[1] - e[0]
    for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
        print("      %-42s x%-4d %.3f ms" % (n, k, t / 1e6))
