#!/usr/bin/env python3
"""Compressed timeline of ONE training step from a rocprofv3 --kernel-trace CSV: the kernels of the busiest queue between two
AdamW launches, merged into runs (consecutive launches whose names share a prefix class), one line per run:
start offset (ms), launches, busy us, idle us in front of / inside the run, name(s).  Side queues are summarised per 1 ms window.
usage: step_timeline.py <kernel_trace.csv> [min_run_us=0]"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    if "multi_tensor_apply" in n:
        return "multi_tensor_apply"
    return n.split("(")[0].replace("void ", "").replace("at::native::", "")[:70]


def klass(n):
    for p in ("spconv_fwd2_split", "spconv_fwd2", "spconv_wgrad", "un_bn_bwd", "un_bn", "un_", "cm_", "cmp_", "vi_", "voxelize", "bqg_", "bq_", "cl_", "cp_",
              "hg_", "td_", "gm_", "edgeconv", "attn_", "ln_", "cap_", "pth_", "seg_", "sec_", "roipool", "stb_", "rocprim", "__amd_rocclr", "xe_"):
        if n.startswith(p):
            return p
    if "elementwise" in n or "reduce_kernel" in n or "Fill" in n:
        return "torch"
    return n[:12]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
adam = [i for i, e in enumerate(ev) if e[2].startswith("adamw_kernel")]
if len(adam) < 3:
    sys.exit("need at least three optimizer launches in the trace")
a, b = adam[-2], adam[-1]
seg = ev[a + 1:b + 1]
t0 = seg[0][0]
perq = defaultdict(list)
for e in seg:
    perq[e[3]].append(e)
main = max(perq, key=lambda q: len(perq[q]))
print("step %.2f ms; main queue %s: %d launches; other queues: %s" % ((seg[-1][1] - t0) / 1e6, main, len(perq[main]),
      ", ".join("%s: %d launches %.2f ms busy" % (q, len(v), sum(x[1] - x[0] for x in v) / 1e6) for q, v in perq.items() if q != main)))
runs = []
prev_end = None
for e in perq[main]:
    gap = 0 if prev_end is None else max(0, e[0] - prev_end)
    k = klass(e[2])
    if runs and runs[-1]["k"] == k and gap < 30000:
        r = runs[-1]
        r["n"] += 1; r["busy"] += e[1] - e[0]; r["gap_in"] += gap; r["end"] = e[1]; r["names"][e[2]] += 1
    else:
        runs.append(dict(k=k, start=e[0], end=e[1], n=1, busy=e[1] - e[0], gap_front=gap, gap_in=0, names=defaultdict(int)))
        runs[-1]["names"][e[2]] += 1
    prev_end = max(prev_end or 0, e[1])
print("%8s %5s %9s %9s %9s  %s" % ("t0 ms", "n", "busy us", "front us", "inside us", "kernels"))
for r in runs:
    names = ", ".join("%s x%d" % (n[:46], c) for n, c in sorted(r["names"].items(), key=lambda x: -x[1])[:3])
    print("%8.3f %5d %9.1f %9.1f %9.1f  %s" % ((r["start"] - t0) / 1e6, r["n"], r["busy"] / 1e3, r["gap_front"] / 1e3, r["gap_in"] / 1e3, names))
tot_busy = sum(r["busy"] for r in runs); tot_gap = sum(r["gap_front"] + r["gap_in"] for r in runs)
print("main queue busy %.2f ms, idle %.2f ms" % (tot_busy / 1e6, tot_gap / 1e6))
for q, v in perq.items():
    if q == main:
        continue
    print("queue %s:" % q)
    w = defaultdict(lambda: [0, 0.0, defaultdict(int)])
    for e in v:
        i = int((e[0] - t0) / 1e6)
        w[i][0] += 1; w[i][1] += (e[1] - e[0]) / 1e3; w[i][2][klass(e[2])] += 1
    for i in sorted(w):
        print("   %2d-%2d ms: %4d launches %8.1f us busy  %s" % (i, i + 1, w[i][0], w[i][1], dict(w[i][2])))
