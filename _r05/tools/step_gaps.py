#!/usr/bin/env python3
"""Where the main queue idles inside one step, from a rocprofv3 --kernel-trace CSV: for the last complete steps (AdamW kernel
to AdamW kernel) the kernels of the busiest queue in launch order, with per kernel name the launches, the summed duration and
the summed idle time in FRONT of it (start − end of the previous kernel on that queue).  A chain of dependent 7 µs kernels
shows up as gaps of the same order as the kernels.  usage: step_gaps.py <kernel_trace.csv> [top=40] [side queue id to list]"""
import csv
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    if "multi_tensor_apply" in n:
        return "multi_tensor_apply"
    head = n.split("(")[0].replace("void ", "").replace("at::native::", "")
    return head[:60]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
    if os.environ.get("D3_GAPS_SMALLGRID"):   # kernels that run long on few workgroups (a chip of 256 CUs mostly idle): name, workgroups, us
        seen = {}
        for r in rows[len(rows) // 2:]:
            try:
                wg = max(1, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) \
                    * max(1, int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
            except (KeyError, ValueError):
                continue
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if us >= float(os.environ["D3_GAPS_SMALLGRID"]) and wg < 200:
                k = (short(r["Kernel_Name"]), wg)
                seen[k] = max(seen.get(k, 0), us)
        for (nm, wg), us in sorted(seen.items(), key=lambda x: -x[1]):
            print("small grid: %-56s %5d workgroups %8.1f us" % (nm, wg, us))
    adam = [i for i, e in enumerate(ev) if e[2].startswith("adamw_kernel")]
    if len(adam) < 3:
        print("fewer than 3 steps in the trace")
        return
    a0, a1 = adam[-2], adam[-1]
    seg = ev[a0 + 1:a1 + 1]
    perq = defaultdict(list)
    for e in seg:
        perq[e[3]].append(e)
    mainq = max(perq, key=lambda q: len(perq[q]))
    es = perq[mainq]
    span = es[-1][1] - es[0][0]
    busy = sum(e[1] - e[0] for e in es)
    print("step span %.2f ms; main queue %s: %d kernels, busy %.2f ms, idle %.2f ms" % (span / 1e6, mainq, len(es), busy / 1e6, (span - busy) / 1e6))
    for q, qs in perq.items():
        if q != mainq:
            print("   queue %s: %d kernels busy %.2f ms" % (q, len(qs), sum(e[1] - e[0] for e in qs) / 1e6))
    if len(sys.argv) > 3:   # launch-order listing of one side queue (and what the main queue runs meanwhile)
        q = sys.argv[3]
        t0 = es[0][0]
        for e in perq.get(q, []):
            during = [m for m in es if m[1] > e[0] and m[0] < e[1]]
            dn = defaultdict(int)
            for m in during:
                dn[m[2][:26]] += min(m[1], e[1]) - max(m[0], e[0])
            print("   q%s @%8.3f ms %8.1f us  %-40s | main: %s" % (q, (e[0] - t0) / 1e6, (e[1] - e[0]) / 1e3, e[2][:40],
                                                                 ", ".join("%s %.0f" % (k, v / 1e3) for k, v in sorted(dn.items(), key=lambda x: -x[1])[:3])))
    if os.environ.get("D3_GAPS_DUMP"):   # every main-queue kernel of the step in launch order: start, duration, idle before it
        with open(os.environ["D3_GAPS_DUMP"], "w") as f:
            pe = es[0][0]
            for e in es:
                f.write("%9.3f %7.1f %7.1f  %s\n" % ((e[0] - es[0][0]) / 1e6, (e[1] - e[0]) / 1e3, max(0, e[0] - pe) / 1e3, e[2]))
                pe = max(pe, e[1])
    if os.environ.get("D3_GAPS_DUMP_ALL"):   # every kernel of the step on every queue in start order (queue, start ms, us, name)
        with open(os.environ["D3_GAPS_DUMP_ALL"], "w") as f:
            qend = {}
            for e in seg:
                idle = max(0, e[0] - qend.get(e[3], e[0])) / 1e3
                f.write("q%-2s %9.3f %7.1f %7.1f  %s\n" % (e[3], (e[0] - seg[0][0]) / 1e6, (e[1] - e[0]) / 1e3, idle, e[2]))
                qend[e[3]] = max(qend.get(e[3], 0), e[1])
    agg = defaultdict(lambda: [0, 0, 0])
    prev_end = es[0][0]
    for e in es:
        gap = max(0, e[0] - prev_end)
        a = agg[e[2]]
        a[0] += 1
        a[1] += e[1] - e[0]
        a[2] += gap
        prev_end = max(prev_end, e[1])
    print("%-60s %5s %9s %9s %7s" % ("kernel", "n", "busy ms", "gap ms", "gap/n us"))
    for n, (k, t, g) in sorted(agg.items(), key=lambda x: -(x[1][1] + x[1][2]))[:top]:
        print("%-60s %5d %9.3f %9.3f %7.1f" % (n, k, t / 1e6, g / 1e6, g / k / 1e3))
    # coarse sections in launch order: runs of kernels cut where the idle gap exceeds 30 us (host-induced)
    big = [(max(0, es[i][0] - max(x[1] for x in es[:i])), es[i - 1][2], es[i][2]) for i in range(1, len(es))]
    big = sorted(big, key=lambda x: -x[0])[:12]
    print("largest single gaps (us): " + "; ".join("%.0f %s->%s" % (g / 1e3, a[:22], b[:22]) for g, a, b in big))


if __name__ == "__main__":
    main()
