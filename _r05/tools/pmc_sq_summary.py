#!/usr/bin/env python3
"""Per-kernel SQ counter shares from a rocprofv3 --pmc pass (tools/gpu_round.sh):
  wave-cycle split   WAIT_ANY (parked on s_waitcnt / barrier) / WAIT_INST_ANY (issue stall) / ACTIVE_INST_ANY (issuing)
  MFMA utilisation   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES  (matrix-core busy share of the time a SQ had work: the
                     north_star's "MFMA utilisation vs gfx950 peak"; MI355X_MICROARCH.md: BUSY counts cycles, the wave
                     counters quad-cycles) and the MFMA op counts per launch (MOPS_BF16 / MOPS_F32, 512-flop units)
  MFMA rate          MOPS * 512 flop / the kernel's average duration (kernel_stats.csv of the --kernel-trace --stats run of the same
                     command), as a fraction of the dense peak (bf16 2.5 PFLOP/s, f32 157 TFLOP/s: MI355X_MICROARCH.md)
usage: pmc_sq_summary.py <kernel_stats.csv> <counter_collection.csv> [...] > summary.txt"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = {}
for r in csv.DictReader(open(sys.argv[1], newline="")):
    dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
for path in sys.argv[2:]:
    with open(path, newline="") as f:
        seen = set()
        for r in csv.DictReader(f):
            name = (r.get("Kernel_Name") or "?").split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]] += float(r.get("Counter_Value") or 0)
            key = (name, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key); cnt[name] += 1
print("kernel | launches | wait_any | wait_inst | active | avg us | MFMA bf16 TFLOP/s (% of 2.5 PF) | MFMA f32 TFLOP/s (% of 157 TF)")
for name in sorted(acc, key=lambda n: -acc[n].get("SQ_WAVE_CYCLES", 0)):
    a = acc[name]
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1.0
    busy = a.get("SQ_BUSY_CYCLES", 0) or 1.0
    n = max(cnt[name], 1)
    d = dur.get(name, 0.0)
    tb = a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) / n * 512 / d / 1e3 if d else 0.0     # flop/ns -> TFLOP/s
    tf = a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / n * 512 / d / 1e3 if d else 0.0
    print("%-46s %5d  %5.1f%%  %5.1f%%  %5.1f%%  %8.1f   %8.1f (%4.1f%%)   %8.1f (%4.1f%%)" % (
        name[:46], n, 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        d / 1e3, tb, 100 * tb / 2500.0, tf, 100 * tf / 157.0))
