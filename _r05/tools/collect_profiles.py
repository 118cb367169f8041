#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_round.sh session (gpurun_out/<tag>/) into profiles/ as <name>_* and merge its PMC traffic
(speaker config) into profiles/pmc_traffic.json.  usage: tools/collect_profiles.py <tag> <name>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
keep = {"bench.json": "bench.json", "bench_exact.json": "bench_exact.json", "kernel_stats.csv": "kernel_stats.csv",
        "pmc_FETCH_SIZE.csv": "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv": "pmc_WRITE_SIZE.csv", "pmc_sq_summary.txt": "pmc_sq_summary.txt",
        "pmc_traffic.json": "pmc_traffic.json", "step_gaps.txt": "step_gaps_speaker.txt", "phases.txt": "phases_speaker.txt",
        "pytest_gpu.log": "pytest_gpu.log", "smoke.log": "smoke.log"}
keep["step_timeline.txt"] = "step_timeline_speaker.txt"
for c in ("detector", "listener", "joint"):
    keep["bench_%s.json" % c] = "bench_%s.json" % c
    keep["kernel_stats_%s.csv" % c] = "kernel_stats_%s.csv" % c
    keep["pmc_%s_FETCH_SIZE.csv" % c] = "pmc_%s_FETCH_SIZE.csv" % c
    keep["pmc_%s_WRITE_SIZE.csv" % c] = "pmc_%s_WRITE_SIZE.csv" % c
for f, t in keep.items():
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (name, t)))
        print("profiles/%s_%s" % (name, t))
pt = os.path.join(src, "pmc_traffic.json")
if os.path.exists(pt):
    cur_path = os.path.join(dst, "pmc_traffic.json")
    cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
    cur["speaker"] = json.load(open(pt))
    cur["speaker"]["measured_by"] = "%s (tools/gpu_round.sh)" % name
    for c in ("detector", "listener", "joint"):
        pc = os.path.join(src, "pmc_traffic_%s.json" % c)
        if os.path.exists(pc) and os.path.getsize(pc) > 2:
            cur[c] = json.load(open(pc))
            cur[c]["measured_by"] = "%s (tools/gpu_round.sh, --config %s)" % (name, c)
    json.dump(cur, open(cur_path, "w"), indent=1)
    print("merged traffic of", [k for k in cur if isinstance(cur[k], dict)], "code_sha", cur["speaker"].get("code_sha"))
