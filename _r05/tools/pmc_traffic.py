#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 --pmc summaries of tools/gpu_round.sh (FETCH_SIZE and WRITE_SIZE,
collected in separate passes).  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies
64 B per 128-B request (MI355X_MICROARCH.md, HBM section).  Template instances are aggregated under the kernel's base
name.  usage: pmc_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
inst = defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})     # per template instance (rocprofv3's kernel name)
for path in sys.argv[1:3]:
    for line in list(open(path))[1:]:
        # kernel names contain commas (template arguments): the four numeric / counter fields are the last four
        name, counter, launches, _mean, total = line.rstrip("\n").rsplit(",", 4)
        base = name.split("<")[0].strip('"').replace("void ", "")
        if base in ("spconv_fwd2_c_kernel", "spconv_fwd2_ks_kernel"):      # (entry points of one kernel body: bench.kernel_family)
            base = "spconv_fwd2_kernel"
        full = name.strip('"').replace("void ", "")
        b = inst[full][counter]
        b[0] += int(launches); b[1] += float(total)
        a = acc[base][counter]
        a[0] += int(launches); a[1] += float(total)
out = {}
for k in ("spconv_fwd2_kernel", "spconv_fwd2_split_kernel", "spconv_wgrad3_kernel", "spconv_wgrad2_kernel", "spconv_wgrad2_wide_kernel", "cl_bfs2_kernel",
          "un_bn_apply_kernel", "un_bn_bwd_apply_kernel", "un_bn_fused_small_kernel", "un_bn_bwd_fused_small_kernel", "hg_gemm_kernel", "hg_gemm_tiled_kernel",
          "hg_gemm_tiled3_kernel", "td_gru4_fwd_kernel", "cl_push_kernel", "cl_union_kernel", "bqg_query_kernel", "bq_scan_kernel"):
    if k not in acc:
        continue
    f, w = acc[k]["FETCH_SIZE"], acc[k]["WRITE_SIZE"]
    if not f[0] or not w[0]:
        continue
    fk, wk = f[1] / f[0], w[1] / w[0]
    out[k] = {"launches_sampled": f[0], "fetch_kib_per_launch": fk, "write_kib_per_launch": wk,
              "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_round.sh), bench.py --steps 2 "
                      "--warmup 1 (the bench default workload unless the file name says otherwise); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE tallies 64 B "
                      "per 128-B request: MI355X_MICROARCH.md HBM section)"}
BASES = ("spconv_fwd2_kernel", "spconv_fwd2_split_kernel", "spconv_wgrad3_kernel", "spconv_wgrad2_kernel", "spconv_wgrad2_wide_kernel",
         "hg_gemm_kernel", "hg_gemm_tiled_kernel", "hg_gemm_tiled3_kernel", "td_gru4_fwd_kernel", "cl_bfs2_kernel", "un_bn_apply_kernel", "un_bn_bwd_apply_kernel",
         "un_bn_fused_small_kernel", "un_bn_bwd_fused_small_kernel")
for full, v in inst.items():
    if full in out or full.split("<")[0] not in BASES + ("spconv_fwd2_c_kernel", "spconv_fwd2_ks_kernel"):
        continue
    f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
    if not f[0] or not w[0]:
        continue
    fk, wk = f[1] / f[0], w[1] / w[0]
    out[full] = {"launches_sampled": f[0], "fetch_kib_per_launch": fk, "write_kib_per_launch": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
# stamp: the state of bench.py + csrc this was measured on (bench.py prints traffic_stale when it differs)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out["code_sha"] = bench.code_sha()
except Exception as e:   # (never lose a measurement over the stamp)
    out["code_sha"] = None
    print("no code_sha: %r" % (e,), file=sys.stderr)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out.items() if isinstance(v, dict)}))
