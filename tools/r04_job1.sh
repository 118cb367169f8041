#!/bin/bash
OUT=gpurun_out/r04_j1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_fullsize_gpu.py tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py -q -x 2>&1 | tail -5 > $OUT/conv_tests.txt
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-fp32 2> $OUT/bench_il1.err | grep '^{' > $OUT/bench_il1.json
D3_C2_INTERLEAVE=0 timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-fp32 2> $OUT/bench_il0.err | grep '^{' > $OUT/bench_il0.json
timeout 300 python tools/level_cost.py > $OUT/level_cost_1.txt 2>&1
timeout 300 python tools/level_cost.py --four > $OUT/level_cost_4.txt 2>&1
timeout 1500 python -m pytest tests/test_bench_heads_workload_gpu.py -q -s --durations=5 > $OUT/heads.txt 2>&1
tail -5 $OUT/conv_tests.txt; python - <<'PY'
import json
for k in ("il1","il0"):
    try:
        d=json.load(open("gpurun_out/r04_j1/bench_%s.json"%k)); r=d["roofline"]
        print(k, "ms/step %.2f"%d["ms_per_step"], r["kernel"], "frac %.3f"%r["frac"], "us %.1f"%r["avg_launch_us"])
        for n,v in list(r["per_kernel"].items())[:10]: print("   %-60s %6.1f us x %5.1f  frac %.3f"%(n, v["avg_launch_us"], v["launches_per_step"], v["frac"]))
    except Exception as e: print(k, "failed", e)
PY
cat $OUT/level_cost_1.txt $OUT/level_cost_4.txt | grep levels; tail -25 $OUT/heads.txt
