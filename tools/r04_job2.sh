#!/bin/bash
OUT=gpurun_out/r04_j2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py tests/test_fullsize_step_gpu.py tests/test_config0_step_gpu.py -q -x 2>&1 | tail -8 > $OUT/tests.txt
for i in 1 2; do
D3_BN_FUSED_ROWS=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/bench_f0_$i.err | grep '^{' > $OUT/bench_f0_$i.json
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/bench_f1_$i.err | grep '^{' > $OUT/bench_f1_$i.json
D3_C2_INTERLEAVE=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/bench_i0_$i.err | grep '^{' > $OUT/bench_i0_$i.json
done
D3_BN_FUSED_ROWS=0 timeout 300 python bench.py --config detector --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/bench_det_f0.err | grep '^{' > $OUT/bench_det_f0.json
timeout 300 python bench.py --config detector --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/bench_det_f1.err | grep '^{' > $OUT/bench_det_f1.json
timeout 300 python tools/level_cost.py > $OUT/level_cost_1.txt 2>&1
timeout 300 python tools/torch_prof.py > $OUT/torch_prof.txt 2>&1
cat $OUT/tests.txt; python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_j2/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms/step %.2f"%d["ms_per_step"], "value %.1f"%d["value"])
    except Exception as e: print(f, "failed", e)
PY
grep levels $OUT/level_cost_1.txt; tail -60 $OUT/torch_prof.txt
