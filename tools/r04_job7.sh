#!/bin/bash
OUT=gpurun_out/r04_j7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py tests/test_fullsize_step_gpu.py tests/test_config0_step_gpu.py -q -x 2>&1 | tail -4 > $OUT/tests.txt
for i in 1 2 3; do
D3_BN_FUSED_BIG=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/spk_b0_$i.err | grep '^{' > $OUT/spk_b0_$i.json
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/spk_b1_$i.err | grep '^{' > $OUT/spk_b1_$i.json
done
D3_BN_FUSED_BIG=0 timeout 300 python bench.py --config detector --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/det_b0.err | grep '^{' > $OUT/det_b0.json
timeout 300 python bench.py --config detector --steps 30 --no-cpu-baseline --no-fp32 2> $OUT/det_b1.err | grep '^{' > $OUT/det_b1.json
cat $OUT/tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_j7/*_b*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms/step %.2f"%d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
