#!/bin/bash
OUT=gpurun_out/r04_j6; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hgemm_gpu.py tests/test_listener_gpu.py tests/test_speaker_gpu.py tests/test_rl_gpu.py tests/test_heads_gpu.py tests/test_pipeline_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_bench_workload_gpu.py -q -x 2>&1 | tail -12 > $OUT/tests.txt
for CFG in listener joint speaker; do
for X in 0 1; do
D3_HG_BF16X3=$X timeout 300 python bench.py --config $CFG --steps 20 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/${CFG}_x$X.err | grep '^{' > $OUT/${CFG}_x$X.json
done; done
cat $OUT/tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_j6/*_x*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], "ms/step %.2f"%d["ms_per_step"], r["kernel"], "%.3f"%r["frac"], {k:(round(v["avg_launch_us"],1),round(v["launches_per_step"],1)) for k,v in r["per_kernel"].items() if "tiled" in k})
    except Exception as e: print(f, "failed", e)
PY
