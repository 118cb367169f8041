#!/bin/bash
OUT=gpurun_out/r04_j14; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_speaker_gpu.py tests/test_rl_gpu.py tests/test_pipeline_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_eval_harness_gpu.py -q -x 2>&1 | tail -8 > $OUT/tests.txt
timeout 300 python bench.py --config joint --steps 20 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/joint.err | grep '^{' > $OUT/joint.json
cat $OUT/tests.txt; python -c "
import json; d=json.load(open('gpurun_out/r04_j14/joint.json')); print('joint ms/step', d['ms_per_step'])"
