#!/bin/bash
OUT=gpurun_out/r04_j17; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_metric_parity_gpu.py tests/test_map_parity_gpu.py tests/test_eval_harness_gpu.py tests/test_pipeline_gpu.py tests/test_config0_step_gpu.py -q -x -s --durations=6 2>&1 | grep -v "^$" | tail -30 > $OUT/tests.txt
cat $OUT/tests.txt
