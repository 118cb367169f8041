#!/bin/bash
OUT=gpurun_out/r04_j13; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hgemm_gpu.py tests/test_listener_gpu.py tests/test_speaker_gpu.py tests/test_rl_gpu.py tests/test_heads_gpu.py tests/test_pipeline_gpu.py tests/test_bench_heads_workload_gpu.py -q -x 2>&1 | tail -5 > $OUT/tests.txt
for CFG in listener joint speaker; do
timeout 600 python tools/ab.py $CFG D3_HG_CLASS_SPLIT=0,1 --rounds 6 --block 10 > $OUT/ab_$CFG.txt 2>&1
done
cat $OUT/tests.txt; grep "ms (" $OUT/ab_*.txt
