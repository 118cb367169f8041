#!/bin/bash
OUT=gpurun_out/r04_j8; mkdir -p $OUT
D3_BENCH_CPROFILE=$OUT/cprof_speaker.txt timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/b.json 2> $OUT/b.err
head -90 $OUT/cprof_speaker.txt | cut -c1-170
