#!/bin/bash
OUT=gpurun_out/r04_j9; mkdir -p $OUT
timeout 600 python tools/ab.py speaker D3_BN_FUSED_BIG=0,1 D3_C2_INTERLEAVE=0,1 D3_KMAP16=0,1 D3_BN_FUSED_ROWS=0,16384 > $OUT/ab_speaker.txt 2>&1
timeout 600 python tools/ab.py detector D3_BN_FUSED_BIG=0,1 D3_C2_INTERLEAVE=0,1 D3_KMAP16=0,1 D3_BN_FUSED_ROWS=0,16384 > $OUT/ab_detector.txt 2>&1
D3_BENCH_CPROFILE=$OUT/cprof_speaker.txt timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/b.json 2> $OUT/b.err
grep "ms (" $OUT/ab_speaker.txt $OUT/ab_detector.txt
head -75 $OUT/cprof_speaker.txt | cut -c1-170
