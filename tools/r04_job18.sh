#!/bin/bash
OUT=gpurun_out/r04_j18; mkdir -p $OUT
timeout 300 python tools/host_vs_gpu.py detector > $OUT/hvg.txt 2>&1
timeout 300 python tools/host_vs_gpu.py speaker 1 >> $OUT/hvg.txt 2>&1
timeout 300 python tools/host_vs_gpu.py speaker >> $OUT/hvg.txt 2>&1
grep "per step" $OUT/hvg.txt
