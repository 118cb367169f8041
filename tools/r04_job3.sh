#!/bin/bash
OUT=gpurun_out/r04_j3; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for CFG in speaker detector; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o bench -- python3 bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/rocprof_$CFG.log 2>&1
cp $(find /tmp/prof_$CFG -name "*kernel_stats.csv") $OUT/kernel_stats_$CFG.csv
python tools/step_timeline.py $(find /tmp/prof_$CFG -name "*kernel_trace.csv") > $OUT/timeline_$CFG.txt 2>&1
python tools/step_gaps.py $(find /tmp/prof_$CFG -name "*kernel_trace.csv") 30 > $OUT/gaps_$CFG.txt 2>&1
done
timeout 300 python bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo; head -5 $OUT/timeline_speaker.txt
