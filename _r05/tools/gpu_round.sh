#!/bin/bash
# One GPU session: full -m gpu suite, smoke, bench (default = BASELINE's PointGroup+speaker config, with CPU baseline and the
# reference-precision line), rocprofv3 kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and the SQ pass of
# the same command; then bench line + kernel stats of the other three configs.
# usage: tools/gpu_round.sh <tag> [skip-tests] [bench args...]  -> everything lands under gpurun_out/<tag>/ ; every step is bounded.
set -u
ulimit -c 0      # (a faulting process must not fill the box's disk with a multi-GB core file)
TAG=${1:-r03}
SKIP=${2:-}
shift; shift
BARGS="$@"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$SKIP" != "skip-tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 > $OUT/smoke.log
fi
timeout 1200 python bench.py $BARGS 2> $OUT/bench.err | grep '^{' > $OUT/bench.json
cp gpurun_out/bench_detail_speaker.json $OUT/bench_detail.json 2>/dev/null
timeout 300 python bench.py --exact --no-cpu-baseline $BARGS 2> $OUT/bench_exact.err | grep '^{' > $OUT/bench_exact.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-ceiling $BARGS > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv") $OUT/kernel_stats.csv
python tools/step_gaps.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv") 45 > $OUT/step_gaps.txt 2>&1
python tools/step_timeline.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv") > $OUT/step_timeline.txt 2>&1
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${TAG}_$CTR -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-ceiling $BARGS > $OUT/pmc_$CTR.log 2>&1
  python tools/pmc_summary.py $(find /tmp/pmc_${TAG}_$CTR -name "*counter_collection.csv") > $OUT/pmc_$CTR.csv
done
# SQ pass: wave-cycle split + MFMA busy / op counts (8 SQ slots)
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pmc_${TAG}_SQ -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-ceiling $BARGS > $OUT/pmc_SQ.log 2>&1
python tools/pmc_sq_summary.py $OUT/kernel_stats.csv $(find /tmp/pmc_${TAG}_SQ -name "*counter_collection.csv") > $OUT/pmc_sq_summary.txt 2>&1
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json > /dev/null 2> $OUT/pmc_traffic.err
timeout 300 python tools/phase_times.py 10 > $OUT/phases.txt 2>&1
for CFG in detector listener joint; do
  timeout 400 python bench.py --config $CFG --no-fp32 --no-cpu-baseline 2> $OUT/bench_$CFG.err | grep '^{' > $OUT/bench_$CFG.json
  cp gpurun_out/bench_detail_$CFG.json $OUT/bench_detail_$CFG.json 2>/dev/null
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$CFG -o bench -- python3 bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/bench_${CFG}_under_rocprof.log 2>&1
  cp $(find /tmp/prof_${TAG}_$CFG -name "*kernel_stats.csv") $OUT/kernel_stats_$CFG.csv
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${TAG}_${CFG}_$CTR -o pmc -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/pmc_${CFG}_$CTR.log 2>&1
    python tools/pmc_summary.py $(find /tmp/pmc_${TAG}_${CFG}_$CTR -name "*counter_collection.csv") > $OUT/pmc_${CFG}_$CTR.csv
  done
  python tools/pmc_traffic.py $OUT/pmc_${CFG}_FETCH_SIZE.csv $OUT/pmc_${CFG}_WRITE_SIZE.csv $OUT/pmc_traffic_$CFG.json > /dev/null 2>> $OUT/pmc_traffic.err
done
cat $OUT/pytest_gpu.log $OUT/smoke.log 2>/dev/null; tail -3 $OUT/bench.err; cut -c1-300 $OUT/bench.json; cut -c1-200 $OUT/bench_exact.json; head -12 $OUT/kernel_stats.csv | cut -c1-150; head -6 $OUT/pmc_FETCH_SIZE.csv; head -4 $OUT/pmc_WRITE_SIZE.csv; head -14 $OUT/pmc_sq_summary.txt; for CFG in detector listener joint; do cut -c1-160 $OUT/bench_$CFG.json; done
