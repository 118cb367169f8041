#!/bin/bash
# One GPU session: full -m gpu suite, smoke, bench (with CPU baseline), rocprofv3 kernel stats and PMC traffic.
# usage: tools/gpu_round.sh <tag> [skip-tests]    -> everything lands under gpurun_out/<tag>/ ; every step is bounded.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 > $OUT/smoke.log
fi
timeout 700 python bench.py 2>&1 | grep '^{' > $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv") $OUT/kernel_stats.csv
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${TAG}_$CTR -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$CTR.log 2>&1
  python tools/pmc_summary.py $(find /tmp/pmc_${TAG}_$CTR -name "*counter_collection.csv") > $OUT/pmc_$CTR.csv
done
cat $OUT/pytest_gpu.log $OUT/smoke.log 2>/dev/null; cut -c1-300 $OUT/bench.json; head -6 $OUT/pmc_FETCH_SIZE.csv; head -4 $OUT/pmc_WRITE_SIZE.csv
