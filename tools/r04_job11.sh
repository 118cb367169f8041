#!/bin/bash
OUT=gpurun_out/r04_j11; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_stem -o pmc -- python3 tools/stem_probe.py 4 > $OUT/probe.log 2>&1
python tools/stem_probe.py --parse $(find /tmp/pmc_stem -name "*counter_collection.csv") > $OUT/stem_fetch.txt 2>&1
cat $OUT/stem_fetch.txt; tail -2 $OUT/probe.log
