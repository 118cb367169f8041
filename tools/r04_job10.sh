#!/bin/bash
OUT=gpurun_out/r04_j10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_stem -o pmc -- python3 tools/stem_probe.py 4 > $OUT/probe.log 2>&1
python tools/stem_probe.py --parse $(find /tmp/pmc_stem -name "*counter_collection.csv") > $OUT/stem_fetch.txt 2>&1
timeout 900 python tools/ab.py speaker "D3_KMAP16+D3_BN_FUSED_ROWS+D3_BN_FUSED_BIG+D3_C2_INTERLEAVE=0/0/0/0,1/16384/1/1" --rounds 10 --block 20 > $OUT/ab_speaker.txt 2>&1
timeout 900 python tools/ab.py detector "D3_KMAP16+D3_BN_FUSED_ROWS+D3_BN_FUSED_BIG+D3_C2_INTERLEAVE=0/0/0/0,1/16384/1/1" --rounds 10 --block 20 > $OUT/ab_detector.txt 2>&1
grep "ms (" $OUT/ab_speaker.txt $OUT/ab_detector.txt; tail -3 $OUT/ab_speaker.txt
cat $OUT/stem_fetch.txt; tail -2 $OUT/probe.log
