#!/bin/bash
OUT=gpurun_out/r04_j15; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_speaker_gpu.py -q -x -k "beam_and_greedy" 2>&1 | tail -3 > $OUT/tests.txt
for CFG in joint listener; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o bench -- python3 bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/rocprof_$CFG.log 2>&1
cp $(find /tmp/prof_$CFG -name "*kernel_stats.csv") $OUT/kernel_stats_$CFG.csv
python tools/step_timeline.py $(find /tmp/prof_$CFG -name "*kernel_trace.csv") > $OUT/timeline_$CFG.txt 2>&1
done
cat $OUT/tests.txt
python - <<'PY'
import csv
for cfg in ('joint','listener'):
    rows=list(csv.DictReader(open('gpurun_out/r04_j15/kernel_stats_%s.csv'%cfg)))
    n=[int(r['Calls']) for r in rows if r['Name'].startswith('adamw_kernel')][0]
    tot=sum(float(r['TotalDurationNs']) for r in rows)/n/1e6; calls=sum(int(r['Calls']) for r in rows)/n
    lib=[r for r in rows if any(k in r['Name'] for k in ('at::native','rocprim','__amd_rocclr','Cijk','void at::','hipcub','MIOpen','miopen'))]
    print(cfg,'steps',n,'kernel ms/step %.2f launches/step %.0f; library: %.2f ms/step %.0f launches'%(tot,calls,sum(float(r['TotalDurationNs']) for r in lib)/n/1e6,sum(int(r['Calls']) for r in lib)/n))
    for r in rows[:16]:
        print('   %-84s %7.1f/step %8.1f us  %.3f ms/step'%(r['Name'][:84],int(r['Calls'])/n,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/n/1e6))
    for r in sorted(lib,key=lambda r:-float(r['TotalDurationNs']))[:8]:
        print('   LIB %-80s %7.1f/step %8.1f us  %.3f ms/step'%(r['Name'][:80],int(r['Calls'])/n,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/n/1e6))
PY
