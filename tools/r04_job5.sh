#!/bin/bash
OUT=gpurun_out/r04_j5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py tests/test_fullsize_step_gpu.py tests/test_config0_step_gpu.py tests/test_bench_workload_gpu.py -q -x 2>&1 | tail -12 > $OUT/tests.txt
for CFG in detector speaker; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o bench -- python3 bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-ceiling > $OUT/rocprof_$CFG.log 2>&1
cp $(find /tmp/prof_$CFG -name "*kernel_stats.csv") $OUT/kernel_stats_$CFG.csv
done
for i in 1 2; do
D3_KMAP16=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/spk_k0_$i.err | grep '^{' > $OUT/det_spk_k0_$i.json
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/spk_k1_$i.err | grep '^{' > $OUT/det_spk_k1_$i.json
D3_BN_FUSED_ROWS=0 timeout 300 python bench.py --config detector --steps 40 --no-cpu-baseline --no-fp32 2> $OUT/det_f0_$i.err | grep '^{' > $OUT/det_f0_$i.json
timeout 300 python bench.py --config detector --steps 40 --no-cpu-baseline --no-fp32 2> $OUT/det_f1_$i.err | grep '^{' > $OUT/det_f1_$i.json
done
cat $OUT/tests.txt
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("gpurun_out/r04_j5/det_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms/step %.2f"%d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
for cfg in ('speaker','detector'):
    rows=list(csv.DictReader(open('gpurun_out/r04_j5/kernel_stats_%s.csv'%cfg)))
    n=[int(r['Calls']) for r in rows if r['Name'].startswith('adamw_kernel')][0]
    tot=sum(float(r['TotalDurationNs']) for r in rows)/n/1e6; calls=sum(int(r['Calls']) for r in rows)/n
    print(cfg,'steps',n,'kernel ms/step %.2f launches/step %.0f'%(tot,calls))
    for r in rows:
        if any(k in r['Name'] for k in ('un_bn',)):
            print('   %-70s %6.1f/step %8.1f us  %.3f ms/step'%(r['Name'][:70],int(r['Calls'])/n,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/n/1e6))
PY
