set -u
OUT=gpurun_out/r03_z19; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_pg_ops_gpu.py tests/test_bench_workload_gpu.py tests/test_fullsize_step_gpu.py tests/test_compat_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -3 > $OUT/pytest.txt
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done > $OUT/bench.txt 2>&1
timeout 300 python tools/phase_times.py 10 > $OUT/phases.txt 2>&1
cat $OUT/pytest.txt $OUT/bench.txt; head -12 $OUT/phases.txt
