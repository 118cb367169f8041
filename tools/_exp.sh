set -u
OUT=gpurun_out/r03_z13; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py tests/test_fullsize_step_gpu.py tests/test_conv_fullsize_gpu.py -q -x 2>&1 | tail -4 > $OUT/pytest.txt
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done > $OUT/bench.txt 2>&1
timeout 300 python tools/fwd2_bench.py 4 30 2>&1 | tail -10 > $OUT/fwd2_bench.txt
cat $OUT/pytest.txt $OUT/bench.txt $OUT/fwd2_bench.txt
