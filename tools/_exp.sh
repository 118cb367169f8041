set -u
OUT=gpurun_out/r03_z7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_fullsize_gpu.py tests/test_sparse_gpu.py tests/test_executor_ops_gpu.py -q -x 2>&1 | tail -4 > $OUT/pytest.txt
timeout 300 python tools/fwd2_bench.py 4 30 > $OUT/fwd2_bench.txt 2>&1
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done > $OUT/bench.txt 2>&1
cat $OUT/pytest.txt $OUT/bench.txt $OUT/fwd2_bench.txt
