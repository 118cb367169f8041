mkdir -p gpurun_out/r04_j28
python -m pytest tests/test_executor_ops_gpu.py tests/test_bench_workload_gpu.py tests/test_fullsize_step_gpu.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r04_j28/tests.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
D3_ACT_GRAD_BF16=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j28_$v -o bench -- python3 bench.py --steps 5 --warmup 2 --settle 5 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j28/bench_$v.log 2>&1
cp $(find /tmp/prof_j28_$v -name "*kernel_stats.csv") gpurun_out/r04_j28/kernel_stats_$v.csv
done
