mkdir -p gpurun_out/r04_j41
python -m pytest tests/test_pg_ops_gpu.py tests/test_bench_workload_gpu.py tests/test_fullsize_step_gpu.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r04_j41/tests.txt
