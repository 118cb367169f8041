set -u
OUT=gpurun_out/r03_z22; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_bench_workload_gpu.py tests/test_metric_parity_gpu.py tests/test_heads_gpu.py tests/test_rl_gpu.py tests/test_speaker_gpu.py tests/test_listener_gpu.py tests/test_eval_harness_gpu.py tests/test_bench_multirank_gpu.py -q -x 2>&1 | tail -4 > $OUT/pytest.txt
cat $OUT/pytest.txt
