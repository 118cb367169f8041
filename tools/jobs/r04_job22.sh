mkdir -p gpurun_out/r04_j22
python -m pytest tests/test_rl_gpu.py tests/test_speaker_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_pipeline_gpu.py tests/test_metric_parity_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_j22/tests.txt
python bench.py --config joint --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 > gpurun_out/r04_j22/bench_joint.json
