mkdir -p gpurun_out/r04_j24
python -m pytest tests/test_rl_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_pipeline_gpu.py tests/test_heads_gpu.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r04_j24/tests.txt
python tools/ab.py joint py:d3net_amd.captioning_loss.LOGP_SUM_TENSOR=0,1 --rounds 8 --block 12 > gpurun_out/r04_j24/ab_joint.txt 2>&1
