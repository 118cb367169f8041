mkdir -p gpurun_out/r04_j21
python -m pytest tests/test_rl_gpu.py tests/test_speaker_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_j21/tests.txt
python tools/ab.py joint py:d3net_amd.speaker.JOINED_DECODES=0,1 --rounds 8 --block 12 > gpurun_out/r04_j21/ab_joint.txt 2>&1
