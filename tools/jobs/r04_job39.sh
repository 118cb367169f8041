mkdir -p gpurun_out/r04_j39
python -m pytest tests/test_rl_gpu.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04_j39/tests.txt
