mkdir -p gpurun_out/r04_j40
python -m pytest tests/test_listener_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04_j40/tests.txt
