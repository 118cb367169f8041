python -m pytest tests/test_sparse_gpu.py tests/test_pipeline_gpu.py tests/test_map_parity_gpu.py -x -q -m gpu 2>&1 | tail -2
python3 tools/step_jitter.py 80
python3 tools/step_jitter.py 80
