python3 tools/opt_host.py 2>&1 | tail -2
python -m pytest tests/test_heads_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
