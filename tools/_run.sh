python -m pytest tests/test_pg_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
