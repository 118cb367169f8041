python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "adamw" 2>&1 | tail -2
