export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_heads_gpu.py tests/test_pg_ops_gpu.py tests/test_pipeline_gpu.py tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
