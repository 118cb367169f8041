export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_pg_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 tools/step_jitter.py 80
