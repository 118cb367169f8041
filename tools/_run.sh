export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
