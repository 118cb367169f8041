export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 30 --warmup 8 2>&1 | tail -1
python bench.py --steps 30 --warmup 8 2>&1 | tail -1
