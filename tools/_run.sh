python bench.py --steps 20 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-210
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-210
