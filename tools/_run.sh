python bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-teacher 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --small 2>&1 | tail -1 | cut -c1-200
