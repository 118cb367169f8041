python bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
python bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], r['kernel'], r['launches_sampled'], round(r['achieved'], 1), round(r['avg_launch_us'], 1), {k: (v['launches_per_step'], round(v['avg_launch_us'], 1)) for k, v in r['other'].items()})"
