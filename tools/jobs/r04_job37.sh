mkdir -p gpurun_out/r04_j37
for r in 1 2 3; do
for st in 13 53 1000003; do
for CFG in speaker joint; do
D3_BENCH_PROF_STRIDE=$st python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CFG stride=$st', round(d['ms_per_step'],3))" >> gpurun_out/r04_j37/ab.txt
done
done
done
