mkdir -p gpurun_out/r04_j18
python -m pytest tests/test_rl_gpu.py tests/test_hgemm_gpu.py tests/test_bench_heads_workload_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_j18/tests.txt
for r in 1 2; do
D3_CONCURRENT_DECODES=0 python bench.py --config joint --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq ', d['ms_per_step'])" >> gpurun_out/r04_j18/ab.txt
D3_CONCURRENT_DECODES=1 python bench.py --config joint --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conc', d['ms_per_step'])" >> gpurun_out/r04_j18/ab.txt
done
