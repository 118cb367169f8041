mkdir -p gpurun_out/r04_j26
python -m pytest tests/test_executor_ops_gpu.py tests/test_bench_workload_gpu.py tests/test_fullsize_step_gpu.py tests/test_unet_gpu.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r04_j26/tests.txt
for r in 1 2 3; do
for v in 0 1; do
D3_ACT_GRAD_BF16=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('gabf=$v', round(d['ms_per_step'],3), d['final_loss'], r['per_kernel'].get('spconv_fwd2_kernel<1, true, true, 4, false, 27, 2, true>',{}).get('avg_launch_us'))" >> gpurun_out/r04_j26/ab.txt
done
done
