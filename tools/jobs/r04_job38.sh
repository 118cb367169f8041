mkdir -p gpurun_out/r04_j38
for r in 1 2 3 4 5; do
for arm in default expandable; do
if [ $arm = expandable ]; then export PYTORCH_HIP_ALLOC_CONF=expandable_segments:True; else unset PYTORCH_HIP_ALLOC_CONF; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print('$arm', round(d['ms_per_step'],3), round(r['spconv_fwd2_kernel<1, true, true, 4, false, 27, 2, true>']['avg_launch_us'],1), round(r['cl_bfs2_kernel']['avg_launch_us'],0))" >> gpurun_out/r04_j38/ab.txt
done
done
