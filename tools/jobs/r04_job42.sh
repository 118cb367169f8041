mkdir -p gpurun_out/r04_j42
python -m pytest tests/test_pg_ops_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r04_j42/tests.txt
run() {
  export D3_CXX_EXTRA="$2"
  rm -f d3net_amd/build/cluster.o d3net_amd/build/cluster.o.stamp
  python -c "from d3net_amd import build as b; b.build()" > gpurun_out/r04_j42/build_$1.log 2>&1
  for r in 1 2; do
  python bench.py --config detector --steps 30 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel']['cl_bfs2_kernel']; print('$1', round(d['ms_per_step'],3), round(k['avg_launch_us'],1))" >> gpurun_out/r04_j42/ab.txt
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel']['cl_bfs2_kernel']; print('$1 speaker', round(d['ms_per_step'],3), round(k['avg_launch_us'],1))" >> gpurun_out/r04_j42/ab.txt
  done
}
run deferred ""
run waiting "-DB2_PREFETCH_WAIT"
run deferred2 ""
