mkdir -p gpurun_out/r04_j33
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # tag, extra defines, cap
  export D3_CXX_EXTRA="$2"
  rm -f d3net_amd/build/unet.o d3net_amd/build/unet.o.stamp
  python -c "from d3net_amd import build as b; b.build()" > gpurun_out/r04_j33/build_$1.log 2>&1
  D3_BN_FUSED_BIG=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j33_$1 -o bench -- python3 bench.py --steps 5 --warmup 2 --settle 5 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j33/bench_$1.log 2>&1
  cp $(find /tmp/prof_j33_$1 -name "*kernel_stats.csv") gpurun_out/r04_j33/kernel_stats_$1.csv
}
run t256_c512 "" 1
run t256_c256 "" 256
run t512_c512 "-DUN_FS_T=512" 1
run t512_c256 "-DUN_FS_T=512" 256
