mkdir -p gpurun_out/r04_j32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 2048 8192; do
D3_BN_FUSED_BIG=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j32_$v -o bench -- python3 bench.py --steps 5 --warmup 2 --settle 5 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j32/bench_$v.log 2>&1
cp $(find /tmp/prof_j32_$v -name "*kernel_stats.csv") gpurun_out/r04_j32/kernel_stats_$v.csv
done
