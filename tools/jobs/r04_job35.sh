mkdir -p gpurun_out/r04_j35
timeout 900 python -m pytest tests/test_executor_ops_gpu.py tests/test_bench_workload_gpu.py tests/test_fullsize_step_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r04_j35/tests.txt
timeout 600 python tools/ab.py speaker D3_BN_BCAST_MIN_WG=0,64 --rounds 6 --block 20 > gpurun_out/r04_j35/ab_speaker.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 64; do
D3_BN_BCAST_MIN_WG=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j35_$v -o bench -- python3 bench.py --steps 5 --warmup 2 --settle 5 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j35/bench_$v.log 2>&1
cp $(find /tmp/prof_j35_$v -name "*kernel_stats.csv") gpurun_out/r04_j35/kernel_stats_$v.csv
done
