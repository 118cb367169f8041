export TMPDIR=/tmp
mkdir -p gpurun_out
python3 tools/phase_times.py 12 2>&1 | grep " ms" | tr '\n' ';'; echo
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > gpurun_out/phase.log 2>&1
cp $(find /tmp/pp -name "*kernel_stats.csv") gpurun_out/stats.csv
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
