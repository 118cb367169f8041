export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -2
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > gpurun_out/phase.log 2>&1
cp $(find /tmp/pp -name "*kernel_stats.csv") gpurun_out/stats.csv
python3 tools/step_jitter.py 80
