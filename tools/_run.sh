export TMPDIR=/tmp; mkdir -p gpurun_out/ph
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_trace.csv"); head -2 $f | cut -c1-600
python tools/backward_timeline.py $f
