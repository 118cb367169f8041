export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > gpurun_out/phase.log 2>&1
python3 tools/backward_timeline.py $(find /tmp/pp -name "*kernel_trace.csv") | tail -19 > gpurun_out/bwd_timeline.txt
python3 tools/prebackward.py $(find /tmp/pp -name "*kernel_trace.csv") > gpurun_out/prebackward.txt 2>&1
