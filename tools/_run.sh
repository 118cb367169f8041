export TMPDIR=/tmp
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
python tools/prebackward.py $(find /tmp/pp -name "*kernel_trace.csv") | head -150
