mkdir -p gpurun_out/r04_j25
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_j25 -o bench -- python3 bench.py --config joint --steps 3 --warmup 2 --settle 5 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j25/bench.log 2>&1
python tools/step_timeline.py $(find /tmp/prof_j25 -name "*kernel_trace.csv") > gpurun_out/r04_j25/timeline_joint.txt 2>&1
python tools/step_gaps.py $(find /tmp/prof_j25 -name "*kernel_trace.csv") 45 > gpurun_out/r04_j25/gaps_joint.txt 2>&1
