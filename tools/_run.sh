mkdir -p gpurun_out/ph; export TMPDIR=/tmp
timeout 300 python tools/phase_times.py 10 > gpurun_out/ph/phase.txt 2>&1; cat gpurun_out/ph/phase.txt | tail -16
timeout 300 python tools/hostprof.py > gpurun_out/ph/hostprof.txt 2>&1; head -45 gpurun_out/ph/hostprof.txt
timeout 300 python -m pytest tests/test_pg_ops_gpu.py -x -q 2>&1 | tail -3
