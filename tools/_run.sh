export TMPDIR=/tmp
python -m pytest tests/test_pg_ops_gpu.py -x -q -m gpu -k "bfs or cluster" 2>&1 | tail -2
python3 tools/phase_times.py 12 2>&1 | grep -E "wall|cl_bfs|clustering" | tr '\n' ';'; echo
python3 tools/phase_times.py 12 2>&1 | grep -E "wall|cl_bfs|clustering" | tr '\n' ';'; echo
python3 tools/step_jitter.py 80
