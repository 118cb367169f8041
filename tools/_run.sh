export TMPDIR=/tmp
python -m pytest tests/test_heads_gpu.py tests/test_map_parity_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -15
python3 tools/phase_times.py 12 2>&1 | grep -E "wall|pr_|proposals|loss" | tr '\n' ';'; echo
