for v in 32768 2048 32768 2048; do echo "side_min_rows=$v"; D3_SIDE_MIN_ROWS=$v python3 tools/step_jitter.py 80; done
