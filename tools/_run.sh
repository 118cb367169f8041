python3 tools/h2d_rate.py 2>&1 | tail -6
