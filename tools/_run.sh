D3_DBG=1 D3_TEACHER=0 python3 tools/phase_times.py 4 2>&1 | grep DBG | tail -5
