python3 tools/step_jitter.py 80
python3 tools/step_jitter.py 80 0.005 off
python3 tools/step_jitter.py 80 0.0002
python3 tools/step_jitter.py 80 0.0002 off
python3 tools/step_jitter.py 80
