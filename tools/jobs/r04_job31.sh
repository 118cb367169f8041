mkdir -p gpurun_out/r04_j31
python tools/hbm_probe.py > gpurun_out/r04_j31/hbm_probe.txt 2>&1
