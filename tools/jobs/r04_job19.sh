mkdir -p gpurun_out/r04_j19
python tools/ab.py joint py:d3net_amd.speaker.CONCURRENT_DECODES=0,1 --rounds 6 --block 15 > gpurun_out/r04_j19/ab_joint.txt 2>&1
