mkdir -p gpurun_out/r04_j30
python tools/ab.py speaker D3_BN_FUSED_BIG=1,1024,2048,4096 --rounds 6 --block 20 > gpurun_out/r04_j30/ab_speaker.txt 2>&1
