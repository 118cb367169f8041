mkdir -p gpurun_out/r04_j16
python -m pytest tests/test_hgemm_gpu.py tests/test_listener_gpu.py tests/test_bench_heads_workload_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_j16/tests.txt
python tools/ab.py joint D3_HG_SPLITK=0,256,1024 --rounds 5 > gpurun_out/r04_j16/ab_joint.txt 2>&1
python tools/ab.py listener D3_HG_SPLITK=0,256,1024 --rounds 5 > gpurun_out/r04_j16/ab_listener.txt 2>&1
python tools/ab.py speaker D3_HG_SPLITK=0,256,1024 --rounds 5 > gpurun_out/r04_j16/ab_speaker.txt 2>&1
