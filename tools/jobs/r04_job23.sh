mkdir -p gpurun_out/r04_j23
python tools/host_profile.py speaker 30 > gpurun_out/r04_j23/host_speaker.txt 2>&1
python tools/host_profile.py joint 15 > gpurun_out/r04_j23/host_joint.txt 2>&1
python -m pytest tests/test_speaker_gpu.py tests/test_pipeline_gpu.py tests/test_bench_workload_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r04_j23/tests.txt
