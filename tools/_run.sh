python -m pytest tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do echo "fuse_small=$v"; D3_FUSE_SMALL_BN=$v python3 tools/step_jitter.py 80; done
