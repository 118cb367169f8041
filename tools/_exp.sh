set -u
OUT=gpurun_out/r03_z20; mkdir -p $OUT; export TMPDIR=/tmp
cp d3net_amd/lib/libd3hip.so /tmp/libd3hip_base.so
for V in base b2_512_6 b2_512_8 b2_1024_4 b2_256_12 base; do
  if [ "$V" = "base" ]; then cp /tmp/libd3hip_base.so d3net_amd/lib/libd3hip.so; else cp d3net_amd/lib/libd3hip_$V.so d3net_amd/lib/libd3hip.so; fi
  timeout 200 python -m pytest tests/test_pg_ops_gpu.py -q -x -k "bfs or cluster" 2>&1 | tail -1
  timeout 300 python tools/phase_times.py 10 2>/dev/null | grep "cl_bfs\|wall" | tr '\n' ' '; echo " <- $V"
done > $OUT/b2.txt 2>&1
cp /tmp/libd3hip_base.so d3net_amd/lib/libd3hip.so
cat $OUT/b2.txt
