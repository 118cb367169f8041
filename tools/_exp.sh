set -u
OUT=gpurun_out/r03_z10; mkdir -p $OUT; export TMPDIR=/tmp
D3_GRU4=1 timeout 900 python -m pytest tests/test_speaker_gpu.py tests/test_listener_gpu.py -q -x 2>&1 | tail -4 > $OUT/pytest.txt
for CFG in "D3_GRU4=0" "D3_GRU4=1"; do
  for i in 1 2; do
  env $CFG timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CFG', d['value'], d['ms_per_step'])"
  done
done > $OUT/gru4.txt 2>&1
cat $OUT/pytest.txt $OUT/gru4.txt
