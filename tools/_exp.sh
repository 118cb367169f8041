set -u
OUT=gpurun_out/r03_z17; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/phase_times.py 10 > $OUT/phases.txt 2>&1
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done > $OUT/bench.txt 2>&1
cat $OUT/phases.txt $OUT/bench.txt
