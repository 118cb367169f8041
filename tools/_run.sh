export TMPDIR=/tmp; mkdir -p gpurun_out/ph
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 6 > /dev/null 2>&1
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])))
steps=10
for r in sorted(rows, key=lambda r:-int(r['TotalDurationNs']))[:14]:
    n=r['Name']
    print("%-50s calls/step %6.1f avg_us %8.1f ms/step %6.3f"%(n[:50], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, int(r['TotalDurationNs'])/steps/1e6))
PY
timeout 300 python -m pytest tests/test_pg_ops_gpu.py -x -q 2>&1 | tail -2
timeout 300 python tools/phase_times.py 10 2>&1 | grep -E "wall|cl_|clustering"
