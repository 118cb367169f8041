export TMPDIR=/tmp
for v in "-DWG2_T=16" "-DWG2_T=32"; do
  rm -f d3net_amd/build/spconv2.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  echo "=== variant [$v]"
  timeout 300 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "wgrad or executor or native" 2>&1 | tail -1
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
t=0
for r in csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])):
    if 'wgrad2_kernel' in r['Name']:
        t+=float(r['TotalDurationNs']); print('  ',r['Name'][5:36], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3))
print('  wgrad total ms/8 steps', t/1e6)
PY
  python3 tools/phase_times.py 12 2>&1 | grep -E "wall|backward" | tr '\n' ';'; echo
done
rm -f d3net_amd/build/spconv2.o*
