export TMPDIR=/tmp
for v in "-DWG2_PF1=4 -DWG2_PF2=2" "-DWG2_PF1=8 -DWG2_PF2=2" "-DWG2_PF1=8 -DWG2_PF2=4" "-DWG2_PF1=2 -DWG2_PF2=1"; do
  rm -f d3net_amd/build/spconv2.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  echo "=== variant [$v]"
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
t=0
for r in csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])):
    if 'wgrad2_kernel' in r['Name']:
        t+=float(r['TotalDurationNs']); print('  ',r['Name'][5:36], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3))
print('  wgrad total ms/8 steps', t/1e6)
PY
done
rm -f d3net_amd/build/spconv2.o*
