export TMPDIR=/tmp
for v in "-DC2_F32_U=8 -DC2_F32_OCCDROP=1" "-DC2_F32_U=4 -DC2_F32_OCCDROP=0" "-DC2_F32_U=4 -DC2_F32_OCCDROP=1" "-DC2_F32_U=6 -DC2_F32_OCCDROP=0"; do
  rm -f d3net_amd/build/spconv2.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  echo "=== variant [$v]"
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])):
    if 'spconv_fwd2_kernel' in r['Name'] and 'false>' in r['Name']:
        print('  ',r['Name'][5:42], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3))
PY
done
rm -f d3net_amd/build/spconv2.o*
