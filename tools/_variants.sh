export TMPDIR=/tmp
for v in "" "-DB2_EXP_NOSTORE"; do
  rm -f d3net_amd/build/cluster.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  echo "=== variant [$v]"; python -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])):
    if r['Name'].startswith('cl_bfs2'): print(r['Calls'], r['AverageNs'])
"
done
