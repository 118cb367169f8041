export TMPDIR=/tmp
for v in "-DB2_TIMING" "-DB2_HGRID=16" "-DB2_HGRID=32" "-DB2_HGRID=16 -DB2_EPT=2" "-DB2_HGRID=16 -DB2_EPT=4"; do
  rm -f d3net_amd/build/cluster.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  echo "=== variant [$v]"
  if [ "$v" = "-DB2_TIMING" ]; then
    D3_BFS_DEBUG=1 timeout 300 python3 tools/phase_times.py 2 2>&1 | grep "bfs2 cluster" | sort -k7 -n -r | head -3
    continue
  fi
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  python -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0])):
    if r['Name'].startswith('cl_bfs2'): print(r['Calls'], r['AverageNs'])
"
  timeout 200 python -m pytest tests/test_pg_ops_gpu.py -x -q -k "bfs or cluster" 2>&1 | tail -1
done
rm -f d3net_amd/build/cluster.o*
