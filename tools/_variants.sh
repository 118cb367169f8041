export TMPDIR=/tmp
for v in "-DBQ_NB=4" "-DBQ_NB=8" "-DBQ_NB=2"; do
  rm -f d3net_amd/build/ballquery.o*
  D3_CXX_EXTRA="$v" python -m d3net_amd.build > /dev/null 2>&1
  echo "=== variant [$v]"
  timeout 200 python -m pytest tests/test_pg_ops_gpu.py -x -q -m gpu -k "ballquery" 2>&1 | tail -1
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/phase_times.py 4 > /dev/null 2>&1
  python3 tools/cluster_timeline.py $(find /tmp/pp -name "*kernel_trace.csv") | grep bq_scan
done
rm -f d3net_amd/build/ballquery.o*
