mkdir -p gpurun_out/cb; export TMPDIR=/tmp; export CONV_BENCH_GROUPS=/tmp/groups.json
for v in "512 8" "1024 16" "2048 32"; do set -- $v
  rm -f d3net_amd/build/spconv2.o*
  D3_CXX_EXTRA="-DWG2_TARGET_WGS=$1 -DWG2_PART_MB=$2" python -m d3net_amd.build > /dev/null 2>&1
  rm -rf /tmp/cbp; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/cbp -o cb -- python3 tools/conv_bench.py 4 5 > /tmp/cb.txt 2>&1
  echo "=== TARGET=$1 MB=$2"; python tools/conv_trace.py $(find /tmp/cbp -name "*kernel_trace.csv") /tmp/groups.json | awk -F'|' '{print $1 "|" $4}'
done
