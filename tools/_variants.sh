mkdir -p gpurun_out/cb; export TMPDIR=/tmp; export CONV_BENCH_GROUPS=/tmp/groups.json
for v in "8 4" "4 4" "8 3"; do set -- $v
  rm -f d3net_amd/build/spconv2.o*
  D3_CXX_EXTRA="-DC2_UBIG=$1 -DC2_OCC_SMALL=$2" python -m d3net_amd.build > /dev/null 2>&1
  rm -rf /tmp/cbp; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/cbp -o cb -- python3 tools/conv_bench.py 2 5 > /tmp/cb.txt 2>&1
  echo "=== U=$1 OCC=$2"; python tools/conv_trace.py $(find /tmp/cbp -name "*kernel_trace.csv") /tmp/groups.json x | grep -E "(fwd2|dgrad2) \{" | sed 's/spconv_pack_kernel.: [0-9.]*, //'
done
