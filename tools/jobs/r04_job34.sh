mkdir -p gpurun_out/r04_j34
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
D3_BN_FUSED_BIG=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_j34_$v -o bench -- python3 bench.py --steps 3 --warmup 2 --settle 3 --no-cpu-baseline --no-fp32 --no-ceiling > gpurun_out/r04_j34/bench_$v.log 2>&1
python - "$(find /tmp/prof_j34_$v -name '*kernel_trace.csv')" > gpurun_out/r04_j34/top_$v.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "un_bn" in n or "vectorized_elementwise" in n:
        d[n.split("(")[0][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v.sort(reverse=True)
    print("%-62s n=%5d total %.2f ms  top: %s" % (k, len(v), sum(v)/1e3, " ".join("%.1f"%x for x in v[:12])))
PY
done
