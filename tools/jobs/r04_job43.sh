mkdir -p gpurun_out/r04_j43
O=gpurun_out/r04_j43/info.txt
for d in /sys/class/drm/card*/device; do echo "$d numa $(cat $d/numa_node 2>/dev/null) vendor $(cat $d/vendor 2>/dev/null)" >> $O; done
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" >> $O
cat /proc/self/status | grep -i "cpus_allowed_list" >> $O
nproc >> $O
python - >> $O <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
N0=$(cat /sys/devices/system/node/node0/cpulist 2>/dev/null)
N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 $N0 node1 $N1" >> $O
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],3))" >> gpurun_out/r04_j43/ab.txt
  if [ -n "$N0" ]; then taskset -c $N0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('node0', round(d['ms_per_step'],3))" >> gpurun_out/r04_j43/ab.txt; fi
  if [ -n "$N1" ]; then taskset -c $N1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-ceiling 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('node1', round(d['ms_per_step'],3))" >> gpurun_out/r04_j43/ab.txt; fi
done
