#!/bin/bash
OUT=gpurun_out/r04_j12; mkdir -p $OUT
for CFG in listener joint; do
timeout 300 python bench.py --config $CFG --steps 20 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/$CFG.err | grep '^{' > $OUT/$CFG.json
done
python - <<'PY'
import json
for c in ("listener","joint"):
    d=json.load(open("gpurun_out/r04_j12/%s.json"%c))
    print(c, d["ms_per_step"])
    for s in d["roofline"]["hg_gemm_shapes"]: print("   ", s)
PY
