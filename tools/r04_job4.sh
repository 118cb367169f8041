#!/bin/bash
OUT=gpurun_out/r04_j4; mkdir -p $OUT
timeout 600 python tools/torch_prof_cfg.py joint > $OUT/torch_prof_joint.txt 2>&1
timeout 600 python tools/torch_prof_cfg.py listener > $OUT/torch_prof_listener.txt 2>&1
timeout 600 python tools/torch_prof_cfg.py speaker > $OUT/torch_prof_speaker.txt 2>&1
for CS in 0 1; do
D3_DIST_WORLD1=1 D3_CHUNK_STREAMS=$CS timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/bench_w1_cs$CS.err | grep '^{' > $OUT/bench_w1_cs$CS.json
D3_DIST_WORLD1=1 D3_CHUNK_STREAMS=$CS timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-fp32 --no-ceiling 2> $OUT/bench_w1_cs${CS}b.err | grep '^{' > $OUT/bench_w1_cs${CS}b.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_j4/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms/step %.2f"%d["ms_per_step"], d["config"].get("grad_sync"))
    except Exception as e: print(f, "failed", e)
PY
tail -70 $OUT/torch_prof_joint.txt
