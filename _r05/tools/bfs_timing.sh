#!/bin/bash
mkdir -p gpurun_out/r04_j20
export D3_CXX_EXTRA="-DB2_TIMING"
rm -f d3net_amd/build/cluster.o d3net_amd/build/cluster.o.stamp
python -c "from d3net_amd import build as b; b.build()" > gpurun_out/r04_j20/build.log 2>&1
D3_BFS_DEBUG=1 timeout 300 python - > gpurun_out/r04_j20/out.txt 2> gpurun_out/r04_j20/bfs_timing.txt <<'PY'
import torch, bench
from d3net_amd import synthetic as S
from d3net_amd.config import default_conf
from d3net_amd.pointgroup import PointGroup
dev=torch.device("cuda",0)
cfg=default_conf(bench.CONF["detector"]); torch.manual_seed(123)
m=PointGroup(cfg).to(dev).train(); m.teacher=True
b=S.make_batch(bench.make_scenes("detector",0),dev)
for _ in range(3):
    loss,d=m.training_step(dict(b)); torch.cuda.synchronize()
PY
grep "bfs2 cluster" gpurun_out/r04_j20/bfs_timing.txt | sort -t' ' -k5 -n -r | head -8
