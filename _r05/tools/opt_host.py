#!/usr/bin/env python3
"""Host time of the optimizer step (no profiler): usage python tools/opt_host.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3net_amd import pointgroup as PG, synthetic as S
from d3net_amd.config import default_conf
from d3net_amd.optim import FusedAdamW
dev = torch.device("cuda", 0)
cfg = default_conf(); torch.manual_seed(123)
model = PG.PointGroup(cfg).to(dev).train(); model.teacher = True
params = [p for p in model.parameters() if p.requires_grad]
for name, opt in (("FusedAdamW", FusedAdamW(params, lr=0.002)), ("torch fused", torch.optim.AdamW(params, lr=0.002, fused=True))):
    occ, sem, inst, _ = S.occupancy_grid((100, 75, 50), 4, (8, 30), (8, 25), 0)
    batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)
    th = []
    for it in range(12):
        d = dict(batch); model.zero_grad(set_to_none=True)
        loss, d = model.training_step(d); loss.backward(); torch.cuda.synchronize()
        t0 = time.perf_counter(); opt.step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        th.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    th = th[4:]
    print("%-12s host %.3f ms  host+gpu %.3f ms" % (name, sum(a for a, _ in th) / len(th), sum(b for _, b in th) / len(th)))
