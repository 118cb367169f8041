#!/usr/bin/env python3
"""torch.profiler view of one training step: host-side op counts / times and device memcpy events.
usage: python tools/torch_prof.py"""
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

dev = torch.device("cuda", 0)
cfg = default_conf()
torch.manual_seed(123)
model = PG.PointGroup(cfg).to(dev).train()
model.teacher = True
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
occ, sem, inst, _ = S.occupancy_grid()
batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)


def step():
    d = dict(batch)
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(d)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
ev = [e for e in prof.events() if "emcpy" in e.name or "copy_" in e.name]
print("copy-like events:", len(ev))
import collections
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous"):
        st = [f for f in (e.stack or []) if "d3net_amd" in f or "tools/" in f]
        c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(40):
    print(v, k)
