#!/usr/bin/env python3
"""GPU-timeline time per phase of the detector training step (cuda events on the main stream, no extra syncs) and the
host-side time of the same phases.  usage: python tools/phase_times.py [steps]"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = default_conf()
torch.manual_seed(123)
model = PG.PointGroup(cfg).to(dev).train()
model.teacher = os.environ.get("D3_TEACHER", "1") != "0"
from d3net_amd.optim import FusedAdamW
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
occ, sem, inst, _ = S.occupancy_grid()
batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)


def step():
    d = dict(batch)
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(d)
    loss.backward()
    PG._mark("backward")
    opt.step()
    PG._mark("optimizer")


for _ in range(4):
    step()
torch.cuda.synchronize()
gpu, host = collections.OrderedDict(), collections.OrderedDict()
t_all = time.perf_counter()
for _ in range(steps):
    PG.PHASES = []
    h0 = time.perf_counter()
    step()
    marks = PG.PHASES
    PG.PHASES = None
    torch.cuda.synchronize()
    for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
        gpu[b] = gpu.get(b, 0.0) + ea.elapsed_time(eb)
wall = (time.perf_counter() - t_all) / steps * 1e3
print("wall %.2f ms/step (with a sync per step)" % wall)
for k, v in gpu.items():
    print("  %-22s %7.2f ms" % (k, v / steps))
print("  %-22s %7.2f ms" % ("sum", sum(gpu.values()) / steps))
