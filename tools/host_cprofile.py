#!/usr/bin/env python3
"""cProfile of the host side of the training step (main thread), 20 steps: top functions by own time."""
import cProfile
import os
import pstats
import sys

import torch

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


for _ in range(6):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(38)
