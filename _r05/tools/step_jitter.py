#!/usr/bin/env python3
"""Per-step wall time distribution of the detector training step (a sync per step).
usage: python tools/step_jitter.py [steps] [switchinterval_s] [gc: on|off]"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if len(sys.argv) > 2:
    sys.setswitchinterval(float(sys.argv[2]))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
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


for _ in range(8):
    step()
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "off":
    gc.collect(); gc.freeze(); gc.disable()
ts = []
for _ in range(steps):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
s = sorted(ts)
print("switch %.4f gc %s: min %.2f  p25 %.2f  median %.2f  p75 %.2f  p95 %.2f  max %.2f  mean %.2f" % (
    sys.getswitchinterval(), "off" if not gc.isenabled() else "on", s[0], s[len(s) // 4], s[len(s) // 2], s[3 * len(s) // 4],
    s[int(len(s) * 0.95)], s[-1], sum(s) / len(s)))
