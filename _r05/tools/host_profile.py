#!/usr/bin/env python3
"""Where does the HOST spend a step?  cProfile over bench.py's step (after warm-up), top functions by own time and by cumulative
time.  The step is host-bound wherever the device waits for launches (tools/step_gaps.py): this names the interpreter-side cost.
usage: python tools/host_profile.py [config=speaker] [steps=30]"""
import cProfile
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "speaker"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
cfg = default_conf(bench.CONF[config])
torch.manual_seed(123)
scenes = bench.make_scenes(config, 0)
chunk = cfg.data.num_des_per_scene
if config == "detector":
    from d3net_amd.pointgroup import PointGroup
    model = PointGroup(cfg).to(dev).train(); det = model
else:
    from d3net_amd.pipeline import PipelineNet
    model = PipelineNet(cfg, bench.make_dataset(len(scenes), chunk, config == "joint")).to(dev).train(); det = model.detector
det.teacher = True
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
opt.register_step_pre_hook(lambda *a: det.drop_stale_grads())
batch = S.make_batch(scenes, dev)
lis = None
if config != "detector":
    batch = S.add_language(batch, dev, chunk=chunk, vocab=bench.VOCAB)
    if config in ("speaker", "joint"):
        batch["lang_len"] = batch["spk_lang_len"]
    if config == "joint":
        lis = S.add_language(S.make_batch(scenes, dev), dev, chunk=chunk, vocab=bench.VOCAB, seed=9)


def step():
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step([dict(batch), dict(lis)] if config == "joint" else dict(batch))
    loss.backward()
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    txt = buf.getvalue()
    print("==== by %s (all numbers are totals over %d steps) ====" % (key, steps))
    print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)
