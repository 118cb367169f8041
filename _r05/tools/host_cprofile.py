#!/usr/bin/env python3
"""cProfile of the host side of the training step (main thread), 20 steps of bench.py's workload: top functions by own time and
by cumulative time.  usage: python tools/host_cprofile.py [--config speaker|detector|listener]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

config = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "speaker"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = default_conf(bench.CONF[config])
torch.manual_seed(123)
scenes = bench.make_scenes(config, 0)
if config == "detector":
    model = PG.PointGroup(cfg).to(dev).train()
    det = model
else:
    from d3net_amd.pipeline import PipelineNet
    model = PipelineNet(cfg, bench.make_dataset(len(scenes), cfg.data.num_des_per_scene, False)).to(dev).train()
    det = model.detector
det.teacher = True
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
batch = S.make_batch(scenes, dev)
if config != "detector":
    batch = S.add_language(batch, dev, chunk=cfg.data.num_des_per_scene, vocab=bench.VOCAB)
    if config == "speaker":
        batch["lang_len"] = batch["spk_lang_len"]


def step():
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(dict(batch))
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
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
