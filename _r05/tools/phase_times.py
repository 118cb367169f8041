#!/usr/bin/env python3
"""GPU-timeline time per phase of one training step (cuda events on the main stream, no extra syncs).
usage: python tools/phase_times.py [steps] [--config speaker|detector|listener]   (workloads = bench.py's)"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if args else 10
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
det.teacher = os.environ.get("D3_TEACHER", "1") != "0"
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
batch = S.make_batch(scenes, dev)
if config != "detector":
    batch = S.add_language(batch, dev, chunk=cfg.data.num_des_per_scene, vocab=bench.VOCAB)
    if config == "speaker":
        batch["lang_len"] = batch["spk_lang_len"]


def step():
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(dict(batch))
    PG._mark("step_forward_end")
    loss.backward()
    PG._mark("backward")
    opt.step()
    PG._mark("optimizer")


for _ in range(4):
    step()
torch.cuda.synchronize()
gpu = collections.OrderedDict()
t_all = time.perf_counter()
for _ in range(steps):
    PG.PHASES = []
    step()
    marks = PG.PHASES
    PG.PHASES = None
    torch.cuda.synchronize()
    for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
        gpu[b] = gpu.get(b, 0.0) + ea.elapsed_time(eb)
wall = (time.perf_counter() - t_all) / steps * 1e3
print("config %s: wall %.2f ms/step (with a sync per step)" % (config, wall))
for k, v in gpu.items():
    print("  %-22s %7.2f ms" % (k, v / steps))
print("  %-22s %7.2f ms" % ("sum", sum(gpu.values()) / steps))
