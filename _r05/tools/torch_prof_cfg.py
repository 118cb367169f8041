#!/usr/bin/env python3
"""torch.profiler view of one training step of a bench config: which python lines launch the library kernels.
usage: python tools/torch_prof_cfg.py [speaker|detector|listener|joint]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "speaker"
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


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
# aten ops (leaf ops that launch kernels) by the innermost repo frame
c = collections.Counter()
t = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_children:
        continue
    st = [f for f in (e.stack or []) if "/d3net_amd/" in f or "bench.py" in f or "/tools/" in f]
    key = (st[0].split("/root/repo/")[-1].split("repo/")[-1] if st else "?")
    c[key] += 1; t[key] += e.cpu_time_total
print("leaf aten ops by innermost repo frame (count, host us):")
for k, v in c.most_common(60):
    print("%5d %9.0f  %s" % (v, t[k], k))
