#!/usr/bin/env python3
"""Is a phase of the step bound by the host (enqueue time) or by the device?  For bench.py's step: host wall time of the forward
call, of loss.backward() and of the optimizer (time until the call returns = everything enqueued), next to the device time of the
same phases (events on the current stream).  usage: python tools/host_vs_gpu.py [detector|speaker] [scenes]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "detector"
nsc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
cfg = default_conf(bench.CONF[config])
torch.manual_seed(123)
scenes = bench.make_scenes(config, 0)
if nsc:
    scenes = scenes[:nsc]
chunk = cfg.data.num_des_per_scene
if config == "detector":
    from d3net_amd.pointgroup import PointGroup
    model = PointGroup(cfg).to(dev).train(); det = model
else:
    from d3net_amd.pipeline import PipelineNet
    model = PipelineNet(cfg, bench.make_dataset(len(scenes), chunk, False)).to(dev).train(); det = model.detector
det.teacher = True
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=0.002)
batch = S.make_batch(scenes, dev)
if config != "detector":
    batch = S.add_language(batch, dev, chunk=chunk, vocab=bench.VOCAB)
    batch["lang_len"] = batch["spk_lang_len"]
E = lambda: torch.cuda.Event(enable_timing=True)
acc = {"fwd_host": 0, "bwd_host": 0, "opt_host": 0, "fwd_gpu": 0, "bwd_gpu": 0, "opt_gpu": 0, "step_wall": 0}
def step(measure):
    e = [E() for _ in range(4)]
    model.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); e[0].record()
    loss, d = model.training_step(dict(batch))
    t1 = time.perf_counter(); e[1].record()
    loss.backward()
    t2 = time.perf_counter(); e[2].record()
    opt.step()
    t3 = time.perf_counter(); e[3].record()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if measure:
        acc["fwd_host"] += t1 - t0; acc["bwd_host"] += t2 - t1; acc["opt_host"] += t3 - t2; acc["step_wall"] += t4 - t0
        acc["fwd_gpu"] += e[0].elapsed_time(e[1]) / 1e3; acc["bwd_gpu"] += e[1].elapsed_time(e[2]) / 1e3; acc["opt_gpu"] += e[2].elapsed_time(e[3]) / 1e3
for _ in range(30):
    step(False)
n = 30
for _ in range(n):
    step(True)
print("%s, %d scene(s): per step (ms)  " % (config, len(scenes)) + "  ".join("%s %.2f" % (k, 1e3 * v / n) for k, v in acc.items()))
