#!/usr/bin/env python3
"""Which host call sites launch LIBRARY kernels (at::native element-wise / fill / copy, rocprim) inside one training step:
torch.profiler with stacks over one step of bench.py's configuration, device kernels attributed to the innermost d3net_amd frame
of the aten op that launched them.
usage: python tools/aten_sites.py [speaker|detector|listener|joint]"""
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
from d3net_amd.pointgroup import InputPrefetcher  # noqa: E402

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
opt.register_step_pre_hook(lambda *a: det.drop_stale_grads())
batch = S.make_batch(scenes, dev)
lis = None
if config != "detector":
    batch = S.add_language(batch, dev, chunk=chunk, vocab=bench.VOCAB)
    if config in ("speaker", "joint"):
        batch["lang_len"] = batch["spk_lang_len"]
    if config == "joint":
        lis = S.add_language(S.make_batch(scenes, dev), dev, chunk=chunk, vocab=bench.VOCAB, seed=9)
feeder = InputPrefetcher(det, (lambda: [dict(batch), dict(lis)]) if config == "joint" else (lambda: dict(batch)))


def step():
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(feeder.next())
    loss.backward()
    opt.step()


for _ in range(8):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = {"aten::empty.memory_format", "aten::empty_strided", "aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::detach", "aten::slice.Tensor",
        "aten::select.int", "aten::t", "aten::transpose.int", "aten::unsqueeze", "aten::squeeze.dim", "aten::expand", "aten::alias", "aten::reshape",
        "aten::_reshape_alias", "aten::permute", "aten::unbind.int", "aten::split.Tensor", "aten::narrow", "aten::empty_like", "aten::new_empty",
        "aten::is_pinned", "aten::record_stream", "aten::squeeze", "aten::lift_fresh", "aten::view.dtype", "aten::_local_scalar_dense"}
sites = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name()
        if name not in SKIP:
            big = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or "device" in (kwargs or {})
            if big:
                fr = [f for f in traceback.extract_stack() if "/d3net_amd/" in f.filename or f.filename.endswith("bench.py")]
                where = "%s:%d %s" % (os.path.relpath(fr[-1].filename, ROOT), fr[-1].lineno, fr[-1].name) if fr else "(autograd engine / other)"
                sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step()
torch.cuda.synchronize()
print("aten ops on device tensors in one step: %d over %d sites" % (sum(sites.values()), len(sites)))
for (name, where), n in sorted(sites.items(), key=lambda kv: (kv[0][1], -kv[1])):
    print("%3d  %-34s %s" % (n, name, where))
