#!/usr/bin/env python3
"""In-process A/B of library switches on bench.py's step: one model, one batch, the switch flipped through d3_tuning_set every
`block` steps for `rounds` rounds (interleaved: clock drift and box noise hit both arms alike).
usage: python tools/ab.py <config> <SWITCH>=a,b [<SWITCH2>=a,b ...] [--block 20] [--rounds 6]"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import _lib, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402
from d3net_amd.optim import FusedAdamW  # noqa: E402

args, skip = [], False
for a in sys.argv[1:]:
    if skip:
        skip = False
    elif a in ("--block", "--rounds"):
        skip = True
    elif not a.startswith("--"):
        args.append(a)
config = args[0]
# NAME=a,b flips one switch (py:module.ATTR=a,b a host-side module attribute); NAME1+NAME2=a,b flips several together (arm values per switch: a:b pairs joined by '/', e.g.
# D3_KMAP16+D3_BN_FUSED_ROWS=0/0,1/16384)
switches = []
for a in args[1:]:
    names, vals = a.split("=")
    names = names.split("+")
    arms = [[int(x) for x in v.split("/")] for v in vals.split(",")]
    arms = [arm * len(names) if len(arm) == 1 else arm for arm in arms]
    switches.append((names, arms))
block = int(sys.argv[sys.argv.index("--block") + 1]) if "--block" in sys.argv else 20
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 6
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


from d3net_amd.pointgroup import InputPrefetcher  # noqa: E402
feeder = InputPrefetcher(det, (lambda: [dict(batch), dict(lis)]) if config == "joint" else (lambda: dict(batch)))


def step():
    model.zero_grad(set_to_none=True)
    loss, d = model.training_step(feeder.next())
    loss.backward()
    opt.step()


L = _lib.lib()


def set_switch(nm, x):
    """a library switch (D3_*: d3_tuning_set) or a module attribute of the host side (py:d3net_amd.speaker.CONCURRENT_DECODES)"""
    if nm.startswith("py:"):
        import importlib
        mod, attr = nm[3:].rsplit(".", 1)
        setattr(importlib.import_module(mod), attr, x)
    else:
        assert L.d3_tuning_set(nm.encode(), x) == 0, nm


for _ in range(40):
    step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
for names, arms in switches:
    name = "+".join(names)
    vals = list(range(len(arms)))
    res = {v: [] for v in vals}
    for r in range(rounds):
        for v in (vals if r % 2 == 0 else vals[::-1]):
            for nm, x in zip(names, arms[v]):
                set_switch(nm, x)
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(block):
                step()
            torch.cuda.synchronize()
            res[v].append(1e3 * (time.perf_counter() - t0) / block)
    for nm, x in zip(names, arms[-1]):
        set_switch(nm, x)
    print("%s (%s): " % (name, config) + "   ".join("%s -> %.3f ms (median %.3f, min %.3f)" % ("/".join(map(str, arms[v])), statistics.mean(res[v]), statistics.median(res[v]), min(res[v]))
                                                    for v in vals), flush=True)
