#!/usr/bin/env python3
"""What do the deep U-Net levels cost?  Detector step (bench.py's canonical scene, or the 4-scene speaker batch with --four) with
the 7-level backbone and with the backbone cut to 5 / 3 / 2 levels: backbone forward and backward phase times.
usage: python tools/level_cost.py [--four]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3net_amd import pointgroup as PG, synthetic as S  # noqa: E402
from d3net_amd.config import default_conf  # noqa: E402

dev = torch.device("cuda", 0)
four = "--four" in sys.argv
scenes = bench.make_scenes("speaker" if four else "detector", 0)
batch = S.make_batch(scenes, dev)
for nlev in (7, 5, 3, 2):
    cfg = default_conf(bench.CONF["detector"], overrides={"model": {"blocks": list(range(1, nlev + 1))}})
    torch.manual_seed(123)
    model = PG.PointGroup(cfg).to(dev).train()
    model.teacher = True

    def step():
        model.zero_grad(set_to_none=True)
        loss, d = model.training_step(dict(batch))
        PG._mark("step_forward_end")
        loss.backward()
        PG._mark("backward")

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    gpu = collections.OrderedDict()
    steps = 10
    for _ in range(steps):
        PG.PHASES = []
        step()
        marks = PG.PHASES
        PG.PHASES = None
        torch.cuda.synchronize()
        for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
            gpu[b] = gpu.get(b, 0.0) + ea.elapsed_time(eb)
    print("levels %d: backbone_fwd %.3f ms, backward %.3f ms, sum %.3f ms" % (nlev, gpu["backbone_fwd"] / steps, gpu["backward"] / steps,
                                                                            sum(gpu.values()) / steps), flush=True)
    del model
