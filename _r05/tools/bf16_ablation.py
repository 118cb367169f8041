#!/usr/bin/env python3
"""Where does the bf16 evaluation error come from?    For a model trained with the bf16 step (seed given; the
round-4 outlier is head_lr 1e-3, seed 1: -1.70 % CIDEr@0.5IoU with the bf16 kernels forced onto the evaluation), the held-out mAP /
CIDEr against the fp32 CPU oracle with the bf16 kernels on BOTH sparse U-Nets, on ONE of them (the other on its reference-precision
twin), and on neither.  The proposal-level heads are fp32 in every variant.
usage: python tools/bf16_ablation.py [n_val=128] [seeds...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_metric_parity_gpu as T  # noqa: E402

dev = torch.device("cuda", 0)
n_val = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seeds = [int(s) for s in sys.argv[2:]] or [1]
for seed in seeds:
    t0 = time.time()
    r = T.run_parity(dev, n_val=n_val, head_lr=1e-3, seed=seed, verbose=False, ablate=(("backbone",), ("score_net",)))
    o = r["oracle"]
    for k, h in r.items():
        if k == "oracle":
            continue
        print("seed %d %-26s mAP %.5f vs %.5f = %+.3f %%   CIDEr %.5f vs %.5f = %+.3f %%   captions %s   proposals %d vs %d   (%.0f s)" %
              (seed, k, h["mAP"], o["mAP"], 100 * (h["mAP"] - o["mAP"]) / o["mAP"], h["cider"], o["cider"],
               100 * (h["cider"] - o["cider"]) / o["cider"], h.get("same_captions"), h["proposals"], o["proposals"], time.time() - t0), flush=True)
