#!/usr/bin/env python3
"""bf16 / reference-precision HIP path vs the fp32 CPU oracle on N held-out scenes for several training seeds
(the measurement behind tests/test_metric_parity_gpu.py's bound).  usage: python tools/metric_parity_seeds.py [n_val] [seeds...]"""
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
seeds = [int(s) for s in sys.argv[2:]] or [0, 1, 2]
for seed in seeds:
    t0 = time.time()
    r = T.run_parity(dev, n_val=n_val, head_lr=1e-3, seed=seed, verbose=False)
    o = r["oracle"]
    for k in ("bf16", "exact"):
        h = r[k]
        print("seed %d %-5s mAP %.5f vs %.5f = %+.3f %%   CIDEr %.5f vs %.5f = %+.3f %%   captions %s   proposals %d vs %d   (%.0f s)" %
              (seed, k, h["mAP"], o["mAP"], 100 * (h["mAP"] - o["mAP"]) / o["mAP"], h["cider"], o["cider"],
               100 * (h["cider"] - o["cider"]) / o["cider"], h["same_captions"], h["proposals"], o["proposals"], time.time() - t0), flush=True)
