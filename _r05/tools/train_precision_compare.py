#!/usr/bin/env python3
"""Does TRAINING with bf16 MFMA operands produce models of the same quality as training at the reference's precision?
The recipe of tests/test_metric_parity_gpu.py (600 steps, 128 held-out scenes) trained twice per seed -- bf16 steps (what bench.py
times) and fp32 steps (minkowski.set_exact) -- from the same initial weights and data; both models are scored by the fp32 CPU oracle
(the evaluation the library ships is fp32 as well).  Training is chaotic (a rounding difference in step 1 is a different model
after 600 steps), so the comparison is statistical: the spread between seeds is the yardstick.
usage: python tools/train_precision_compare.py [n_val=128] [seeds...]"""
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
rows = []
for seed in seeds:
    for name, ex in (("bf16-trained", False), ("fp32-trained", True)):
        t0 = time.time()
        r = T.run_parity(dev, n_val=n_val, head_lr=1e-3, seed=seed, verbose=False, exact_too=False, train_exact=ex)
        o = r["oracle"]
        rows.append((seed, name, o["mAP"], o["cider"]))
        print("seed %d %-13s oracle mAP@0.5 %.4f  CIDEr@0.5IoU %.4f  proposals %d   (%.0f s)" % (seed, name, o["mAP"], o["cider"], o["proposals"], time.time() - t0), flush=True)
for name in ("bf16-trained", "fp32-trained"):
    m = [r[2] for r in rows if r[1] == name]; c = [r[3] for r in rows if r[1] == name]
    print("%-13s mean mAP %.4f (min %.4f max %.4f)   mean CIDEr %.4f (min %.4f max %.4f)" % (name, sum(m) / len(m), min(m), max(m), sum(c) / len(c), min(c), max(c)))
