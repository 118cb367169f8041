#!/usr/bin/env python3
"""Operating-point sweep for tests/test_metric_parity_gpu.py: trains at several noise levels / step counts and prints the held-out
detector diagnostics (no oracle).  usage: python tools/metric_parity_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_metric_parity_gpu as T  # noqa: E402

dev = torch.device("cuda", 0)
for sigma, steps in ((1.0, 600), (2.0, 600), (3.0, 600), (4.0, 600)):
    print("==== sigma %.1f, %d steps" % (sigma, steps), flush=True)
    r = T.run_parity(dev, sigma=sigma, steps=steps, exact_too=False, with_oracle=False)
    print("bf16: mAP %.4f CIDEr %.4f proposals %d" % (r["bf16"]["mAP"], r["bf16"]["cider"], r["bf16"]["proposals"]), flush=True)
