#!/usr/bin/env python3
"""d3_hgemm on the batched shapes of the captioner's classifier and its gradients (992 rows = 31 steps x 32 captions, hidden 512,
vocabulary 3004), the three operand forms.
usage: python tools/hgemm_bench.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hgemm_gpu import _prob, _seg  # noqa: E402
from d3net_amd import _lib  # noqa: E402
from d3net_amd._lib import GemmProb  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()


def run(p, iters=30):
    arr = (GemmProb * 1)(p)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        L.d3_hgemm(arr, 1, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.d3_hgemm(arr, 1, st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


R, H, V = 992, 512, 3004
x = torch.randn(R, H, device=dev); W = torch.randn(V, H, device=dev); out = torch.empty(R, V, device=dev)
dlog = torch.randn(R, V, device=dev); dx = torch.empty(R, H, device=dev); dW = torch.empty(V, H, device=dev)
cases = [("logits = x W^T          (992 x 3004, K 512)", _prob([_seg(x, W, H)], R, V, out), 2.0 * R * V * H),
         ("dx = dlog W             (992 x 512, K 3004, B k-major)", _prob([_seg(dlog, W, V, b_km=True)], R, H, dx), 2.0 * R * V * H),
         ("dW = dlog^T x           (3004 x 512, K 992, both k-major)", _prob([_seg(dlog, x, R, a_km=True, b_km=True)], V, H, dW), 2.0 * R * V * H)]
for name, p, fl in cases:
    us = run(p)
    print("%-62s %8.1f us  %6.1f TFLOP/s" % (name, us, fl / us / 1e6))
