#!/usr/bin/env python3
"""What does THIS box's HBM deliver to plain streaming kernels?  Library copy / fill / sum over 50 ... 800 MB buffers, best of 20,
in GB/s of bytes moved (copy counts read + write).  The yardstick next to the executor's streaming kernels (BatchNorm apply passes:
~3.4 TB/s) and MI355X_MICROARCH.md's 8 TB/s peak.  usage: python tools/hbm_probe.py"""
import torch

dev = torch.device("cuda", 0)
E = lambda: torch.cuda.Event(enable_timing=True)


def best(fn, n=20):
    for _ in range(3):
        fn()
    t = []
    for _ in range(n):
        a, b = E(), E()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b) * 1e-3)
    return min(t)


for mb in (50, 100, 200, 400, 800):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    h = torch.empty(n, dtype=torch.bfloat16, device=dev)
    tc = best(lambda: y.copy_(x))
    tf = best(lambda: y.fill_(1.0))
    ts = best(lambda: x.sum())
    tm = best(lambda: torch.add(x, y, out=y))          # 2 reads + 1 write
    tb = best(lambda: h.copy_(x))                        # 4 B read + 2 B write (the BatchNorm apply's shape)
    print("%4d MB: copy %.0f GB/s (%.1f us)   fill %.0f   sum %.0f   add(2r+1w) %.0f   f32->bf16 %.0f" % (
        mb, 2 * n * 4 / tc / 1e9, tc * 1e6, n * 4 / tf / 1e9, n * 4 / ts / 1e9, 3 * n * 4 / tm / 1e9, n * 6 / tb / 1e9))
