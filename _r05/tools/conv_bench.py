#!/usr/bin/env python3
"""Per-layer micro-benchmark of the sparse convolution kernels on the canonical scene's coordinate levels:
forward / data gradient / weight gradient, first-generation (spconv.hip) vs second-generation (spconv2.hip) kernels,
with a cross-check of the two results.  usage: python tools/conv_bench.py [levels] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3net_amd import minkowski as ME, synthetic as S  # noqa: E402


GROUPS = []   # (label) per marker-separated group of launches, for tools/conv_trace.py
_MARK = None


def timeit(fn, iters, label=""):
    global _MARK
    for _ in range(3):
        fn()
    if _MARK is None:
        _MARK = torch.arange(17, device="cuda")
    _MARK.flip(0)            # separator kernel in the rocprofv3 kernel trace
    GROUPS.append(label)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    _MARK.flip(0)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    occ, sem, inst, _ = S.occupancy_grid()
    scene = S.scene_from_grid(occ, sem, inst)
    batch = S.make_batch([scene], dev)
    cm = ME.CoordinateManager(batch["voxel_locs"].int().contiguous())
    ts = 1
    print("%-26s %8s | %9s %9s | %9s %9s | %9s %9s | %s" % ("layer", "rows", "fwd1", "fwd2", "dgrad1", "dgrad2", "wgrad1", "wgrad2", "maxrel f/d/w"))
    for lev in range(nlev):
        C = 16 * (lev + 1)
        nbr = cm.k3(ts)
        M = nbr.size(0)
        cases = [("L%d k3 %d->%d" % (lev, C, C), nbr, nbr, M, M, 27, C, C, ME.D3_CONV_FLIPK),
                 ("L%d k3 %d->%d" % (lev, 2 * C, C), nbr, nbr, M, M, 27, 2 * C, C, ME.D3_CONV_FLIPK),
                 ("L%d k1 %d->%d" % (lev, 2 * C, C), None, None, M, M, 1, 2 * C, C, 0)]
        if lev + 1 < nlev:
            child, up, Mo = cm.down(ts)
            cases.append(("L%d down %d->%d" % (lev, C, C + 16), child, up, M, Mo, 8, C, C + 16, 0))
            cases.append(("L%d up %d->%d" % (lev + 1, C + 16, C), up, child, Mo, M, 8, C + 16, C, 0))
        for name, tf, tb, Min, Mout, K, Cin, Cout, bfl in cases:
            torch.manual_seed(lev)
            xf = torch.randn(Min, Cin, device=dev)
            xb = xf.to(torch.bfloat16)
            W = torch.randn(K, Cin, Cout, device=dev) * 0.1
            dy = torch.randn(Mout, Cout, device=dev)
            res = {}
            for gen in (1, 2):
                ME._GEN2 = gen == 2
                f = lambda: ME._conv_call(xb, tf, W, Mout, K, Cin, Cout, ME.D3_CONV_XBF16)
                d = lambda: ME._conv_call(dy, tb, W, Min, K, Cout, Cin, bfl | ME.D3_CONV_TRANSW)
                w = lambda: ME._conv_wgrad(xb, tf, tb, dy, W, Mout, bfl, ME.D3_CONV_XBF16)
                res[gen] = (timeit(f, iters, "%s|fwd%d" % (name, gen)), timeit(d, iters, "%s|dgrad%d" % (name, gen)),
                            timeit(w, iters, "%s|wgrad%d" % (name, gen)), f(), d(), w())
            rel = [float((res[1][i] - res[2][i]).abs().max() / (res[1][i].abs().max() + 1e-20)) for i in (3, 4, 5)]
            print("%-26s %8d | %9.1f %9.1f | %9.1f %9.1f | %9.1f %9.1f | %.1e %.1e %.1e" %
                  (name, Mout, res[1][0], res[2][0], res[1][1], res[2][1], res[1][2], res[2][2], rel[0], rel[1], rel[2]))
        ts *= 2
    if os.environ.get("CONV_BENCH_GROUPS"):
        import json
        json.dump({"iters": iters, "groups": GROUPS}, open(os.environ["CONV_BENCH_GROUPS"], "w"))


if __name__ == "__main__":
    main()
