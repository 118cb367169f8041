#!/usr/bin/env python3
"""A/B micro-benchmark of `spconv_fwd2_kernel` (csrc/spconv2.hip) on the coordinate levels of bench.py's 4-scene batch:
pre-packed weights, bf16 / fp32 gathers, 4-wave workgroups vs 16 waves around one LDS copy of the weights
(switch D3_C2_NW16_KB flipped through d3_tuning_set), results cross-checked.
usage: python tools/fwd2_bench.py [scenes=4] [iters=30]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3net_amd import _lib, minkowski as ME, synthetic as S  # noqa: E402
from d3net_amd.pointgroup_ops import _ptr, _stream  # noqa: E402

XBF16 = 32


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    scenes = []
    for b in range(nsc):   # bench.py make_scenes("speaker")
        occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=b)
        scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + b, feat_seed=2 + b))
    batch = S.make_batch(scenes, dev)
    cm = ME.CoordinateManager(batch["voxel_locs"].int().contiguous())
    shapes = {0: [(136, 16), (16, 16), (32, 16)], 1: [(32, 32), (64, 32), (32, 64)], 2: [(48, 48), (96, 48), (48, 96)]}
    modes = (("nw4", 1 << 20), ("nw16", 24))
    print("%-18s %8s %8s | %-28s | %9s %9s | %9s %9s | maxrel" % ("layer", "rows", "W KB", "plan nw4 -> nw16 (waves, grid, wlds)", "bf16 nw4", "bf16 nw16",
                                                                   "fp32 nw4", "fp32 nw16"))
    ts = 1
    for lev in range(3):
        nbr = cm.k3(ts)
        M = nbr.size(0)
        for cin, cout in shapes[lev]:
            torch.manual_seed(lev)
            xf = torch.randn(M, cin, device=dev)
            xb = xf.to(torch.bfloat16)
            W = (torch.randn(27, cin, cout, device=dev) * 0.1).contiguous()
            wp = torch.empty(L.d3_spconv_pack_bytes(27, cin, cout), dtype=torch.uint8, device=dev)
            assert L.d3_spconv_pack(_ptr(W), _ptr(wp), 27, cin, cout, 0, _stream()) == 0
            out = torch.empty(M, cout, device=dev)

            def run(x, fl):
                rc = L.d3_spconv_fwd2(_ptr(x), x.stride(0), _ptr(nbr), _ptr(wp), _ptr(out), cout, None, 0, None, M, M, 27, cin, cout, fl, _stream())
                assert rc == 0, rc
            res, plans, outs = {}, [], {}
            for name, kb in modes:
                assert L.d3_tuning_set(b"D3_C2_NW16_KB", kb) == 0
                p = (C.c_int * 6)()
                assert L.d3_spconv_fwd2_plan(M, 27, cin, cout, p) == 0
                plans.append("%d/%d/%d" % (p[1], p[2], p[3]))
                res[name, "bf16"] = timeit(lambda: run(xb, XBF16), iters)
                outs[name] = out.clone()
                res[name, "fp32"] = timeit(lambda: run(xf, 0), iters)
            assert L.d3_tuning_set(b"D3_C2_NW16_KB", 24) == 0
            rel = float((outs["nw4"] - outs["nw16"]).abs().max() / outs["nw4"].abs().max())
            print("%-18s %8d %8.1f | %-28s | %9.1f %9.1f | %9.1f %9.1f | %.1e" %
                  ("L%d k3 %d->%d" % (lev, cin, cout), M, wp.numel() / 1024, " -> ".join(plans), res["nw4", "bf16"], res["nw16", "bf16"],
                   res["nw4", "fp32"], res["nw16", "fp32"], rel))
        if lev < 2:
            cm.down(ts)      # creates the next coordinate level
        ts *= 2


if __name__ == "__main__":
    main()
