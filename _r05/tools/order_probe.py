#!/usr/bin/env python3
"""Does the ROW ORDER of a coordinate level matter to the gather convolutions?  The level-0 / level-1 K = 27 convolutions of
bench.py's 4-scene batch (csrc/spconv2.hip: spconv_fwd2_kernel through d3_spconv_fwd2) timed with the voxel rows in raster order
(what the synthetic scenes give: x, y, z with z fastest), Morton order, 4^3 / 8^3 brick order, and shuffled.
usage: python tools/order_probe.py [scenes=4] [iters=40]"""
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


def spread(v, bits=10):
    """interleave zeros: bit i of v -> bit 3 i"""
    out = torch.zeros_like(v)
    for i in range(bits):
        out |= ((v >> i) & 1) << (3 * i)
    return out


def orders(c):
    b, x, y, z = (c[:, i].long() for i in range(4))
    n = c.size(0)
    res = {"raster": torch.arange(n, device=c.device)}
    key = (b << 30) | (spread(x) << 2) | (spread(y) << 1) | spread(z)
    res["morton"] = torch.argsort(key, stable=True)
    for B in (4, 8):
        kb = (((b * 256 + x // B) * 256 + y // B) * 256 + z // B) * (B ** 3) + ((x % B) * B + (y % B)) * B + (z % B)
        res["brick%d" % B] = torch.argsort(kb, stable=True)
    # z-slab order: (b, z // 4, x, y, z % 4): floors / table tops are horizontal
    res["shuffled"] = torch.randperm(n, device=c.device)
    return res


def main():
    nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    scenes = []
    for b in range(nsc):   # bench.py make_scenes("speaker")
        occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=b)
        scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + b, feat_seed=2 + b))
    batch = S.make_batch(scenes, dev)
    c0 = batch["voxel_locs"].int().contiguous()
    print("%-10s | %-16s %9s | %s" % ("order", "layer", "rows", "fwd bf16 us   wgrad us"))
    for name, perm in orders(c0).items():
        cm = ME.CoordinateManager(c0[perm].contiguous())
        ts = 1
        for lev, shapes in ((0, [(16, 16), (136, 16)]), (1, [(32, 32)])):
            nbr = cm.k3(ts)
            M = nbr.size(0)
            for cin, cout in shapes:
                torch.manual_seed(lev)
                xb = torch.randn(M, cin, device=dev).to(torch.bfloat16)
                W = (torch.randn(27, cin, cout, device=dev) * 0.1).contiguous()
                wp = torch.empty(L.d3_spconv_pack_bytes(27, cin, cout), dtype=torch.uint8, device=dev)
                assert L.d3_spconv_pack(_ptr(W), _ptr(wp), 27, cin, cout, 0, _stream()) == 0
                out = torch.empty(M, cout, device=dev)

                def run():
                    rc = L.d3_spconv_fwd2(_ptr(xb), xb.stride(0), _ptr(nbr), _ptr(wp), _ptr(out), cout, None, 0, None, M, M, 27, cin, cout, XBF16, _stream())
                    assert rc == 0, rc
                t_f = timeit(run, iters)
                # weight gradient (x bf16, dy fp32)
                t_w = float("nan")
                try:
                    dy = torch.randn(M, cout, device=dev)
                    dW = torch.empty(27, cin, cout, device=dev)
                    wsb = L.d3_spconv_wgrad2_ws_bytes(M, M, 27, cin, cout, XBF16)
                    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=dev)

                    def runw():
                        rc = L.d3_spconv_wgrad2(_ptr(xb), xb.stride(0), _ptr(nbr), _ptr(dy), cout, _ptr(dW), M, M, 27, cin, cout, cin, XBF16, _ptr(ws), ws.numel(), _stream())
                        assert rc == 0, rc
                    t_w = timeit(runw, iters)
                except Exception as e:      # (signature drift: the forward numbers are what the probe is for)
                    t_w = float("nan"); print("wgrad skipped:", repr(e)[:120])
                print("%-10s | L%d k3 %3d->%-3d    %9d | %10.1f %10.1f" % (name, lev, cin, cout, M, t_f, t_w), flush=True)
            if lev == 0:
                cm.down(ts)
            ts *= 2


if __name__ == "__main__":
    main()
