#!/usr/bin/env python3
"""A/B micro-benchmark of `spconv_fwd3_kernel` (csrc/spconv3.hip, lane table) against `spconv_fwd2_kernel` (dense / 16-bit table) on
the coordinate levels of bench.py's 4-scene batch; outputs cross-checked (same bf16 operands, fp32 accumulation: only the
summation order differs).
usage: python tools/fwd3_bench.py [scenes=4] [iters=30]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3net_amd import _lib, minkowski as ME, synthetic as S  # noqa: E402
from d3net_amd.pointgroup_ops import _ptr, _stream  # noqa: E402

XBF16, OUTBF16 = 32, 512


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
    shapes = {0: [(16, 16), (32, 16), (16, 32)], 1: [(32, 32), (64, 32), (32, 64)], 2: [(48, 48)]}
    print("%-16s %8s | %9s %9s %9s | %9s %9s | %9s %9s | maxrel (plain res bnbwd)" %
          ("layer", "rows", "fwd2", "fwd2 t16", "fwd3", "fwd2 res", "fwd3 res", "fwd2 bnb", "fwd3 bnb"))
    ts = 1
    for lev in range(3):
        nbr = cm.k3(ts)
        M = nbr.size(0)
        n16 = torch.empty(M * 27 + 2, dtype=torch.int16, device=dev)
        ok16 = torch.zeros(1, dtype=torch.int32, device=dev)
        assert L.d3_kmap_k3_pack16(_ptr(nbr), M, _ptr(n16), _ptr(ok16), _stream()) == 0
        tq = torch.empty(L.d3_kmap_k3_q16_bytes(M), dtype=torch.uint8, device=dev)
        okq = torch.zeros(1, dtype=torch.int32, device=dev)
        assert L.d3_kmap_k3_packq(_ptr(nbr), M, _ptr(tq), _ptr(okq), _stream()) == 0
        assert int(okq[0]) == 1 and int(ok16[0]) == 1
        nt = (M + 15) // 16
        recs = tq[nt * 1024:].view(torch.int32).view(nt, 8)
        print("level %d: %d rows, present fraction %.3f, live gather groups per tile %.2f of 7" % (lev, M, float((nbr >= 0).float().mean()), float(recs[:, 7].float().mean())), flush=True)
        for cin, cout in shapes[lev]:
            torch.manual_seed(lev)
            xb = torch.randn(M, cin, device=dev).to(torch.bfloat16)
            W = (torch.randn(27, cin, cout, device=dev) * 0.1).contiguous()
            wp = torch.empty(L.d3_spconv_pack_bytes(27, cin, cout), dtype=torch.uint8, device=dev)
            assert L.d3_spconv_pack(_ptr(W), _ptr(wp), 27, cin, cout, 0, _stream()) == 0
            res = torch.randn(M, cout, device=dev)
            bnx = torch.randn(M, cout, device=dev)
            mean, var = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
            gamma, beta = torch.randn(cout, device=dev), torch.randn(cout, device=dev) * 0.1
            np2 = L.d3_spconv_fwd2_nparts(M, 27, cin, cout)
            np3 = L.d3_spconv_fwd3_nparts(M, cin, cout)
            assert np3 > 0
            part2 = torch.zeros(np2, 2, (cout + 15) // 16 * 16, device=dev)
            part3 = torch.zeros(np3, 2, cout, device=dev)
            o2, o3 = torch.empty(M, cout, device=dev), torch.empty(M, cout, device=dev)

            def f2(r=None, t16=False):
                if t16:
                    L.d3_tuning_set(b"D3_KMAP16", 1)
                rc = L.d3_spconv_fwd2(_ptr(xb), cin, _ptr(nbr), _ptr(wp), _ptr(o2), cout, _ptr(r) if r is not None else None, cout,
                                      _ptr(part2), M, M, 27, cin, cout, XBF16, _stream())
                assert rc == 0, rc

            def f3(r=None):
                rc = L.d3_spconv_fwd3(_ptr(xb), cin, _ptr(tq), _ptr(wp), _ptr(o3), cout, _ptr(r) if r is not None else None, cout,
                                      _ptr(part3), None, M, M, cin, cout, 0, _stream())
                assert rc == 0, rc

            def b2():
                rc = L.d3_spconv_fwd2_bnbwd(_ptr(xb), cin, _ptr(nbr), _ptr(wp), _ptr(o2), cout, _ptr(part2), _ptr(bnx), cout, _ptr(mean), _ptr(var),
                                            _ptr(gamma), _ptr(beta), 1e-4, 1, M, M, 27, cin, cout, XBF16, _stream())
                assert rc == 0, rc

            def b3():
                rc = L.d3_spconv_fwd3_bnbwd(_ptr(xb), cin, _ptr(tq), _ptr(wp), _ptr(o3), cout, _ptr(part3), None, _ptr(bnx), cout, _ptr(mean), _ptr(var),
                                            _ptr(gamma), _ptr(beta), 1e-4, 1, M, M, cin, cout, 0, _stream())
                assert rc == 0, rc

            def rel(a, b):
                return float((a - b).abs().max() / b.abs().max())

            t = {}
            rels = []
            t["f2"] = timeit(f2, iters)
            part2.zero_(); f2()
            f3()
            rels.append(max(rel(o3, o2), rel(part3[:L.d3_spconv_last_nparts()].sum(0), part2.sum(0)[:, :cout])))
            t["f3"] = timeit(f3, iters)
            t["f2r"] = timeit(lambda: f2(res), iters)
            part2.zero_(); f2(res)
            f3(res)
            rels.append(max(rel(o3, o2), rel(part3[:L.d3_spconv_last_nparts()].sum(0), part2.sum(0)[:, :cout])))
            t["f3r"] = timeit(lambda: f3(res), iters)
            t["b2"] = timeit(b2, iters)
            part2.zero_(); b2()
            b3()
            rels.append(max(rel(o3, o2), rel(part3[:L.d3_spconv_last_nparts()].sum(0), part2.sum(0)[:, :cout])))
            t["b3"] = timeit(b3, iters)
            print("%-16s %8d | %9.1f %9s %9.1f | %9.1f %9.1f | %9.1f %9.1f | %.1e %.1e %.1e  (grid %d)" %
                  ("L%d %d->%d" % (lev, cin, cout), M, t["f2"], "-", t["f3"], t["f2r"], t["f3r"], t["b2"], t["b3"], *rels, L.d3_spconv_last_nparts()), flush=True)
        if lev < 2:
            cm.down(ts)      # creates the next coordinate level
        ts *= 2


if __name__ == "__main__":
    main()
