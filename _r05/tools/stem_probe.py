#!/usr/bin/env python3
"""HBM-side traffic of the stem convolution (K = 27, 136 -> 16, bf16 rows of 272 B) under variations, to be run under
`rocprofv3 --pmc FETCH_SIZE` (then: tools/stem_probe.py --parse <counter_collection.csv>): 3 launches each of
  A interleaved tile groups, scan-ordered rows      B contiguous tile ranges, scan-ordered rows
  C interleaved, rows shuffled                      D interleaved, 4-wave workgroups (weights from L2)
  E interleaved, 16 -> 16 convolution on the same rows (32-byte rows) for scale
  F as A with the offsets of a tile split over 4 waves (D3_C2_KSPLIT=1, round 5; A - E run with D3_C2_KSPLIT=0, D3_C2_COMPACT=0)
  G as A with the tile's dead offsets dropped (D3_C2_COMPACT=1, round 5)     H as E with D3_C2_COMPACT=1     I / J: 32 -> 32 without / with
usage: python tools/stem_probe.py [scenes=4]"""
import csv
import os
import sys

if "--parse" in sys.argv:
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--parse") + 1])))
    rows = [r for r in rows if "spconv_fwd2_" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for i, r in enumerate(rows):
        print("%2d %-10s %10.1f KiB  %s" % (i, r["Counter_Name"], float(r["Counter_Value"]), r["Kernel_Name"].split("(")[0][-60:]))
    sys.exit(0)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3net_amd import _lib, minkowski as ME, synthetic as S  # noqa: E402
from d3net_amd.pointgroup_ops import _ptr, _stream  # noqa: E402

XBF16 = 32
nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
L = _lib.lib()
scenes = []
for b in range(nsc):
    occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=b)
    scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + b, feat_seed=2 + b))
batch = S.make_batch(scenes, dev)
coords = batch["voxel_locs"].int().contiguous()


def conv(coords, cin, cout, n=3):
    cm = ME.CoordinateManager(coords)
    nbr = cm.k3(1)
    M = nbr.size(0)
    x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(27, cin, cout, device=dev) * 0.1).contiguous()
    wp = torch.empty(L.d3_spconv_pack_bytes(27, cin, cout), dtype=torch.uint8, device=dev)
    assert L.d3_spconv_pack(_ptr(W), _ptr(wp), 27, cin, cout, 0, _stream()) == 0
    out = torch.empty(M, cout, device=dev)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        assert L.d3_spconv_fwd2(_ptr(x), x.stride(0), _ptr(nbr), _ptr(wp), _ptr(out), cout, None, 0, None, M, M, 27, cin, cout, XBF16, _stream()) == 0
        ev[i + 1].record()
    torch.cuda.synchronize()
    print("   %d -> %d: %s us per launch" % (cin, cout, ", ".join("%.1f" % (1e3 * ev[i].elapsed_time(ev[i + 1])) for i in range(n))))
    return out


L.d3_tuning_set(b"D3_C2_KSPLIT", 0); L.d3_tuning_set(b"D3_C2_COMPACT", 0)
torch.manual_seed(0)
L.d3_tuning_set(b"D3_C2_INTERLEAVE", 1); oA = conv(coords, 136, 16, 6)                                  # A
L.d3_tuning_set(b"D3_C2_INTERLEAVE", 0); conv(coords, 136, 16)                                        # B
L.d3_tuning_set(b"D3_C2_INTERLEAVE", 1)
perm = torch.from_numpy(np.random.default_rng(0).permutation(coords.size(0))).to(dev)
conv(coords[perm].contiguous(), 136, 16)                                                              # C
L.d3_tuning_set(b"D3_C2_NW16_KB", 1 << 20); conv(coords, 136, 16); L.d3_tuning_set(b"D3_C2_NW16_KB", 24)   # D
torch.manual_seed(1)
oE = conv(coords, 16, 16, 6)                                                                          # E
torch.manual_seed(0)
L.d3_tuning_set(b"D3_C2_KSPLIT", 1); oF = conv(coords, 136, 16, 6)                                      # F
print("F vs A: max |diff| %.3e (max |A| %.3e)" % (float((oF - oA).abs().max()), float(oA.abs().max())))
L.d3_tuning_set(b"D3_C2_KSPLIT", 0); L.d3_tuning_set(b"D3_C2_COMPACT", 1)
torch.manual_seed(0)
oG = conv(coords, 136, 16, 6)                                                                         # G
print("G vs A: max |diff| %.3e" % float((oG - oA).abs().max()))
torch.manual_seed(1)
oH = conv(coords, 16, 16, 6)                                                                          # H
print("H vs E: max |diff| %.3e (max |E| %.3e)" % (float((oH - oE).abs().max()), float(oE.abs().max())))
L.d3_tuning_set(b"D3_C2_COMPACT", 0); torch.manual_seed(2); oI = conv(coords, 32, 32, 6)              # I
L.d3_tuning_set(b"D3_C2_COMPACT", 1); torch.manual_seed(2); oJ = conv(coords, 32, 32, 6)              # J
print("J vs I: max |diff| %.3e" % float((oJ - oI).abs().max()))
print("rows", coords.size(0), "input MB", coords.size(0) * 136 * 2 / 1e6)
