#!/usr/bin/env python3
"""Times the forward convolutions of the backbone's first levels (4-scene bench batch) with the library given by D3_SO (a variant built by
tools/probes/build_variant.sh) -- one process per variant: python tools/probes/conv_variants.py [so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    for so in sys.argv[1:]:
        print("==", so, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, D3_SO=so), check=False)
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch  # noqa: E402
from d3net_amd import _lib  # noqa: E402
so = os.environ.get("D3_SO", "default")
if so != "default":
    _lib.SO_PATH = so
from d3net_amd import minkowski as ME, synthetic as S  # noqa: E402
from d3net_amd.pointgroup_ops import _ptr, _stream  # noqa: E402

XBF16 = 32
dev = torch.device("cuda", 0)
L = _lib.lib()
scenes = []
for b in range(4):
    occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=b)
    scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + b, feat_seed=2 + b))
batch = S.make_batch(scenes, dev)
cm = ME.CoordinateManager(batch["voxel_locs"].int().contiguous())
shapes = {0: [(16, 16), (32, 16), (136, 16), (16, 32)], 1: [(32, 32), (64, 32)]}
ts = 1
for lev in (0, 1):
    nbr = cm.k3(ts)
    M = nbr.size(0)
    for cin, cout in shapes[lev]:
        torch.manual_seed(lev * 10 + cin)
        x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
        W = (torch.randn(27, cin, cout, device=dev) * 0.1).contiguous()
        wp = torch.empty(L.d3_spconv_pack_bytes(27, cin, cout), dtype=torch.uint8, device=dev)
        assert L.d3_spconv_pack(_ptr(W), _ptr(wp), 27, cin, cout, 0, _stream()) == 0
        out = torch.empty(M, cout, device=dev)

        def run():
            assert L.d3_spconv_fwd2(_ptr(x), x.stride(0), _ptr(nbr), _ptr(wp), _ptr(out), cout, None, 0, None, M, M, 27, cin, cout, XBF16, _stream()) == 0
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            run()
        b_.record()
        torch.cuda.synchronize()
        print("level %d  %3d -> %2d  rows %7d  %7.1f us   checksum %.6e" % (lev, cin, cout, M, a.elapsed_time(b_) / 30 * 1e3, float(out.double().sum())), flush=True)
    cm.down(ts)      # (builds the next level's coordinates)
    ts *= 2
