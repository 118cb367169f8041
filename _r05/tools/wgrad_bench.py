#!/usr/bin/env python3
"""Weight-gradient kernels per layer shape on the bench workload's coordinate levels (N scenes of the speaker config):
third-generation kernel (spconv_wgrad3_kernel) vs the generic kernel with transposing LDS reads vs its first staging scheme,
selected per call through D3_WG3 / D3_WG2_TR, results cross-checked against the first scheme.
usage: python tools/wgrad_bench.py [scenes=4] [levels=3] [iters=20]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3net_amd import minkowski as ME, synthetic as S  # noqa: E402

MODES = (("gen3", {"D3_WG3": 1, "D3_WG2_TR": 1}), ("gen2+tr", {"D3_WG3": 0, "D3_WG2_TR": 1}), ("gen2", {"D3_WG3": 0, "D3_WG2_TR": 0}))


def set_mode(env):
    """library switches are flipped through the C ABI (csrc/tuning.hip parses the environment only once)"""
    from d3net_amd import _lib
    for k, v in env.items():
        assert _lib.lib().d3_tuning_set(k.encode(), v) == 0


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
    nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda", 0)
    scenes = []
    for b in range(nsc):   # bench.py make_scenes("speaker")
        occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=b)
        scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + b, feat_seed=2 + b))
    batch = S.make_batch(scenes, dev)
    cm = ME.CoordinateManager(batch["voxel_locs"].int().contiguous())
    ts = 1
    print("%-22s %8s %8s | %s | %s" % ("layer", "rows", "alg MB", " ".join("%9s" % m for m, _ in MODES), "GB/s(gen3)  maxrel vs gen2"))
    for lev in range(nlev):
        C = 16 * (lev + 1)
        nbr = cm.k3(ts)
        M = nbr.size(0)
        cases = [("L%d k3 %d->%d" % (lev, C, C), nbr, nbr, M, M, 27, C, C, ME.D3_CONV_FLIPK),
                 ("L%d k3 %d->%d" % (lev, 2 * C, C), nbr, nbr, M, M, 27, 2 * C, C, ME.D3_CONV_FLIPK)]
        if lev == 0:
            cases.append(("L0 stem 136->16", nbr, nbr, M, M, 27, 136, 16, ME.D3_CONV_FLIPK))
        if lev + 1 < nlev + 1:
            child, up, Mo = cm.down(ts)
            cases.append(("L%d down %d->%d" % (lev, C, C + 16), child, up, M, Mo, 8, C, C + 16, 0))
            cases.append(("L%d up %d->%d" % (lev + 1, C + 16, C), up, child, Mo, M, 8, C + 16, C, 0))
        for name, tf, tb, Min, Mout, K, Cin, Cout, bfl in cases:
            torch.manual_seed(lev)
            xb = torch.randn(Min, Cin, device=dev).to(torch.bfloat16)
            W = torch.zeros(K, Cin, Cout, device=dev)
            dy = torch.randn(Mout, Cout, device=dev)
            res, tms = {}, {}
            for mode, env in MODES:
                set_mode(env)
                w = lambda: ME._conv_wgrad(xb, tf, tb, dy, W, Mout, bfl, ME.D3_CONV_XBF16)
                tms[mode] = timeit(w, iters)
                res[mode] = w()
            set_mode(MODES[0][1])
            dyb = dy.to(torch.bfloat16)
            wb = lambda: ME._conv_wgrad(xb, tf, tb, dyb, W, Mout, bfl, ME.D3_CONV_XBF16 | ME.D3_CONV_DYBF16)
            t_b = timeit(wb, iters)
            rel_b = float((wb() - res["gen3"]).abs().max() / (res["gen3"].abs().max() + 1e-20))
            alg = 2.0 * Min * Cin + 4.0 * Mout * Cout + 4.0 * K * Cin * Cout + 4.0 * (Min if Cin > Cout else Mout) * K
            ref = res["gen2"]
            rel = [float((res[m] - ref).abs().max() / (ref.abs().max() + 1e-20)) for m in ("gen3", "gen2+tr")]
            print("%-22s %8d %8.1f | %s | %8.0f   %.1e %.1e | dy bf16: %7.1f us (vs gen3 %.1e)" %
                  (name, Mout, alg / 1e6, " ".join("%9.1f" % tms[m] for m, _ in MODES), alg / tms["gen3"] / 1e3, rel[0], rel[1], t_b, rel_b))
        ts *= 2


if __name__ == "__main__":
    main()
