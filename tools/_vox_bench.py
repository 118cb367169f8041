import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from d3net_amd import pointgroup_ops as P, synthetic as S
dev = torch.device("cuda", 0)
occ, sem, inst, _ = S.occupancy_grid()
batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)
v2p = batch["v2p_map"]; N = batch["locs"].shape[0]; M = v2p.shape[0]
print("N", N, "M", M, "maxActive", v2p.shape[1] - 1, "mean active", float(v2p[:, 0].float().mean()))
for C in (16, 134):
    f = torch.randn(N, C, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): o = P.voxelization(f, v2p, 4)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print("voxelization fwd C=%d: %.1f us" % (C, (t1 - t0) / 20 * 1e6))
