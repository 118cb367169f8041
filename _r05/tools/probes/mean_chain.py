"""sec_mean_pc_kernel alone: one cluster of n points (and the 4-scene mix: 8 clusters of 33,721 + 460 of 1,500)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3net_amd import pointgroup_ops as P
dev = torch.device("cuda", 0)
def run(sizes):
    S = sum(sizes); N = 750000
    coords = torch.rand(N, 3, device=dev) * 8
    pts = torch.randint(0, N, (S,), device=dev, dtype=torch.int32)
    cid = torch.repeat_interleave(torch.arange(len(sizes), device=dev, dtype=torch.int32), torch.tensor(sizes, device=dev))
    idx = torch.stack([cid, pts], 1).contiguous()
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=dev)
    for _ in range(3): P.cluster_coords_stats(coords, idx, off)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): P.cluster_coords_stats(coords, idx, off)
    b.record(); torch.cuda.synchronize()
    print(len(sizes), "clusters, largest", max(sizes), ": %.1f us per call (mean + min/max kernels)" % (a.elapsed_time(b) * 100))
run([33721]); run([1000]); run([33721] * 8 + [1500] * 460); run([100000])
