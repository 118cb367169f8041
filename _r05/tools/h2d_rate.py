#!/usr/bin/env python3
"""Host -> device time of one canonical scene batch (pinned host tensors, one stream): the PCIe-inclusive note of DESIGN §6."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3net_amd import synthetic as S
dev = torch.device("cuda", 0)
occ, sem, inst, _ = S.occupancy_grid()
batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)
host = {k: v.cpu().pin_memory() for k, v in batch.items() if torch.is_tensor(v)}
nbytes = sum(v.numel() * v.element_size() for v in host.values())
dst = {k: torch.empty_like(v, device=dev) for k, v in host.items()}
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k, v in host.items():
        dst[k].copy_(v, non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("batch tensors: %d, %.1f MB, H2D %.2f ms = %.1f GB/s" % (len(host), nbytes / 1e6, (t1 - t0) * 1e3, nbytes / (t1 - t0) / 1e9))
for k, v in sorted(host.items(), key=lambda kv: -kv[1].numel() * kv[1].element_size())[:5]:
    print("   %-18s %-22s %.1f MB" % (k, tuple(v.shape), v.numel() * v.element_size() / 1e6))
