"""The reference-side binding d3net_amd.compat.PG_OP (the stub INTEGRATION.md describes) called with the reference's
own calling convention: caller-allocated / caller-resized tensors (functions/pointgroup_ops.py:24-32,57,136-141,165-176)."""
import numpy as np
import pytest
import torch

from oracle import pg_oracle as o

pytestmark = pytest.mark.gpu


def test_pg_op_stub_like_the_reference_wrapper(dev):
    from d3net_amd.compat import PG_OP
    rng = np.random.default_rng(0)
    # Voxelization_Idx.forward: CPU coords, empty outputs that the native resizes
    coords = torch.from_numpy(rng.integers(0, 9, (5000, 4)).astype(np.int64)); coords[:, 0] %= 2
    output_coords = coords.new(); input_map = torch.IntTensor(5000).zero_(); output_map = input_map.new()
    PG_OP.voxelize_idx(coords, output_coords, input_map, output_map, 2, 4)
    roc, rim, rom = o.voxelization_idx(coords.numpy(), 2, 4)
    assert np.array_equal(output_coords.numpy(), roc) and np.array_equal(input_map.numpy(), rim) and np.array_equal(output_map.numpy(), rom)
    # Voxelization.forward
    feats = torch.from_numpy(rng.standard_normal((5000, 6)).astype(np.float32)).to(dev)
    M, mA = rom.shape[0], rom.shape[1] - 1
    out = torch.zeros((M, 6), device=dev)
    PG_OP.voxelize_fp(feats, out, output_map.to(dev), 4, M, mA, 6)
    assert np.array_equal(out.cpu().numpy(), o.voxelization(feats.cpu().numpy(), rom, 4))
    # BallQueryBatchP.forward with its retry loop
    n = 4000
    xyz_np = (rng.random((n, 3)) * np.array([1, 1, 0.2])).astype(np.float32)
    xyz = torch.from_numpy(xyz_np).to(dev)
    bi = torch.zeros(n, dtype=torch.int32, device=dev); bo = torch.tensor([0, n], dtype=torch.int32, device=dev)
    meanActive = 2  # too small on purpose
    while True:
        idx = torch.zeros(n * meanActive, dtype=torch.int32, device=dev); start_len = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        nActive = PG_OP.ballquery_batch_p(xyz, bi, bo, idx, start_len, n, meanActive, 0.05)
        if nActive <= n * meanActive:
            break
        meanActive = int(nActive // n + 1)
    ridx, rsl = o.ballquery_batch_p(xyz_np, np.zeros(n, np.int32), np.array([0, n], np.int32), 0.05, 2)
    assert np.array_equal(idx[:nActive].cpu().numpy(), ridx) and np.array_equal(start_len.cpu().numpy(), rsl)
    # BFSCluster.forward: CPU tensors, outputs resized by the native
    sem = torch.from_numpy(rng.integers(1, 3, n).astype(np.int32))
    ci, co = sem.new(), sem.new()
    PG_OP.bfs_cluster(sem, idx[:nActive].cpu(), start_len.cpu(), ci, co, n, 10)
    rci, rco = o.bfs_cluster(sem.numpy(), ridx, rsl, 10)
    assert np.array_equal(ci.numpy(), rci) and np.array_equal(co.numpy(), rco)
