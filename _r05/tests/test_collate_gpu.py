"""device-side `sparse_collate_fn` against the reference's stacking rules (lib/dataset/pipeline.py:
917-994) restated with numpy and the oracle's voxelization_idx (C restatement of src/voxelize/voxelize.cpp): bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sample(seed, n, ninst):
    rng = np.random.default_rng(seed)
    ids = rng.integers(-1, ninst, n).astype(np.int64)
    return {"locs": rng.random((n, 3)).astype(np.float32) * 3, "locs_scaled": rng.integers(0, 40, (n, 3)).astype(np.int64),
            "feats": rng.standard_normal((n, 6)).astype(np.float32), "sem_labels": rng.integers(-1, 20, n).astype(np.int32),
            "instance_ids": ids, "instance_info": rng.random((n, 12)).astype(np.float32),
            "instance_num_point": rng.integers(1, 50, ninst).astype(np.int32), "num_instance": np.array(ninst),
            "gt_bbox": rng.random((128, 8, 3)).astype(np.float32), "scene_id": "scene%04d_00" % seed}


def test_sparse_collate_matches_reference_contract(dev):
    from d3net_amd.collate import sparse_collate_fn
    from oracle import pg_oracle as pg
    samples = [_sample(1, 3000, 5), _sample(2, 2500, 3), _sample(3, 10, 1)]
    keep = [s["instance_ids"].copy() for s in samples]
    out = sparse_collate_fn(samples, dev)
    assert all(np.array_equal(k, s["instance_ids"]) for k, s in zip(keep, samples))      # samples are not modified
    ns = [3000, 2500, 10]
    assert out["batch_offsets"].cpu().tolist() == [0, 3000, 5500, 5510] and out["batch_offsets"].dtype == torch.int32
    ls = np.concatenate([np.concatenate([np.full((n, 1), i), s["locs_scaled"]], 1) for i, (n, s) in enumerate(zip(ns, samples))])
    assert np.array_equal(out["locs_scaled"].cpu().numpy(), ls) and out["locs_scaled"].dtype == torch.int64
    ids, tot = [], 0
    for s in samples:
        ii = s["instance_ids"].copy(); ii[ii != -1] += tot; tot += int(s["num_instance"]); ids.append(ii)
    assert np.array_equal(out["instance_ids"].cpu().numpy(), np.concatenate(ids))
    assert out["instance_offsets"].cpu().tolist() == [0, 5, 8, 9]
    assert out["instance_num_point"].dtype == torch.int32 and out["sem_labels"].dtype == torch.int64
    assert out["gt_bbox"].shape == (3, 128, 8, 3) and out["scene_id"] == [s["scene_id"] for s in samples]
    vl, p2v, v2p = pg.voxelization_idx(ls, 3, 4)
    assert np.array_equal(out["voxel_locs"].cpu().numpy(), vl)
    assert np.array_equal(out["p2v_map"].cpu().numpy(), p2v) and np.array_equal(out["v2p_map"].cpu().numpy(), v2p)
