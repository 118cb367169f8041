"""`sparse_collate_fn` with the input voxelisation on the device.

The reference's loader stacks the per-scene sample dicts and then calls the single-threaded CPU `voxelization_idx` on
the stacked `locs_scaled` inside every DataLoader worker (reference: lib/dataset/pipeline.py:917-994, the call at :992).
This drop-in keeps the stacking contract (same keys, dtypes, the batch index in column 0 of `locs_scaled`, instance ids
offset by the running instance count, `batch_offsets` / `instance_offsets`, GT proposal lists) and moves the stacked
point tensors to `device` where `voxel_locs` / `p2v_map` / `v2p_map` are produced by the HIP operator -- bit-identical to
the CPU natives (first-occurrence voxel order, ascending point lists: tests/test_pg_ops_gpu.py) -- so the batch arrives
in the layout `PointGroup.feed` consumes without a host hash pass.  Use it as the DataLoader's `collate_fn` with
`num_workers=0` / in the main process (a worker process must not touch the GPU), or call it on the worker's CPU output.
"""
import numpy as np
import torch

from . import pointgroup_ops

_POINT_KEYS = ("locs", "locs_scaled", "feats", "sem_labels", "instance_ids", "instance_info", "instance_num_point",
               "num_instance", "gt_proposals_idx", "gt_proposals_offset")


def _t(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def scannet_collate_fn(batch):
    """generic part (reference: lib/dataset/pipeline.py:888-915): stack arrays / tensors, recurse into dicts, keep lists"""
    data = {}
    for key in batch[0].keys():
        if key in _POINT_KEYS:
            continue
        v = batch[0][key]
        if isinstance(v, (np.ndarray, torch.Tensor)):
            data[key] = torch.stack([_t(s[key]) for s in batch], 0)
        elif isinstance(v, dict):
            data[key] = sparse_collate_fn([s[key] for s in batch])
        else:
            data[key] = [s[key] for s in batch]
    return data


def sparse_collate_fn(batch, device=None, mode=4):
    data = scannet_collate_fn(batch)
    if "locs" not in batch[0]:
        return data
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    locs, locs_scaled, feats, sem, ids, info, npt = [], [], [], [], [], [], []
    batch_offsets, instance_offsets = [0], [0]
    gt_idx, gt_off = [], []
    total_inst = total_pts = 0
    for i, b in enumerate(batch):
        n = b["locs_scaled"].shape[0]
        locs.append(_t(b["locs"]))
        locs_scaled.append(torch.cat([torch.full((n, 1), i, dtype=torch.int64), _t(b["locs_scaled"]).long()], 1))
        feats.append(_t(b["feats"]))
        batch_offsets.append(batch_offsets[-1] + n)
        if "gt_proposals_idx" in b:
            gi = np.array(b["gt_proposals_idx"], copy=True)
            gi[:, 0] += total_inst; gi[:, 1] += total_pts
            gt_idx.append(torch.from_numpy(gi))
            go = np.array(b["gt_proposals_offset"], copy=True)
            if gt_off:
                gt_off.append(torch.from_numpy(go + int(gt_off[-1][-1]))[1:])
            else:
                gt_off.append(torch.from_numpy(go))
        if "instance_ids" in b:
            ii = np.array(b["instance_ids"], copy=True)      # (the reference shifts the sample's array in place)
            ii[ii != -1] += total_inst
            ninst = int(np.asarray(b["num_instance"]).item())
            total_inst += ninst; total_pts += len(ii)
            ids.append(torch.from_numpy(ii)); sem.append(_t(b["sem_labels"]))
            info.append(_t(b["instance_info"])); npt.append(_t(b["instance_num_point"]))
            instance_offsets.append(instance_offsets[-1] + ninst)
    data["locs"] = torch.cat(locs, 0).to(torch.float32).to(device)
    data["locs_scaled"] = torch.cat(locs_scaled, 0).to(device)
    data["feats"] = torch.cat(feats, 0).to(device)
    data["batch_offsets"] = torch.tensor(batch_offsets, dtype=torch.int32, device=device)
    if ids:
        data["sem_labels"] = torch.cat(sem, 0).long().to(device)
        data["instance_ids"] = torch.cat(ids, 0).long().to(device)
        data["instance_info"] = torch.cat(info, 0).to(torch.float32).to(device)
        data["instance_num_point"] = torch.cat(npt, 0).int().to(device)
        data["instance_offsets"] = torch.tensor(instance_offsets, dtype=torch.int32, device=device)
    if gt_idx:
        data["gt_proposals_idx"] = torch.cat(gt_idx, 0).to(torch.int32).to(device)
        data["gt_proposals_offset"] = torch.cat(gt_off, 0).to(torch.int32).to(device)
    # the loader's voxelisation, on the device (reference :992 runs the CPU native here)
    data["voxel_locs"], data["p2v_map"], data["v2p_map"] = pointgroup_ops.voxelization_idx(data["locs_scaled"], len(batch), mode)
    return data
