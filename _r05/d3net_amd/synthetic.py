"""Synthetic ScanNet-shaped scenes with the tensor contract of the reference's `sparse_collate_fn`
(reference: lib/dataset/pipeline.py:917-994; key list in SURVEY.md section 8(a) row A0).

`canonical_scene()` is the benchmark scene of SURVEY.md section 8(d) / Appendix C: a 200x150x100 grid of
2 cm cells (4 m x 3 m x 2 m room): floor, four walls and 8 hollow box shells drawn with
numpy.random.default_rng(0) -> 142,920 occupied voxels (4.8 % occupancy).  numpy only (host side);
`to_device` moves a batch to the GPU.
"""
import numpy as np

VOXEL = 0.02  # metres; data.scale = 50 (reference: conf/pointgroup.yaml:24)


def occupancy_grid(dims=(200, 150, 100), n_boxes=8, side=(15, 60), height=(15, 50), seed=0):
    """Boolean occupancy + per-cell (semantic, instance) labels.  Draw order fixed by SURVEY Appendix C."""
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    occ = np.zeros((X, Y, Z), bool)
    sem = np.full((X, Y, Z), -1, np.int64)
    inst = np.full((X, Y, Z), -1, np.int64)
    occ[:, :, 0] = True
    sem[:, :, 0] = 1  # floor
    inst[:, :, 0] = -1
    for sl in (np.s_[0], np.s_[-1], np.s_[:, 0], np.s_[:, -1]):
        occ[sl] = True
        sem[sl] = 0  # walls
        inst[sl] = -1
    boxes = []
    for i in range(n_boxes):
        sx, sy, sz = rng.integers(side[0], side[1]), rng.integers(side[0], side[1]), rng.integers(height[0], height[1])
        x0, y0 = rng.integers(2, X - sx - 2), rng.integers(2, Y - sy - 2)
        z0 = 1
        b = np.zeros((sx, sy, sz), bool)
        b[[0, -1]] = True
        b[:, [0, -1]] = True
        b[:, :, -1] = True
        occ[x0:x0 + sx, y0:y0 + sy, z0:z0 + sz] |= b
        sem[x0:x0 + sx, y0:y0 + sy, z0:z0 + sz][b] = 2 + (i % 18)
        inst[x0:x0 + sx, y0:y0 + sy, z0:z0 + sz][b] = i
        boxes.append((int(x0), int(y0), int(z0), int(sx), int(sy), int(sz)))
    return occ, sem, inst, boxes


def scene_from_grid(occ, sem, inst, n_feat=131, extra_frac=0.15, seed=1, feat_seed=2):
    """One scene: points (one per voxel centre + a jittered second point in `extra_frac` of voxels)."""
    vox = np.argwhere(occ)  # raster order (x, then y, then z)
    M = vox.shape[0]
    rng = np.random.default_rng(seed)
    extra = rng.random(M) < extra_frac
    jitter = rng.random((int(extra.sum()), 3)) * 0.98 + 0.01
    pts_vox = np.concatenate([vox, vox[extra]], 0)
    frac = np.concatenate([np.full((M, 3), 0.5), jitter], 0)
    # interleave so that the two points of a voxel are not adjacent (order = voxel raster order of first points,
    # then the extra points): keeps the first-occurrence voxel order equal to the raster order
    locs = ((pts_vox + frac) * VOXEL).astype(np.float32)
    N = locs.shape[0]
    sem_labels = sem[pts_vox[:, 0], pts_vox[:, 1], pts_vox[:, 2]].astype(np.int64)
    instance_ids = inst[pts_vox[:, 0], pts_vox[:, 1], pts_vox[:, 2]].astype(np.int64)
    feats = np.random.default_rng(feat_seed).standard_normal((N, n_feat)).astype(np.float32)
    return dict(locs=locs, locs_vox=pts_vox.astype(np.int64), feats=feats, sem_labels=sem_labels,
                instance_ids=instance_ids)


def instance_info(locs, instance_ids):
    """reference: lib/dataset/pipeline.py:711-772 (_getInstanceInfo): per point (mean xyz, mean xyz(dup), min xyz, max xyz)
    of its instance, and points per instance.  Rows of unlabelled points stay 0 for the mean columns."""
    N = locs.shape[0]
    info = np.zeros((N, 12), np.float32)
    n_inst = int(instance_ids.max()) + 1 if (instance_ids >= 0).any() else 0
    num_point = np.zeros(n_inst, np.int32)
    for i in range(n_inst):
        m = instance_ids == i
        if not m.any():
            continue
        xyz = locs[m]
        mean, mn, mx = xyz.mean(0), xyz.min(0), xyz.max(0)
        info[m, 0:3] = mean
        info[m, 3:6] = mean
        info[m, 6:9] = mn
        info[m, 9:12] = mx
        num_point[i] = int(m.sum())
    return info, num_point


def collate(scenes, max_num_instance=128):
    """Stack scenes the way `sparse_collate_fn` does (host side; voxelisation indices come from the
    caller -- the device `voxelization_idx` in the product, the oracle in tests)."""
    locs, locs_scaled, feats, sem, ins, info, npt, offs = [], [], [], [], [], [], [], [0]
    gt_centers, gt_sizes, gt_sem, gt_mask = [], [], [], []
    total_inst = 0
    for b, s in enumerate(scenes):
        n = s["locs"].shape[0]
        locs.append(s["locs"])
        locs_scaled.append(np.concatenate([np.full((n, 1), b, np.int64), s["locs_vox"]], 1))
        feats.append(s["feats"])
        sem.append(s["sem_labels"])
        ii = s["instance_ids"].copy()
        inf, num = instance_info(s["locs"], ii)
        ii[ii >= 0] += total_inst
        total_inst += len(num)
        ins.append(ii); info.append(inf); npt.append(num)
        offs.append(offs[-1] + n)
        c = np.zeros((max_num_instance, 3), np.float32); z = np.zeros((max_num_instance, 3), np.float32)
        l = np.zeros(max_num_instance, np.int64); m = np.zeros(max_num_instance, np.float32)
        for i in range(min(len(num), max_num_instance)):
            sel = s["instance_ids"] == i
            if sel.any():
                mn, mx = s["locs"][sel].min(0), s["locs"][sel].max(0)
                c[i] = (mn + mx) / 2; z[i] = mx - mn; l[i] = s["sem_labels"][sel][0]; m[i] = 1
        gt_centers.append(c); gt_sizes.append(z); gt_sem.append(l); gt_mask.append(m)
    return dict(
        locs=np.concatenate(locs), locs_scaled=np.concatenate(locs_scaled), feats=np.concatenate(feats),
        sem_labels=np.concatenate(sem), instance_ids=np.concatenate(ins), instance_info=np.concatenate(info),
        instance_num_point=np.concatenate(npt).astype(np.int32), batch_offsets=np.array(offs, np.int32),
        center_label=np.stack(gt_centers), size_label=np.stack(gt_sizes), sem_cls_label=np.stack(gt_sem),
        box_label_mask=np.stack(gt_mask))


def canonical_scene(n_feat=131):
    occ, sem, inst, _ = occupancy_grid()
    return scene_from_grid(occ, sem, inst, n_feat=n_feat)


def small_scene(dims=(48, 40, 24), n_boxes=3, seed=0, n_feat=131, side=(6, 14), height=(5, 12)):
    occ, sem, inst, _ = occupancy_grid(dims, n_boxes, side, height, seed)
    return scene_from_grid(occ, sem, inst, n_feat=n_feat, seed=seed + 1, feat_seed=seed + 2)


def to_device(batch, device):
    import torch
    out = {}
    for k, v in batch.items():
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device) if isinstance(v, np.ndarray) else v
    return out


def make_batch(scenes, device, mode=4):
    """collate + device voxelisation: the tensors `PointGroup.feed` consumes (sparse_collate_fn's contract,
    reference lib/dataset/pipeline.py:917-994; the reference computes voxel_locs / p2v_map / v2p_map with the CPU
    voxelization_idx inside the loader, :992 -- here the same operator runs on the device)."""
    import torch
    from . import pointgroup_ops
    batch = to_device(collate(scenes), device)
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(batch["locs_scaled"].contiguous(), len(scenes), mode)
    batch["voxel_locs"], batch["p2v_map"], batch["v2p_map"] = voxel_locs, p2v_map, v2p_map
    return batch


def make_vocabulary(size=3004):
    """vocabulary dict in the reference's format (word2idx / idx2word with pad_, unk, sos, eos first)"""
    words = ["pad_", "unk", "sos", "eos"] + ["w%d" % i for i in range(size - 4)]
    return {"word2idx": {w: i for i, w in enumerate(words)}, "idx2word": {str(i): w for i, w in enumerate(words)},   # str keys: the reference loads it from json
            "special_tokens": {"bos_token": "sos", "eos_token": "eos", "unk_token": "unk", "pad_token": "pad_"}}   # lib/dataset/pipeline.py:440-447


def add_language(batch, device, chunk=8, max_spk_len=30, max_lis_len=126, vocab=3004, seed=3):
    """Synthetic ScanRefer-shaped language tensors for the speaker / listener heads:
    random token ids in [4, V), caption length U[8, 30], description length U[10, 126], GloVe-like N(0,1) embeddings,
    referred object uniform over the scene's GT boxes.  Keys as written by the reference loader
    (lib/dataset/pipeline.py:282-318)."""
    import torch
    rng = np.random.default_rng(seed)
    B = batch["center_label"].shape[0]
    centers, sizes = batch["center_label"].cpu().numpy(), batch["size_label"].cpu().numpy()
    n_obj = batch["box_label_mask"].cpu().numpy().sum(1).astype(int)
    sgn = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)
    gt_bbox = centers[:, :, None, :] + sgn[None, None] * sizes[:, :, None, :] / 2
    T = max_lis_len + 2
    lang_len = rng.integers(10, max_lis_len + 1, (B, chunk)).astype(np.int64)
    spk_len = rng.integers(8, max_spk_len + 1, (B, chunk)).astype(np.int64)
    lang_ids = np.zeros((B, chunk, max_spk_len + 2), np.int64)
    ref_label = np.zeros((B, chunk, 128), np.float32)
    ref_corner = np.zeros((B, chunk, 8, 3), np.float32)
    cat = np.zeros((B, chunk), np.int64)
    for b in range(B):
        for c in range(chunk):
            n = spk_len[b, c] + 2
            lang_ids[b, c, 0] = 2; lang_ids[b, c, 1:n - 1] = rng.integers(4, vocab, n - 2); lang_ids[b, c, n - 1] = 3
            o = rng.integers(0, max(n_obj[b], 1))
            ref_label[b, c, o] = 1; ref_corner[b, c] = gt_bbox[b, o]
            cat[b, c] = int(batch["sem_cls_label"][b, o]) % 18
    out = dict(lang_feat=rng.standard_normal((B, chunk, T, 300)).astype(np.float32), lang_len=lang_len, lang_ids=lang_ids,
               annotated=np.ones((B, chunk), np.int64), ref_box_label=ref_label, ref_box_corner_label=ref_corner, object_cat=cat,
               gt_bbox=gt_bbox.astype(np.float32), istrain=np.ones(B, np.int64),
               scene_object_rotations=np.tile(np.eye(3, dtype=np.float32), (B, 128, 1, 1)),
               scene_object_rotation_masks=batch["box_label_mask"].cpu().numpy().astype(np.float32))
    # the speaker's lang_len is the caption length (+2), the listener's the description length: the reference feeds
    # two different batches; a single synthetic batch carries the caption lengths under `lang_len` for mode 1
    out["gt_bbox_label"] = batch["box_label_mask"].cpu().numpy().astype(np.float32)      # valid GT boxes (lib/dataset/pipeline.py:300)
    out["gt_bbox_object_id"] = np.tile(np.arange(gt_bbox.shape[1], dtype=np.int64), (B, 1))
    batch["scene_id"] = ["scene%04d_00" % b for b in range(B)]
    out["id"] = np.arange(B, dtype=np.int64)                                  # scene index into `chunked_data`
    out["chunk_ids"] = np.tile(np.arange(chunk, dtype=np.int64), (B, 1))
    for k, v in out.items():
        batch[k] = torch.from_numpy(v).to(device)
    batch["spk_lang_len"] = torch.from_numpy(spk_len + 2).to(device)
    return batch


def make_language_corpus(B, chunk=8, vocab=3004, max_spk_len=30, objects_per_scene=8, seed=5):
    """Synthetic ScanRefer-shaped annotation store for the self-critical reward (lib/captioning/loss_helper.py:15-96):
    `chunked_data[scene][chunk]` -> {scene_id, object_id}; `organized[scene_id][object_id]` -> list of {token: [...]}
    (2-5 tokenised descriptions per object, words drawn from the vocabulary)."""
    rng = np.random.default_rng(seed)
    words = ["w%d" % i for i in range(vocab - 4)]
    chunked, organized = [], {}
    for b in range(B):
        sid = "scene%04d_00" % b
        organized[sid] = {str(o): [{"token": [words[i] for i in rng.integers(0, len(words), int(rng.integers(6, max_spk_len)))]}
                                   for _ in range(int(rng.integers(2, 6)))] for o in range(objects_per_scene)}
        chunked.append([{"scene_id": sid, "object_id": str(int(rng.integers(0, objects_per_scene)))} for _ in range(chunk)])
    return chunked, organized


def corpus_raw_data(organized):
    """the flat description list the evaluation corpus is built from (`dataset.raw_data`: scene_id, object_id, token)"""
    return [{"scene_id": sid, "object_id": oid, "token": d["token"]} for sid, objs in organized.items() for oid, ds in objs.items() for d in ds]
