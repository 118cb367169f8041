"""fp32 CPU restatement of the reference's PointGroup detector step (TEST INFRASTRUCTURE ONLY).

Follows model/pointgroup.py line by line -- feed :466-479, forward :266-370, clusters_voxelization :125-178,
convert_stack_to_batch :223-263, get_object_assignments :216-221, parse_feed_ret :481-510, loss :387-463 --
on CPU tensors, with the native operators replaced by oracle/pg_oracle.py (C restatement of PG_OP) and the
MinkowskiEngine layers by oracle/sparse_oracle.py.  The reference class itself cannot be imported
(MinkowskiEngine, pytorch_lightning and PG_OP are absent) and hard-codes .cuda(); PARITY UNPINNED for the
MinkowskiEngine part (see sparse_oracle.py), everything else is restated from the source text.

Weights come from a state dict with the reference's key layout, so the same tensors
drive this oracle and the HIP model.  Host RNG draws (`torch.rand(3)` x2, `torch.randperm(128)`) can be injected.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import pg_oracle as pg
from . import sparse_oracle as so


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


class PointGroupOracle:
    def __init__(self, cfg, state_dict, training=True):
        self.cfg = cfg
        # fp32 leaf copies so that autograd provides the backward oracle
        self.p = {k: v.detach().cpu().clone().float().requires_grad_(v.dtype.is_floating_point and "running" not in k)
                  for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.training = training
        self.m = cfg.model.m
        self.teacher = False

    # ------------------------------------------------------------------------------- network pieces
    def _bn_sparse(self, x, name, relu=True):
        if not self.training:   # model.eval(): running statistics
            return so.bn_relu(x, self.p[name + ".bn.weight"], self.p[name + ".bn.bias"], 1e-4, relu,
                              running=(self.p[name + ".bn.running_mean"], self.p[name + ".bn.running_var"]), training=False)
        return so.bn_relu(x, self.p[name + ".bn.weight"], self.p[name + ".bn.bias"], 1e-4, relu,
                          training=True)  # batch statistics (training mode), running buffers untouched

    def _unet(self, x, cm, planes, prefix):
        params = {k[len(prefix) + 1:]: v for k, v in self.p.items() if k.startswith(prefix + ".")}
        return so.OracleUNet(params, planes, block_reps=self.cfg.model.block_reps if prefix == "backbone" else 2,
                             prefix="1" if prefix == "backbone" else "0", training=self.training).forward(x, cm)

    def backbone(self, voxel_feats, voxel_coords):
        cm = so.OracleCoords(voxel_coords)
        h = so.conv_k3(voxel_feats, self.p["backbone.0.kernel"], cm.get_k3(1))
        h = self._unet(h, cm, [self.m * c for c in self.cfg.model.blocks], "backbone")
        return self._bn_sparse(h, "backbone.2")

    def score_net(self, voxel_feats, voxel_coords):
        cm = so.OracleCoords(voxel_coords)
        h = self._unet(voxel_feats, cm, [self.m * c for c in self.cfg.model.cluster_blocks], "score_net")
        return self._bn_sparse(h, "score_net.1")

    def offset_net(self, x):
        p = self.p
        h = F.linear(x, p["offset_net.0.weight"], p["offset_net.0.bias"])
        if self.training:
            h = F.batch_norm(h, None, None, p["offset_net.1.weight"], p["offset_net.1.bias"], True, 0.1, 1e-4)
        else:
            h = F.batch_norm(h, p["offset_net.1.running_mean"], p["offset_net.1.running_var"], p["offset_net.1.weight"],
                             p["offset_net.1.bias"], False, 0.1, 1e-4)
        return F.linear(torch.relu(h), p["offset_net.3.weight"], p["offset_net.3.bias"])

    # ---------------------------------------------------------------- reference :125-178
    def clusters_voxelization(self, clusters_idx, clusters_offset, feats, coords, fullscale, scale, mode, rand=None):
        c_idxs = _t(clusters_idx[:, 1]).long()
        cid = _t(clusters_idx[:, 0]).long()
        clusters_feats = feats[c_idxs]
        clusters_coords = coords[c_idxs]
        mean = _t(pg.sec_mean(clusters_coords.detach().numpy(), clusters_offset))
        clusters_coords = clusters_coords - torch.index_select(mean, 0, cid)
        cmin = _t(pg.sec_min(clusters_coords.numpy(), clusters_offset))
        cmax = _t(pg.sec_max(clusters_coords.numpy(), clusters_offset))
        clusters_size = cmax - cmin
        clusters_center = (cmax + cmin) / 2 + mean
        clusters_scale = 1 / ((cmax - cmin) / fullscale).max(1)[0] - 0.01
        clusters_scale = torch.clamp(clusters_scale, min=None, max=scale)
        min_xyz = cmin * clusters_scale.unsqueeze(-1)
        max_xyz = cmax * clusters_scale.unsqueeze(-1)
        clusters_scale = torch.index_select(clusters_scale, 0, cid)
        clusters_coords = clusters_coords * clusters_scale.unsqueeze(-1)
        rng = max_xyz - min_xyz
        r0, r1 = (torch.rand(3), torch.rand(3)) if rand is None else (rand[0], rand[1])
        offset = - min_xyz + torch.clamp(fullscale - rng - 0.001, min=0) * r0 + torch.clamp(fullscale - rng + 0.001, max=0) * r1
        clusters_coords = clusters_coords + torch.index_select(offset, 0, cid)
        assert clusters_coords.shape.numel() == ((clusters_coords >= 0) * (clusters_coords < fullscale)).sum()
        clusters_coords = torch.cat([cid.view(-1, 1), clusters_coords.long()], 1)
        vc, p2v, v2p = pg.voxelization_idx(clusters_coords.numpy(), int(clusters_idx[-1, 0]) + 1, mode)
        # voxelization (mean pooling) with autograd: rows of v2p
        vf = _Voxelize.apply(clusters_feats, v2p, mode)
        return (vf, vc), p2v, (clusters_center, clusters_size)

    # ---------------------------------------------------------------- reference :466-479 + :266-370
    def feed(self, d, epoch=0, rand=None, perms=None):
        cfg = self.cfg
        d = dict(d)
        d["epoch"] = epoch
        if cfg.model.use_coords:
            d["feats"] = torch.cat((d["feats"], d["locs"]), 1)
        d["voxel_feats"] = _Voxelize.apply(d["feats"], d["v2p_map"].numpy(), cfg.data.mode)
        batch_size = len(d["batch_offsets"]) - 1

        out = self.backbone(d["voxel_feats"], d["voxel_locs"].numpy())
        pt_feats = out[d["p2v_map"].long()]
        semantic_scores = F.linear(pt_feats, self.p["sem_seg.weight"], self.p["sem_seg.bias"])
        semantic_preds = semantic_scores.max(1)[1]
        d["semantic_scores"] = semantic_scores
        pt_offsets = self.offset_net(pt_feats)
        d["pt_offsets"] = pt_offsets

        if self.teacher:
            semantic_preds = d["sem_labels"].clamp(min=0)
            cl_off = (d["instance_info"][:, 0:3] - d["locs"]).detach()
            cl_off = torch.where((d["instance_ids"] >= 0).unsqueeze(1), cl_off, torch.zeros_like(cl_off))
        else:
            cl_off = pt_offsets
        batch_idxs = d["locs_scaled"][:, 0].int()
        object_idxs = torch.nonzero(semantic_preds > 0, as_tuple=False).view(-1)
        batch_idxs_ = batch_idxs[object_idxs]
        bo = np.zeros(batch_size + 1, np.int32)
        for i in range(batch_size):
            bo[i + 1] = bo[i] + int((batch_idxs_ == i).sum())
        coords_ = d["locs"][object_idxs]
        sem_ = semantic_preds[object_idxs].int().numpy()
        th = cfg.cluster.cluster_npoint_thre

        idx_s, sl_s = pg.ballquery_batch_p((coords_ + cl_off[object_idxs]).detach().numpy(), batch_idxs_.numpy(), bo,
                                           cfg.cluster.cluster_radius, cfg.cluster.cluster_shift_meanActive)
        pis, pos = pg.bfs_cluster(sem_, idx_s, sl_s, th)
        pis[:, 1] = object_idxs.numpy()[pis[:, 1]]
        bid_s = batch_idxs.numpy()[pis[:, 1]]
        idx_o, sl_o = pg.ballquery_batch_p(coords_.numpy(), batch_idxs_.numpy(), bo, cfg.cluster.cluster_radius,
                                           cfg.cluster.cluster_meanActive)
        pi, po = pg.bfs_cluster(sem_, idx_o, sl_o, th)
        pi[:, 1] = object_idxs.numpy()[pi[:, 1]]
        bid = batch_idxs.numpy()[pi[:, 1]]
        pis[:, 0] += (po.shape[0] - 1)
        pos = pos + po[-1]
        proposals_idx = np.concatenate((pi, pis), 0)
        proposals_offset = np.concatenate((po, pos[1:]))
        bid_all = np.concatenate((bid, bid_s[1:]))
        d["_debug"] = dict(idx_shift=idx_s, start_len_shift=sl_s, idx=idx_o, start_len=sl_o)

        (vf, vc), p2v, (center, size) = self.clusters_voxelization(
            proposals_idx, proposals_offset, pt_feats, d["locs"], cfg.train.score_fullscale, cfg.train.score_scale,
            cfg.train.score_mode, rand)
        score_feats = self.score_net(vf, vc)
        pt_score_feats = score_feats[_t(p2v).long()]
        psf = _RoiPool.apply(pt_score_feats, proposals_offset)
        scores = F.linear(psf, self.p["score_linear.weight"], self.p["score_linear.bias"])
        d["proposal_scores"] = (scores, proposals_idx, proposals_offset)
        P = proposals_offset.shape[0] - 1
        npoint = torch.zeros(P)
        pid0 = _t(proposals_idx[:, 0])
        for i in range(P):
            npoint[i] = (pid0 == i).sum()
        sig = torch.sigmoid(scores.view(-1))
        mask = torch.logical_and(sig > cfg.test.TEST_SCORE_THRESH, npoint > cfg.test.TEST_NPOINT_THRESH)
        d["proposals_npoint"] = npoint
        d["proposal_thres_mask"] = mask
        d["proposals_batchId"] = _t(bid_all)[_t(proposals_offset[:-1]).long()][mask]
        d["proposal_feats"] = psf[mask]
        d["proposal_objectness_scores"] = sig[mask]
        crop = torch.zeros(P, 9)
        crop[:, :3] = center
        crop[:, 3:6] = size
        crop[:, 7] = semantic_preds[_t(proposals_idx[proposals_offset[:-1], 1]).long()].float()
        crop[:, 8] = sig
        d["proposal_crop_bbox"] = crop[mask]
        return self.convert_stack_to_batch(d, perms)

    # ---------------------------------------------------------------- reference :223-263
    def convert_stack_to_batch(self, d, perms=None):
        cfg = self.cfg
        B = len(d["batch_offsets"]) - 1
        K = cfg.model.max_num_proposal
        pf = d["proposal_feats"]
        keys = {"proposal_feats_batched": (K, self.m), "proposal_bbox_batched": (K, 8, 3), "proposal_center_batched": (K, 3),
                "proposal_sem_cls_batched": (K,), "proposal_scores_batched": (K,), "proposal_batch_mask": (K,)}
        for k, s in keys.items():
            d[k] = torch.zeros((B,) + s)
        pb = d["proposal_crop_bbox"].detach().numpy()
        l, w, h = pb[:, 3:4], pb[:, 4:5], pb[:, 5:6]           # lib/utils/bbox.py:54-74 with heading 0 (R = I), float64
        corners = np.zeros((pb.shape[0], 8, 3))
        corners[:, :, 0] = np.concatenate((l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2), -1)
        corners[:, :, 1] = np.concatenate((w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2), -1)
        corners[:, :, 2] = np.concatenate((h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2), -1)
        corners += np.expand_dims(pb[:, :3].astype(np.float64), -2)
        corners = torch.tensor(corners).float()
        for b in range(B):
            idx = torch.nonzero(d["proposals_batchId"] == b).squeeze(-1)
            n = min(len(idx), K)
            d["proposal_feats_batched"][b, :n] = pf[idx][:n]
            d["proposal_bbox_batched"][b, :n] = corners[idx][:n]
            d["proposal_center_batched"][b, :n] = d["proposal_crop_bbox"][idx, :3][:n]
            d["proposal_sem_cls_batched"][b, :n] = d["proposal_crop_bbox"][idx, 7][:n]
            d["proposal_scores_batched"][b, :n] = d["proposal_objectness_scores"][idx][:n]
            d["proposal_batch_mask"][b, :n] = 1
            perm = torch.randperm(K) if perms is None else perms[b]
            for k in keys:
                d[k][b] = d[k][b][perm]
        diff = d["proposal_center_batched"].unsqueeze(2) - d["center_label"].unsqueeze(1)
        d["object_assignment"] = diff.abs().sum(-1).min(2)[1]          # nn_distance(l1=True) (:216-221)
        return d

    # ---------------------------------------------------------------- reference :387-463
    def loss(self, d):
        cfg = self.cfg
        semantic_scores, semantic_labels = d["semantic_scores"], d["sem_labels"]
        semantic_loss = F.cross_entropy(semantic_scores, semantic_labels, ignore_index=cfg.data.ignore_label)
        pt_offsets, coords, info, ids = d["pt_offsets"], d["locs"], d["instance_info"], d["instance_ids"]
        gt_offsets = info[:, 0:3] - coords
        pt_dist = torch.sum(torch.abs(pt_offsets - gt_offsets), dim=-1)
        valid = (ids != cfg.data.ignore_label).float()
        offset_norm_loss = torch.sum(pt_dist * valid) / (torch.sum(valid) + 1e-6)
        g_ = gt_offsets / (torch.norm(gt_offsets, p=2, dim=1).unsqueeze(-1) + 1e-8)
        p_ = pt_offsets / (torch.norm(pt_offsets, p=2, dim=1).unsqueeze(-1) + 1e-8)
        offset_dir_loss = torch.sum(-(g_ * p_).sum(-1) * valid) / (torch.sum(valid) + 1e-6)
        scores, proposals_idx, proposals_offset = d["proposal_scores"]
        ious = _t(pg.get_iou(proposals_idx[:, 1], proposals_offset, ids.numpy(), d["instance_num_point"].numpy()))
        gt_ious, _ = ious.max(1)
        fg, bg = cfg.train.fg_thresh, cfg.train.bg_thresh
        fg_mask, bg_mask = gt_ious > fg, gt_ious < bg
        interval = (fg_mask == 0) & (bg_mask == 0)
        gt_scores = (fg_mask > 0).float()
        gt_scores[interval] = gt_ious[interval] * (1 / (fg - bg)) + bg / (bg - fg)
        score_loss = F.binary_cross_entropy_with_logits(scores.view(-1), gt_scores, reduction="none").mean()
        w = cfg.train.loss_weight
        total = w[0] * semantic_loss + w[1] * offset_norm_loss + w[2] * offset_dir_loss + w[3] * score_loss
        d.update(semantic_loss=semantic_loss, offset_norm_loss=offset_norm_loss, offset_dir_loss=offset_dir_loss,
                 score_loss=score_loss, total_loss=total, gt_ious=gt_ious)
        return d


class _Voxelize(torch.autograd.Function):
    """PG_OP.voxelize_fp / voxelize_bp through the C oracle (reference: functions/pointgroup_ops.py:42-75)."""

    @staticmethod
    def forward(ctx, feats, v2p, mode):
        ctx.v2p, ctx.mode, ctx.N = v2p, mode, feats.shape[0]
        return _t(pg.voxelization(feats.detach().numpy(), v2p, mode))

    @staticmethod
    def backward(ctx, g):
        return _t(pg.voxelization_bp(g.contiguous().numpy(), ctx.v2p, ctx.N, ctx.mode)), None, None


class _RoiPool(torch.autograd.Function):
    """PG_OP.roipool_fp / roipool_bp through the C oracle (reference: functions/pointgroup_ops.py:185-221)."""

    @staticmethod
    def forward(ctx, feats, offsets):
        out, mx = pg.roipool(feats.detach().numpy(), offsets)
        ctx.mx, ctx.off, ctx.S = mx, offsets, feats.shape[0]
        return _t(out)

    @staticmethod
    def backward(ctx, g):
        return _t(pg.roipool_bp(g.contiguous().numpy(), ctx.off, ctx.mx, ctx.S)), None
