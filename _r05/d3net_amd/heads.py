"""Point-level head operators of the detector on MI355X (csrc/heads.hip): drop-in replacements for the three library
calls that dominate the heads at N ~ 165k points -- the weight gradient of a tall-skinny `nn.Linear`
(reference: model/pointgroup.py:77-85 `sem_seg`, `offset_net`), `F.cross_entropy` over (N, 20) logits (:389-390) and the
backward of the voxel -> point gather `output.features[p2v_map]` (:272).  Small inputs fall through to the library.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._lib import check
from .pointgroup_ops import _on, _ptr, _stream, _workspace

TALL_ROWS = 8192   # below this the library GEMM is fine


class _TallLinear(Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W.t()) if b is not None else x @ W.t()

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ W if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        db = torch.empty(W.size(0), dtype=W.dtype, device=W.device) if ctx.has_bias else None
        L = _lib.lib()
        O, I = W.shape
        ws = _workspace(L.d3_tall_wgrad_ws_bytes(I, O), x.device, "tallw")
        with _on(x.device):
            check(L.d3_tall_wgrad(_ptr(x), _ptr(dy), _ptr(dW), _ptr(db) if db is not None else None, x.size(0), I, O,
                                  _ptr(ws), ws.numel(), _stream()), "tall_wgrad")
        return dx, dW, db


def linear(module, x):
    """module(x) for an nn.Linear; tall-skinny inputs use the HIP weight-gradient kernel in the backward"""
    W = module.weight
    if x.is_cuda and x.dim() == 2 and x.size(0) >= TALL_ROWS and W.size(0) <= 32 and W.size(1) <= 32 \
            and x.dtype == torch.float32 and torch.is_grad_enabled():
        return _TallLinear.apply(x.contiguous(), W, module.bias)
    return module(x)


class _PointHeads(Function):
    """sem_seg + offset_net of PointGroup in three launches (csrc/heads.hip: d3_point_heads_fwd); the backward keeps the
    existing pieces: HIP weight gradients for the three tall Linear layers, library ops for the small data gradients and
    the batch-norm backward"""

    @staticmethod
    def forward(ctx, x, Ws, bs, W0, b0, gamma, beta, W3, b3, bn):
        ctx.set_materialize_grads(False)      # (an output nobody differentiates through arrives as None, not as a zero tensor: one fill launch less each)
        N, m = x.shape
        Cc = Ws.size(0)
        dev = x.device
        x = x.contiguous()
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        scores, h, y, off, stat = f(N, Cc), f(N, m), f(N, m), f(N, 3), f(2 * m)
        preds = torch.empty(N, dtype=torch.int64, device=dev)
        L = _lib.lib()
        ws = _workspace(L.d3_point_heads_ws_bytes(), dev, "ptheads")
        training = bool(bn.training or bn.running_mean is None)
        upd = training and bn.track_running_stats and bn.running_mean is not None
        with _on(dev):
            check(L.d3_point_heads_fwd(_ptr(x), N, m, Cc, _ptr(Ws.contiguous()), _ptr(bs.contiguous()), _ptr(W0.contiguous()),
                                       _ptr(b0.contiguous()), _ptr(gamma.contiguous()), _ptr(beta.contiguous()), _ptr(W3.contiguous()),
                                       _ptr(b3.contiguous()), float(bn.eps), float(bn.momentum), int(training),
                                       _ptr(bn.running_mean) if (upd or not training) else None,
                                       _ptr(bn.running_var) if (upd or not training) else None,
                                       _ptr(bn.num_batches_tracked) if upd and bn.num_batches_tracked is not None else None,
                                       _ptr(scores), _ptr(preds), _ptr(h), _ptr(y), _ptr(off), _ptr(stat), _ptr(ws), ws.numel(), _stream()),
                  "point_heads_fwd")
        ctx.save_for_backward(x, Ws, W0, gamma, W3, h, y, stat)
        ctx.bn_args = (training, float(bn.eps), bn.running_mean, bn.running_var)
        ctx.mark_non_differentiable(preds)
        return scores, preds, off

    @staticmethod
    def backward(ctx, g_s, _g_p, g_o):
        x, Ws, W0, gamma, W3, h, y, stat = ctx.saved_tensors
        training, eps, rm, rv = ctx.bn_args
        m = x.size(1)
        L = _lib.lib()
        dev = x.device

        def wgrad(inp, dy, O, I):
            dW = torch.empty((O, I), dtype=torch.float32, device=dev)
            db = torch.empty(O, dtype=torch.float32, device=dev)
            ws = _workspace(L.d3_tall_wgrad_ws_bytes(I, O), dev, "tallw")
            with _on(dev):
                check(L.d3_tall_wgrad(_ptr(inp), _ptr(dy), _ptr(dW), _ptr(db), inp.size(0), I, O, _ptr(ws), ws.numel(), _stream()),
                      "tall_wgrad")
            return dW, db

        N = x.size(0)
        dx = dh = None
        dWs = dbs = dW0 = db0 = dg = dbeta = dW3 = db3 = None
        if g_o is not None:
            g_o = g_o.contiguous()
            dW3, db3 = wgrad(y, g_o, 3, m)
            dy = torch.empty_like(y)
            with _on(dev):      # (g_o W3) * (y > 0) in one pass
                check(L.d3_point_heads_dy(_ptr(g_o), _ptr(W3.contiguous()), _ptr(y), N, _ptr(dy), _stream()), "point_heads_dy")
            dh, dg, dbeta = torch.ops.aten.native_batch_norm_backward(dy, h, gamma, rm, rv, stat[:m], stat[m:], training, eps,
                                                                      [True, True, True])
            dh = dh.contiguous()
            dW0, db0 = wgrad(x, dh, m, m)
        if g_s is not None:
            g_s = g_s.contiguous()
            dWs, dbs = wgrad(x, g_s, Ws.size(0), m)
        if ctx.needs_input_grad[0] and (dh is not None or g_s is not None):
            dx = torch.empty_like(x)
            with _on(dev):      # dh W0 + g_s Ws in one pass
                check(L.d3_point_heads_dx(_ptr(dh) if dh is not None else None, _ptr(W0.contiguous()),
                                          _ptr(g_s) if g_s is not None else None, _ptr(Ws.contiguous()), N, Ws.size(0), _ptr(dx),
                                          _stream()), "point_heads_dx")
        return dx, dWs, dbs, dW0, db0, dg, dbeta, dW3, db3, None


def point_heads(sem_seg, offset_net, x):
    """(semantic_scores, semantic_preds, pt_offsets) of PointGroup's two point-level heads (model/pointgroup.py:77-85,
    277-283); the fused HIP path for the shipped shape (m = 16, <= 32 classes, Linear-BN-ReLU-Linear offset head)"""
    on = offset_net
    bn = on[1]
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(1) == 16 and x.size(0) >= TALL_ROWS
            and isinstance(bn, torch.nn.BatchNorm1d) and bn.affine and bn.momentum is not None and sem_seg.weight.size(0) <= 32
            and on[0].weight.shape == (16, 16) and on[3].weight.shape == (3, 16) and sem_seg.bias is not None
            and on[0].bias is not None and on[3].bias is not None and isinstance(on[2], torch.nn.ReLU)):
        return _PointHeads.apply(x, sem_seg.weight, sem_seg.bias, on[0].weight, on[0].bias, bn.weight, bn.bias, on[3].weight,
                                 on[3].bias, bn)
    semantic_scores = linear(sem_seg, x)
    return semantic_scores, semantic_scores.max(1)[1], linear(on[3], on[2](on[1](linear(on[0], x))))


class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, z, label, ignore_index):
        z = z.contiguous()
        N, Cc = z.shape
        grad = torch.empty_like(z)
        out = torch.empty(2, dtype=torch.float32, device=z.device)
        L = _lib.lib()
        ws = _workspace(L.d3_cross_entropy_ws_bytes(), z.device, "ce")
        with _on(z.device):
            check(L.d3_cross_entropy(_ptr(z), _ptr(label), _ptr(grad), _ptr(out), N, Cc, int(ignore_index), _ptr(ws),
                                     ws.numel(), _stream()), "cross_entropy")
        ctx.save_for_backward(grad, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        grad, out = ctx.saved_tensors
        return grad * (g / out[1].clamp(min=1.0)), None, None


def cross_entropy(z, label, ignore_index=-100):
    """nn.functional.cross_entropy(z, label, ignore_index=...) (mean reduction) in one HIP pass for big (N, C<=64) logits"""
    if z.is_cuda and z.dim() == 2 and z.size(0) >= TALL_ROWS and z.size(1) <= 64 and z.dtype == torch.float32 \
            and label.dtype == torch.int64:
        return _CrossEntropy.apply(z, label.contiguous(), ignore_index)
    return torch.nn.functional.cross_entropy(z, label, ignore_index=ignore_index)


def _gather_rows(feats, idx):
    """feats[idx] (idx int64) through d3_gather_rows; library index_select for shapes the kernel does not take"""
    if not (feats.is_cuda and feats.dim() == 2 and feats.dtype == torch.float32 and feats.is_contiguous() and feats.size(1) % 4 == 0
            and idx.dtype == torch.int64 and idx.dim() == 1):
        return feats.index_select(0, idx)
    out = torch.empty((idx.numel(), feats.size(1)), dtype=feats.dtype, device=feats.device)
    with _on(feats.device):
        check(_lib.lib().d3_gather_rows(_ptr(feats), _ptr(idx.contiguous()), _ptr(out), idx.numel(), feats.size(1), _stream()), "gather_rows")
    return out


class _Devoxelize(Function):
    """feats[p2v] with the backward as a rule-ordered per-voxel sum over v2p (deterministic, no atomics) instead of the
    library's sort-based index_put backward"""

    @staticmethod
    def forward(ctx, feats, p2v, v2p):
        ctx.save_for_backward(v2p)
        ctx.M = feats.size(0)
        return _gather_rows(feats, p2v)

    @staticmethod
    def backward(ctx, dpt):
        (v2p,) = ctx.saved_tensors
        dpt = dpt.contiguous()
        M, Cc = ctx.M, dpt.size(1)
        dv = torch.zeros((M, Cc), dtype=dpt.dtype, device=dpt.device)
        with _on(dpt.device):
            check(_lib.lib().d3_point_recover_bp(_ptr(dpt), _ptr(dv), _ptr(v2p), M, v2p.size(1) - 1, Cc, _stream()),
                  "point_recover_bp")
        return dv, None, None


def devoxelize(feats, p2v, v2p):
    if feats.is_cuda and v2p is not None and v2p.dtype == torch.int32 and v2p.is_contiguous() and feats.dtype == torch.float32:
        return _Devoxelize.apply(feats, p2v.long(), v2p)
    return feats[p2v.long()]


class _OffsetLoss(Function):
    @staticmethod
    def forward(ctx, pt_offsets, coords, instance_info, instance_ids, ignore_label):
        pt = pt_offsets.contiguous()
        N = pt.size(0)
        g1, g2 = torch.empty_like(pt), torch.empty_like(pt)
        out = torch.empty(3, dtype=torch.float32, device=pt.device)
        L = _lib.lib()
        ws = _workspace(L.d3_offset_loss_ws_bytes(), pt.device, "ol")
        with _on(pt.device):
            check(L.d3_offset_loss(_ptr(pt), _ptr(coords), _ptr(instance_info), instance_info.size(1), _ptr(instance_ids),
                                   int(ignore_label), _ptr(g1), _ptr(g2), _ptr(out), N, _ptr(ws), ws.numel(), _stream()),
                  "offset_loss")
        ctx.save_for_backward(g1, g2, out)
        return out

    @staticmethod
    def backward(ctx, g):
        g1, g2, out = ctx.saved_tensors
        den = out[2] + 1e-6
        return torch.addcmul(g1 * (g[0] / den), g2, g[1] / den), None, None, None, None


def offset_losses(pt_offsets, coords, instance_info, instance_ids, ignore_label):
    """(offset_norm_loss, offset_dir_loss, sum(valid)) of PointGroup.loss (reference: model/pointgroup.py:397-420)"""
    if pt_offsets.is_cuda and pt_offsets.size(0) >= TALL_ROWS and pt_offsets.dtype == torch.float32 \
            and coords.is_contiguous() and instance_info.is_contiguous() and instance_ids.dtype == torch.int64 \
            and coords.dtype == torch.float32 and instance_info.dtype == torch.float32:
        o = _OffsetLoss.apply(pt_offsets, coords, instance_info, instance_ids.contiguous(), ignore_label)
        return o[0], o[1], o[2].detach()
    gt_offsets = instance_info[:, 0:3] - coords
    pt_dist = torch.sum(torch.abs(pt_offsets - gt_offsets), dim=-1)
    valid = (instance_ids != ignore_label).float()
    norm_loss = torch.sum(pt_dist * valid) / (torch.sum(valid) + 1e-6)
    gt_ = gt_offsets / (torch.norm(gt_offsets, p=2, dim=1).unsqueeze(-1) + 1e-8)
    pt_ = pt_offsets / (torch.norm(pt_offsets, p=2, dim=1).unsqueeze(-1) + 1e-8)
    dir_loss = torch.sum(-(gt_ * pt_).sum(-1) * valid) / (torch.sum(valid) + 1e-6)
    return norm_loss, dir_loss, valid.sum()


class _ScoreLoss(Function):
    @staticmethod
    def forward(ctx, scores, ious, fg, bg):
        x = scores.reshape(-1).contiguous()
        P = x.numel()
        gt_iou, ds = torch.empty_like(x), torch.empty_like(x)
        out = torch.empty(1 + P, dtype=torch.float32, device=x.device)   # [loss | per-proposal terms]
        with _on(x.device):
            check(_lib.lib().d3_score_loss(_ptr(x), _ptr(ious), P, ious.size(1), float(fg), float(bg), _ptr(gt_iou), _ptr(ds),
                                           _ptr(out), _stream()), "score_loss")
        ctx.save_for_backward(ds)
        ctx.shape = scores.shape
        ctx.mark_non_differentiable(gt_iou)
        return out[0], gt_iou

    @staticmethod
    def backward(ctx, g, _):
        (ds,) = ctx.saved_tensors
        return (ds * g).view(ctx.shape), None, None, None


def score_loss(scores, ious, fg_thresh, bg_thresh):
    """(score_loss, gt_ious) of PointGroup.loss (reference: model/pointgroup.py:436-452): row maxima of the IoU matrix,
    soft labels between the two thresholds, mean BCE-with-logits."""
    if scores.is_cuda and scores.dtype == torch.float32 and ious.dtype == torch.float32 and ious.dim() == 2 \
            and ious.size(1) > 0 and scores.numel() > 0 and ious.is_contiguous():
        return _ScoreLoss.apply(scores, ious, fg_thresh, bg_thresh)
    gt_ious, _ = ious.max(1)
    fg_mask = gt_ious > fg_thresh
    bg_mask = gt_ious < bg_thresh
    k = 1 / (fg_thresh - bg_thresh)
    b = bg_thresh / (bg_thresh - fg_thresh)
    gt_scores = torch.where(~fg_mask & ~bg_mask, gt_ious * k + b, fg_mask.float())
    return torch.nn.functional.binary_cross_entropy_with_logits(scores.view(-1), gt_scores, reduction="none").mean(), gt_ious


class _StackToBatch(Function):
    @staticmethod
    def forward(ctx, pf, scores, crop, bids, perm, center_label, B, K):
        ctx.set_materialize_grads(False)      # (an output nobody differentiates through arrives as None, not as a zero tensor: one fill launch less each)
        P, m = pf.shape
        dev = pf.device
        pf_c, sc_c = pf.contiguous(), scores.contiguous()
        flat = torch.zeros(B * K * (m + 24 + 3 + 3), dtype=torch.float32, device=dev)   # one fill for the six outputs
        o = 0
        outs = []
        for w in (m, 24, 3, 1, 1, 1):
            outs.append(flat[o:o + B * K * w]); o += B * K * w
        slot = torch.empty(max(P, 1), dtype=torch.int64, device=dev)
        G = 0 if center_label is None else center_label.size(1)
        assign = torch.empty((B, K), dtype=torch.int64, device=dev) if G > 0 else None
        with _on(dev):
            check(_lib.lib().d3_stack_to_batch(_ptr(pf_c), _ptr(crop), _ptr(sc_c), _ptr(bids), _ptr(perm),
                                               _ptr(center_label) if G > 0 else None, G, P, m, B, K, *[_ptr(t) for t in outs],
                                               _ptr(slot), _ptr(assign) if G > 0 else None, _stream()), "stack_to_batch")
        ctx.save_for_backward(slot[:P])
        ctx.dims = (P, m)
        res = (outs[0].view(B, K, m), outs[1].view(B, K, 8, 3), outs[2].view(B, K, 3), outs[3].view(B, K), outs[4].view(B, K),
               outs[5].view(B, K), assign if assign is not None else torch.empty(0, dtype=torch.int64, device=dev))
        ctx.mark_non_differentiable(res[1], res[2], res[3], res[5], res[6])
        return res

    @staticmethod
    def backward(ctx, g_feats, g_bbox, g_center, g_sem, g_scores, g_mask, g_assign):
        (slot,) = ctx.saved_tensors
        P, m = ctx.dims
        ok = slot >= 0
        idx = slot.clamp(min=0)
        gp = g_feats.reshape(-1, m).index_select(0, idx) * ok.unsqueeze(1).to(g_feats.dtype) if g_feats is not None else None
        gs = g_scores.reshape(-1).index_select(0, idx) * ok.to(g_scores.dtype) if g_scores is not None else None
        return gp, gs, None, None, None, None, None, None


def stack_to_batch(pf, scores, crop, bids, perm, center_label, B, K):
    """Fused PointGroup.convert_stack_to_batch + get_object_assignments (reference model/pointgroup.py:216-263); returns
    None when the shapes are outside the kernel's limits (the caller then uses the library-op path).
    -> (feats (B,K,m), corners (B,K,8,3), centres (B,K,3), sem_cls, scores, mask (B,K), object_assignment (B,K) or empty)"""
    if not (pf.is_cuda and pf.dtype == torch.float32 and crop.dtype == torch.float32 and crop.is_contiguous() and crop.size(1) == 9
            and scores.dtype == torch.float32 and bids.dtype == torch.int32 and bids.is_contiguous()
            and perm.dtype == torch.int64 and perm.is_contiguous() and pf.size(0) <= 4096 and B * K <= 8192):
        return None
    if center_label is not None and not (center_label.is_cuda and center_label.dtype == torch.float32
                                         and center_label.is_contiguous() and center_label.dim() == 3 and center_label.size(0) == B):
        return None
    return _StackToBatch.apply(pf, scores, crop, bids, perm, center_label, B, K)


class _GatherRows(Function):
    """feats[idx] whose backward is one atomic scatter-add launch (deterministic for <= 2 addends per row: the cluster
    feature gather, where a point is in at most one cluster of each of the two cluster sets)"""

    @staticmethod
    def forward(ctx, feats, idx):
        ctx.save_for_backward(idx)
        ctx.rows = feats.size(0)
        return _gather_rows(feats, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.zeros((ctx.rows, g.size(1)), dtype=g.dtype, device=g.device)
        with _on(g.device):
            check(_lib.lib().d3_scatter_add_rows(_ptr(g), _ptr(idx), _ptr(out), g.size(0), g.size(1), _stream()), "scatter_add_rows")
        return out, None


def gather_cluster_rows(feats, idx):
    """feats[idx] for the cluster feature gather (idx int64, every row index at most twice)"""
    if feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and feats.requires_grad:
        return _GatherRows.apply(feats, idx.contiguous())
    return feats[idx]
