"""Listener (grounding) head on MI355X: `LangModule`, `MultiHeadAttention`, `TransformerMatchModule`, `ListenerNet`
with the reference's constructors, `data_dict` keys and state-dict layout
(reference: model/lang_module.py:8-178, model/transformer/attention.py:7-77,134-176, model/match_module.py:143-336,
model/listener.py:10-54).

What runs where: projections / 1x1 convs / LayerNorm / BatchNorm1d are plain library GEMMs and elementwise ops
(hipBLASLt / MIOpen through torch); the attention core (scores + distance bias + key mask + softmax + PV, forward and
backward) is the hand-written kernel `d3_attn_fwd/bwd` (csrc/attention.hip), which consumes the UN-replicated distance
weights and (B,T) masks instead of the (B*C,4,128,128) copies the reference builds with `.repeat`
(model/match_module.py:191-197,324-326), on fp32 MFMA tiles.  The packed-sequence GRU of `LangModule` is the native
`d3_gru_seq_forward/backward` (csrc/topdown.hip; `LangModule.native`).
"""
import ctypes as C
import math
import random

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import _lib
from . import nativelinear as NL
from ._lib import check
from .pointgroup_ops import _on, _ptr, _stream


# ------------------------------------------------------------------------------------ attention core
class AttentionCoreFunction(Function):
    """softmax(q k^T / sqrt(dk) + bias, key-masked) v on (B, n, h*d) projections."""

    @staticmethod
    def forward(ctx, q, k, v, bias, mask, h, bias_div):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, nq, hdk = q.shape
        nk = k.shape[1]
        dk, dv = hdk // h, v.shape[2] // h
        out = torch.empty((B, nq, h * dv), dtype=torch.float32, device=q.device)
        P = torch.empty((B, h, nq, nk), dtype=torch.float32, device=q.device)
        if bias is not None:
            bias = bias.contiguous()
            assert bias.shape == (B // bias_div, h, nq, nk) and B % bias_div == 0
        if mask is not None:
            mask = mask.contiguous().float()
            assert mask.shape == (B, nk)
        with _on(q.device):
            check(_lib.lib().d3_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(bias) if bias is not None else None,
                                         _ptr(mask) if mask is not None else None, _ptr(out), _ptr(P), B, h, nq, nk,
                                         dk, dv, bias_div, _stream()), "attn_fwd")
        ctx.save_for_backward(q, k, v, P)
        ctx.dims = (B, h, nq, nk, dk, dv)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, P = ctx.saved_tensors
        B, h, nq, nk, dk, dv = ctx.dims
        dout = dout.contiguous()
        dq, dk_, dv_ = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dS = torch.empty_like(P)
        with _on(q.device):
            check(_lib.lib().d3_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(P), _ptr(dout), _ptr(dS), _ptr(dq), _ptr(dk_),
                                         _ptr(dv_), B, h, nq, nk, dk, dv, _stream()), "attn_bwd")
        return dq, dk_, dv_, None, None, None, None   # the distance weights are detached in the reference


class ScaledDotProductAttention(nn.Module):
    """(reference: model/transformer/attention.py:7-77)"""

    def __init__(self, d_model, d_k, d_v, h):
        super().__init__()
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        for fc in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def forward(self, queries, keys, values, key_mask=None, attention_weights=None, way="add", weights_div=1):
        """key_mask (B, nk) with 0 = masked (the reference passes its (B,h,nq,nk) replica);
        attention_weights (B/weights_div, h, nq, nk), added to the scaled scores (way == "add")."""
        if attention_weights is not None and way != "add":
            raise NotImplementedError("only the additive weights the reference uses (match_module.py:238) are implemented")
        # the three projections share one launch, forward and backward (d3net_amd/nativelinear.py over csrc/hgemm.hip)
        q, k, v = NL.linear_multi([(queries, self.fc_q.weight, self.fc_q.bias), (keys, self.fc_k.weight, self.fc_k.bias),
                                   (values, self.fc_v.weight, self.fc_v.bias)])
        out = AttentionCoreFunction.apply(q, k, v, attention_weights, key_mask, self.h, weights_div)
        return NL.linear(out, self.fc_o.weight, self.fc_o.bias)


class MultiHeadAttention(nn.Module):
    """post-LN residual attention layer with dropout (reference: model/transformer/attention.py:134-176)."""

    def __init__(self, d_model, d_k, d_v, h, dropout=.1):
        super().__init__()
        self.attention = ScaledDotProductAttention(d_model=d_model, d_k=d_k, d_v=d_v, h=h)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, queries, keys, values, key_mask=None, attention_weights=None, way="add", weights_div=1):
        out = self.attention(queries, keys, values, key_mask, attention_weights, way, weights_div)
        return NL.add_layer_norm(queries, self.dropout(out), self.layer_norm)      # LayerNorm(queries + dropout(out)), one pass


# ------------------------------------------------------------------------------------------ language
class GRUSeqFunction(torch.autograd.Function):
    """nn.GRU over a packed batch (model/lang_module.py:51-55) as one native call each way (csrc/topdown.hip:
    d3_gru_seq_forward / _backward): x (N,T,I), lens (N) -> hiddens (N,T,H) zero-padded, last (N,H)."""

    @staticmethod
    def forward(ctx, x, lens, Wih, Whh, bih, bhh):
        L = _lib.lib()
        x, Wih, Whh, bih, bhh = (t.contiguous() for t in (x, Wih, Whh, bih, bhh))
        lens32 = lens.to(device=x.device, dtype=torch.int32).contiguous()
        N, T, I = x.shape
        H = Whh.shape[1]
        hiddens = torch.empty((N, T, H), dtype=torch.float32, device=x.device)
        last = torch.empty((N, H), dtype=torch.float32, device=x.device)
        ws = torch.empty(L.d3_gru_seq_ws_bytes(N, T, I, H), dtype=torch.uint8, device=x.device)
        with _on(x.device):
            check(L.d3_gru_seq_forward(_ptr(x), _ptr(lens32), _ptr(Wih), _ptr(Whh), _ptr(bih), _ptr(bhh), N, T, I, H, _ptr(hiddens),
                                       _ptr(last), _ptr(ws), ws.numel(), _stream()), "gru_seq_forward")
        ctx.save_for_backward(x, lens32, Wih, Whh, ws)
        ctx.dims = (N, T, I, H)
        return hiddens, last

    @staticmethod
    def backward(ctx, d_hiddens, d_last):
        L = _lib.lib()
        x, lens32, Wih, Whh, ws = ctx.saved_tensors
        N, T, I, H = ctx.dims
        dev = x.device
        d_hiddens = d_hiddens.contiguous() if d_hiddens is not None else None
        d_last = d_last.contiguous() if d_last is not None else None
        dWih, dWhh = torch.empty_like(Wih), torch.empty_like(Whh)
        dbih, dbhh = torch.empty(3 * H, dtype=torch.float32, device=dev), torch.empty(3 * H, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws2 = torch.empty(L.d3_gru_seq_bwd_ws_bytes(N, T, I, H), dtype=torch.uint8, device=dev)
        with _on(dev):
            check(L.d3_gru_seq_backward(_ptr(x), _ptr(lens32), _ptr(Wih), _ptr(Whh), N, T, I, H,
                                        _ptr(d_hiddens) if d_hiddens is not None else None, _ptr(d_last) if d_last is not None else None,
                                        _ptr(ws), _ptr(dWih), _ptr(dWhh), _ptr(dbih), _ptr(dbhh), _ptr(dx) if dx is not None else None,
                                        _ptr(ws2), ws2.numel(), _stream()), "gru_seq_backward")
        return dx, None, dWih, dWhh, dbih, dbhh


class LangModule(nn.Module):
    """GRU description encoder + language classifier (reference: model/lang_module.py:8-178)."""

    def __init__(self, cfg, emb_size=300, hidden_size=256):
        super().__init__()
        self.num_text_classes = cfg.model.num_bbox_class
        self.use_lang_classifier = cfg.model.use_lang_classifier
        self.use_bidir = cfg.model.use_bidir
        self.emb_size, self.hidden_size = emb_size, hidden_size
        self.gru = nn.GRU(input_size=emb_size, hidden_size=hidden_size, batch_first=True, bidirectional=self.use_bidir)
        self.native = True    # csrc/topdown.hip packed-sequence GRU; False: nn.GRU through the BLAS / MIOpen libraries (tests)
        if self.use_lang_classifier:
            self.lang_cls = nn.Sequential(nn.Linear(hidden_size, self.num_text_classes), nn.Dropout())

    def _encode(self, word_embs, lang_len):
        """(B,C,T,300), (B,C) -> padded hiddens (B*C,T,H), last hidden (B*C,H), mask (B*C,T), scores"""
        B, Cn, T, _ = word_embs.shape
        embs = word_embs.reshape(-1, T, self.emb_size)
        lens = lang_len.reshape(-1)
        if self.native and embs.is_cuda:
            # one native call: no host copy of the lengths, no packing / unpacking, hiddens already zero-padded to T (:67-68)
            g = self.gru
            pad, last = GRUSeqFunction.apply(embs.float(), lens, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
            steps = torch.arange(T, device=lens.device).unsqueeze(0)
            if self.use_bidir:
                # the reverse direction (:15-24, 58-61) is the same recurrence over every description read backwards: position t of
                # sample n <-> position len_n - 1 - t (padding stays where it is); its final state is the one at the first word.
                # The two directions are averaged, as the reference does.
                L_ = lens.clamp(max=T).unsqueeze(1).to(steps.dtype)      # (a length beyond T would gather out of range)
                rev = torch.where(steps < L_, L_ - 1 - steps, steps)                       # (N, T), an involution
                gidx = rev.unsqueeze(-1)
                e_rev = embs.float().gather(1, gidx.expand(-1, -1, self.emb_size))
                pad_r, last_r = GRUSeqFunction.apply(e_rev, lens, g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse,
                                                     g.bias_hh_l0_reverse)
                pad = (pad + pad_r.gather(1, gidx.expand(-1, -1, self.hidden_size))) / 2
                last = (last + last_r) / 2
            masks = (steps < lens.unsqueeze(1)).float()
            scores = self.lang_cls(last) if self.use_lang_classifier else None
            return pad, last, masks, scores
        packed = pack_padded_sequence(embs, lens.cpu(), batch_first=True, enforce_sorted=False)
        hiddens, last = self.gru(packed)
        hiddens, _ = pad_packed_sequence(hiddens, batch_first=True)
        last = last.permute(1, 0, 2).contiguous().flatten(start_dim=1)
        if self.use_bidir:  # average the two directions (:59-61)
            H = self.hidden_size
            hiddens = (hiddens[:, :, :H] + hiddens[:, :, H:]) / 2
            last = (last[:, :H] + last[:, H:]) / 2
        pad = hiddens.new_zeros(B * Cn, T, self.hidden_size)
        pad[:, :hiddens.shape[1]] = hiddens                                  # zero padding up to T (:67-68)
        masks = (torch.arange(T, device=lens.device).unsqueeze(0) < lens.unsqueeze(1)).float()
        scores = self.lang_cls(last) if self.use_lang_classifier else None
        return pad, last, masks, scores

    def forward(self, data_dict, use_rl=False):
        if use_rl:
            s = self._encode(data_dict["lang_feat"]["sampled"], data_dict["lang_len"]["sampled"])
            with torch.no_grad():
                b = self._encode(data_dict["lang_feat"]["baseline"], data_dict["lang_len"]["baseline"])
            for key, i in (("lang_hiddens", 0), ("lang_emb", 1), ("lang_masks", 2), ("lang_scores", 3)):
                data_dict[key] = {"sampled": s[i], "baseline": b[i]}
        else:
            hid, last, masks, scores = self._encode(data_dict["lang_feat"], data_dict["lang_len"])
            data_dict["lang_masks"], data_dict["lang_hiddens"], data_dict["lang_emb"] = masks, hid, last
            if self.use_lang_classifier:
                data_dict["lang_scores"] = scores
        return data_dict


# --------------------------------------------------------------------------------------------- match
import os as _os
_CONV1D_LIB = _os.environ.get("D3_CONV1D_LIB") == "1"      # (A/B measurements: the convolution-library path)


class PointwiseConv1d(nn.Conv1d):
    """`nn.Conv1d(cin, cout, 1)` with the same parameters and state-dict keys, computed as the channel GEMM it is.  Through the
    convolution library a kernel-size-1 Conv1d on (B*C, 128, 128) runs a generic convolution forward (84-360 us per call) and a
    NAIVE weight-gradient kernel (`naive_conv_ab_nonpacked_wrw...`: 0.33 ms per call in the listener step, 1.85 ms in the
    joint one: profiles/r02_ac_kernel_stats_*.csv); as a matmul it is a few microseconds of batched GEMM either way."""

    def forward(self, x):
        if _CONV1D_LIB:
            return super().forward(x)
        y = torch.matmul(self.weight.squeeze(-1), x)            # (cout, cin) @ (B, cin, L) -> (B, cout, L)
        return y if self.bias is None else y + self.bias.view(1, -1, 1)

    def forward_channels_last(self, x):
        """x (B, L, cin) -> (B, L, cout): the same product with the channels in the last dimension -- ONE tall GEMM over the
        B * L positions on csrc/hgemm.hip instead of a batched matmul per item"""
        return NL.linear(x, self.weight.squeeze(-1), self.bias)


def pointwise_stack_channels_last(seq, x):
    """run an nn.Sequential of PointwiseConv1d / BatchNorm1d / PReLU (model/match_module.py:160-169 `features_concat`,
    `match`) on a channels-LAST tensor (B, L, C): BatchNorm1d over (B, C, L) normalises every channel over the B * L
    positions -- nn.BatchNorm1d on the (B * L, C) rows; PReLU's per-channel slope broadcasts over the last dimension."""
    B, Lp = x.shape[:2]
    for m in seq:
        if isinstance(m, PointwiseConv1d):
            x = m.forward_channels_last(x)
        elif isinstance(m, nn.BatchNorm1d):
            x = m(x.reshape(B * Lp, -1)).view(B, Lp, -1)
        elif isinstance(m, nn.PReLU):
            x = torch.nn.functional.prelu(x.reshape(B * Lp, -1), m.weight).view(B, Lp, -1)
        else:
            raise NotImplementedError(type(m))
    return x


class TransformerMatchModule(nn.Module):
    """(reference: model/match_module.py:143-336)"""

    def __init__(self, cfg, lang_size=256, hidden_size=128, head=4, depth=2, use_dist_weight_matrix=True):
        super().__init__()
        self.use_dist_weight_matrix = use_dist_weight_matrix
        self.num_proposals = cfg.model.max_num_proposal
        self.lang_size, self.hidden_size, self.head = lang_size, hidden_size, head
        self.depth = depth - 1
        self.det_channel = cfg.model.m
        self.chunk_size = cfg.data.num_des_per_scene
        self.features_concat = nn.Sequential(
            PointwiseConv1d(self.det_channel, hidden_size, 1), nn.BatchNorm1d(hidden_size), nn.PReLU(hidden_size),
            PointwiseConv1d(hidden_size, hidden_size, 1))
        self.match = nn.Sequential(
            PointwiseConv1d(hidden_size, hidden_size, 1), nn.BatchNorm1d(hidden_size), nn.PReLU(),
            PointwiseConv1d(hidden_size, hidden_size, 1), nn.BatchNorm1d(hidden_size), nn.PReLU(),
            PointwiseConv1d(hidden_size, 1, 1))
        self.lang_fc = nn.Sequential(nn.Linear(lang_size, hidden_size), nn.ReLU(), nn.Dropout(p=0.1), nn.LayerNorm(hidden_size))
        self.lang_self_attn = MultiHeadAttention(d_model=hidden_size, d_k=16, d_v=16, h=head)
        dh = hidden_size // head
        self.self_attn = nn.ModuleList(MultiHeadAttention(hidden_size, dh, dh, head) for _ in range(depth))
        self.cross_attn = nn.ModuleList(MultiHeadAttention(hidden_size, dh, dh, head) for _ in range(depth))

    def multiplex_attention(self, v_features, l_features, l_masks, dist_weights, weights_div):
        """v (B*C,K,128), l (B*C,T,256), l_masks (B*C,T), dist_weights (B',h,K,K) shared by weights_div items"""
        if _CONV1D_LIB:
            l_features = self.lang_fc(l_features)
        else:   # Linear -> ReLU (one hgemm problem with the ReLU epilogue) -> Dropout -> LayerNorm (csrc/layernorm.hip)
            fc, _, drop, ln = self.lang_fc
            l_features = NL.add_layer_norm(drop(NL.linear(l_features, fc.weight, fc.bias, relu=True)), None, ln)
        l_features = self.lang_self_attn(l_features, l_features, l_features, key_mask=l_masks)
        v_features = self.cross_attn[0](v_features, l_features, l_features, key_mask=l_masks)
        for i in range(self.depth):
            v_features = self.self_attn[i + 1](v_features, v_features, v_features, attention_weights=dist_weights,
                                               weights_div=weights_div)
            v_features = self.cross_attn[i + 1](v_features, l_features, l_features, key_mask=l_masks)
        if _CONV1D_LIB:
            return self.match(v_features.permute(0, 2, 1).contiguous()).squeeze(1)       # (B*C, K)
        return pointwise_stack_channels_last(self.match, v_features).squeeze(-1)          # (B*C, K), no transposes

    def _dist_weights(self, centers):
        """row-normalised inverse centre distances, one copy per head (:220-238); detached"""
        diff = centers[:, None, :, :] - centers[:, :, None, :]
        dist = torch.sqrt(torch.sum(diff.pow(2), dim=-1))[:, None, :, :]
        w = 1 / (dist + 1e-2)
        w = w / torch.sum(w, dim=2, keepdim=True)
        return w.expand(-1, self.head, -1, -1).contiguous().detach()

    def _copy_paste(self, features, obj_masks):
        """train-time augmentation: fill the empty proposal slots of each scene with real proposal features of the
        batch (:266-291), same indexing as the reference."""
        B, K = obj_masks.shape
        out = features.clone()
        lens = obj_masks.sum(1)
        flat = features.reshape(B * K, -1)[obj_masks.reshape(-1)]
        total = flat.shape[0]
        pool = flat.repeat(2, 1)
        j = 0
        for i in range(B):
            empty = torch.where(~obj_masks[i])[0]
            n_i = int(lens[i])
            j += n_i
            n = empty.shape[0] if empty.shape[0] < total - n_i else total - n_i
            out[i, empty[:n], :] = pool[j:j + n, :]
        return out

    def forward(self, data_dict, use_rl=False):
        centers = data_dict["proposal_center_batched"]
        dist_weights = self._dist_weights(centers) if self.use_dist_weight_matrix else None
        if _CONV1D_LIB:
            feats = self.features_concat(data_dict["proposal_feats_batched"].permute(0, 2, 1)).permute(0, 2, 1)
        else:
            feats = pointwise_stack_channels_last(self.features_concat, data_dict["proposal_feats_batched"])
        B, K = feats.shape[:2]
        masks = data_dict["proposal_batch_mask"].float()
        feats = self.self_attn[0](feats, feats, feats, attention_weights=dist_weights)     # no proposal mask (:260)
        data_dict["random"] = random.random()
        feature0 = feats
        if data_dict["istrain"][0] == 1 and data_dict["random"] < 0.5:
            feature0 = self._copy_paste(feats, masks.bool())
        Cn = self.chunk_size
        if use_rl:
            topn = data_dict["sampled_topn"]
            v = feature0.unsqueeze(1).repeat(1, topn * Cn, 1, 1).reshape(-1, K, self.hidden_size)
            div = topn * Cn
            sampled = self.multiplex_attention(v, data_dict["lang_hiddens"]["sampled"], data_dict["lang_masks"]["sampled"],
                                               dist_weights, div)
            with torch.no_grad():
                baseline = self.multiplex_attention(v, data_dict["lang_hiddens"]["baseline"],
                                                    data_dict["lang_masks"]["baseline"], dist_weights, div)
            data_dict["cluster_ref"] = {"sampled": sampled, "baseline": baseline}
        else:
            v = feature0[:, None].repeat(1, Cn, 1, 1).reshape(-1, K, self.hidden_size)
            data_dict["cluster_ref"] = self.multiplex_attention(v, data_dict["lang_hiddens"], data_dict["lang_masks"],
                                                                dist_weights, Cn)
        return data_dict


class ListenerNet(nn.Module):
    """(reference: model/listener.py:10-54)"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.match_type = cfg.model.match_type
        self.lang = LangModule(cfg)
        if self.match_type != "Transformer":
            raise NotImplementedError("only match_type: Transformer (the shipped default, conf/pointgroup.yaml:79)")
        self.match = TransformerMatchModule(cfg)

    def forward(self, data_dict, use_rl=False):
        data_dict = self.lang(data_dict, use_rl)
        return self.match(data_dict, use_rl)


# ---------------------------------------------------------------------------------------------- loss
def aabb_iou_to_gt(pred_corners, gt_corners):
    """AABB IoU of every proposal box (N,K,8,3) with its sample's GT box (N,8,3) -> (N,K)
    (lib/utils/bbox.py:247-271 get_aabb3d_iou_batch), on the device."""
    pmin, pmax = pred_corners.min(2)[0], pred_corners.max(2)[0]
    gmin, gmax = gt_corners.min(1)[0].unsqueeze(1), gt_corners.max(1)[0].unsqueeze(1)
    inter = (torch.minimum(pmax, gmax) - torch.maximum(pmin, gmin)).clamp(min=0).prod(-1)
    vol_p = (pmax - pmin).prod(-1)
    vol_g = (gmax - gmin).prod(-1)
    return inter / (vol_p + vol_g - inter + 1e-8)


def softmax_ranking_loss(inputs, targets, reduce=True):
    """lib/grounding/loss.py:6-25 (the 1e-8 inside and outside the softmax included)"""
    probs = torch.softmax(inputs + 1e-8, dim=1)
    loss = -torch.sum(torch.log(probs + 1e-8) * targets, dim=1)
    return loss.mean() if reduce else loss


def _pseudo_gt(data_dict, N, K, repeat):
    """proposal boxes repeated to the N description rows and their IoU with the referred boxes; the pseudo-GT is the
    proposal with the highest IoU (lib/grounding/loss_helper.py:47-58, 148-158) -- without the per-sample host loops"""
    corners = data_dict["proposal_bbox_batched"]
    corners = corners.unsqueeze(1).repeat(1, repeat, 1, 1, 1).reshape(N, K, 8, 3)
    gt = data_dict["ref_box_corner_label"].reshape(N, 8, 3)
    ious = aabb_iou_to_gt(corners, gt)
    return ious, ious.argmax(1)


def get_grounding_loss(data_dict, is_frozen=False, use_rl=False):
    """lib/grounding/loss_helper.py:12-229 (loss="cross_entropy"): softmax ranking loss against the pseudo-GT, accuracy
    and IoU rates.  use_rl: `cluster_ref` holds the listener's scores for the sampled and the greedy (baseline)
    captions; both unreduced losses are kept for the speaker's reward and the sampled one is the listener's loss."""
    if use_rl:
        sampled, baseline = data_dict["cluster_ref"]["sampled"], data_dict["cluster_ref"]["baseline"]
        N, K = sampled.shape
        B = data_dict["proposal_bbox_batched"].shape[0]
        ious, label_idx = _pseudo_gt(data_dict, N, K, N // B)     # rows: (scene, sample, chunk); boxes depend on scene only
        labels = torch.zeros_like(sampled).scatter_(1, label_idx.unsqueeze(1), 1.0)
        data_dict["cluster_labels"] = labels
        s_loss = softmax_ranking_loss(sampled, labels, reduce=False)
        b_loss = softmax_ranking_loss(baseline, labels, reduce=False)
        s_idx, b_idx = sampled.argmax(-1), baseline.argmax(-1)
        rows = torch.arange(N, device=sampled.device)
        s_ious, best_ious = ious[rows, s_idx], ious[rows, label_idx]
        data_dict["ref_loss"] = s_loss.mean()
        data_dict["ref_sampled_loss"], data_dict["ref_baseline_loss"] = s_loss, b_loss
        data_dict["ref_acc_mean"] = data_dict["ref_sampled_acc"] = (s_idx == label_idx).sum().float() / N
        data_dict["ref_sampled_acc_all"] = (s_idx == label_idx).float()
        data_dict["ref_baseline_acc"] = (b_idx == label_idx).sum().float() / N
        data_dict["ref_baseline_acc_all"] = (b_idx == label_idx).float()
        data_dict["ref_iou_mean"] = s_ious.mean()
        data_dict["best_ious_mean"] = best_ious.mean()
        data_dict["ref_iou_rate_0.25"] = (s_ious >= 0.25).float().mean()
        data_dict["ref_iou_rate_0.5"] = (s_ious >= 0.5).float().mean()
        return data_dict["ref_loss"], data_dict
    preds = data_dict["cluster_ref"]
    N, K = preds.shape
    ious, label_idx = _pseudo_gt(data_dict, N, K, N // data_dict["proposal_bbox_batched"].shape[0])
    labels = torch.zeros_like(preds).scatter_(1, label_idx.unsqueeze(1), 1.0)
    loss = softmax_ranking_loss(preds, labels)
    data_dict["cluster_labels"] = labels
    pred_idx = preds.argmax(-1)
    acc = (pred_idx == label_idx).sum().float() / N
    rows = torch.arange(N, device=preds.device)
    ref_ious, best_ious = ious[rows, pred_idx], ious[rows, label_idx]
    data_dict["ref_loss"] = loss if not is_frozen else preds.new_zeros(())
    data_dict["ref_acc_mean"] = acc
    data_dict["ref_iou_mean"] = ref_ious.mean()
    data_dict["best_ious_mean"] = best_ious.mean()
    data_dict["ref_iou_rate_0.25"] = (ref_ious >= 0.25).float().mean()
    data_dict["ref_iou_rate_0.5"] = (ref_ious >= 0.5).float().mean()
    return data_dict["ref_loss"], data_dict


def get_lobjcls_loss(data_dict, is_frozen=False, use_rl=False):
    """lib/grounding/loss_helper.py:231-302"""
    targets = (data_dict["ref_cat_label"] if "ref_cat_label" in data_dict else data_dict["object_cat"]).reshape(-1)
    if use_rl:
        sampled, baseline = data_dict["lang_scores"]["sampled"], data_dict["lang_scores"]["baseline"]
        assert targets.shape[0] == sampled.shape[0]
        s_loss = nn.functional.cross_entropy(sampled, targets.long(), reduction="none")
        b_loss = nn.functional.cross_entropy(baseline, targets.long(), reduction="none")
        data_dict["lang_loss"] = s_loss.mean()
        data_dict["sampled_lang_loss"], data_dict["baseline_lang_loss"] = s_loss, b_loss
        data_dict["lang_acc"] = data_dict["lang_sampled_acc"] = (sampled.argmax(-1) == targets).sum().float() / targets.shape[0]
        data_dict["lang_baseline_acc"] = (baseline.argmax(-1) == targets).sum().float() / targets.shape[0]
        return data_dict["lang_loss"], data_dict
    preds = data_dict["lang_scores"]
    loss = nn.functional.cross_entropy(preds, targets)
    data_dict["lang_loss"] = loss if not is_frozen else preds.new_zeros(())
    data_dict["lang_acc"] = (preds.argmax(-1) == targets).sum().float() / targets.shape[0]
    return data_dict["lang_loss"], data_dict
