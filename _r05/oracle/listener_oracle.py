"""fp32 CPU restatement of the reference's listener path (TEST INFRASTRUCTURE ONLY): LangModule, MultiHeadAttention,
TransformerMatchModule and the non-RL grounding / language-classification losses, as plain functions over a state dict
with the reference's key layout.

Follows model/lang_module.py:139-176, model/transformer/attention.py:42-77,161-176, model/match_module.py:189-336,
lib/grounding/loss.py:6-25, lib/grounding/loss_helper.py:133-214,276-292 and lib/utils/bbox.py:247-271.
PINNED: tests/test_oracle_listener.py checks it against tests/golden/listener_golden.npz, which was produced by
running the reference's own modules (tests/golden/gen_listener_golden.py).  Dropout is the identity (p = 0 in the
fixtures; it cannot be reproduced across implementations)."""
import numpy as np
import torch
import torch.nn.functional as F


def mha(p, pre, q_in, k_in, v_in, h, d_k, d_v, mask=None, weights=None):
    """MultiHeadAttention.forward: layer_norm(queries + attention(...))   (attention.py:161-176, 42-77)"""
    B, nq, _ = q_in.shape
    nk = k_in.shape[1]
    a = pre + ".attention."
    q = F.linear(q_in, p[a + "fc_q.weight"], p[a + "fc_q.bias"]).view(B, nq, h, d_k).permute(0, 2, 1, 3)
    k = F.linear(k_in, p[a + "fc_k.weight"], p[a + "fc_k.bias"]).view(B, nk, h, d_k).permute(0, 2, 3, 1)
    v = F.linear(v_in, p[a + "fc_v.weight"], p[a + "fc_v.bias"]).view(B, nk, h, d_v).permute(0, 2, 1, 3)
    att = torch.matmul(q, k) / np.sqrt(d_k)
    if weights is not None:
        att = att + weights
    if mask is not None:
        att = att.masked_fill(mask == 0, -np.inf)
    att = torch.softmax(att, -1)
    out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(B, nq, h * d_v)
    out = F.linear(out, p[a + "fc_o.weight"], p[a + "fc_o.bias"])
    return F.layer_norm(q_in + out, (q_in.shape[-1],), p[pre + ".layer_norm.weight"], p[pre + ".layer_norm.bias"])


def lang_module(p, lang_feat, lang_len):
    """LangModule.forward, non-RL branch (lang_module.py:139-176); unidirectional nn.GRU(300 -> 256) written out"""
    B, Cn, T, E = lang_feat.shape
    H = 256
    embs = lang_feat.reshape(-1, T, E)
    lens = lang_len.reshape(-1)
    # nn.GRU over packed sequences == the cell recurrence with the state frozen (and the output zero) past each
    # sequence's length.  Gate order in weight_ih/hh: [r, z, n]; n uses r * (W_hn h + b_hn).
    w_ih, w_hh, b_ih, b_hh = (p["lang.gru." + k] for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"))
    h = torch.zeros(B * Cn, H)
    outs = []
    for t in range(T):
        gi = F.linear(embs[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (1 - z) * n + z * h
        alive = (t < lens).float().unsqueeze(1)
        h = alive * hn + (1 - alive) * h
        outs.append(alive * hn)
    hid = torch.stack(outs, 1)[:, :int(lens.max())]
    last = h
    pad = torch.zeros(B * Cn, T, 256)
    pad[:, :hid.shape[1]] = hid
    masks = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).float()
    scores = F.linear(last, p["lang.lang_cls.0.weight"], p["lang.lang_cls.0.bias"])
    return pad, last, masks, scores


def _bn1d(p, pre, x, training):
    return F.batch_norm(x, p[pre + ".running_mean"].clone(), p[pre + ".running_var"].clone(), p[pre + ".weight"],
                        p[pre + ".bias"], training, 0.1, 1e-5)


def match_module(p, d, lang_hiddens, lang_masks, chunk, training, rnd, head=4, hidden=128):
    """TransformerMatchModule.forward, non-RL branch (match_module.py:212-336)"""
    centers = d["proposal_center_batched"]
    K = centers.shape[1]
    A = centers[:, None, :, :].repeat(1, K, 1, 1)
    Bm = centers[:, :, None, :].repeat(1, 1, K, 1)
    dist = torch.sqrt(torch.sum((A - Bm).pow(2), dim=-1))[:, None, :, :]
    w = 1 / (dist + 1e-2)
    w = w / torch.sum(w, dim=2, keepdim=True)
    dist_weights = torch.cat([w for _ in range(head)], dim=1)
    x = d["proposal_feats_batched"].permute(0, 2, 1)
    m = "match.features_concat."
    x = F.conv1d(x, p[m + "0.weight"], p[m + "0.bias"])
    x = _bn1d(p, m + "1", x, training)
    x = F.prelu(x, p[m + "2.weight"])
    x = F.conv1d(x, p[m + "3.weight"], p[m + "3.bias"]).permute(0, 2, 1)
    B = x.shape[0]
    feats = mha(p, "match.self_attn.0", x, x, x, head, hidden // head, hidden // head, None, dist_weights)
    feature0 = feats.clone()
    if training and rnd < 0.5:                                   # copy-paste augmentation (:266-291)
        obj = d["proposal_batch_mask"].bool()
        lens = obj.sum(1)
        pool = feats.reshape(B * K, -1)[obj.reshape(-1)].repeat(2, 1)
        total = pool.shape[0] // 2
        j = 0
        for i in range(B):
            empty = torch.where(~obj[i])[0]
            j += int(lens[i])
            n = len(empty) if len(empty) < total - int(lens[i]) else total - int(lens[i])
            feature0[i, empty[:n]] = pool[j:j + n]
    v = feature0[:, None].repeat(1, chunk, 1, 1).reshape(-1, K, hidden)
    dw = dist_weights[:, None].repeat(1, chunk, 1, 1, 1).reshape(-1, head, K, K)
    N, T, _ = lang_hiddens.shape
    l = F.linear(lang_hiddens, p["match.lang_fc.0.weight"], p["match.lang_fc.0.bias"])
    l = F.layer_norm(torch.relu(l), (hidden,), p["match.lang_fc.3.weight"], p["match.lang_fc.3.bias"])
    self_mask = lang_masks.reshape(N, 1, 1, -1).repeat(1, head, T, 1)
    l = mha(p, "match.lang_self_attn", l, l, l, head, 16, 16, self_mask)
    cross_mask = lang_masks.reshape(N, 1, 1, -1).repeat(1, head, K, 1)
    v = mha(p, "match.cross_attn.0", v, l, l, head, hidden // head, hidden // head, cross_mask)
    v = mha(p, "match.self_attn.1", v, v, v, head, hidden // head, hidden // head, None, dw)
    v = mha(p, "match.cross_attn.1", v, l, l, head, hidden // head, hidden // head, cross_mask)
    x = v.permute(0, 2, 1).contiguous()
    m = "match.match."
    x = F.prelu(_bn1d(p, m + "1", F.conv1d(x, p[m + "0.weight"], p[m + "0.bias"]), training), p[m + "2.weight"])
    x = F.prelu(_bn1d(p, m + "4", F.conv1d(x, p[m + "3.weight"], p[m + "3.bias"]), training), p[m + "5.weight"])
    return F.conv1d(x, p[m + "6.weight"], p[m + "6.bias"]).squeeze(1)


def aabb_iou(c1, c2):
    """lib/utils/bbox.py:247-271 on numpy (N,8,3) arrays"""
    mn1, mx1, mn2, mx2 = c1.min(1), c1.max(1), c2.min(1), c2.max(1)
    inter = np.maximum(np.minimum(mx1, mx2) - np.maximum(mn1, mn2), 0).prod(1)
    return inter / ((mx1 - mn1).prod(1) + (mx2 - mn2).prod(1) - inter + 1e-8)


def grounding_loss(d, cluster_ref):
    """lib/grounding/loss_helper.py:133-214 (non-RL, cross_entropy)"""
    N, K = cluster_ref.shape
    corners = d["proposal_bbox_batched"]
    chunk = N // corners.shape[0]
    corners = corners.unsqueeze(1).repeat(1, chunk, 1, 1, 1).reshape(N, K, 8, 3)
    gt = d["ref_box_corner_label"].reshape(N, 8, 3)
    labels = np.zeros((N, K))
    ious_all = []
    for i in range(N):
        ious = aabb_iou(corners[i].numpy(), gt[i].unsqueeze(0).repeat(K, 1, 1).numpy())
        labels[i, ious.argmax()] = 1
        ious_all.append(ious)
    labels = torch.FloatTensor(labels)
    probs = F.softmax(cluster_ref + 1e-8, dim=1)
    loss = (-torch.sum(torch.log(probs + 1e-8) * labels, dim=1)).mean()
    lab, pred = labels.argmax(-1), cluster_ref.argmax(-1)
    ious = torch.tensor([ious_all[i][pred[i]] for i in range(N)]).float()
    best = torch.tensor([ious_all[i][lab[i]] for i in range(N)]).float()
    return dict(ref_loss=loss, cluster_labels=labels, ref_acc_mean=(pred == lab).sum().float() / N,
                ref_iou_mean=ious.mean(), best_ious_mean=best.mean(),
                rate25=float((ious >= 0.25).sum()) / N, rate5=float((ious >= 0.5).sum()) / N)


def listener_step(p, d, chunk, training, rnd):
    hid, emb, masks, scores = lang_module(p, d["lang_feat"], d["lang_len"])
    ref = match_module(p, d, hid, masks, chunk, training, rnd)
    out = grounding_loss(d, ref)
    targets = d["object_cat"].reshape(-1)
    out.update(cluster_ref=ref, lang_scores=scores, lang_emb=emb, lang_hiddens=hid, lang_masks=masks,
               lang_loss=F.cross_entropy(scores, targets), lang_acc=(scores.argmax(-1) == targets).sum().float() / len(targets))
    return out
