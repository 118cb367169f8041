"""fp32 CPU restatement of the reference's speaker path (TEST INFRASTRUCTURE ONLY): the top-down captioner
(model/caption_module.py:72-133 step, :352-414 greedy/trim, :416-508 select_target, :510-687 training driver,
:689-770 evaluation driver, :800-842 _query_locals, :866-885 _add_relation_feat) and the EdgeConv graph module
(model/graph_module.py:21-114, 184-324), as plain functions over a state dict with the reference's key layout.

PINNED (captioner): tests/test_oracle_speaker.py checks it against tests/golden/speaker_golden.npz, produced by running
the reference's own TopDownSceneCaptionModule (tests/golden/gen_speaker_golden.py).
PARITY UNPINNED (graph module): model/graph_module.py needs torch_geometric (third party, unpinned: README.md:68,
"pytorch-1.8.0/1.8.1" era, 1.x API), absent here and from the reference tree; its arithmetic is restated from the
in-repo source (message :101-108, aggr "add", edge order = row-major non-zeros via scipy COO :273-277) and PyG's
documented source_to_target flow (x_j = x[edge_index[0]], x_i = x[edge_index[1]], aggregation at edge_index[1]).
`_query_locals` is shared with the captioner and therefore pinned."""
import random

import numpy as np
import torch
import torch.nn.functional as F


def aabb_iou_np(c1, c2):
    mn1, mx1, mn2, mx2 = c1.min(1), c1.max(1), c2.min(1), c2.max(1)
    inter = np.maximum(np.minimum(mx1, mx2) - np.maximum(mn1, mn2), 0).prod(1)
    return inter / ((mx1 - mn1).prod(1) + (mx2 - mn2).prod(1) - inter + 1e-8)


def gru_cell(p, pre, x, h):
    H = h.shape[1]
    gi = F.linear(x, p[pre + ".weight_ih"], p[pre + ".bias_ih"]); gh = F.linear(h, p[pre + ".weight_hh"], p[pre + ".bias_hh"])
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]); z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def step(p, word, hiddens, target_feat, obj_feats, object_masks):
    """caption_module.py:72-133"""
    h1, h2 = hiddens
    V = p["embeddings"].shape[0]
    onehot = torch.zeros(word.shape[0], V).scatter_(1, word.unsqueeze(1), 1)
    x = torch.matmul(onehot, p["embeddings"])
    x = F.linear(torch.cat([x, h2, target_feat], -1), p["map_topdown.weight"], p["map_topdown.bias"])
    h1 = gru_cell(p, "recurrent_cell_1", x, h1)
    comb = torch.tanh(F.linear(obj_feats, p["map_feat.weight"]) + F.linear(h1, p["map_hidd.weight"]).unsqueeze(1))
    scores = F.linear(comb, p["attend.weight"]).masked_fill(object_masks == 0, 0)
    masks = F.softmax(scores, dim=1)
    attended = (obj_feats * masks).sum(1)
    h2 = gru_cell(p, "recurrent_cell_2", F.linear(torch.cat([attended, h1], -1), p["map_lang.weight"], p["map_lang.bias"]), h2)
    out = F.linear(torch.relu(F.linear(h2, p["classifier.0.weight"], p["classifier.0.bias"])), p["classifier.2.weight"], p["classifier.2.bias"])
    return out, (h1, h2), masks


# Ties in the neighbour selection.  The reference takes `torch.topk(dist, num_locals, largest=False)`; among EQUAL distances
# its choice is implementation-defined (and differs between the CPU and CUDA kernels of the library).  Equal distances are
# not exotic: padded slots all sit at 1e30, and the two clustering branches report most objects twice with identical boxes.
# "topk" keeps the library call (what the golden vectors were produced with); "index" resolves ties towards the lower slot
# index (a stable ascending sort) -- a deterministic member of the reference's set of admissible results, used by the tests
# that compare whole pipelines.
TIE_RULE = "topk"


def query_locals(corners, target_ids, object_masks, num_locals, include_self=True, thr=0.5):
    """caption_module.py:800-842 / graph_module.py:184-227, "corner" mode, one target per sample"""
    N, K = object_masks.shape
    centers = (corners.min(2)[0] + corners.max(2)[0]) / 2
    tc = torch.gather(corners, 1, target_ids.view(-1, 1, 1, 1).repeat(1, 1, 8, 3))
    diff = tc.squeeze(1).unsqueeze(2).repeat(1, 1, K, 1) - centers.unsqueeze(1).repeat(1, 8, 1, 1)
    dist = torch.sqrt(torch.sum(diff ** 2, dim=-1) + 1e-8).min(1)[0]
    dist = dist.masked_fill(object_masks == 0, 1e30)
    iou = aabb_iou_np(tc.repeat(1, K, 1, 1).view(-1, 8, 3).numpy(), corners.reshape(-1, 8, 3).numpy())
    dist = dist.masked_fill(torch.from_numpy(iou).float().view(N, K) >= thr, 1e30)
    selfm = torch.zeros(N, K).scatter_(1, target_ids.view(-1, 1), 1)
    dist = dist.masked_fill(selfm == 1, 0 if include_self else 1e30)
    if TIE_RULE == "index":
        ids = torch.sort(dist, dim=1, stable=True)[1][:, :num_locals]
    else:
        _, ids = torch.topk(dist, num_locals, largest=False, dim=1)
    return torch.zeros(N, K).scatter_(1, ids, 1)


def add_relation_feat(rel_feats, adj, obj_feats, target_ids, L):
    """caption_module.py:866-885"""
    N, K, Fs = obj_feats.shape
    rel = torch.gather(rel_feats, 1, target_ids.view(N, 1, 1, 1).repeat(1, 1, L, Fs)).squeeze(1)
    rows = torch.gather(adj, 1, target_ids.view(N, 1, 1).repeat(1, 1, K)).squeeze(1)
    m = rows.unsqueeze(-1).repeat(1, 1, Fs) == 1
    return obj_feats + torch.zeros(obj_feats.shape).masked_scatter(m, rel)


def select_target(objness, centers, corners, center_lab, corner_lab, ref_lab, ref_corner, annotated):
    """caption_module.py:416-508 (use_oracle False)"""
    N, K, _ = centers.shape
    ids, ious, labs = [], [], []
    for n in range(N):
        if annotated[n] == 1:
            io = aabb_iou_np(corners[n].numpy(), ref_corner[n].unsqueeze(0).repeat(K, 1, 1).numpy())
            t = int(io.argmax()); ids.append(t); ious.append(float(io[t])); labs.append(int(ref_lab[n].argmax(-1)))
        else:
            allids = torch.arange(K)
            valid = allids[objness[n] == 1]
            t = random.choice(valid) if len(valid) > 0 else random.choice(allids)
            d = ((centers[n].unsqueeze(1) - center_lab[n].unsqueeze(0)) ** 2).sum(-1)
            a = int(d.min(1)[1][t])
            io = aabb_iou_np(corners[n, t].unsqueeze(0).numpy(), corner_lab[n, a].unsqueeze(0).numpy())[0]
            ids.append(int(t)); ious.append(float(io)); labs.append(a)
    return torch.tensor(ids), torch.tensor(ious, dtype=torch.float32), torch.tensor(labs)


def forward_sample_batch(p, d, cfg, K, L):
    """caption_module.py:510-687, XE branch with teacher forcing"""
    word_ids = d["lang_ids"].reshape(-1, cfg.data.max_spk_len + 2)
    lens = d["lang_len"].reshape(-1)
    N = lens.shape[0]
    Cn = N // d["center_label"].shape[0]
    rep = lambda t: t.unsqueeze(1).repeat(1, Cn, *([1] * (t.dim() - 1))).reshape(N, *t.shape[1:])
    obj_feats, centers, corners, masks = rep(d["bbox_feature"]), rep(d["proposal_center_batched"]), rep(d["proposal_bbox_batched"]), rep(d["proposal_batch_mask"])
    tids, tious, labs = select_target(masks, centers, corners, rep(d["center_label"]), rep(d["gt_bbox"]), d["ref_box_label"].reshape(-1, 128),
                                      d["ref_box_corner_label"].reshape(-1, 8, 3), d["annotated"].reshape(-1))
    tf = torch.gather(obj_feats, 1, tids.view(N, 1, 1).repeat(1, 1, 128)).squeeze(1)
    valid = query_locals(corners, tids, masks, L).unsqueeze(-1)
    obj_feats = add_relation_feat(rep(d["edge_feature"]), rep(d["adjacent_mat"]), obj_feats, tids, L)
    h = (torch.zeros(N, 512), torch.zeros(N, 512))
    outs, attn = [], []
    step_id, word = 0, word_ids[:, 0]
    num_words = int(lens.max())
    while True:
        o, h, m = step(p, word, h, tf, obj_feats, valid)
        outs.append(o.unsqueeze(1)); attn.append(m)
        step_id += 1
        if step_id == num_words - 1:
            break
        word = word_ids[:, step_id]
    good = tious > cfg.data.min_iou_threshold
    return dict(lang_cap=torch.cat(outs, 1), topdown_attn=torch.cat(attn, -1), valid_masks=valid, assigned=labs,
                pred_ious=tious[good].mean() if good.sum() > 0 else torch.zeros(()), good=good)


def forward_scene_batch(p, d, cfg, K, L, sos):
    """caption_module.py:689-770: proposal by proposal, step by step (the relation features do not reach step())"""
    obj_feats, masks, corners = d["bbox_feature"], d["proposal_batch_mask"], d["proposal_bbox_batched"]
    B = obj_feats.shape[0]
    outs, valids, attn = [], [], []
    for prop in range(K):
        tids = torch.full((B,), prop, dtype=torch.long)
        tf = obj_feats[:, prop]
        v = query_locals(corners, tids, masks, L)
        valids.append(v.unsqueeze(1))
        h = (torch.zeros(B, 512), torch.zeros(B, 512))
        word = torch.full((B,), sos, dtype=torch.long)
        po, pm = [], []
        for _ in range(cfg.data.max_spk_len + 1):
            o, h, m = step(p, word, h, tf, obj_feats, v.unsqueeze(-1))
            word = o.argmax(-1); po.append(word.unsqueeze(1)); pm.append(m)
        outs.append(torch.cat(po, 1).unsqueeze(1)); attn.append(torch.cat(pm, -1).unsqueeze(1))
    return dict(lang_cap=torch.cat(outs, 1), valid_masks=torch.cat(valids, 1), topdown_attn=torch.cat(attn, 1))


def greedy_decode(p, tf, obj_feats, valid, max_len, sos, eos, pad):
    """caption_module.py:350-414"""
    N = tf.shape[0]
    h = (torch.zeros(N, 512), torch.zeros(N, 512)); word = torch.full((N,), sos, dtype=torch.long)
    ids, lps = [], []
    for _ in range(max_len):
        o, h, _ = step(p, word, h, tf, obj_feats, valid)
        lp, word = F.log_softmax(o, -1).max(-1)
        ids.append(word.unsqueeze(1)); lps.append(lp.unsqueeze(1))
    ids, lps = torch.cat(ids, 1), torch.cat(lps, 1)
    out = []
    for n in range(N):
        t = 0
        for t in range(max_len):
            if ids[n, t] == eos or ids[n, t] == pad:
                break
        out.append((ids[n, :t], lps[n, :t]))
    return out


# ------------------------------------------------------------------------------------------ graph module
def edge_conv(p, pre, x, edge_index):
    """EdgeConv.propagate/message (graph_module.py:43-108), aggr add at edge_index[1]"""
    x_j, x_i = x[edge_index[0]], x[edge_index[1]]
    e = torch.cat([x_i, x_j - x_i], 1)
    msg = F.linear(torch.relu(F.linear(e, p[pre + ".map_edge.0.weight"], p[pre + ".map_edge.0.bias"])), p[pre + ".map_edge.2.weight"], p[pre + ".map_edge.2.bias"])
    out = torch.zeros(x.shape[0], msg.shape[1])
    for k in range(edge_index.shape[1]):
        out[edge_index[1, k]] = out[edge_index[1, k]] + msg[k]
    return out, msg


def graph_module(p, d, num_layers, L, num_bins=6, out_size=128):
    """GraphModule.forward (graph_module.py:252-324)"""
    x_all = F.linear(d["proposal_feats_batched"], p["map_input.weight"], p["map_input.bias"])
    masks, corners = d["proposal_batch_mask"], d["proposal_bbox_batched"]
    B, K, _ = x_all.shape
    adj = torch.zeros(B, K, K)
    for o in range(K):
        adj[:, o] = query_locals(corners, torch.full((B,), o, dtype=torch.long), masks, L, include_self=False)
    new = torch.zeros(B, K, out_size); edge_feats = torch.zeros(B, K, L, out_size); edge_idx = torch.zeros(B, 2, K * L)
    preds = torch.zeros(B, K * L, num_bins + 1); nsrc = torch.zeros(B, dtype=torch.long); ntar = torch.zeros(B, dtype=torch.long)
    for b in range(B):
        valid = masks[b] == 1
        sub = adj[b][valid, :][:, valid].numpy()
        rows, cols = np.nonzero(sub)                                   # scipy.sparse.coo_matrix(dense) order
        ei = torch.from_numpy(np.stack([rows, cols])).long()
        x = x_all[b, valid]
        node, msg = x, None
        for l in range(num_layers):
            node, msg = edge_conv(p, "gc_layers.%d" % l, node, ei)
        try:
            ns = len(set(rows.tolist())); nt = int(msg.shape[0] / ns)
            nsrc[b], ntar[b] = ns, nt
            m = msg[:ns * nt]
            edge_feats[b, :ns, :nt] = m.view(ns, nt, out_size)
            edge_idx[b, :, :ns * nt] = ei[:, :ns * nt]
            _, last = edge_conv(p, "edge_layer", node, ei)
            preds[b, :ns * nt] = F.linear(last, p["edge_predict.weight"], p["edge_predict.bias"])
        except Exception:
            pass
        new[b, valid] = x + node
    return dict(bbox_feature=new, adjacent_mat=adj, edge_index=edge_idx, edge_feature=edge_feats, num_edge_source=nsrc,
                num_edge_target=ntar, edge_orientations=preds[:, :, :-1], edge_distances=preds[:, :, -1])
