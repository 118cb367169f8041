"""Dense-captioning evaluation: assignment of the per-proposal captions to the GT boxes and
CIDEr@kIoU corpus scoring -- the reference's `lib/captioning/eval_helper.py:102-307` (`assign_dense_caption`,
`eval_caption_step`, the candidate filtering of `eval_caption_epoch`) and `lib/utils/bbox.py:571-757`
(`generalized_box3d_iou` for axis-aligned boxes, `box3d_iou_batch_tensor`).

The (B, K1, K2) generalised-IoU cost matrix is computed batched on whatever device the boxes live on (the reference loops
over the batch in python to mask the padded GT columns); the Hungarian assignment stays scipy on the host, as in the
reference; CIDEr is d3net_amd.cider (bit-identical to lib/capeval/cider).  BLEU / ROUGE / METEOR are not restated (METEOR
needs a Java runtime).  Pinned to golden vectors produced by the reference's own functions
(tests/golden/gen_caption_eval_golden.py).

Quirk kept on purpose: the reference's "footprint" rectangle of a box is read from corner columns (x, z) of corners 2 and 0
and the height from the z of corners 0 and 4 -- conventions inherited from a y-up code base -- so for this repo's z-up corner
order the intersection term is usually zero and the cost is dominated by the enclosing-volume term.  Assignments must match
the reference's, so the arithmetic is restated as it is, not as it was presumably meant.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .cider import cider_scores

_EPS = 1e-8


def _edge_volume(c):
    """box volume from three edge lengths (bbox.py:571-592), (..., 8, 3) -> (...)"""
    def edge(i, j):
        return torch.sqrt((c[..., i, :] - c[..., j, :]).pow(2).sum(-1).clamp(min=1e-6))
    return edge(0, 1) * edge(1, 2) * edge(0, 4)


def generalized_box3d_iou(corners1, corners2, nums_k2=None):
    """(B,K1,8,3), (B,K2,8,3) -> (B,K1,K2) generalised IoU, axis-aligned path of bbox.py:645-757 (rotated_boxes=False)"""
    a, b = corners1.float(), corners2.float()
    B, K1, K2 = a.shape[0], a.shape[1], b.shape[1]
    height = (torch.minimum(a[:, :, 0, 2].unsqueeze(2), b[:, :, 0, 2].unsqueeze(1)) -
              torch.maximum(a[:, :, 4, 2].unsqueeze(2), b[:, :, 4, 2].unsqueeze(1))).clamp(min=0)
    cols = [0, 2]
    lt = torch.maximum(a[:, :, 2][..., cols].unsqueeze(2), b[:, :, 2][..., cols].unsqueeze(1))
    rb = torch.minimum(a[:, :, 0][..., cols].unsqueeze(2), b[:, :, 0][..., cols].unsqueeze(1))
    wh = (rb - lt).clamp(min=0)
    valid = torch.ones((B, 1, K2), dtype=a.dtype, device=a.device)
    if nums_k2 is not None:
        valid = (torch.arange(K2, device=a.device).view(1, 1, K2) < nums_k2.view(B, 1, 1)).to(a.dtype)
    inter_vol = wh[..., 0] * wh[..., 1] * valid * height
    lo = torch.minimum(a.min(2).values.unsqueeze(2), b.min(2).values.unsqueeze(1))
    hi = torch.maximum(a.max(2).values.unsqueeze(2), b.max(2).values.unsqueeze(1))
    enclosing = (hi - lo).abs().prod(-1)
    v1, v2 = _edge_volume(a).clamp(min=_EPS), _edge_volume(b).clamp(min=_EPS)
    sum_vols = v1.unsqueeze(2) + v2.unsqueeze(1)
    good = ((enclosing > 2 * _EPS) & (sum_vols > 4 * _EPS)).to(a.dtype)
    union = (sum_vols - inter_vol).clamp(min=_EPS)
    giou = (inter_vol / union - (1 - union / enclosing)) * good
    return giou * valid


def box3d_iou(c1, c2):
    """AABB IoU of paired boxes (N,8,3) x (N,8,3) -> (N) (bbox.py:273-305)"""
    c1, c2 = c1.float(), c2.float()
    lo1, hi1, lo2, hi2 = c1.min(1).values, c1.max(1).values, c2.min(1).values, c2.max(1).values
    inter = (torch.minimum(hi1, hi2) - torch.maximum(lo1, lo2)).clamp(min=0).prod(-1)
    return inter / ((hi1 - lo1).prod(-1) + (hi2 - lo2).prod(-1) - inter + 1e-8)


def decode_caption(tokens, idx2word, special_tokens):
    out = [special_tokens["bos_token"]]
    for t in tokens.tolist():
        w = idx2word[str(t)]
        out.append(w)
        if w == special_tokens["eos_token"]:
            break
    if special_tokens["eos_token"] not in out:
        out.append(special_tokens["eos_token"])
    return " ".join(out)


def assign_dense_caption(pred_captions, pred_boxes, gt_boxes, gt_box_ids, gt_box_masks, gt_scene_list, idx2word,
                         special_tokens, strategy="giou"):
    """eval_helper.py:102-246: Hungarian assignment of proposals to GT boxes, the matched proposal's caption per GT box"""
    B, ngt = gt_box_ids.shape
    nactual = gt_box_masks.sum(1).long()
    if strategy == "giou":
        cost = -generalized_box3d_iou(pred_boxes, gt_boxes, nactual)
    elif strategy == "center":
        cost = torch.cdist(pred_boxes.mean(2).float(), gt_boxes.mean(2).float())
    else:
        raise ValueError("invalid strategy.")
    cost = cost.detach().cpu().numpy()
    per_gt = torch.zeros((B, ngt), dtype=torch.int64)
    for b in range(B):
        n = int(nactual[b])
        if n > 0:
            rows, cols = linear_sum_assignment(cost[b, :, :n])
            per_gt[b, torch.from_numpy(cols)] = torch.from_numpy(rows)
    per_gt = per_gt.to(pred_boxes.device)
    matched = torch.gather(pred_boxes, 1, per_gt[:, :, None, None].expand(B, ngt, 8, 3))
    ious = box3d_iou(matched.reshape(-1, 8, 3), gt_boxes.reshape(-1, 8, 3)).reshape(B, ngt).cpu()
    caps = torch.gather(pred_captions, 1, per_gt[:, :, None].expand(B, ngt, pred_captions.shape[2])).cpu()
    matched_h, gt_h, masks, ids = matched.cpu(), gt_boxes.cpu(), gt_box_masks.cpu(), gt_box_ids.cpu()
    candidates = {}
    for b in range(B):
        for g in range(ngt):
            if masks[b, g] == 0:
                continue
            key = "{}|{}".format(gt_scene_list[b], str(ids[b, g].item()))
            candidates[key] = {"caption": decode_caption(caps[b, g], idx2word, special_tokens), "iou": ious[b, g].item(),
                               "box": matched_h[b, g].numpy().tolist(), "gt_box": gt_h[b, g].numpy().tolist()}
    return candidates


def eval_caption_step(data_dict, dataset_vocabulary):
    """eval_helper.py:248-262"""
    return assign_dense_caption(data_dict["lang_cap"], data_dict["proposal_bbox_batched"], data_dict["gt_bbox"],
                                data_dict["gt_bbox_object_id"], data_dict["gt_bbox_label"], data_dict["scene_id"],
                                dataset_vocabulary["idx2word"], dataset_vocabulary["special_tokens"])


def prepare_corpus(raw_data, candidates, max_len=30):
    """eval_helper.py:35-62: reference descriptions ("sos ... eos") of the scenes that have candidates"""
    scenes = {k.split("|")[0] for k in candidates}
    corpus = {}
    for d in raw_data:
        if d["scene_id"] not in scenes:
            continue
        corpus.setdefault("{}|{}".format(d["scene_id"], d["object_id"]), []).append("sos " + " ".join(d["token"][:max_len]) + " eos")
    return corpus


def score_captions(candidates, raw_data, max_len=30, min_iou=0.5):
    """CIDEr@min_iou as eval_caption_epoch computes it (eval_helper.py:264-296): captions whose matched box has IoU <
    min_iou, and undetected objects, count as the empty caption "sos eos".  -> (mean, per-object scores, keys)"""
    corpus = prepare_corpus(raw_data, candidates, max_len)
    kept = {k: v["caption"] for k, v in candidates.items() if v["iou"] >= min_iou}
    keys = list(corpus.keys())
    mean, scores = cider_scores([corpus[k] for k in keys], [kept.get(k, "sos eos") for k in keys])
    return mean, scores, keys


def eval_caption_epoch(candidates, raw_data, max_len=30, min_iou=0.5):
    """Corpus scores of one validation epoch in the reference's return format (eval_helper.py:264-340):
    (bleu, cider, rouge, meteor) with bleu = ([BLEU-1..4], [per-entry lists]) and the others (mean, per-entry scores).
    Captions whose matched box has IoU < min_iou, and undetected objects, count as the empty caption "sos eos".  METEOR is a
    Java subprocess in the reference (lib/capeval/meteor) and no Java runtime exists here: it is reported as (0.0, zeros)."""
    from .caption_metrics import bleu_scores, rouge_l_scores
    corpus = prepare_corpus(raw_data, candidates, max_len)
    kept = {k: v["caption"] for k, v in candidates.items() if v["iou"] >= min_iou}
    keys = list(corpus.keys())
    refs, cands = [corpus[k] for k in keys], [kept.get(k, "sos eos") for k in keys]
    if not keys:
        z = (0.0, np.zeros(0))
        return ([0.0] * 4, [[] for _ in range(4)]), z, z, z
    bleu = bleu_scores(refs, cands, 4)
    cider = cider_scores(refs, cands)
    rouge = rouge_l_scores(refs, cands)
    return bleu, cider, rouge, (0.0, np.zeros(len(keys)))
