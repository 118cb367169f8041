"""Captioning losses: cross-entropy caption loss, the self-critical (CIDEr + listener reward) caption loss and the
edge-orientation loss (reference: lib/captioning/loss_helper.py:15-96, :98-224, :226-307, :309-334)."""
import numpy as np
import torch
import torch.nn.functional as F

from .cider import cider_scores


def compute_caption_reward(data_dict, cap_tables, sample_topn, idx2word, dataset_data, organized_data):
    """(loss_helper.py:15-96) CIDEr of every sampled caption against ALL ground-truth descriptions of its object, one
    scorer call for the whole batch (so the idf statistics are those of the batch).  Unannotated entries score 0.
    The reference also runs BLEU-4 here and multiplies it by a hard-coded weight of 0 (:83-88): not computed.
    Token ids leave the device in one transfer per call rather than one `.item()` per token."""
    assert len(cap_tables[0]) == sample_topn
    chunk_ids, annotated = data_dict["chunk_ids"], data_dict["annotated"].reshape(-1)
    Cn = chunk_ids.shape[1]
    dataset_ids = data_dict["id"].unsqueeze(1).repeat(1, Cn).reshape(-1).tolist()
    chunk_ids = chunk_ids.reshape(-1).tolist()
    N = len(dataset_ids)
    scores = torch.zeros(N, sample_topn, device=annotated.device)
    valid = (annotated == 1).nonzero().view(-1)
    if valid.shape[0] == 0:
        return scores
    valid_l = valid.tolist()
    lens = [len(cap_tables[n][k]) for n in valid_l for k in range(sample_topn)]
    flat = torch.cat([cap_tables[n][k].reshape(-1) for n in valid_l for k in range(sample_topn)]).tolist() if sum(lens) else []
    refs, cands, pos = [], [], 0
    ref_cache = {}
    for n in valid_l:
        raw = dataset_data[dataset_ids[n]][chunk_ids[n]]
        key = (raw["scene_id"], raw["object_id"])
        gt = ref_cache.get(key)
        if gt is None:
            gt = ref_cache[key] = [" ".join(d["token"] + ["eos"]) for d in organized_data[key[0]][key[1]]]
        for k in range(sample_topn):
            l = lens[len(cands)]
            tokens = [idx2word[str(t)] for t in flat[pos:pos + l]]
            pos += l
            if "eos" not in tokens:
                tokens.append("eos")
            refs.append(gt); cands.append(" ".join(tokens))
    _, cider = cider_scores(refs, cands)
    scores[valid] = torch.from_numpy(cider).to(scores).view(len(valid_l), sample_topn)
    return scores


def _rl_cap_loss(data_dict, loss_opt):
    """(loss_helper.py:110-176) REINFORCE with the greedy caption as baseline; reward = caption_weight * (CIDEr_sampled -
    CIDEr_greedy) + listener_weight * (the listener's loss improvement, detached).
    NOTE as in the reference, the listener losses arrive in (scene, sample, chunk) row order (the moderator moves the
    sample axis out, pipeline.py:835-838) and are `.view`ed as (scene*chunk, sample) without moving it back (:143-146)."""
    topn = loss_opt.get("sample_topn", 1)
    caps, logprobs, base_caps = data_dict["lang_cap"], data_dict["lang_logprob"], data_dict["baseline_cap"]
    good = data_dict["good_bbox_masks"].long()
    annotated = data_dict["annotated"].reshape(-1)
    logp = torch.stack([lp.sum() for beams in logprobs for lp in beams])
    args = (topn, loss_opt.get("idx2word"), loss_opt.get("train_dataset_data"), loss_opt.get("organized_data"))
    sampled = compute_caption_reward(data_dict, caps, *args).type_as(logp)
    baseline = compute_caption_reward(data_dict, base_caps, *args).type_as(logp)
    good = good.unsqueeze(1).repeat(1, topn)
    annotated = annotated.unsqueeze(1).repeat(1, topn)
    cap_reward = sampled - baseline
    shape = cap_reward.shape
    ref_reward = -(data_dict["ref_sampled_loss"].detach().view(shape) - data_dict["ref_baseline_loss"].detach().view(shape))
    lang_reward = -(data_dict["sampled_lang_loss"].detach().view(shape) - data_dict["baseline_lang_loss"].detach().view(shape))
    listener_reward = loss_opt.get("ref_reward_weight", 1) * ref_reward + loss_opt.get("lang_reward_weight", 1) * lang_reward
    rewards = loss_opt.get("caption_reward_weight", 1) * cap_reward + loss_opt.get("listener_reward_weight", 1) * listener_reward
    ngood = good.sum() + 1e-8
    cap_loss = (-rewards.view(-1) * logp * good.view(-1)).sum() / ngood
    cap_acc = (sampled * good * annotated).sum() / ((good * annotated).sum() + 1e-8)
    data_dict["cap_rwd"] = (cap_reward * good).sum() / ngood
    data_dict["loc_rwd"] = (listener_reward * good).sum() / ngood
    data_dict["ttl_rwd"] = (rewards * good).sum() / ngood
    data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
    return cap_loss, data_dict


def compute_cap_loss(data_dict, loss_opt={}):
    """(loss_helper.py:177-224) XE over the descriptions whose target box is good (IoU > min_iou_threshold).
    Same value as the reference's `pred[good]` selection, without its host round trip: the targets of the other
    descriptions are set to the ignored index 0, and an all-bad batch gives 0 (the reference's else branch)."""
    if loss_opt.get("use_rl", False):
        return _rl_cap_loss(data_dict, loss_opt)
    max_len = loss_opt.get("max_len", 30)
    pred = data_dict["lang_cap"]
    num_words = pred.shape[1] + 1                                   # == int(lang_len.max()) (the captioner ran num_words - 1 steps)
    target = data_dict["lang_ids"].reshape(-1, max_len)[:, 1:num_words]
    good = data_dict["good_bbox_masks"]
    V = pred.shape[2]
    t = torch.where(good.unsqueeze(1), target, torch.zeros_like(target)).reshape(-1)
    m = t != 0
    cnt = m.sum()
    denom = cnt.clamp(min=1).to(pred.dtype)
    p = pred.reshape(-1, V)
    cap_loss = F.cross_entropy(p, t, ignore_index=0, reduction="sum") / denom
    cap_acc = ((p.argmax(-1) == t) & m).sum().to(pred.dtype) / denom
    z = data_dict["bbox_feature"].new_zeros(())
    data_dict["cap_rwd"], data_dict["loc_rwd"], data_dict["ttl_rwd"] = z, z, z
    data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
    return cap_loss, data_dict


def radian_to_label(radians, num_bins=6):
    """(loss_helper.py:226-242)"""
    boundaries = torch.arange(np.pi / num_bins, np.pi - 1e-8, np.pi / num_bins).type_as(radians)
    return torch.bucketize(radians, boundaries)


def compute_node_orientation_loss(data_dict, num_bins=6):
    """(loss_helper.py:244-307) relative rotation of the GT objects assigned to the two ends of every graph edge.
    All scenes at once on the padded (B, K*L) edge tensors: the reference loops over the scenes and slices the first
    n = n_source * n_target edges of each (a host round trip per scene); here edges >= n get weight 0."""
    assign = data_dict["object_assignment"]
    edge_indices, edge_preds = data_dict["edge_index"], data_dict["edge_orientations"]
    nsrc, ntar = data_dict["num_edge_source"], data_dict["num_edge_target"]
    B, K = assign.shape
    E = edge_indices.shape[2]
    rots = torch.gather(data_dict["scene_object_rotations"], 1, assign.view(B, K, 1, 1).repeat(1, 1, 3, 3))
    rot_masks = torch.gather(data_dict["scene_object_rotation_masks"], 1, assign)
    n = (nsrc * ntar).view(B, 1)
    live = (torch.arange(E, device=assign.device).view(1, E) < n).to(rot_masks.dtype)
    src, tar = edge_indices[:, 0].long(), edge_indices[:, 1].long()                      # (B,E); padded entries are 0
    rs = torch.gather(rots, 1, src.view(B, E, 1, 1).expand(-1, -1, 3, 3))
    rt = torch.gather(rots, 1, tar.view(B, E, 1, 1).expand(-1, -1, 3, 3))
    rel = torch.matmul(rs, rt.transpose(3, 2))
    rel = torch.acos(torch.clamp(0.5 * (torch.diagonal(rel, dim1=-2, dim2=-1).sum(-1) - 1), -1, 1))
    labels = radian_to_label(rel, num_bins).reshape(-1)
    masks = (torch.gather(rot_masks, 1, src) * torch.gather(rot_masks, 1, tar) * live).reshape(-1)
    preds = edge_preds.reshape(B * E, -1)
    loss = (F.cross_entropy(preds, labels, reduction="none") * masks).sum() / (masks.sum() + 1e-8)
    acc = ((preds.argmax(-1) == labels).to(masks.dtype) * (masks == 1).to(masks.dtype)).sum() / (masks.sum().float() + 1e-8)
    return loss, acc


def get_captioning_loss(data_dict, caption, orientation, num_bins, loss_opt):
    """(loss_helper.py:309-334 `get_loss`)"""
    z = data_dict["bbox_feature"].new_zeros(())
    if caption:
        _, data_dict = compute_cap_loss(data_dict, loss_opt)
    else:
        data_dict["cap_loss"], data_dict["cap_acc"], data_dict["pred_ious"] = z, z, z
    if orientation:
        data_dict["ori_loss"], data_dict["ori_acc"] = compute_node_orientation_loss(data_dict, num_bins)
    else:
        data_dict["ori_loss"], data_dict["ori_acc"] = z, z
    return data_dict["cap_loss"] + 0.1 * data_dict["ori_loss"], data_dict
