"""Grounding evaluation: the reference's `lib/grounding/eval_helper.py:28-137 get_eval`
on batched tensors -- the referred-box prediction, its AABB IoU with the GT box, the pseudo-GT ("best") IoU, Acc@0.25 /
Acc@0.5, the unique / multiple and "others" masks and the language-classification accuracy -- without the reference's
per-sample python loop and per-sample device->host copies (eval_helper.py:92-116: one `.cpu().numpy()` IoU per
description).  Pinned to golden vectors produced by the reference's own function (tests/golden/gen_grounding_eval_golden.py).
"""
import torch

from .listener import aabb_iou_to_gt


def get_eval(data_dict, grounding=True, use_lang_classifier=False):
    mask = (data_dict["proposal_batch_mask"].long() == 1).float()                  # (B, K)
    cluster_ref = data_dict["cluster_ref"]                                        # (B*C, K)
    N, K = cluster_ref.shape
    chunk = N // mask.shape[0]
    pred_masks = mask.unsqueeze(1).repeat(1, chunk, 1).reshape(N, K)
    labels = data_dict["cluster_labels"].float()

    # classification accuracy of the arg-max proposal (over ALL slots, as the reference: eval_helper.py:51-60)
    top = torch.argmax(cluster_ref, 1)
    corrects = (labels.gather(1, top.unsqueeze(1)).squeeze(1) == 1).float()
    ref_acc = corrects / (1.0 + 1e-8)
    data_dict["ref_acc"] = ref_acc.cpu().numpy().tolist()
    data_dict["ref_acc_mean"] = ref_acc.mean()

    # localisation: arg-max over the valid proposals, IoU with the referred box; pseudo-GT IoU
    masked = cluster_ref * pred_masks
    pred_ref = torch.argmax(masked, 1)
    data_dict["cluster_ref"] = masked
    corners = data_dict["proposal_bbox_batched"].unsqueeze(1).repeat(1, chunk, 1, 1, 1).reshape(N, K, 8, 3)
    gt = data_dict["ref_box_corner_label"].reshape(N, 8, 3)
    all_ious = aabb_iou_to_gt(corners, gt)                                         # (N, K)
    ious = all_ious.gather(1, pred_ref.unsqueeze(1)).squeeze(1)
    gt_ref = torch.argmax(labels, 1)
    best_ious = all_ious.gather(1, gt_ref.unsqueeze(1)).squeeze(1)
    rows = torch.arange(N, device=corners.device)
    object_cat = data_dict["object_cat"].reshape(-1)

    if grounding and use_lang_classifier:
        data_dict["lang_acc"] = (torch.argmax(data_dict["lang_scores"], 1) == object_cat).float().mean()
    else:
        data_dict["lang_acc"] = torch.zeros(1)[0].type_as(ious)

    data_dict["ref_iou"] = ious
    data_dict["best_ious"] = best_ious
    data_dict["ref_iou_mean"] = ious.mean()
    data_dict["best_ious_mean"] = best_ious.mean()
    data_dict["ref_iou_rate_0.25"] = float((ious >= 0.25).sum()) / N
    data_dict["ref_iou_rate_0.5"] = float((ious >= 0.5).sum()) / N
    data_dict["ref_multiple_mask"] = data_dict["unique_multiple"].reshape(-1).cpu().tolist()
    data_dict["ref_others_mask"] = (object_cat == 17).long().cpu().tolist()
    data_dict["pred_bboxes"] = corners[rows, pred_ref]
    data_dict["gt_bboxes"] = gt
    return data_dict
