"""Captioning losses on the device: cross-entropy caption loss and the edge-orientation loss
(reference: lib/captioning/loss_helper.py:98-224 non-RL branch, :226-307, :309-334).  The self-critical (CIDEr reward)
branch is not built yet."""
import numpy as np
import torch
import torch.nn.functional as F


def compute_cap_loss(data_dict, loss_opt={}):
    """(loss_helper.py:177-224) XE over the descriptions whose target box is good (IoU > min_iou_threshold)"""
    if loss_opt.get("use_rl", False):
        raise NotImplementedError("self-critical caption loss (CIDEr reward, loss_helper.py:110-176) is not built yet")
    max_len = loss_opt.get("max_len", 30)
    pred = data_dict["lang_cap"]
    num_words = int(data_dict["lang_len"].reshape(-1).max())
    target = data_dict["lang_ids"].reshape(-1, max_len)[:, 1:num_words]
    good = data_dict["good_bbox_masks"]
    if bool(good.sum() > 0):
        V = pred.shape[2]
        p, t = pred[good].reshape(-1, V), target[good].reshape(-1)
        cap_loss = F.cross_entropy(p, t, ignore_index=0)
        m = t != 0
        cap_acc = (p.argmax(-1)[m] == t[m]).sum().float() / m.sum().float()
    else:
        cap_loss, cap_acc = pred.new_zeros(()), pred.new_zeros(())
    z = data_dict["bbox_feature"].new_zeros(())
    data_dict["cap_rwd"], data_dict["loc_rwd"], data_dict["ttl_rwd"] = z, z, z
    data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
    return cap_loss, data_dict


def radian_to_label(radians, num_bins=6):
    """(loss_helper.py:226-242)"""
    boundaries = torch.arange(np.pi / num_bins, np.pi - 1e-8, np.pi / num_bins).type_as(radians)
    return torch.bucketize(radians, boundaries)


def compute_node_orientation_loss(data_dict, num_bins=6):
    """(loss_helper.py:244-307) relative rotation of the GT objects assigned to the two ends of every graph edge"""
    assign = data_dict["object_assignment"]
    edge_indices, edge_preds = data_dict["edge_index"], data_dict["edge_orientations"]
    nsrc, ntar = data_dict["num_edge_source"], data_dict["num_edge_target"]
    B, K = assign.shape
    rots = torch.gather(data_dict["scene_object_rotations"], 1, assign.view(B, K, 1, 1).repeat(1, 1, 3, 3))
    rot_masks = torch.gather(data_dict["scene_object_rotation_masks"], 1, assign)
    preds, labels, masks = [], [], []
    for b in range(B):
        n = int(nsrc[b]) * int(ntar[b])
        src, tar = edge_indices[b, 0, :n].long(), edge_indices[b, 1, :n].long()
        rel = torch.matmul(rots[b][src], rots[b][tar].transpose(2, 1))
        rel = torch.acos(torch.clamp(0.5 * (torch.diagonal(rel, dim1=-2, dim2=-1).sum(-1) - 1), -1, 1))
        preds.append(edge_preds[b, :n]); labels.append(radian_to_label(rel, num_bins)); masks.append(rot_masks[b][src] * rot_masks[b][tar])
    preds, labels, masks = torch.cat(preds), torch.cat(labels), torch.cat(masks)
    loss = (F.cross_entropy(preds, labels, reduction="none") * masks).sum() / (masks.sum() + 1e-8)
    hit = preds.argmax(-1)
    acc = (hit[masks == 1] == labels[masks == 1]).sum().float() / (masks.sum().float() + 1e-8)
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
