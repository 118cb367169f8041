"""Detection evaluator: class-aware 3D NMS -> per-class precision/recall -> VOC AP -> mAP, as in the reference
(lib/det/ap_helper.py:24-150 parse_predictions, :152-193 parse_groundtruths, :195-249 APCalculator;
lib/det/nms.py:110-150 nms_3d_faster_samecls; lib/det/eval_det.py:21-52 voc_ap, :74-158 eval_det_cls, :165-204 eval_det;
lib/det/box_util.py:97-121 box3d_iou).  Host-side numpy like the reference (it is an epoch-end metric, not a per-step
kernel), vectorised over boxes instead of the reference's per-box python loops; float64 arithmetic as numpy does there.
Used for the mAP@0.5 parity report (HIP detector vs CPU oracle on identical weights and scenes)."""
import numpy as np

POST_DICT = {"remove_empty_box": False, "use_3d_nms": True, "nms_iou": 0.25, "use_old_type_nms": False, "cls_nms": True,
             "per_class_proposal": True, "conf_thresh": 0.09}   # scripts/eval.py:132-143, model/pipeline.py:75-87


def nms_3d_faster_samecls(boxes, overlap_threshold, old_type=False):
    """boxes (n,8) = [x1,y1,z1,x2,y2,z2,score,cls] -> kept indices, highest score first (nms.py:110-150)"""
    x1, y1, z1, x2, y2, z2, score, cls = (boxes[:, i] for i in range(8))
    area = (x2 - x1) * (y2 - y1) * (z2 - z1)
    I = np.argsort(score)
    pick = []
    while I.size != 0:
        i = I[-1]
        pick.append(i)
        rest = I[:-1]
        l = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        w = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        h = np.maximum(0, np.minimum(z2[i], z2[rest]) - np.maximum(z1[i], z1[rest]))
        inter = l * w * h
        o = inter / area[rest] if old_type else inter / (area[i] + area[rest] - inter + 1e-8)
        o = o * (cls[i] == cls[rest])
        I = rest[o <= overlap_threshold]
    return pick


def nms_pred_mask_device(data_dict, nms_iou=0.25, old_type=False, numpy_tie_order=False):
    """the class-aware 3D NMS of all scenes in one launch on the device (csrc/nms.hip) -> pred_mask (B,K) float tensor;
    same picks as the host loop below (float64 arithmetic).  Exactly tied scores: numpy's argsort (the reference's visiting
    order) leaves their order to its sort implementation; `numpy_tie_order` computes that order on the host and hands it to
    the kernel, otherwise ties go to the later proposal first."""
    import ctypes as C
    import torch
    from . import _lib
    boxes = data_dict["proposal_bbox_batched"].detach().float()
    B, K = boxes.shape[:2]
    cls = data_dict["proposal_sem_cls_batched"].detach().float() - 2
    cls = torch.where(cls < 0, torch.full_like(cls, 17.0), cls)
    b8 = torch.cat([boxes.min(2).values, boxes.max(2).values, data_dict["proposal_scores_batched"].detach().float().unsqueeze(-1),
                    cls.unsqueeze(-1)], -1).contiguous()
    valid = (data_dict["proposal_batch_mask"].detach() == 1).float().contiguous()
    pick = torch.empty((B, K), dtype=torch.float32, device=boxes.device)
    visit = None
    if numpy_tie_order:
        sc, va = b8[:, :, 6].cpu().numpy().astype(np.float64), valid.cpu().numpy() == 1
        vis = np.full((B, K), -1, np.int32)
        for i in range(B):
            inds = np.where(va[i])[0]
            o = inds[np.argsort(sc[i][va[i]])[::-1]]          # nms.py:122: I = np.argsort(score), visited from the end
            vis[i, :len(o)] = o
        visit = torch.from_numpy(vis).to(boxes.device)
    with torch.cuda.device(boxes.device):
        _lib.check(_lib.lib().d3_nms3d_samecls(C.c_void_p(b8.data_ptr()), C.c_void_p(valid.data_ptr()),
                                               C.c_void_p(visit.data_ptr()) if visit is not None else None, B, K, float(nms_iou), int(old_type),
                                               C.c_void_p(pick.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nms3d_samecls")
    return pick


def parse_predictions(data_dict, config_dict=POST_DICT, num_class=18, device_nms=None, numpy_tie_order=False):
    """-> per scene list of (class, corners (8,3), score) after class-aware 3D NMS and the confidence threshold.
    device_nms (default: when the proposals live on the GPU): the NMS of all scenes runs as one kernel; the per-scene numpy
    loop is the host form the reference has."""
    cfg = dict(POST_DICT); cfg.update(config_dict or {})
    assert cfg["use_3d_nms"] and cfg["cls_nms"] and not cfg["remove_empty_box"], "only the configuration the reference uses"
    g = lambda k: data_dict[k].detach().cpu().numpy()
    boxes = g("proposal_bbox_batched")
    cls = g("proposal_sem_cls_batched") - 2
    cls[cls < 0] = 17
    nonempty, prob = g("proposal_batch_mask"), g("proposal_scores_batched")
    B, K = prob.shape
    pred_mask = np.zeros((B, K))
    if device_nms is None:
        device_nms = bool(getattr(data_dict["proposal_bbox_batched"], "is_cuda", False)) and K <= 256
    if device_nms:
        pred_mask = nms_pred_mask_device(data_dict, cfg["nms_iou"], cfg["use_old_type_nms"], numpy_tie_order).cpu().numpy().astype(np.float64)
    for i in range(B if not device_nms else 0):
        b = np.zeros((K, 8))
        b[:, 0:3], b[:, 3:6] = boxes[i].min(1), boxes[i].max(1)
        b[:, 6], b[:, 7] = prob[i], cls[i]
        inds = np.where(nonempty[i] == 1)[0]
        if len(inds) == 0:
            continue
        pick = nms_3d_faster_samecls(b[nonempty[i] == 1], cfg["nms_iou"], cfg["use_old_type_nms"])
        pred_mask[i, inds[pick]] = 1
    data_dict["pred_mask"] = pred_mask
    out = []
    for i in range(B):
        cur = []
        if cfg["per_class_proposal"]:
            for c in range(num_class):
                cur += [(c, boxes[i, j], prob[i, j]) for j in range(K)
                        if pred_mask[i, j] == 1 and cls[i, j] == c and prob[i, j] > cfg["conf_thresh"]]
        else:
            cur = [(cls[i, j], boxes[i, j], prob[i, j]) for j in range(K) if pred_mask[i, j] == 1 and prob[i, j] > cfg["conf_thresh"]]
        out.append(cur)
    data_dict["batch_pred_map_cls"] = out
    return out


def parse_groundtruths(data_dict, config_dict=None):
    g = lambda k: data_dict[k].detach().cpu().numpy()
    corners, mask, cls = g("gt_bbox"), g("gt_bbox_label"), g("sem_cls_label")
    out = [[(cls[i, j], corners[i, j]) for j in range(corners.shape[1]) if mask[i, j] == 1] for i in range(corners.shape[0])]
    data_dict["batch_gt_map_cls"] = out
    return out


def box3d_iou(c1, c2):
    """AABB IoU from (8,3) corners (box_util.py:97-121)"""
    mn1, mx1, mn2, mx2 = c1.min(0), c1.max(0), c2.min(0), c2.max(0)
    inter = np.maximum(np.minimum(mx1, mx2) - np.maximum(mn1, mn2), 0).prod()
    return inter / ((mx1 - mn1).prod() + (mx2 - mn2).prod() - inter + 1e-8)


def voc_ap(rec, prec):
    """area under the monotone precision envelope (eval_det.py:21-52, use_07_metric=False)"""
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def eval_det_cls(pred, gt, ovthresh):
    """one class: pred {scene: [(box, score)]}, gt {scene: [box]} (eval_det.py:74-158)"""
    recs, npos = {}, 0
    for sid, boxes in gt.items():
        recs[sid] = {"bbox": np.array(boxes), "det": [False] * len(boxes)}
        npos += len(boxes)
    for sid in pred:
        recs.setdefault(sid, {"bbox": np.array([]), "det": []})
    ids, conf, BB = [], [], []
    for sid in pred:
        for box, score in pred[sid]:
            ids.append(sid); conf.append(score); BB.append(box)
    conf, BB = np.array(conf), np.array(BB)
    order = np.argsort(-conf)
    BB = BB[order, ...] if len(order) else BB
    ids = [ids[x] for x in order]
    nd = len(ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = recs[ids[d]]
        bb = BB[d].astype(float)
        ovmax, jmax = -np.inf, -1
        G = R["bbox"].astype(float)
        for j in range(G.shape[0] if G.size > 0 else 0):
            iou = box3d_iou(bb, G[j])
            if iou > ovmax:
                ovmax, jmax = iou, j
        if ovmax > ovthresh and not R["det"][jmax]:
            tp[d] = 1.; R["det"][jmax] = 1
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos + 1e-8)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec)


def eval_det(pred_all, gt_all, ovthresh=0.25):
    """(eval_det.py:165-204)"""
    pred, gt = {}, {}
    for sid, items in pred_all.items():
        for c, box, score in items:
            pred.setdefault(c, {}).setdefault(sid, []).append((box, score))
            gt.setdefault(c, {}).setdefault(sid, [])
    for sid, items in gt_all.items():
        for c, box in items:
            gt.setdefault(c, {}).setdefault(sid, []).append(box)
    rec, prec, ap = {}, {}, {}
    for c in gt:
        rec[c], prec[c], ap[c] = eval_det_cls(pred.get(c, {}), gt[c], ovthresh)
    return rec, prec, ap


class APCalculator:
    """(ap_helper.py:195-249)"""

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None):
        self.ap_iou_thresh, self.class2type_map = ap_iou_thresh, class2type_map
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls):
        assert len(batch_pred_map_cls) == len(batch_gt_map_cls)
        for p, g in zip(batch_pred_map_cls, batch_gt_map_cls):
            self.gt_map_cls[self.scan_cnt], self.pred_map_cls[self.scan_cnt] = g, p
            self.scan_cnt += 1

    def compute_metrics(self):
        rec, prec, ap = eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=self.ap_iou_thresh)
        name = lambda k: self.class2type_map[k] if self.class2type_map else str(k)
        ret = {"%s Average Precision" % name(k): ap[k] for k in sorted(ap)}
        ret["mAP"] = np.mean(list(ap.values()))
        recs = []
        for k in sorted(ap):
            r = rec[k][-1] if len(rec[k]) else 0
            ret["%s Recall" % name(k)] = r
            recs.append(r)
        ret["AR"] = np.mean(recs)
        return ret

    def reset(self):
        self.gt_map_cls, self.pred_map_cls, self.scan_cnt = {}, {}, 0
