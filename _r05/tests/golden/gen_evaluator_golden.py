#!/usr/bin/env python3
"""Generates tests/golden/evaluator_golden.npz by RUNNING THE REFERENCE's own detection evaluator
(lib/det/ap_helper.py parse_predictions / parse_groundtruths / APCalculator, lib/det/eval_det.py, lib/det/nms.py).
`data.scannet.model_util_scannet` (imported by ap_helper for a function this path never calls) pulls in dataset
constants and mesh I/O; it is registered as a placeholder exposing only the name ap_helper imports.
Run in the build container only.  Inputs are rebuilt in the tests from `evaluator_inputs()`."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def evaluator_inputs(B=3, K=128, seed=21):
    rng = np.random.default_rng(seed)
    sgn = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)
    gt = np.zeros((B, 128, 8, 3), np.float32); gt_mask = np.zeros((B, 128), np.float32); gt_cls = np.zeros((B, 128), np.int64)
    pred = np.zeros((B, K, 8, 3), np.float32); pm = np.zeros((B, K), np.float32); pc = np.zeros((B, K), np.float32); ps = np.zeros((B, K), np.float32)
    for b in range(B):
        n = int(rng.integers(8, 20))
        c = rng.random((n, 3)).astype(np.float32) * np.array([4, 3, 2], np.float32); s = rng.random((n, 3)).astype(np.float32) * 0.8 + 0.3
        gt[b, :n] = c[:, None] + sgn[None] * s[:, None] / 2; gt_mask[b, :n] = 1; gt_cls[b, :n] = rng.integers(0, 18, n)
        slots = rng.permutation(K)[:3 * n]
        for q, sl in enumerate(slots):          # noisy copies (duplicates -> NMS), a few false positives
            o = q % n
            jit = rng.normal(0, 0.06 if q < 2 * n else 0.5, 3).astype(np.float32)
            pred[b, sl] = gt[b, o] + jit
            pm[b, sl] = 1
            pc[b, sl] = gt_cls[b, o] + 2 if rng.random() > 0.15 else float(rng.integers(0, 20))   # semantic ids (+2 offset, 0/1 -> "others")
            ps[b, sl] = float(rng.random()) if rng.random() > 0.1 else 0.05
    return dict(proposal_bbox_batched=pred, proposal_sem_cls_batched=pc, proposal_batch_mask=pm, proposal_scores_batched=ps,
                gt_bbox=gt, gt_bbox_label=gt_mask, sem_cls_label=gt_cls)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    m = types.ModuleType("data.scannet.model_util_scannet"); m.extract_pc_in_box3d = None
    sys.modules["data"] = types.ModuleType("data"); sys.modules["data.scannet"] = types.ModuleType("data.scannet")
    sys.modules["data.scannet.model_util_scannet"] = m
    from lib.det.ap_helper import parse_predictions, parse_groundtruths, APCalculator
    inp = evaluator_inputs()
    d = {k: torch.from_numpy(v) for k, v in inp.items()}
    cfg = {"remove_empty_box": False, "use_3d_nms": True, "nms_iou": 0.25, "use_old_type_nms": False, "cls_nms": True,
           "per_class_proposal": True, "conf_thresh": 0.09, "dataset_config": types.SimpleNamespace(num_class=18)}
    out = {}
    preds = parse_predictions(d, cfg); gts = parse_groundtruths(d, cfg)
    out["pred_mask"] = d["pred_mask"].astype(np.uint8)
    out["n_pred"] = np.array([len(p) for p in preds]); out["n_gt"] = np.array([len(g) for g in gts])
    out["pred_cls0"] = np.array([p[0] for p in preds[0]]); out["pred_score0"] = np.array([p[2] for p in preds[0]])
    for thr in (0.25, 0.5):
        ap = APCalculator(thr)
        ap.step(preds, gts)
        m_ = ap.compute_metrics()
        out["mAP@%s" % thr], out["AR@%s" % thr] = np.float64(m_["mAP"]), np.float64(m_["AR"])
        out["AP@%s" % thr] = np.array([m_["%d Average Precision" % k] for k in sorted(int(x.split()[0]) for x in m_ if x.endswith("Average Precision"))])
    np.savez_compressed(os.path.join(HERE, "evaluator_golden.npz"), **out)
    print("wrote evaluator_golden.npz", {k: (v.shape, float(v) if v.shape == () else None) for k, v in out.items()})


if __name__ == "__main__":
    main()
