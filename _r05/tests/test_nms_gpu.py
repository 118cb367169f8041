"""Device-side non-maximum suppressions (csrc/nms.hip) -- SURVEY.md section 8(f) rank 1:
  * class-aware 3D box NMS of parse_predictions against the REFERENCE's own pred_mask (tests/golden/evaluator_golden.npz,
    generated from lib/det/ap_helper.py + nms.py) and against the host restatement on random crowded boxes;
  * instance-mask IoUs + greedy NMS of PointGroup.test (model/pointgroup.py:577-601, lib/utils/eval.py:75-97) against the
    reference's dense formulation (mask matrix product + get_nms_instances restated in numpy)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_box_nms_matches_reference_golden_and_host_form(dev):
    from gen_evaluator_golden import evaluator_inputs
    from d3net_amd import evaluator as ev
    g = np.load(os.path.join(HERE, "golden", "evaluator_golden.npz"))
    d = {k: torch.from_numpy(v).to(dev) for k, v in evaluator_inputs().items()}
    # device tensors -> one NMS launch for all scenes.  The golden proposals contain exactly tied scores (28 distinct values
    # among 33 boxes): the visiting order of ties is numpy's, handed to the kernel
    preds = ev.parse_predictions(d, numpy_tie_order=True)
    assert np.array_equal(d["pred_mask"].astype(np.uint8), g["pred_mask"])
    assert [len(p) for p in preds] == g["n_pred"].tolist()
    ap = ev.APCalculator(0.5)
    ap.step(preds, ev.parse_groundtruths(d))
    assert abs(ap.compute_metrics()["mAP"] - float(g["mAP@0.5"])) < 1e-12
    # crowded random scenes: many overlapping boxes, few classes
    rng = np.random.default_rng(0)
    B, K = 6, 128
    ctr = rng.random((B, K, 1, 3)) * 2
    size = rng.random((B, K, 1, 3)) * 0.8 + 0.1
    sgn = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)
    dd = {"proposal_bbox_batched": torch.from_numpy((ctr + sgn * size / 2).astype(np.float32)),
          "proposal_sem_cls_batched": torch.from_numpy(rng.integers(0, 5, (B, K)).astype(np.float32)),
          "proposal_scores_batched": torch.from_numpy(rng.random((B, K)).astype(np.float32)),
          "proposal_batch_mask": torch.from_numpy((rng.random((B, K)) < 0.8).astype(np.float32))}
    dd["proposal_batch_mask"][5] = 0                                                   # a scene without proposals
    host = dict(dd); ev.parse_predictions(host, device_nms=False)
    devd = {k: v.to(dev) for k, v in dd.items()}; ev.parse_predictions(devd)
    assert np.array_equal(host["pred_mask"], devd["pred_mask"]) and 0 < host["pred_mask"].sum() < (dd["proposal_batch_mask"] == 1).sum()


def _nms_instances(cross_ious, scores, thr):
    """lib/utils/eval.py:75-97 (stable argsort for reproducible ties)"""
    ixs = np.argsort(-scores, kind="stable")
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        rem = np.where(cross_ious[i, ixs[1:]] > thr)[0] + 1
        ixs = np.delete(ixs, rem)
        ixs = np.delete(ixs, 0)
    return np.array(pick, np.int64)


def test_instance_mask_iou_and_nms_match_dense_reference(dev):
    from d3net_amd import _lib
    rng = np.random.default_rng(1)
    N, P1, P2 = 20000, 40, 40
    # two clusterings of the same points (a point is in at most one cluster of each), overlapping heavily
    lab1 = rng.integers(-1, P1, N); lab2 = np.where(rng.random(N) < 0.7, (lab1 + 3) % P2, rng.integers(-1, P2, N)); lab2[lab1 < 0] = -1
    idx, off = [], [0]
    for c in range(P1):
        pts = np.nonzero(lab1 == c)[0]; idx += [(c, p) for p in pts]; off.append(off[-1] + len(pts))
    for c in range(P2):
        pts = np.nonzero(lab2 == c)[0]; idx += [(P1 + c, p) for p in pts]; off.append(off[-1] + len(pts))
    idx = np.array(idx, np.int32); off = np.array(off, np.int32)
    P = P1 + P2
    scores = rng.random(P).astype(np.float32)
    keep = (rng.random(P) < 0.85)
    # reference formulation (model/pointgroup.py:577-589)
    mask = np.zeros((P, N), np.float32); mask[idx[:, 0], idx[:, 1]] = 1
    inter = mask @ mask.T
    npnt = mask.sum(1)
    ref_iou = inter / (npnt[:, None] + npnt[None, :] - inter)
    kept = np.nonzero(keep)[0]
    ref_pick = kept[_nms_instances(ref_iou[np.ix_(kept, kept)], scores[kept], 0.3)]
    L = _lib.lib()
    t = lambda a: torch.from_numpy(a).to(dev)
    cidx, offd, sc, kp = t(idx), t(off), t(scores), t(keep.astype(np.uint8))
    ious = torch.empty((P, P), dtype=torch.float32, device=dev)
    member = torch.empty(2 * N, dtype=torch.int32, device=dev)
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    order = torch.empty(P, dtype=torch.int32, device=dev); picked = torch.empty(P, dtype=torch.int32, device=dev)
    p_ = lambda x: C.c_void_p(x.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.d3_instance_cross_iou(p_(cidx), p_(offd), idx.shape[0], P, N, p_(ious), p_(member), p_(flags), st) == 0
    assert L.d3_nms_matrix(p_(ious), p_(sc), p_(kp), P, 0.3, p_(order), p_(picked), p_(flags[1:]), st) == 0
    over, n = flags.tolist()
    assert over == 0
    assert np.allclose(ious.cpu().numpy(), ref_iou, rtol=1e-6, atol=0)
    assert np.array_equal(picked[:n].cpu().numpy(), ref_pick) and 3 < n < len(kept)


def test_predict_instances_runs_on_a_scene(dev):
    """PointGroup.predict_instances (the predictions PointGroup.test writes out): picks are a subset of the thresholded
    proposals, no two picked proposals overlap above the NMS threshold, scores descend"""
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    cfg = default_conf(overrides={"model": {"blocks": [1, 2, 3]}})
    torch.manual_seed(0)
    model = PointGroup(cfg).to(dev).eval()
    model.teacher = True
    with torch.no_grad():
        model.score_linear.bias.fill_(2.0)
    out = model.predict_instances(S.make_batch([S.small_scene(dims=(44, 36, 20), n_boxes=4, seed=3)], dev))
    pick, sc = out["pick"].cpu().numpy(), out["scores"].cpu().numpy()
    assert len(pick) >= 4 and (np.diff(sc) <= 1e-7).all()
    io = out["cross_ious"].cpu().numpy()[np.ix_(pick, pick)]
    assert (io[~np.eye(len(pick), dtype=bool)] <= cfg.test.TEST_NMS_THRESH + 1e-6).all()
