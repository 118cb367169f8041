"""SURVEY.md section 8(f) rank 4 ON THE DEVICE: the golden fixtures produced by the reference's own evaluation harness
(lib/captioning/eval_helper.py:102-307, lib/utils/bbox.py:645-881, lib/capeval/*; lib/grounding/eval_helper.py:28-137 --
tests/golden/gen_caption_eval_golden.py, gen_grounding_eval_golden.py) pushed through `d3net_amd.caption_eval` /
`d3net_amd.grounding_eval` with CUDA tensors, as `PipelineNet.validation_step` feeds them: GIoU cost matrix 1e-5, identical
Hungarian assignments and captions, CIDEr / BLEU-1..4 / ROUGE-L @0.25 / @0.5 IoU and Acc@kIoU to rounding.  (The CPU-tensor
twins of these tests are tests/test_caption_eval.py and tests/test_grounding_eval.py.)"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_dense_caption_assignment_and_corpus_scores_with_device_tensors(dev):
    from gen_caption_eval_golden import caption_inputs
    from d3net_amd import caption_eval as ce, caption_metrics as cm
    g = np.load(os.path.join(HERE, "golden", "caption_eval_golden.npz"))
    inp = caption_inputs()
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    gious = ce.generalized_box3d_iou(t["pred_boxes"], t["gt_boxes"], t["gt_box_masks"].sum(1).long())
    assert gious.is_cuda
    assert np.allclose(gious.cpu().numpy(), g["gious"], rtol=1e-5, atol=1e-6)
    # the entry point validation_step calls (model/pipeline.py:457-643 -> eval_helper.py:248-262)
    d = dict(lang_cap=t["pred_captions"], proposal_bbox_batched=t["pred_boxes"], gt_bbox=t["gt_boxes"], gt_bbox_object_id=t["gt_box_ids"],
             gt_bbox_label=t["gt_box_masks"], scene_id=inp["scene_list"])
    cands = ce.eval_caption_step(d, inp["vocab"])
    keys = sorted(cands)
    assert keys == g["keys"].tolist()
    assert np.allclose([cands[k]["iou"] for k in keys], g["ious"], rtol=1e-5, atol=1e-7)
    assert [cands[k]["caption"] for k in keys] == g["captions"].tolist()          # identical assignments -> identical captions
    for thr in (0.25, 0.5):
        bleu, cider, rouge, meteor = ce.eval_caption_epoch(cands, inp["raw"], max_len=30, min_iou=thr)
        assert abs(cider[0] - float(g["cider_%s" % thr])) <= 1e-9 * abs(float(g["cider_%s" % thr]))
        assert np.allclose(cider[1], g["cider_scores_%s" % thr], rtol=1e-9, atol=1e-12)
        assert np.allclose(bleu[0], g["bleu_%s" % thr], rtol=1e-12) and np.allclose(bleu[1], g["bleu_list_%s" % thr], rtol=1e-12)
        assert abs(rouge[0] - float(g["rouge_%s" % thr])) < 1e-12 and np.allclose(rouge[1], g["rouge_scores_%s" % thr], rtol=1e-12)
        mean, scores, ckeys = ce.score_captions(cands, inp["raw"], max_len=30, min_iou=thr)
        assert ckeys == g["corpus_keys"].tolist() and abs(mean - cider[0]) < 1e-15


def test_grounding_get_eval_with_device_tensors(dev):
    from gen_grounding_eval_golden import eval_inputs
    from d3net_amd.grounding_eval import get_eval
    g = np.load(os.path.join(HERE, "golden", "grounding_eval_golden.npz"))
    d = {k: torch.from_numpy(v).to(dev) for k, v in eval_inputs().items()}
    d = get_eval(d, grounding=True, use_lang_classifier=True)
    assert d["ref_iou"].is_cuda and d["pred_bboxes"].is_cuda
    assert np.allclose(np.array(d["ref_acc"], np.float32), g["ref_acc"])
    for k in ("ref_acc_mean", "ref_iou", "best_ious", "ref_iou_mean", "best_ious_mean", "lang_acc", "pred_bboxes", "cluster_ref"):
        assert np.allclose(d[k].cpu().numpy(), g[k], rtol=1e-5, atol=1e-6), k
    assert abs(d["ref_iou_rate_0.25"] - float(g["rate25"])) < 1e-6 and abs(d["ref_iou_rate_0.5"] - float(g["rate5"])) < 1e-6   # Acc@kIoU
    assert d["ref_multiple_mask"] == g["multiple"].tolist() and d["ref_others_mask"] == g["others"].tolist()
