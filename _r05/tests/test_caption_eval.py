"""d3net_amd.caption_eval (batched GIoU cost, Hungarian assignment, CIDEr@kIoU) against golden vectors from the REFERENCE's
own lib/captioning/eval_helper.py + lib/capeval/cider (tests/golden/gen_caption_eval_golden.py).  Host-side: runs on CPU."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_assignment_and_cider_match_reference_golden():
    from gen_caption_eval_golden import caption_inputs
    from d3net_amd import caption_eval as ce
    g = np.load(os.path.join(HERE, "golden", "caption_eval_golden.npz"))
    inp = caption_inputs()
    t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
    gious = ce.generalized_box3d_iou(t["pred_boxes"], t["gt_boxes"], t["gt_box_masks"].sum(1).long())
    assert np.allclose(gious.numpy(), g["gious"], rtol=1e-5, atol=1e-6)
    cands = ce.assign_dense_caption(t["pred_captions"], t["pred_boxes"], t["gt_boxes"], t["gt_box_ids"], t["gt_box_masks"],
                                    inp["scene_list"], inp["vocab"]["idx2word"], inp["vocab"]["special_tokens"])
    keys = sorted(cands)
    assert keys == g["keys"].tolist()
    assert np.allclose([cands[k]["iou"] for k in keys], g["ious"], rtol=1e-5, atol=1e-7)
    assert [cands[k]["caption"] for k in keys] == g["captions"].tolist()
    for thr in (0.25, 0.5):
        mean, scores, ckeys = ce.score_captions(cands, inp["raw"], max_len=30, min_iou=thr)
        assert ckeys == g["corpus_keys"].tolist()
        assert abs(mean - float(g["cider_%s" % thr])) <= 5e-3 * abs(float(g["cider_%s" % thr]))   # the north-star bound: 0.5 %
        assert np.allclose(scores, g["cider_scores_%s" % thr], rtol=1e-9, atol=1e-12)            # and in fact to rounding
        from d3net_amd import caption_metrics as cm
        corpus = ce.prepare_corpus(inp["raw"], cands, 30)
        kept = {k: v["caption"] for k, v in cands.items() if v["iou"] >= thr}
        refs, hyps = [corpus[k] for k in ckeys], [kept.get(k, "sos eos") for k in ckeys]
        bleu, bleu_list = cm.bleu_scores(refs, hyps)
        assert np.allclose(bleu, g["bleu_%s" % thr], rtol=1e-12) and np.allclose(bleu_list, g["bleu_list_%s" % thr], rtol=1e-12)
        rouge, rouges = cm.rouge_l_scores(refs, hyps)
        assert abs(rouge - float(g["rouge_%s" % thr])) < 1e-12 and np.allclose(rouges, g["rouge_scores_%s" % thr], rtol=1e-12)
    assert float(g["cider_0.25"]) > float(g["cider_0.5"]) > 0   # the fixture has matches on both sides of the thresholds
