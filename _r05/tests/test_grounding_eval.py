"""d3net_amd.grounding_eval.get_eval (batched, loop-free) against golden vectors from the REFERENCE's own
lib/grounding/eval_helper.get_eval (tests/golden/gen_grounding_eval_golden.py).  Host-side torch code: runs on CPU."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_get_eval_matches_reference_golden():
    from gen_grounding_eval_golden import eval_inputs
    from d3net_amd.grounding_eval import get_eval
    g = np.load(os.path.join(HERE, "golden", "grounding_eval_golden.npz"))
    d = {k: torch.from_numpy(v) for k, v in eval_inputs().items()}
    d = get_eval(d, grounding=True, use_lang_classifier=True)
    assert np.allclose(np.array(d["ref_acc"], np.float32), g["ref_acc"])
    for k in ("ref_acc_mean", "ref_iou", "best_ious", "ref_iou_mean", "best_ious_mean", "lang_acc", "pred_bboxes", "cluster_ref"):
        assert np.allclose(d[k].numpy(), g[k], rtol=1e-5, atol=1e-6), k
    assert abs(d["ref_iou_rate_0.25"] - float(g["rate25"])) < 1e-6 and abs(d["ref_iou_rate_0.5"] - float(g["rate5"])) < 1e-6
    assert d["ref_multiple_mask"] == g["multiple"].tolist() and d["ref_others_mask"] == g["others"].tolist()
    assert 0.2 < float(g["rate25"]) < 0.9     # the fixture exercises both outcomes
