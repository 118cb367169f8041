"""d3net_amd.evaluator (host numpy, like the reference's) against the reference's own APCalculator pipeline
(golden: tests/golden/evaluator_golden.npz from lib/det/ap_helper.py + eval_det.py + nms.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_nms_ap_map_match_reference():
    from gen_evaluator_golden import evaluator_inputs
    from d3net_amd import evaluator as ev
    g = np.load(os.path.join(HERE, "golden", "evaluator_golden.npz"))
    d = {k: torch.from_numpy(v) for k, v in evaluator_inputs().items()}
    preds, gts = ev.parse_predictions(d), ev.parse_groundtruths(d)
    assert np.array_equal(d["pred_mask"].astype(np.uint8), g["pred_mask"])
    assert [len(p) for p in preds] == g["n_pred"].tolist() and [len(x) for x in gts] == g["n_gt"].tolist()
    assert np.array_equal(np.array([p[0] for p in preds[0]]), g["pred_cls0"]) and np.allclose([p[2] for p in preds[0]], g["pred_score0"])
    for thr in (0.25, 0.5):
        ap = ev.APCalculator(thr)
        ap.step(preds, gts)
        m = ap.compute_metrics()
        assert abs(m["mAP"] - float(g["mAP@%s" % thr])) < 1e-12 and abs(m["AR"] - float(g["AR@%s" % thr])) < 1e-12
        got = np.array([m["%d Average Precision" % k] for k in sorted(int(x.split()[0]) for x in m if x.endswith("Average Precision"))])
        assert np.allclose(got, g["AP@%s" % thr], atol=1e-12)
    assert 0.05 < float(g["mAP@0.5"]) < 0.99   # a non-degenerate case
