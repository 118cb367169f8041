"""mAP@0.5 parity: same weights + same synthetic scenes -> HIP detector vs fp32 CPU oracle detector,
both scored by the evaluator restated from lib/det (tests/test_evaluator.py pins it to the reference's own).
Bound: |mAP_hip - mAP_oracle| <= 0.5 % of the oracle's (north_star), for the exact-fp32 and the bf16-MFMA paths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SGN = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)


def _gt_keys(batch):
    c, s = batch["center_label"].cpu().numpy(), batch["size_label"].cpu().numpy()
    cls = batch["sem_cls_label"].cpu().numpy() - 2
    cls[cls < 0] = 17
    return dict(gt_bbox=torch.from_numpy(c[:, :, None] + SGN[None, None] * s[:, :, None] / 2),
                gt_bbox_label=batch["box_label_mask"].cpu(), sem_cls_label=torch.from_numpy(cls))


def test_map_parity_hip_vs_oracle(dev):
    from d3net_amd import synthetic as S, minkowski as ME, evaluator as ev
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg = default_conf(overrides={"model": {"blocks": [1, 2, 3]}})
    torch.manual_seed(1)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = True
    with torch.no_grad():   # confident objectness so that proposals pass TEST_SCORE_THRESH whatever the random ScoreNet says
        model.score_linear.bias.fill_(3.0)
    calc = {k: ev.APCalculator(0.5) for k in ("oracle", "exact", "bf16")}
    for seed in (3, 4, 5):
        scene = S.small_scene(dims=(44, 36, 20), n_boxes=4, seed=seed)
        rand = torch.rand(2, 3); perms = [torch.randperm(cfg.model.max_num_proposal)]
        host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in S.make_batch([scene], dev).items()}
        gt = _gt_keys(host)
        orc = PointGroupOracle(cfg, model.state_dict()); orc.teacher = True
        with torch.no_grad():
            od = orc.feed(host, 0, rand=rand, perms=perms)
        od.update(gt)
        calc["oracle"].step(ev.parse_predictions(od), ev.parse_groundtruths(od))
        for name, exact in (("exact", True), ("bf16", False)):
            ME.set_exact(exact)
            try:
                b = S.make_batch([scene], dev); b["cluster_rand"], b["slot_perms"] = rand, perms
                with torch.no_grad():
                    d = model.feed(b, 0)
            finally:
                ME.set_exact(False)
            d.update(gt)
            calc[name].step(ev.parse_predictions(d), ev.parse_groundtruths(d))
    m = {k: v.compute_metrics()["mAP"] for k, v in calc.items()}
    assert m["oracle"] > 0.2, m       # a meaningful operating point, not 0 == 0
    for k in ("exact", "bf16"):
        assert abs(m[k] - m["oracle"]) <= 0.005 * m["oracle"] + 1e-12, m
