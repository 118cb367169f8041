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


def test_map_parity_of_a_trained_detector_on_its_own_predictions(dev):
    """the teacher-driven test above fixes the clusters, so bf16 can only permute score ranks.  Here the detector is
    TRAINED (400 AdamW steps on 12 synthetic scenes, bf16 MFMA path + native executor, as bench.py runs it) until its own
    semantic / offset predictions cluster, then evaluated with `teacher = False` and no score bias: HIP bf16 executor vs the
    fp32 CPU oracle on the same weights and scenes, mAP@0.5 through the evaluator pinned to the reference's
    (scripts/eval.py:128-166, lib/det/ap_helper.py:195-249).  Bound: |mAP_hip - mAP_oracle| <= 0.5 % of the oracle's."""
    from d3net_amd import synthetic as S, evaluator as ev
    from d3net_amd.config import default_conf
    from d3net_amd.optim import FusedAdamW
    from d3net_amd.pointgroup import PointGroup
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg = default_conf(overrides={"model": {"blocks": [1, 2, 3]}})
    torch.manual_seed(2)
    model = PointGroup(cfg).to(dev).train()
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=4e-3, weight_decay=1e-4)
    opt.register_step_pre_hook(lambda *a: model.drop_stale_grads())
    scenes = [S.small_scene(dims=(44, 36, 20), n_boxes=4, seed=s) for s in range(20, 32)]
    for s in scenes:                      # learnable inputs: the class is readable from the features (the random ones carry nothing)
        onehot = np.eye(20, dtype=np.float32)[np.clip(s["sem_labels"], 0, 19)]
        s["feats"][:, :20] = onehot * 2 + 0.1 * s["feats"][:, :20]
    batches = [S.make_batch(scenes[i:i + 2], dev) for i in range(0, 12, 2)]
    for it in range(400):
        model.zero_grad(set_to_none=True)
        loss, _ = model.training_step(dict(batches[it % len(batches)]))
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    calc = {k: ev.APCalculator(0.5) for k in ("oracle", "bf16")}
    orc = PointGroupOracle(cfg, model.state_dict())
    n_prop = {"oracle": 0, "bf16": 0}
    for bi in range(0, 12, 2):
        pair = scenes[bi:bi + 2]
        rand = torch.rand(2, 3); perms = [torch.randperm(cfg.model.max_num_proposal) for _ in pair]
        host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in S.make_batch(pair, dev).items()}
        gt = _gt_keys(host)
        with torch.no_grad():
            od = orc.feed(host, 0, rand=rand, perms=perms)
        od.update(gt)
        calc["oracle"].step(ev.parse_predictions(od), ev.parse_groundtruths(od))
        n_prop["oracle"] += int(od["proposal_batch_mask"].sum())
        b = S.make_batch(pair, dev); b["cluster_rand"], b["slot_perms"] = rand, perms
        with torch.no_grad():
            d = model.feed(b, 0)
        d.update(gt)
        calc["bf16"].step(ev.parse_predictions(d, device_nms=False), ev.parse_groundtruths(d))
        n_prop["bf16"] += int(d["proposal_batch_mask"].sum())
    m = {k: v.compute_metrics()["mAP"] for k, v in calc.items()}
    print("trained detector: mAP@0.5 oracle %.4f, HIP bf16 %.4f; proposals %s; final loss %.3f" % (m["oracle"], m["bf16"], n_prop, float(loss)))
    assert m["oracle"] > 0.2, m           # the detector detects with its own predictions
    assert abs(m["bf16"] - m["oracle"]) <= 0.005 * m["oracle"] + 1e-12, m
