"""PipelineNet modes 0 / 1 / 2 end to end on the device (detector -> speaker / listener, losses, backward)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**model):
    from d3net_amd.config import default_conf
    base = {"model": {"blocks": [1, 2, 3], "num_graph_steps": 2, "num_locals": 10, "use_relation": True, "use_orientation": True,
                      "match_type": "Transformer", "use_lang_classifier": True, "use_bidir": False, "num_bbox_class": 18,
                      "loss_type": "cross_entropy"},
            "data": {"num_des_per_scene": 4, "max_spk_len": 30, "max_lis_len": 126, "min_iou_threshold": 0.25, "num_ori_bins": 6},
            "train": {"use_rl": False, "sample_topn": 1}}
    base["model"].update(model)
    return default_conf(overrides=base)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pipeline_modes_run_and_train(dev, mode):
    from d3net_amd import synthetic as S
    from d3net_amd.pipeline import PipelineNet
    flags = {0: dict(no_captioning=True, no_grounding=True), 1: dict(no_captioning=False, no_grounding=True),
             2: dict(no_captioning=True, no_grounding=False)}[mode]
    cfg = _cfg(**flags)
    V = 200
    ds = {"train": types.SimpleNamespace(vocabulary=S.make_vocabulary(V), glove=np.random.default_rng(0).standard_normal((V, 300)).astype(np.float32))}
    net = PipelineNet(cfg, ds).to(dev).train()
    assert net.mode == mode
    net.detector.teacher = True
    scenes = [S.small_scene(dims=(40, 32, 20), n_boxes=3, seed=s) for s in (3, 4)]
    batch = S.add_language(S.make_batch(scenes, dev), dev, chunk=4, vocab=V)
    if mode == 1:
        batch["lang_len"] = batch["spk_lang_len"]
    loss, d = net.training_step(batch)
    assert torch.isfinite(loss)
    loss.backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)
    if mode == 1:
        assert d["lang_cap"].shape[0] == 8 and d["bbox_feature"].shape == (2, 128, 128) and "train_loss/captioning_loss" in net.logged
    if mode == 2:
        assert d["cluster_ref"].shape == (8, 128) and 0 <= float(d["ref_acc_mean"]) <= 1 and "train_score/ref_iou_rate_0.5" in net.logged
    opt, _ = net.configure_optimizers()
    opt[0].step()


def test_validation_hooks_and_inference_forward(dev):
    """PipelineNet.validation_step / validation_epoch_end / forward with the reference's call pattern
    (scripts/train.py:338-365 through Lightning; scripts/eval.py:154,205-209): mode 1 returns the dense-caption candidates
    of a batch and scores them over the epoch (CIDEr / BLEU / ROUGE keys), mode 2 logs the grounding scores, mode 0 the
    detector losses; forward() is the inference chain."""
    from d3net_amd import synthetic as S
    from d3net_amd.pipeline import PipelineNet
    V = 200
    scenes = [S.small_scene(dims=(40, 32, 20), n_boxes=3, seed=s) for s in (3, 4)]
    chunked, organized = S.make_language_corpus(2, chunk=4, vocab=V, objects_per_scene=3)
    for mode, flags in ((0, dict(no_captioning=True, no_grounding=True)), (1, dict(no_captioning=False, no_grounding=True)),
                        (2, dict(no_captioning=True, no_grounding=False))):
        cfg = _cfg(**flags)
        tr = types.SimpleNamespace(vocabulary=S.make_vocabulary(V), glove=np.random.default_rng(0).standard_normal((V, 300)).astype(np.float32),
                                   chunked_data=chunked, organized=organized, raw_data=S.corpus_raw_data(organized))
        net = PipelineNet(cfg, {"train": tr, "val": tr}).to(dev).eval()
        net.detector.teacher = True
        batch = S.add_language(S.make_batch(scenes, dev), dev, chunk=4, vocab=V)
        if mode == 1:
            batch["lang_len"] = batch["spk_lang_len"]          # the speaker's lang_len is the caption length
        out = net.validation_step(dict(batch), 0)
        if mode == 0:
            assert out is None and "val_loss/total_loss" in net.logged
        elif mode == 1:
            assert isinstance(out, dict) and len(out) > 0
            k, v = next(iter(out.items()))
            assert k.startswith("scene000") and v["caption"].startswith("sos") and 0.0 <= v["iou"] <= 1.0 and len(v["box"]) == 8
            log = net.validation_epoch_end([out])
            assert set(log) == {"bleu-1", "bleu-2", "bleu-3", "bleu-4", "cider", "meteor", "rouge"} and "val_score/cider" in net.logged
            assert all(np.isfinite(float(x)) for x in log.values())
        else:
            assert out is None and 0.0 <= float(net.logged["val_score/ref_iou_rate_0.5"]) <= 1.0 and "val_score/lang_acc" in net.logged
        with torch.no_grad():
            d = net(dict(batch))
        assert "proposal_bbox_batched" in d and (mode != 1 or "lang_cap" in d) and (mode != 2 or d["cluster_ref"].shape == (8, 128))
