"""The benchmark's `--config listener` (BASELINE configs[3]) and `--config joint` (configs[4]) workloads under pytest: ONE bench
scene (the 40-box synthetic ScanNet scene of `bench.make_scenes`, ~160 k voxels, T = 128, V = 3004, 8 descriptions) through
`PipelineNet` mode 2 / mode 3 with the reference-precision kernels, against the fp32 CPU oracle chain

    PointGroupOracle -> listener_oracle.listener_step                                             (mode 2)
    PointGroupOracle -> speaker_oracle.graph_module -> rl_oracle.rl_sample_batch (beam search + greedy baseline)
                     -> rl_oracle.caption_reward (CIDEr-D) -> rl_oracle.moderator -> rl_oracle.rl_listener -> RL losses   (mode 3)

and the bf16 step `bench.py` times beside it.  Dropout layers are set to p = 0 (a random mask cannot be reproduced across
implementations; the golden fixtures do the same) and the match module's copy-paste coin comes from a seeded `random`.
Reference: model/listener.py:34-54, model/lang_module.py:139-176, model/match_module.py:212-336, model/pipeline.py:229-274,759-892,
model/caption_module.py:431-569, lib/grounding/loss_helper.py:133-214, lib/captioning/loss_helper.py:15-176.
"""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def l2err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def close(a, b, rel=1e-3, name=""):
    a, b = float(a), float(b)
    assert abs(a - b) <= rel * abs(b) + 1e-6, (name, a, b)


def _net(dev, config):
    import bench
    from d3net_amd.config import default_conf
    from d3net_amd.pipeline import PipelineNet
    cfg = default_conf(bench.CONF[config])
    torch.manual_seed(cfg.general.manual_seed)
    scene = bench.make_scenes(config, 0)[0]
    net = PipelineNet(cfg, bench.make_dataset(1, cfg.data.num_des_per_scene, config == "joint")).to(dev).train()
    net.detector.teacher = True
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return cfg, net, scene, bench.VOCAB


def _host(batch):
    return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}


def test_one_bench_scene_listener_step_equals_oracle_chain(dev):
    """mode 2 at bench size: identical proposals and `cluster_ref` arg-max / `cluster_labels`, `cluster_ref` / `lang_scores`
    rtol 1e-3, `ref_loss` / `lang_loss` / detector loss 1e-3 with the exact-fp32 kernels; the bf16 step beside it."""
    from d3net_amd import synthetic as S, minkowski as ME
    from oracle import listener_oracle as lo
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg, net, scene, V = _net(dev, "listener")
    assert net.mode == 2
    chunk = cfg.data.num_des_per_scene
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]
    mk = lambda: S.add_language(S.make_batch([scene], dev), dev, chunk=chunk, vocab=V)
    host = _host(mk())
    assert host["lang_feat"].shape[2] == 128 and int(host["lang_len"].max()) > 100        # T = 128 (conf/pointgroup_grounding.yaml)

    torch.set_num_threads(min(16, torch.get_num_threads()))
    orc = PointGroupOracle(cfg, net.detector.state_dict())
    orc.teacher = True
    od = orc.loss(orc.feed(host, 0, rand=rand, perms=perms))
    lp = {k: v.detach().cpu().clone() for k, v in net.listener.state_dict().items()}
    rnd = random.Random(11).random()
    with torch.no_grad():
        od.update({k: v for k, v in host.items() if k not in od})
        ol = lo.listener_step(lp, od, chunk, True, rnd)
    n_prop = int(od["proposal_batch_mask"].sum())
    assert n_prop >= 30, n_prop

    def hip_step(exact):
        net.zero_grad(set_to_none=True)
        ME.set_exact(exact)
        try:
            batch = mk()
            batch["cluster_rand"], batch["slot_perms"] = rand, perms
            random.seed(11)
            loss, d = net.training_step(batch)
            loss.backward()
        finally:
            ME.set_exact(False)
        torch.cuda.synchronize()
        assert abs(d["random"] - rnd) < 1e-12
        return loss, d

    loss, d = hip_step(True)
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2]), "cluster offsets differ"
    assert torch.equal(d["proposal_batch_mask"].cpu(), od["proposal_batch_mask"])
    assert l2err(d["proposal_feats_batched"], od["proposal_feats_batched"]) < 1e-3
    close(d["total_loss"][0], od["total_loss"], name="detector loss")
    ref, oref = d["cluster_ref"].detach().cpu(), ol["cluster_ref"]
    assert ref.shape == oref.shape == (chunk, cfg.model.max_num_proposal)
    scale = float(oref.abs().max())
    assert torch.allclose(ref, oref, rtol=1e-3, atol=1e-3 * scale), (float((ref - oref).abs().max()), scale)
    assert torch.equal(ref.argmax(-1), oref.argmax(-1)), "grounded proposal (cluster_ref arg-max)"
    assert torch.equal(d["cluster_labels"].cpu().argmax(-1), ol["cluster_labels"].argmax(-1)), "best-IoU proposal labels"
    assert torch.allclose(d["lang_scores"].detach().cpu(), ol["lang_scores"], rtol=1e-3, atol=1e-4)
    assert l2err(d["lang_emb"], ol["lang_emb"]) < 1e-4
    for k in ("ref_loss", "lang_loss", "ref_acc_mean", "lang_acc", "ref_iou_mean", "best_ious_mean"):
        close(d[k], ol[k], name=k)
    close(d["ref_iou_rate_0.25"], ol["rate25"], name="ref_iou_rate_0.25"); close(d["ref_iou_rate_0.5"], ol["rate5"], name="ref_iou_rate_0.5")
    print("bench scene, listener, exact fp32 vs oracle chain: %d proposals, ref_loss %.6f / %.6f, lang_loss %.6f / %.6f, cluster_ref rel-L2 %.2e"
          % (n_prop, float(d["ref_loss"]), float(ol["ref_loss"]), float(d["lang_loss"]), float(ol["lang_loss"]), l2err(ref, oref)))

    # ---- the bf16 step bench.py times
    loss_b, db = hip_step(False)
    assert np.array_equal(db["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1])
    close(db["total_loss"][0], od["total_loss"], 2e-2, "bf16 detector loss")
    close(db["ref_loss"], ol["ref_loss"], 2e-2, "bf16 ref_loss")
    close(db["lang_loss"], ol["lang_loss"], 1e-3, "bf16 lang_loss")        # (the language encoder does not see the backbone)
    same = int((db["cluster_ref"].argmax(-1).cpu() == oref.argmax(-1)).sum())
    assert same >= chunk - 1, same
    print("bench scene, listener, bf16 executor: ref_loss %.6f, %d / %d arg-max equal, cluster_ref rel-L2 vs oracle %.2e"
          % (float(db["ref_loss"]), same, chunk, l2err(db["cluster_ref"], oref)))


def test_one_bench_scene_joint_step_equals_oracle_chain(dev):
    """mode 3 (self-critical speaker-listener) at bench size: beam-3 / top-3 sampled captions and the greedy baseline token for
    token, CIDEr rewards, the moderator's listener inputs, RL grounding / language / caption losses 1e-3, then the second
    (listener) batch; the bf16 step beside it."""
    from d3net_amd import synthetic as S, minkowski as ME
    from oracle import listener_oracle as lo, rl_oracle as rlo, speaker_oracle as spo
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg, net, scene, V = _net(dev, "joint")
    assert net.mode == 3 and cfg.train.use_rl
    chunk, topn, beam = cfg.data.num_des_per_scene, cfg.train.sample_topn, cfg.train.beam_size
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]

    def mk():
        spk = S.add_language(S.make_batch([scene], dev), dev, chunk=chunk, vocab=V)
        spk["lang_len"] = spk["spk_lang_len"]
        lis = S.add_language(S.make_batch([scene], dev), dev, chunk=chunk, vocab=V, seed=9)
        for b in (spk, lis):
            b["cluster_rand"], b["slot_perms"] = rand, perms
        return spk, lis

    hs, hl = (_host(b) for b in mk())
    torch.set_num_threads(min(16, torch.get_num_threads()))
    orc = PointGroupOracle(cfg, net.detector.state_dict())
    orc.teacher = True
    sp = {k: v.detach().cpu().clone() for k, v in net.speaker.state_dict().items()}
    lp = {k: v.detach().cpu().clone() for k, v in net.listener.state_dict().items()}
    cp = {k[len("caption."):]: v for k, v in sp.items() if k.startswith("caption.")}
    opt = dict(net.loss_opt)
    r1, r2 = random.Random(21).random(), None
    spo.TIE_RULE = "index"
    try:
        with torch.no_grad():
            od = orc.loss(orc.feed(hs, 0, rand=rand, perms=perms))
            od.update({k: v for k, v in hs.items() if k not in od})
            od["lang_len"] = hs["lang_len"]
            od.update(spo.graph_module({k[len("graph."):]: v for k, v in sp.items() if k.startswith("graph.")}, od, cfg.model.num_graph_steps,
                                       cfg.model.num_locals))
            osp = rlo.rl_sample_batch(cp, od, cfg, cfg.model.max_num_proposal, cfg.model.num_locals, beam, topn)
            dd = dict(od); dd.update(osp)
            mod = rlo.moderator(dd, net.embeddings.cpu(), cfg.data.max_spk_len + 2)
            st = random.Random(21)
            r1 = st.random()
            olis = rlo.rl_listener(lp, od, mod, chunk, True, r1)
            ogr = rlo.rl_grounding_loss(od, mod, olis["cluster_ref"])
            olc = rlo.rl_lobjcls_loss(mod, olis["lang_scores"])
            ocl = rlo.rl_cap_loss(od, osp, ogr, olc, opt)
            # second batch: plain listener step
            od2 = orc.loss(orc.feed(hl, 0, rand=rand, perms=perms))
            od2.update({k: v for k, v in hl.items() if k not in od2})
            r2 = st.random()
            ol2 = lo.listener_step(lp, od2, chunk, True, r2)
    finally:
        spo.TIE_RULE = "topk"
    assert int(osp["good_bbox_masks"].sum()) > 0, "no description refers to a detected box: the caption reward would be vacuous"

    def hip_step(exact):
        net.zero_grad(set_to_none=True)
        ME.set_exact(exact)
        try:
            spk, lis = mk()
            random.seed(21)
            loss, out = net.training_step([spk, lis])
            loss.backward()
        finally:
            ME.set_exact(False)
        torch.cuda.synchronize()
        return loss, out["speaker"], out["listener"]

    loss, s, l = hip_step(True)
    assert np.array_equal(s["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert torch.equal(s["assigned_bbox_id_labels"].cpu(), osp["assigned_bbox_id_labels"])
    assert torch.equal(s["good_bbox_masks"].cpu(), osp["good_bbox_masks"])
    # sampled (beam) and baseline (greedy) captions, token for token
    N = chunk
    assert len(s["lang_cap"]) == N and len(s["lang_cap"][0]) == topn
    same = total = 0
    for n in range(N):
        for k in range(topn):
            a, b = s["lang_cap"][n][k].cpu().tolist(), osp["lang_cap"][n][k].tolist()
            total += 1; same += int(a == b)
            a, b = s["baseline_cap"][n][k].cpu().tolist(), osp["baseline_cap"][n][k].tolist()
            total += 1; same += int(a == b)
    assert same == total, "sampled / baseline captions: %d of %d identical" % (same, total)
    for k in ("sampled", "baseline"):
        assert torch.equal(s["lang_len"][k].cpu(), mod["lang_len"][k]) and torch.allclose(s["lang_feat"][k].cpu(), mod["lang_feat"][k])
        a, b = s["cluster_ref"][k].detach().cpu(), olis["cluster_ref"][k]
        scale = float(b.abs().max())
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * scale), (k, float((a - b).abs().max()), scale)
        assert torch.equal(a.argmax(-1), b.argmax(-1)), "cluster_ref arg-max (%s)" % k
    assert torch.allclose(s["ref_box_corner_label"].cpu(), mod["ref_box_corner_label"], atol=1e-5)
    assert torch.equal(s["ref_cat_label"].cpu().long(), mod["ref_cat_label"].long())
    for k in ("ref_loss", "ref_acc_mean", "ref_iou_mean", "best_ious_mean"):
        close(s[k], ogr[k], name="speaker batch " + k)
    close(s["lang_loss"], olc["lang_loss"], name="speaker batch lang_loss")
    assert torch.allclose(s["sampled_scores"].cpu().double(), ocl["sampled_scores"].double(), rtol=1e-9, atol=1e-12), "CIDEr reward (sampled)"
    assert torch.allclose(s["baseline_scores"].cpu().double(), ocl["baseline_scores"].double(), rtol=1e-9, atol=1e-12), "CIDEr reward (baseline)"
    for k in ("cap_loss", "cap_rwd", "loc_rwd", "ttl_rwd", "cap_acc"):
        a, b = float(s[k]), float(ocl[k])
        assert abs(a - b) <= 1e-3 * abs(b) + 1e-5, (k, a, b)
    close(s["total_loss"][0], od["total_loss"], name="detector loss (speaker batch)")
    # second batch
    assert np.array_equal(l["proposal_scores"][1].cpu().numpy(), od2["proposal_scores"][1])
    a, b = l["cluster_ref"].detach().cpu(), ol2["cluster_ref"]
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max())) and torch.equal(a.argmax(-1), b.argmax(-1))
    close(l["ref_loss"], ol2["ref_loss"], name="listener batch ref_loss"); close(l["lang_loss"], ol2["lang_loss"], name="listener batch lang_loss")
    total_o = (float(od["total_loss"]) + float(ocl["cap_loss"]) + 0.1 * float(s["ori_loss"]) + float(ogr["ref_loss"]) + float(olc["lang_loss"])
               + float(od2["total_loss"]) + float(ol2["ref_loss"]) + float(ol2["lang_loss"]))
    close(loss, total_o, name="step loss (orientation term from the device)")
    print("bench scene, joint, exact fp32 vs oracle chain: %d / %d captions identical, cap_loss %.6f / %.6f, ref_loss %.6f / %.6f, reward %.4f / %.4f"
          % (same, total, float(s["cap_loss"]), float(ocl["cap_loss"]), float(s["ref_loss"]), float(ogr["ref_loss"]), float(s["ttl_rwd"]), float(ocl["ttl_rwd"])))

    # ---- the bf16 step bench.py times: same proposals; the sampled captions may differ where beams are nearly tied
    loss_b, sb, lb = hip_step(False)
    assert np.array_equal(sb["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1])
    close(sb["total_loss"][0], od["total_loss"], 2e-2, "bf16 detector loss")
    close(lb["ref_loss"], ol2["ref_loss"], 2e-2, "bf16 listener-batch ref_loss")
    eq = sum(int(sb["baseline_cap"][n][0].cpu().tolist() == osp["baseline_cap"][n][0].tolist()) for n in range(N))
    print("bench scene, joint, bf16 executor: loss %.6f (exact %.6f), %d / %d greedy captions identical to the oracle's" % (float(loss_b), float(loss), eq, N))
    assert torch.isfinite(loss_b) and abs(float(loss_b) - float(loss)) <= 0.05 * abs(float(loss))
