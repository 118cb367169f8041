"""Pins oracle/rl_oracle.py (self-critical speaker-listener path) against golden vectors produced by the REFERENCE's
own modules (tests/golden/rl_golden.npz, generator tests/golden/gen_rl_golden.py)."""
import os
import random
import sys

import numpy as np
import torch

from oracle import rl_oracle as rlo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def unpad(mat, lens):
    return [[torch.from_numpy(mat[n, k, :lens[n, k]].copy()) for k in range(mat.shape[1])] for n in range(mat.shape[0])]


def setup():
    import gen_rl_golden as R
    import gen_speaker_golden as S
    from gen_listener_golden import golden_weights, make_cfg as listener_cfg
    from d3net_amd.listener import ListenerNet                 # state-dict layouts only (CPU, no kernels run)
    from d3net_amd.speaker import TopDownSceneCaptionModule
    g = np.load(os.path.join(HERE, "golden", "rl_golden.npz"))
    gs = np.load(os.path.join(HERE, "golden", "speaker_golden.npz"))
    cfg, vocab, emb = S.make_cfg(), R.vocab_str(), S.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=S.K, num_locals=S.L, use_relation=True)
    p = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"})
    p["embeddings"] = torch.from_numpy(emb)
    lp = golden_weights(ListenerNet(listener_cfg()).state_dict())
    d = {k: torch.from_numpy(v) for k, v in S.speaker_inputs().items()}
    d["adjacent_mat"] = torch.from_numpy(gs["adjacent_mat"].astype(np.float32))
    dataset_data, organized, ids, chunk_ids = R.rl_corpus()
    d["id"], d["chunk_ids"] = torch.from_numpy(ids), torch.from_numpy(chunk_ids)
    d["proposal_sem_cls_batched"] = torch.from_numpy(R.sem_cls())
    opt = dict(use_rl=True, sample_topn=R.TOPN, idx2word=vocab["idx2word"], train_dataset_data=dataset_data,
               organized_data=organized, **R.OPT_W)
    return R, S, g, cfg, vocab, p, lp, d, opt


def test_cider_matches_reference_bit_exactly():
    import gen_rl_golden as R
    g = np.load(os.path.join(HERE, "golden", "rl_golden.npz"))
    gts, res = R.cider_cases()
    mean, scores = rlo.cider(gts, res)
    assert np.array_equal(scores, g["cider/scores"]) and mean == g["cider/mean"]
    assert scores[4] == 0.0 and abs(scores[12] - 10.0) < 1e-12      # one-word candidate; candidate == its single reference


def test_beam_search_matches_reference():
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    si = {k: torch.from_numpy(v) for k, v in S.step_inputs().items()}
    with torch.no_grad():
        done = rlo.beam_decode(p, si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN, 2, 3)
    for n in range(8):
        assert len(done[n]) == R.BEAM
        for k in range(R.BEAM):
            l = g["beam/len"][n, k]
            assert np.array_equal(done[n][k]["seq"].numpy(), g["beam/seq"][n, k, :l]), (n, k)
            assert np.allclose(done[n][k]["logps"].numpy(), g["beam/logps"][n, k, :l], atol=2e-5)
            assert abs(done[n][k]["p"] - g["beam/p"][n, k]) < 1e-4


def test_rl_training_chain_matches_reference():
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    pp = {k: v.clone().requires_grad_(k != "embeddings") for k, v in p.items()}
    lpp = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in lp.items()}
    random.seed(5)
    spk = rlo.rl_sample_batch(pp, d, cfg, S.K, S.L, R.BEAM, R.TOPN)
    N = 8
    for n in range(N):
        for k in range(R.TOPN):
            l = g["rl/lang_cap_len"][n, k]
            assert np.array_equal(spk["lang_cap"][n][k].numpy(), g["rl/lang_cap"][n, k, :l])
            assert np.allclose(spk["lang_logprob"][n][k].detach().numpy(), g["rl/lang_logprob"][n, k, :l], atol=2e-5)
            bl = g["rl/baseline_len"][n, k]
            assert np.array_equal(spk["baseline_cap"][n][k].numpy(), g["rl/baseline_cap"][n, k, :bl])
    assert np.array_equal(spk["assigned_bbox_id_labels"].numpy(), g["rl/assigned"]) and np.array_equal(spk["good_bbox_masks"].numpy(), g["rl/good"])
    # caption rewards
    s_sc = rlo.caption_reward(d, spk["lang_cap"], R.TOPN, vocab["idx2word"], opt["train_dataset_data"], opt["organized_data"])
    b_sc = rlo.caption_reward(d, spk["baseline_cap"], R.TOPN, vocab["idx2word"], opt["train_dataset_data"], opt["organized_data"])
    assert np.array_equal(s_sc.numpy(), g["reward/sampled"]) and np.array_equal(b_sc.numpy(), g["reward/baseline"])
    assert (g["reward/sampled"][[2, 5]] == 0).all() and g["reward/sampled"].max() > 0.1           # unannotated rows score 0
    # moderator -> listener (RL branch) -> losses
    dd = dict(d); dd.update(spk)
    mod = rlo.moderator(dd, p["embeddings"], S.MAXLEN + 2)
    lis = rlo.rl_listener(lpp, d, mod, 4, True, random.Random(3).random())
    for k in ("sampled", "baseline"):
        assert np.allclose(lis["cluster_ref"][k].detach().numpy(), g["lis/cluster_ref/" + k], rtol=1e-3, atol=2e-4), k
        assert np.allclose(lis["lang_scores"][k].detach().numpy(), g["lis/lang_scores/" + k], rtol=1e-4, atol=1e-5), k
    gr = rlo.rl_grounding_loss(d, mod, lis["cluster_ref"])
    lc = rlo.rl_lobjcls_loss(mod, lis["lang_scores"])
    cl = rlo.rl_cap_loss(d, spk, gr, lc, opt)
    assert np.array_equal(gr["cluster_labels"].numpy().argmax(-1), g["lis/cluster_labels"])
    for src, keys in ((gr, ("ref_loss", "ref_sampled_loss", "ref_baseline_loss", "ref_acc_mean", "ref_baseline_acc", "ref_iou_mean", "best_ious_mean")),
                      (lc, ("lang_loss", "sampled_lang_loss", "baseline_lang_loss", "lang_acc", "lang_baseline_acc")),
                      (cl, ("cap_loss", "cap_acc", "cap_rwd", "loc_rwd", "ttl_rwd"))):
        for k in keys:
            assert np.allclose(src[k].detach().numpy(), g["loss/" + k], rtol=1e-3, atol=1e-4), k
    assert abs(gr["rate25"] - g["loss/ref_iou_rate_0.25"]) < 1e-6 and abs(gr["rate5"] - g["loss/ref_iou_rate_0.5"]) < 1e-6
    (cl["cap_loss"] + gr["ref_loss"] + lc["lang_loss"]).backward()
    for k in g.files:
        if k.startswith("grad/cap/"):
            ref, got = g[k], pp[k[len("grad/cap/"):]].grad.numpy()[:32]
        elif k.startswith("grad/lis/"):
            ref, got = g[k], lpp[k[len("grad/lis/"):]].grad.numpy()[:32]
        else:
            continue
        assert np.abs(ref).max() > 0, k
        assert np.allclose(got, ref, rtol=2e-3, atol=1e-6 + 2e-3 * np.abs(ref).max()), k


def test_moderator_contract():
    """`moderator` is the one unpinned restatement: check its documented contract directly."""
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    caps = unpad(g["rl/lang_cap"].astype(np.int64), g["rl/lang_cap_len"])
    base = unpad(g["rl/baseline_cap"].astype(np.int64), g["rl/baseline_len"])
    dd = dict(d, lang_cap=caps, baseline_cap=base, assigned_bbox_id_labels=torch.from_numpy(g["rl/assigned"]))
    T = S.MAXLEN + 2
    mod = rlo.moderator(dd, p["embeddings"], T)
    B, Cn, topn = 2, 4, R.TOPN
    assert mod["lang_feat"]["sampled"].shape == (B * topn, Cn, T, 300) and mod["lang_len"]["sampled"].shape == (B * Cn, topn)
    for n in range(B * Cn):
        for k in range(topn):
            toks = [2] + caps[n][k].tolist()
            if 3 not in toks:
                toks.append(3)
            assert mod["lang_len"]["sampled"][n, k] == len(toks)
            b, c = divmod(n, Cn)
            row = mod["lang_feat"]["sampled"][b * topn + k, c]
            assert torch.equal(row[:len(toks)], p["embeddings"][toks]) and torch.equal(row[len(toks):], p["embeddings"][0].expand(T - len(toks), -1))
            a = int(g["rl/assigned"][n])
            assert torch.equal(mod["ref_box_corner_label"][b * topn + k, c], d["proposal_bbox_batched"][b, a])
            cat = float(d["proposal_sem_cls_batched"][b, a]) - 2
            assert float(mod["ref_cat_label"][b * topn + k, c]) == (17 if cat < 0 else cat)
