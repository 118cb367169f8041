"""Host-side pieces of the self-critical path (CIDEr reward, beam search driver, moderator, RL losses) of d3net_amd
against the reference-generated golden vectors and the oracle.  These are plain host / torch code (no HIP kernel is
involved), so they are checked on the CPU here; tests/test_rl_gpu.py runs the whole chain on the GPU."""
import os
import random
import sys
import types

import numpy as np
import torch

from d3net_amd import cider as pcider
from d3net_amd.captioning_loss import compute_cap_loss, compute_caption_reward
from d3net_amd.listener import get_grounding_loss, get_lobjcls_loss
from d3net_amd.pipeline import PipelineNet
from d3net_amd.speaker import TopDownSceneCaptionModule
from oracle import rl_oracle as rlo
from test_oracle_rl import setup, unpad

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cider_bit_exact_and_cached_paths():
    import gen_rl_golden as R
    g = np.load(os.path.join(HERE, "golden", "rl_golden.npz"))
    gts, res = R.cider_cases()
    keys = list(gts)
    mean, scores = pcider.cider_scores([gts[k] for k in keys], [res[k][0] for k in keys])
    assert np.array_equal(scores, g["cider/scores"]) and mean == float(g["cider/mean"])
    # random corpora against the oracle, including duplicated (candidate, reference set) pairs
    rng = np.random.default_rng(0)
    words = ["a", "b", "c", "d", "e", "f"]
    for trial in range(5):
        refs = [[" ".join(rng.choice(words, int(rng.integers(1, 8)))) for _ in range(int(rng.integers(1, 4)))] for _ in range(10)]
        cands = [" ".join(rng.choice(words, int(rng.integers(1, 8)))) for _ in range(10)]
        refs += refs[:3]; cands += cands[:3]
        _, got = pcider.cider_scores(refs, cands)
        _, want = rlo.cider({str(i): r for i, r in enumerate(refs)}, {str(i): [c] for i, c in enumerate(cands)})
        assert np.array_equal(got, want)


def _caption_module(S, vocab, p):
    cap = TopDownSceneCaptionModule(S.make_cfg(), vocab, S.make_embeddings(), num_proposals=S.K, num_locals=S.L, use_relation=True)
    cap.load_state_dict(p)
    return cap


def test_beam_decode_matches_reference_and_is_differentiable():
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    cap = _caption_module(S, vocab, p)
    si = {k: torch.from_numpy(v) for k, v in S.step_inputs().items()}
    done = cap.beam_decode(si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN)
    for n in range(8):
        assert len(done[n]) == R.BEAM
        for k in range(R.BEAM):
            l = g["beam/len"][n, k]
            assert np.array_equal(done[n][k]["seq"].numpy(), g["beam/seq"][n, k, :l]), (n, k)
            assert np.allclose(done[n][k]["logps"].detach().numpy(), g["beam/logps"][n, k, :l], atol=2e-5)
            assert abs(done[n][k]["p"] - g["beam/p"][n, k]) < 1e-4
    # gradients of the chosen-token log-probs equal the oracle's (one sample / one beam at a time) search
    loss = sum(b["logps"].sum() for s in done for b in s[:2])
    loss.backward()
    pp = {k: v.clone().requires_grad_(k != "embeddings") for k, v in p.items()}
    od = rlo.beam_decode(pp, si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN, 2, 3)
    sum(b["logps"].sum() for s in od for b in s[:2]).backward()
    for name, prm in cap.named_parameters():
        ref = pp[name].grad
        assert torch.allclose(prm.grad, ref, rtol=1e-3, atol=1e-5 + 1e-3 * float(ref.abs().max())), name


def test_caption_reward_moderator_and_rl_losses_match_reference():
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    caps = unpad(g["rl/lang_cap"].astype(np.int64), g["rl/lang_cap_len"])
    base = unpad(g["rl/baseline_cap"].astype(np.int64), g["rl/baseline_len"])
    args = (R.TOPN, vocab["idx2word"], opt["train_dataset_data"], opt["organized_data"])
    assert np.array_equal(compute_caption_reward(d, caps, *args).numpy(), g["reward/sampled"])
    assert np.array_equal(compute_caption_reward(d, base, *args).numpy(), g["reward/baseline"])
    none = dict(d, annotated=torch.zeros_like(d["annotated"]))
    assert float(compute_caption_reward(none, caps, *args).abs().sum()) == 0
    # moderator == oracle moderator
    dd = dict(d, lang_cap=caps, baseline_cap=base, assigned_bbox_id_labels=torch.from_numpy(g["rl/assigned"]))
    T = S.MAXLEN + 2
    want = rlo.moderator(dd, p["embeddings"], T)
    got = PipelineNet.moderator(types.SimpleNamespace(embeddings=p["embeddings"]), dict(dd), T)
    assert got["sampled_topn"] == R.TOPN
    for k in ("sampled", "baseline"):
        assert torch.equal(got["lang_feat"][k], want["lang_feat"][k]) and torch.equal(got["lang_len"][k], want["lang_len"][k])
    assert torch.equal(got["ref_box_corner_label"], want["ref_box_corner_label"]) and torch.equal(got["ref_cat_label"], want["ref_cat_label"])
    # RL losses on the reference's own listener outputs
    lps = [[torch.from_numpy(g["rl/lang_logprob"][n, k, :g["rl/lang_cap_len"][n, k]].copy()).requires_grad_() for k in range(R.TOPN)] for n in range(8)]
    got["cluster_ref"] = {k: torch.from_numpy(g["lis/cluster_ref/" + k]) for k in ("sampled", "baseline")}
    got["lang_scores"] = {k: torch.from_numpy(g["lis/lang_scores/" + k]) for k in ("sampled", "baseline")}
    got["lang_logprob"], got["good_bbox_masks"] = lps, torch.from_numpy(g["rl/good"])
    _, got = get_grounding_loss(got, use_rl=True)
    _, got = get_lobjcls_loss(got, use_rl=True)
    _, got = compute_cap_loss(got, opt)
    assert np.array_equal(got["cluster_labels"].numpy().argmax(-1), g["lis/cluster_labels"])
    for k in ("ref_loss", "ref_sampled_loss", "ref_baseline_loss", "ref_acc_mean", "ref_baseline_acc", "ref_iou_mean",
              "best_ious_mean", "lang_loss", "sampled_lang_loss", "baseline_lang_loss", "lang_acc", "lang_baseline_acc",
              "cap_loss", "cap_acc", "cap_rwd", "loc_rwd", "ttl_rwd", "ref_iou_rate_0.25", "ref_iou_rate_0.5"):
        assert np.allclose(got[k].detach().numpy(), g["loss/" + k], rtol=1e-5, atol=1e-6), k
    # d cap_loss / d logprob = -reward * good / #good, the REINFORCE weight
    got["cap_loss"].backward()
    assert all(lp.grad is not None for row in lps for lp in row)
