"""GPU parity of the self-critical (RL) speaker -> moderator -> listener chain against the golden vectors produced by
the reference's own modules (tests/golden/rl_golden.npz), and PipelineNet mode 3 end to end.
Tolerance: fp32; differences come from library GEMM / GRU summation order and the HIP attention core: rtol 1e-3,
atol 2e-4 on outputs; token sequences (beam search / greedy argmax over well-separated scores) must be identical."""
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)


def test_rl_chain_matches_reference_golden(dev):
    from test_oracle_rl import setup
    from gen_listener_golden import make_cfg as listener_cfg
    from d3net_amd.captioning_loss import compute_cap_loss
    from d3net_amd.listener import ListenerNet, get_grounding_loss, get_lobjcls_loss
    from d3net_amd.pipeline import PipelineNet
    from d3net_amd.speaker import TopDownSceneCaptionModule, query_locals_all
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    cap = TopDownSceneCaptionModule(cfg, vocab, S.make_embeddings(), num_proposals=S.K, num_locals=S.L, use_relation=True)
    cap.load_state_dict(p)
    cap = cap.to(dev)
    net = ListenerNet(listener_cfg())
    net.load_state_dict(lp)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = net.to(dev).train()
    d = {k: v.to(dev) for k, v in d.items()}
    d["adjacent_mat"] = query_locals_all(d["proposal_bbox_batched"], d["proposal_batch_mask"], S.L, include_self=False)
    # beam search alone
    si = {k: torch.from_numpy(v).to(dev) for k, v in S.step_inputs().items()}
    done = cap.beam_decode(si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN)
    for n in range(8):
        for k in range(R.BEAM):
            l = g["beam/len"][n, k]
            assert np.array_equal(done[n][k]["seq"].cpu().numpy(), g["beam/seq"][n, k, :l]), (n, k)
            assert np.allclose(done[n][k]["logps"].detach().cpu().numpy(), g["beam/logps"][n, k, :l], atol=2e-4)
            assert abs(done[n][k]["p"] - g["beam/p"][n, k]) < 1e-3
    # the training chain
    random.seed(5)
    dd = cap(dict(d), use_tf=True, use_rl=True, is_eval=False, beam_opt={"train_beam_size": R.BEAM, "train_sample_topn": R.TOPN})
    for n in range(8):
        for k in range(R.TOPN):
            l, bl = g["rl/lang_cap_len"][n, k], g["rl/baseline_len"][n, k]
            assert np.array_equal(dd["lang_cap"][n][k].cpu().numpy(), g["rl/lang_cap"][n, k, :l])
            assert np.allclose(dd["lang_logprob"][n][k].detach().cpu().numpy(), g["rl/lang_logprob"][n, k, :l], atol=2e-4)
            assert np.array_equal(dd["baseline_cap"][n][k].cpu().numpy(), g["rl/baseline_cap"][n, k, :bl])
    assert np.array_equal(dd["assigned_bbox_id_labels"].cpu().numpy(), g["rl/assigned"]) and np.array_equal(dd["good_bbox_masks"].cpu().numpy(), g["rl/good"])
    dd = PipelineNet.moderator(types.SimpleNamespace(embeddings=cap.embeddings), dd, S.MAXLEN + 2)
    dd["istrain"] = torch.tensor([1])
    random.seed(3)
    dd = net(dd, use_rl=True)
    for k in ("sampled", "baseline"):
        for name in ("cluster_ref", "lang_scores"):
            ref, got = g["lis/%s/%s" % (name, k)], dd[name][k].detach().cpu().numpy()
            assert np.allclose(got, ref, rtol=1e-3, atol=2e-4), (name, k, float(np.abs(got - ref).max()))
        assert not dd["cluster_ref"]["baseline"].requires_grad and dd["cluster_ref"]["sampled"].requires_grad
    _, dd = get_grounding_loss(dd, use_rl=True)
    _, dd = get_lobjcls_loss(dd, use_rl=True)
    _, dd = compute_cap_loss(dd, opt)
    assert np.array_equal(dd["cluster_labels"].cpu().numpy().argmax(-1), g["lis/cluster_labels"])
    for k in ("ref_loss", "ref_sampled_loss", "ref_baseline_loss", "ref_acc_mean", "ref_baseline_acc", "ref_iou_mean",
              "best_ious_mean", "lang_loss", "sampled_lang_loss", "baseline_lang_loss", "lang_acc", "lang_baseline_acc",
              "cap_loss", "cap_acc", "cap_rwd", "loc_rwd", "ttl_rwd", "ref_iou_rate_0.25", "ref_iou_rate_0.5"):
        ref, got = g["loss/" + k], dd[k].detach().cpu().numpy()
        assert np.allclose(got, ref, rtol=1e-3, atol=2e-4), (k, float(np.abs(got - ref).max()))
    (dd["cap_loss"] + dd["ref_loss"] + dd["lang_loss"]).backward()
    cp, lpn = dict(cap.named_parameters()), dict(net.named_parameters())
    for k in g.files:
        if k.startswith("grad/cap/"):
            ref, got = g[k], cp[k[len("grad/cap/"):]].grad.cpu().numpy()[:32]
        elif k.startswith("grad/lis/"):
            ref, got = g[k], lpn[k[len("grad/lis/"):]].grad.cpu().numpy()[:32]
        else:
            continue
        assert np.allclose(got, ref, rtol=5e-3, atol=1e-6 + 2e-3 * np.abs(ref).max()), (k, float(np.abs(got - ref).max()))


def test_joined_decode_chain_equals_the_two_separate_decodes(dev):
    """model/caption_module.py:588-633: the beam search and the greedy baseline of one self-critical step decode the same samples
    with the same parameters; the library runs them as one chain of launches, the greedy sample as one more row per sample
    (csrc/topdown.hip d3_topdown_beam_greedy; speaker.JOINED_DECODES).  A row of the decode step never reads another row, so
    captions and baselines must be identical and the log-probabilities equal to the last bits' summation order (the GEMM's tile
    class may differ with the row count): 1e-5."""
    from test_oracle_rl import setup
    from d3net_amd import speaker as SP
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    cap = SP.TopDownSceneCaptionModule(cfg, vocab, S.make_embeddings(), num_proposals=S.K, num_locals=S.L, use_relation=True)
    cap.load_state_dict(p)
    cap = cap.to(dev)
    d = {k: v.to(dev) for k, v in d.items()}
    d["adjacent_mat"] = SP.query_locals_all(d["proposal_bbox_batched"], d["proposal_batch_mask"], S.L, include_self=False)
    outs = []
    old = SP.JOINED_DECODES
    try:
        for flag in (False, True, True):
            SP.JOINED_DECODES = flag
            random.seed(5)
            dd = cap(dict(d), use_tf=True, use_rl=True, is_eval=False, beam_opt={"train_beam_size": R.BEAM, "train_sample_topn": R.TOPN})
            torch.cuda.synchronize()
            outs.append(([[t.cpu() for t in row] for row in dd["lang_cap"]], [[t.detach().cpu() for t in row] for row in dd["lang_logprob"]],
                         [[t.cpu() for t in row] for row in dd["baseline_cap"]]))
    finally:
        SP.JOINED_DECODES = old
    for o in outs[1:]:
        for k, (a, b) in enumerate(zip(outs[0], o)):
            assert len(a) == len(b)
            for ra, rb in zip(a, b):
                assert len(ra) == len(rb)
                for x, y in zip(ra, rb):
                    assert x.shape == y.shape and (torch.equal(x, y) if k != 1 else float((x - y).abs().max()) < 1e-5)
    # greedy-only and beam-only entry points (evaluation paths) still agree with the joined chain's rows
    si = {k: torch.from_numpy(v).to(dev) for k, v in S.step_inputs().items()}
    done = cap.beam_decode(si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN)
    gr, glp = cap.greedy_decode(si["target"], si["obj"], si["mask"].unsqueeze(-1) if si["mask"].dim() == 2 else si["mask"], S.MAXLEN + 1)
    done2, (gr2, glp2) = cap._beam_decode_native(si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN, None, greedy_len=S.MAXLEN + 1)
    for n in range(len(done)):
        assert len(done[n]) == len(done2[n])
        for x, y in zip(done[n], done2[n]):
            assert torch.equal(x["seq"], y["seq"]) and abs(x["p"] - y["p"]) < 1e-4
        assert torch.equal(gr[n][0], gr2[n][0]) and float((glp[n][0] - glp2[n][0]).abs().max()) < 1e-5


@pytest.mark.parametrize("b,extra", [(1, 0), (2, 3), (4, 1), (8, 0)])
def test_joined_decode_chain_at_other_beam_widths_and_lengths(dev, b, extra):
    """d3_topdown_beam_greedy's contract (include/d3hip.h): any 1 <= b <= 8, greedy length >= beam length (the greedy rows go on alone
    once the beams have ended, or stop with them).  Against the two separate library loops on three samples."""
    from test_oracle_rl import setup
    from d3net_amd import speaker as SP
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    cap = SP.TopDownSceneCaptionModule(cfg, vocab, S.make_embeddings(), num_proposals=S.K, num_locals=S.L, use_relation=True)
    cap.load_state_dict(p)
    cap = cap.to(dev)
    si = {k: torch.from_numpy(v).to(dev)[:3].contiguous() for k, v in S.step_inputs().items()}
    T = 7
    done = cap.beam_decode(si["target"], si["obj"], si["mask"], b, T)
    gr, glp = cap.greedy_decode(si["target"], si["obj"], si["mask"], T + extra)
    done2, (gr2, glp2) = cap._beam_decode_native(si["target"], si["obj"], si["mask"], b, T, None, greedy_len=T + extra)
    for n in range(3):
        assert len(done[n]) == len(done2[n]) >= 1
        for x, y in zip(done[n], done2[n]):
            assert torch.equal(x["seq"], y["seq"]) and abs(x["p"] - y["p"]) < 1e-4
            assert float((x["logps"] - y["logps"]).abs().max()) < 1e-5
        assert torch.equal(gr[n][0], gr2[n][0]) and float((glp[n][0] - glp2[n][0]).abs().max()) < 1e-5


def test_pipeline_mode3_runs_and_trains(dev):
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pipeline import PipelineNet
    Cn, V = 2, 200
    cfg = default_conf(overrides={
        "model": {"blocks": [1, 2, 3], "num_graph_steps": 2, "num_locals": 10, "use_relation": True, "use_orientation": True,
                  "match_type": "Transformer", "use_lang_classifier": True, "use_bidir": False, "num_bbox_class": 18,
                  "loss_type": "cross_entropy", "no_captioning": False, "no_grounding": False},
        "data": {"num_des_per_scene": Cn, "max_spk_len": 30, "max_lis_len": 126, "min_iou_threshold": 0.25, "num_ori_bins": 6},
        "train": {"use_rl": True, "sample_topn": 2, "beam_size": 2}})
    chunked, organized = S.make_language_corpus(2, chunk=Cn, vocab=V)
    ds = {"train": types.SimpleNamespace(vocabulary=S.make_vocabulary(V), glove=np.random.default_rng(0).standard_normal((V, 300)).astype(np.float32),
                                         chunked_data=chunked, organized=organized)}
    net = PipelineNet(cfg, ds).to(dev).train()
    assert net.mode == 3
    net.detector.teacher = True
    scenes = [S.small_scene(dims=(40, 32, 20), n_boxes=3, seed=s) for s in (3, 4)]
    spk = S.add_language(S.make_batch(scenes, dev), dev, chunk=Cn, vocab=V)
    spk["lang_len"] = spk["spk_lang_len"]
    lis = S.add_language(S.make_batch(scenes, dev), dev, chunk=Cn, vocab=V, seed=9)
    loss, out = net.training_step([spk, lis])
    assert torch.isfinite(loss)
    loss.backward()
    grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(g).all() for g in grads.values())
    assert any(n.startswith("speaker.caption") and float(g.abs().sum()) > 0 for n, g in grads.items())
    assert any(n.startswith("listener.match") and float(g.abs().sum()) > 0 for n, g in grads.items())
    s = out["speaker"]
    assert len(s["lang_cap"]) == 2 * Cn and len(s["lang_cap"][0]) == 2
    assert s["lang_feat"]["sampled"].shape == (2 * 2, Cn, 32, 300) and s["cluster_ref"]["sampled"].shape == (2 * 2 * Cn, 128)
    for k in ("train_score/cap_rwd", "train_score/loc_rwd", "train_score/ttl_rwd", "train_loss/captioning_loss", "train_score/ref_iou_rate_0.5"):
        assert k in net.logged
    opt, _ = net.configure_optimizers()
    opt[0].step()


def test_device_cider_reward_matches_reference_golden_and_host_scorer(dev):
    """csrc/cider.hip against (a) the reward golden vectors produced by the reference's own `compute_caption_reward`
    (lib/captioning/loss_helper.py:15-96) and (b) the host scorer (pinned bit-exact to the reference's CiderScorer) on a random
    corpus with repeated words (tf > 1, clipping), candidates with and without "eos", empty candidates, reference words outside
    the vocabulary, duplicated reference sets and topn > 1.  float64 scores: 1e-12 relative (device log / pow)."""
    from test_oracle_rl import setup, unpad
    from d3net_amd import captioning_loss as CL, cider as pcider
    R, S, g, cfg, vocab, p, lp, d, opt = setup()
    caps = unpad(g["rl/lang_cap"].astype(np.int64), g["rl/lang_cap_len"])
    base = unpad(g["rl/baseline_cap"].astype(np.int64), g["rl/baseline_len"])
    args = (R.TOPN, vocab["idx2word"], opt["train_dataset_data"], opt["organized_data"])
    dd = {k: v.to(dev) for k, v in d.items()}
    on = lambda t: [[c.to(dev) for c in row] for row in t]
    CL._CORPORA.clear()
    got_s = CL.compute_caption_reward(dict(dd), on(caps), *args)
    got_b = CL.compute_caption_reward(dict(dd), on(base), *args)
    hit = [v for (oid, _dev), v in CL._CORPORA.items() if oid == id(opt["organized_data"])]
    assert hit and hit[0][0] is opt["organized_data"] and hit[0][1].ok, "device path not taken"
    assert np.allclose(got_s.cpu().numpy(), g["reward/sampled"], rtol=1e-6, atol=1e-7)
    assert np.allclose(got_b.cpu().numpy(), g["reward/baseline"], rtol=1e-6, atol=1e-7)

    # random corpus, float64 scores against the host scorer
    rng = np.random.default_rng(1)
    V = 40
    idx2word = {str(i): "w%d" % i for i in range(V)}
    idx2word[str(V - 1)] = "eos"
    organized = {}
    for s in range(3):
        organized["s%d" % s] = {}
        for o in range(4):
            descs = []
            for _ in range(int(rng.integers(1, 6))):
                toks = ["w%d" % int(t) for t in rng.integers(0, 12, int(rng.integers(1, 30)))]      # small alphabet: repeated n-grams
                if rng.random() < 0.3:
                    toks[int(rng.integers(0, len(toks)))] = "oov%d" % int(rng.integers(0, 3))         # not in the vocabulary
                descs.append({"token": toks})
            organized["s%d" % s][str(o)] = descs
    corpus = CL.CiderCorpus(organized, idx2word, dev)
    assert corpus.ok
    topn = 2
    keys = [("s%d" % int(rng.integers(0, 3)), str(int(rng.integers(0, 4)))) for _ in range(9)]
    keys += keys[:2]                                                                                   # duplicated sets
    cands = []
    for _ in range(len(keys) * topn):
        l = int(rng.integers(0, 14))
        t = rng.integers(0, 12, l)
        if l and rng.random() < 0.5:
            t[-1] = V - 1                                                                                # ends in eos
        cands.append(torch.from_numpy(t.astype(np.int64)).to(dev))
    out = CL._cider_device(corpus, [corpus.sets[k] for k in keys], cands, topn)
    assert out is not None
    refs, cs = [], []
    for i, k in enumerate(keys):
        gt = [" ".join(dd_["token"] + ["eos"]) for dd_ in organized[k[0]][k[1]]]
        for j in range(topn):
            toks = [idx2word[str(int(t))] for t in cands[i * topn + j].tolist()]
            if "eos" not in toks:
                toks.append("eos")
            refs.append(gt); cs.append(" ".join(toks))
    _, want = pcider.cider_scores(refs, cs)
    got = out.cpu().numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=1e-13), np.abs(got - want).max()
    assert want.max() > 0.1                                                                            # a non-trivial case

    # long sentences: several 64-position rounds per n-gram order in cd_vec_kernel (descriptions go up to 126 tokens)
    organized2 = {"s0": {str(o): [{"token": ["w%d" % int(t) for t in rng.integers(0, 9, int(rng.integers(60, 150)))]}
                                  for _ in range(3)] for o in range(2)}}
    corpus2 = CL.CiderCorpus(organized2, idx2word, dev)
    assert corpus2.ok
    keys2 = [("s0", "0"), ("s0", "1"), ("s0", "0")]
    cands2 = [torch.from_numpy(rng.integers(0, 9, int(l)).astype(np.int64)).to(dev) for l in (70, 129, 3)]
    out2 = CL._cider_device(corpus2, [corpus2.sets[k] for k in keys2], cands2, 1)
    if out2 is not None:                                   # (a set beyond the LDS hash falls back to the host scorer: None)
        refs2 = [[" ".join(dd_["token"] + ["eos"]) for dd_ in organized2[k[0]][k[1]]] for k in keys2]
        cs2 = [" ".join([idx2word[str(int(t))] for t in c.tolist()] + ["eos"]) for c in cands2]
        _, want2 = pcider.cider_scores(refs2, cs2)
        assert np.allclose(out2.cpu().numpy(), want2, rtol=1e-12, atol=1e-13), np.abs(out2.cpu().numpy() - want2).max()
    else:
        assert 4 * sum(len(d_["token"]) + 1 for d_ in organized2["s0"]["0"]) > CL.CiderCorpus.MAX_SET_NGRAMS


def test_native_beam_search_with_teacher_forced_replay_equals_library_search(dev):
    """TopDownSceneCaptionModule.beam_decode on the native decode step + one teacher-forced replay of the returned beams
    (d3net_amd/speaker.py:_beam_decode_native) against the library-op search that differentiates through every step: identical
    token sequences, chosen-token log-probabilities to 2e-4, parameter gradients of a loss over the returned beams to 1e-3."""
    from test_oracle_rl import setup
    from d3net_amd.speaker import TopDownSceneCaptionModule
    R, S, g, cfg, vocab, p, lp, d, opt = setup()

    def build(native):
        cap = TopDownSceneCaptionModule(cfg, vocab, S.make_embeddings(), num_proposals=S.K, num_locals=S.L, use_relation=True)
        cap.load_state_dict(p)
        cap = cap.to(dev)
        cap.native = native
        return cap

    si = {k: torch.from_numpy(v).to(dev) for k, v in S.step_inputs().items()}
    res = {}
    for native in (True, False):
        cap = build(native)
        done = cap.beam_decode(si["target"], si["obj"], si["mask"], R.BEAM, S.MAXLEN, topn=2)
        sum(b["logps"].sum() * (1.0 + 0.1 * j) for s_ in done for j, b in enumerate(s_)).backward()
        res[native] = (done, {n: q.grad.clone() for n, q in cap.named_parameters() if q.grad is not None})
    (dn, gn), (dl, gl) = res[True], res[False]
    assert len(dn) == len(dl)
    for a, b in zip(dn, dl):
        assert len(a) == len(b) == 2
        for x, y in zip(a, b):
            assert torch.equal(x["seq"], y["seq"]) and abs(x["p"] - y["p"]) < 1e-3
            assert torch.allclose(x["logps"], y["logps"], atol=2e-4)
    assert set(gn) == set(gl)
    for n in gl:
        assert torch.allclose(gn[n], gl[n], rtol=1e-3, atol=1e-5 + 1e-3 * float(gl[n].abs().max())), n
