"""Pins oracle/speaker_oracle.py (captioner part) against golden vectors produced by the REFERENCE's own
TopDownSceneCaptionModule (tests/golden/speaker_golden.npz, generator tests/golden/gen_speaker_golden.py)."""
import os
import random
import sys

import numpy as np
import torch

from oracle import speaker_oracle as spo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def load():
    import gen_speaker_golden as G
    from gen_listener_golden import golden_weights
    from d3net_amd.speaker import TopDownSceneCaptionModule   # state-dict layout only (CPU, no kernels run)
    g = np.load(os.path.join(HERE, "golden", "speaker_golden.npz"))
    cfg, vocab, emb = G.make_cfg(), G.make_vocab(), G.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=G.K, num_locals=G.L, use_relation=True)
    p = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"})
    p["embeddings"] = torch.from_numpy(emb)
    d = {k: torch.from_numpy(v) for k, v in G.speaker_inputs().items()}
    d["adjacent_mat"] = torch.from_numpy(g["adjacent_mat"].astype(np.float32))
    return G, g, cfg, vocab, p, d, cap


def test_caption_param_count():
    G, g, cfg, vocab, p, d, cap = load()
    # 5,107,108 parameters at vocabulary 3004 -> the vocabulary-dependent part is classifier.2 (512*V + V)
    n = sum(x.numel() for x in cap.parameters())
    assert n - (512 * G.V + G.V) + (512 * 3004 + 3004) == 5107108


def test_query_locals_and_step_match_reference():
    G, g, cfg, vocab, p, d, _ = load()
    for t in (0, 5, 77, 127):
        tid = torch.full((2,), t, dtype=torch.long)
        assert np.array_equal(spo.query_locals(d["proposal_bbox_batched"], tid, d["proposal_batch_mask"], G.L, False).numpy(), g["adjacent_mat"][:, t])
        assert np.array_equal(spo.query_locals(d["proposal_bbox_batched"], tid, d["proposal_batch_mask"], G.L, True).numpy(), g["locals_incl_self"][:, t])
    si = {k: torch.from_numpy(v) for k, v in G.step_inputs().items()}
    o, h, m = spo.step(p, si["word"], (si["h1"], si["h2"]), si["target"], si["obj"], si["mask"])
    for got, key in ((o, "step/out"), (h[0], "step/h1"), (h[1], "step/h2"), (m, "step/attn")):
        assert np.allclose(got.numpy(), g[key], rtol=1e-4, atol=1e-5), key


def test_training_and_eval_drivers_match_reference():
    G, g, cfg, vocab, p, d, _ = load()
    pp = {k: v.clone().requires_grad_(k != "embeddings") for k, v in p.items()}
    random.seed(5)
    out = spo.forward_sample_batch(pp, d, cfg, G.K, G.L)
    assert np.array_equal(out["assigned"].numpy(), g["xe/assigned"]) and np.array_equal(out["good"].numpy(), g["xe/good"])
    assert np.array_equal(out["valid_masks"].numpy(), g["xe/valid_masks"])
    assert np.allclose(out["lang_cap"].detach().numpy(), g["xe/lang_cap"], rtol=1e-4, atol=2e-5)
    assert np.allclose(out["topdown_attn"].detach().numpy(), g["xe/topdown_attn"], rtol=1e-4, atol=1e-6)
    assert abs(float(out["pred_ious"]) - float(g["xe/pred_ious"])) < 1e-6
    logits = out["lang_cap"]
    tgt = d["lang_ids"].reshape(-1, G.MAXLEN + 2)[:, 1:logits.shape[1] + 1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, G.V), tgt.reshape(-1), ignore_index=0)
    assert abs(float(loss) - float(g["xe/loss"])) < 1e-5
    loss.backward()
    for k in g.files:
        if k.startswith("xe/grad/"):
            ref = g[k]; got = pp[k[len("xe/grad/"):]].grad.numpy()[:32]
            assert np.allclose(got, ref, rtol=1e-3, atol=1e-6 + 1e-3 * np.abs(ref).max()), k
    with torch.no_grad():
        ev = spo.forward_scene_batch(p, d, cfg, G.K, G.L, vocab["word2idx"]["sos"])
    assert np.array_equal(ev["valid_masks"].numpy(), g["eval/valid_masks"])
    assert np.array_equal(ev["lang_cap"].numpy(), g["eval/lang_cap"])
    assert np.allclose(ev["topdown_attn"].sum(-1).numpy(), g["eval/topdown_attn_sum"], rtol=1e-4, atol=1e-5)
    si = {k: torch.from_numpy(v) for k, v in G.step_inputs().items()}
    gd = spo.greedy_decode(p, si["target"], si["obj"], si["mask"], G.MAXLEN + 1, 2, 3, 0)
    assert [len(x[0]) for x in gd] == g["greedy/len"].tolist()
    assert np.array_equal(gd[0][0].numpy(), g["greedy/ids0"]) and np.allclose(gd[0][1].numpy(), g["greedy/lp0"], atol=1e-5)
