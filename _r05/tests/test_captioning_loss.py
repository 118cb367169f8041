"""d3net_amd.captioning_loss (pure tensor logic, runs on CPU) against the reference's own loss functions
(golden: tests/golden/speaker_golden.npz, xe/cap_loss, ori/*)."""
import os
import random
import sys

import numpy as np
import torch

from oracle import speaker_oracle as spo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_cap_and_orientation_loss_match_reference():
    import gen_speaker_golden as G
    from gen_listener_golden import golden_weights
    from d3net_amd.captioning_loss import compute_cap_loss, compute_node_orientation_loss
    from d3net_amd.speaker import TopDownSceneCaptionModule
    g = np.load(os.path.join(HERE, "golden", "speaker_golden.npz"))
    cfg, vocab, emb = G.make_cfg(), G.make_vocab(), G.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=G.K, num_locals=G.L, use_relation=True)
    p = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"}); p["embeddings"] = torch.from_numpy(emb)
    d = {k: torch.from_numpy(v) for k, v in G.speaker_inputs().items()}
    d["adjacent_mat"] = torch.from_numpy(g["adjacent_mat"].astype(np.float32))
    random.seed(5)
    out = spo.forward_sample_batch(p, d, cfg, G.K, G.L)      # pinned oracle provides the logits on CPU
    dd = dict(d, lang_cap=out["lang_cap"], good_bbox_masks=out["good"])
    loss, dd = compute_cap_loss(dd, {"use_rl": False, "max_len": G.MAXLEN + 2})
    assert abs(float(loss) - float(g["xe/cap_loss"])) < 1e-5 and abs(float(dd["cap_acc"]) - float(g["xe/cap_acc"])) < 1e-6
    ol, oa = compute_node_orientation_loss({k: torch.from_numpy(v) for k, v in G.orientation_inputs().items()}, 6)
    assert abs(float(ol) - float(g["ori/loss"])) < 1e-5 and abs(float(oa) - float(g["ori/acc"])) < 1e-6
