#!/usr/bin/env python3
"""Generates tests/golden/speaker_golden.npz by RUNNING THE REFERENCE's own caption module on CPU
(model/caption_module.py: step, select_target, _query_locals, _add_relation_feat, _forward_sample_batch,
_forward_scene_batch, greedy_decode).  model/graph_module.py cannot be imported (torch_geometric is absent), so the
graph outputs the captioner consumes (bbox_feature, edge_feature, adjacent_mat) are synthetic inputs here; the
adjacency is built with the reference's own `_query_locals(include_self=False)`, which is the same function
GraphModule uses.  Run in the build container only.  Weights: `golden_weights()` of gen_listener_golden.py."""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_listener_golden import golden_weights, listener_inputs  # noqa: E402

REF = "/root/reference"
V, MAXLEN, K, L = 60, 10, 128, 10


def make_vocab():
    words = ["pad_", "unk", "sos", "eos"] + ["w%d" % i for i in range(V - 4)]
    return {"word2idx": {w: i for i, w in enumerate(words)}, "idx2word": {i: w for i, w in enumerate(words)}}


def make_embeddings():
    return np.random.default_rng(7).standard_normal((V, 300)).astype(np.float32)


def make_cfg(Cn=4):
    ns = types.SimpleNamespace
    return ns(model=ns(max_num_proposal=K, m=16, num_locals=L, use_relation=True, no_detection=False, num_graph_steps=2,
                       use_orientation=True, no_captioning=False),
              data=ns(num_des_per_scene=Cn, max_spk_len=MAXLEN, min_iou_threshold=0.25))


def speaker_inputs(B=2, Cn=4, seed=1):
    rng = np.random.default_rng(seed)
    base = listener_inputs(B=B, Cn=Cn, seed=seed)
    d = {k: base[k] for k in ("proposal_center_batched", "proposal_bbox_batched", "proposal_batch_mask", "ref_box_corner_label")}
    d["proposal_feats_batched"] = base["proposal_feats_batched"]
    d["bbox_feature"] = rng.standard_normal((B, K, 128)).astype(np.float32) * d["proposal_batch_mask"][..., None]
    d["edge_feature"] = rng.standard_normal((B, K, L, 128)).astype(np.float32) * 0.5
    lens = rng.integers(4, MAXLEN + 3, (B, Cn)).astype(np.int64); lens[0, 0] = MAXLEN + 2
    ids = np.zeros((B, Cn, MAXLEN + 2), np.int64)
    for b in range(B):
        for c in range(Cn):
            n = lens[b, c]
            ids[b, c, 0] = 2; ids[b, c, 1:n - 1] = rng.integers(4, V, n - 2); ids[b, c, n - 1] = 3
    d["lang_ids"], d["lang_len"] = ids, lens
    ann = np.ones((B, Cn), np.int64); ann[0, 2] = 0; ann[1, 1] = 0
    d["annotated"] = ann
    # GT boxes: 20 objects per scene (jittered copies of valid proposals); referred object = one of them
    gt_c = np.zeros((B, 128, 3), np.float32); gt_b = np.zeros((B, 128, 8, 3), np.float32); ref_lab = np.zeros((B, Cn, 128), np.float32)
    for b in range(B):
        valid = np.nonzero(d["proposal_batch_mask"][b])[0]
        for o in range(20):
            src = valid[o % len(valid)]
            gt_b[b, o] = d["proposal_bbox_batched"][b, src] + rng.normal(0, 0.03, (1, 3)).astype(np.float32)
            gt_c[b, o] = (gt_b[b, o].min(0) + gt_b[b, o].max(0)) / 2
        for c in range(Cn):
            o = rng.integers(0, 20)
            ref_lab[b, c, o] = 1
            d["ref_box_corner_label"][b, c] = gt_b[b, o]
    d["center_label"], d["gt_bbox"], d["ref_box_label"] = gt_c, gt_b, ref_lab
    return d


def step_inputs(seed=11):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return dict(h1=f(8, 512) * 0.1, h2=f(8, 512) * 0.1, word=rng.integers(0, V, 8).astype(np.int64), target=f(8, 128),
                obj=f(8, K, 128), mask=(rng.random((8, K, 1)) > 0.5).astype(np.float32))


def orientation_inputs(seed=13, B=2, E=60):
    rng = np.random.default_rng(seed)
    def rotz(a):
        c, s_ = np.cos(a), np.sin(a)
        return np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    rots = np.stack([np.stack([rotz(rng.uniform(0, np.pi)) for _ in range(128)]) for _ in range(B)])
    ei = np.zeros((B, 2, K * L), np.float32)
    ei[:, :, :E] = rng.integers(0, K, (B, 2, E))
    return dict(object_assignment=rng.integers(0, 128, (B, K)).astype(np.int64), edge_index=ei,
                edge_orientations=rng.standard_normal((B, K * L, 6)).astype(np.float32),
                num_edge_source=np.array([6, 5], np.int64), num_edge_target=np.array([10, 10], np.int64),
                scene_object_rotations=rots, scene_object_rotation_masks=(rng.random((B, 128)) > 0.3).astype(np.float32))


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType("model"); pkg.__path__ = [os.path.join(REF, "model")]; sys.modules["model"] = pkg
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    from model.caption_module import TopDownSceneCaptionModule

    cfg, vocab = make_cfg(), make_vocab()
    emb = make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=K, num_locals=L, use_relation=True, use_oracle=False)
    sd = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"})
    sd["embeddings"] = torch.from_numpy(emb)
    cap.load_state_dict(sd)
    inp = speaker_inputs()
    d = {k: torch.from_numpy(v) for k, v in inp.items()}
    d["lang_feat"] = torch.zeros(inp["annotated"].shape[0], 1)   # only its batch dimension is read (caption_module.py:702)
    out = {}
    # adjacency with the reference's own _query_locals (include_self=False), target by target
    masks = d["proposal_batch_mask"]
    adj = torch.zeros(masks.shape[0], K, K)
    loc = torch.zeros(masks.shape[0], K, K)
    for t in range(K):
        tid = torch.full((masks.shape[0],), t, dtype=torch.long)
        adj[:, t] = cap._query_locals(d["proposal_bbox_batched"], tid, masks, include_self=False)
        loc[:, t] = cap._query_locals(d["proposal_bbox_batched"], tid, masks, include_self=True)
    d["adjacent_mat"] = adj
    out["adjacent_mat"], out["locals_incl_self"] = adj.numpy(), loc.numpy()
    # one decode step
    si = {k: torch.from_numpy(v) for k, v in step_inputs().items()}
    h, w, tf, of, om = (si["h1"], si["h2"]), si["word"], si["target"], si["obj"], si["mask"]
    so, lp, hh, mk = cap.step(w, h, tf, of, om)
    out.update({"step/out": so.detach().numpy(), "step/h1": hh[0].detach().numpy(), "step/h2": hh[1].detach().numpy(),
                "step/attn": mk.detach().numpy()})
    # XE training forward + backward
    random.seed(5)
    dd = cap(dict(d), use_tf=True, use_rl=False, is_eval=False)
    logits = dd["lang_cap"]
    tgt = d["lang_ids"].reshape(-1, MAXLEN + 2)[:, 1:logits.shape[1] + 1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), ignore_index=0)
    loss.backward()
    out.update({"xe/lang_cap": logits.detach().numpy(), "xe/topdown_attn": dd["topdown_attn"].detach().numpy(),
                "xe/valid_masks": dd["valid_masks"].numpy(), "xe/assigned": dd["assigned_bbox_id_labels"].numpy(),
                "xe/pred_ious": np.float32(dd["pred_ious"].detach()), "xe/good": dd["good_bbox_masks"].numpy(), "xe/loss": np.float32(loss.detach())})
    # the reference's own caption loss (lib/captioning/loss_helper.py:177-224) and orientation loss (:244-307)
    from lib.captioning.loss_helper import compute_cap_loss, compute_node_orientation_loss
    dd["lang_len"], dd["lang_ids"] = d["lang_len"], d["lang_ids"]
    _, dd = compute_cap_loss(dd, {"use_rl": False, "max_len": MAXLEN + 2})
    out["xe/cap_loss"], out["xe/cap_acc"] = np.float32(dd["cap_loss"].detach()), np.float32(dd["cap_acc"])
    oi = orientation_inputs()
    ol, oa = compute_node_orientation_loss({k: torch.from_numpy(v) for k, v in oi.items()}, 6)
    out["ori/loss"], out["ori/acc"] = np.float32(ol), np.float32(oa)
    for n in ("map_topdown.weight", "recurrent_cell_1.weight_hh", "map_feat.weight", "attend.weight", "map_lang.bias",
              "recurrent_cell_2.weight_ih", "classifier.2.weight"):
        out["xe/grad/" + n] = dict(cap.named_parameters())[n].grad.numpy()[:32].copy()   # first 32 rows only
    # evaluation decode (all proposals) and greedy decode
    with torch.no_grad():
        de = cap(dict(d), is_eval=True)
    out["eval/lang_cap"], out["eval/valid_masks"] = de["lang_cap"].numpy(), de["valid_masks"].numpy()
    out["eval/topdown_attn_sum"] = de["topdown_attn"].sum(-1).numpy()
    g_ids, g_lp = cap.greedy_decode(tf, of, om, MAXLEN + 1)
    out["greedy/len"] = np.array([len(x[0]) for x in g_ids]); out["greedy/ids0"] = g_ids[0][0].numpy(); out["greedy/lp0"] = g_lp[0][0].numpy()
    out["eval/lang_cap"] = out["eval/lang_cap"].astype(np.int16); out["xe/topdown_attn"] = out["xe/topdown_attn"].astype(np.float32)
    for k in ("adjacent_mat", "locals_incl_self", "eval/valid_masks", "xe/valid_masks"):
        out[k] = out[k].astype(np.uint8)
    # inputs are NOT stored: tests rebuild them with speaker_inputs() / step_inputs() (numpy generators, seeded)
    np.savez_compressed(os.path.join(HERE, "speaker_golden.npz"), **out)
    print("wrote speaker_golden.npz", {k: v.shape for k, v in out.items() if "grad" not in k and "step/in" not in k})


if __name__ == "__main__":
    main()
