#!/usr/bin/env python3
"""Generates tests/golden/listener_*.npz by RUNNING THE REFERENCE's own python modules on CPU
(model/lang_module.py, model/match_module.py, model/transformer/attention.py, lib/grounding/loss_helper.py).

Run in the build container only (it needs /root/reference): `python tests/golden/gen_listener_golden.py`.
The fixtures hold inputs and expected outputs; weights are NOT stored -- both this script and the tests rebuild
them with `golden_weights()` below (a deterministic function of parameter name and shape).
Two I/O-only third-party modules the reference imports at module scope but never uses on this path (trimesh,
plyfile) are absent from the image and are registered as empty placeholder modules for the import."""
import os
import random
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def golden_weights(state_dict, scale=1.0):
    """deterministic weights: N(0,1)*s seeded by crc32(name); BN/LN weights around 1, running_var positive"""
    out = {}
    for name, t in state_dict.items():
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        if not t.dtype.is_floating_point:
            out[name] = t.clone()
            continue
        a = rng.standard_normal(tuple(t.shape)).astype(np.float32)
        if name.endswith("running_var"):
            a = np.abs(a) * 0.5 + 0.5
        elif name.endswith(("layer_norm.weight", "lang_fc.3.weight")) or (".1.weight" in name or ".4.weight" in name) and t.dim() == 1:
            a = 1.0 + 0.1 * a
        elif t.dim() >= 2:
            a = a * (scale / np.sqrt(t.shape[1] if t.dim() > 1 else 1))
        else:
            a = a * 0.1
        out[name] = torch.from_numpy(np.asarray(a).astype(np.float32))   # float64 products round to fp32 like load_state_dict
    return out


def listener_inputs(B=2, Cn=4, T=24, K=128, m=16, seed=0):
    rng = np.random.default_rng(seed)
    n_valid = [37, 15]   # >= num_locals + 1 valid proposals per scene: top-k among the 1e30 ties of FEWER candidates is implementation-defined
    mask = np.zeros((B, K), np.float32)
    centers = np.zeros((B, K, 3), np.float32)
    corners = np.zeros((B, K, 8, 3), np.float32)
    feats = np.zeros((B, K, m), np.float32)
    sgn = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)
    for b in range(B):
        slots = rng.permutation(K)[:n_valid[b]]
        mask[b, slots] = 1
        c = rng.random((n_valid[b], 3)).astype(np.float32) * np.array([4, 3, 2], np.float32)
        s = (rng.random((n_valid[b], 3)).astype(np.float32) * 0.8 + 0.2)
        centers[b, slots] = c
        corners[b, slots] = c[:, None, :] + sgn[None] * s[:, None, :] / 2
        feats[b, slots] = rng.standard_normal((n_valid[b], m)).astype(np.float32)
    lang_len = rng.integers(5, T + 1, (B, Cn)).astype(np.int64)
    lang_len[0, 0] = T
    lang_feat = rng.standard_normal((B, Cn, T, 300)).astype(np.float32)
    # referred boxes: a jittered copy of a valid proposal box of the scene
    ref = np.zeros((B, Cn, 8, 3), np.float32)
    for b in range(B):
        valid = np.nonzero(mask[b])[0]
        for c in range(Cn):
            ref[b, c] = corners[b, valid[rng.integers(len(valid))]] + rng.normal(0, 0.05, (1, 3)).astype(np.float32)
    object_cat = rng.integers(0, 18, (B, Cn)).astype(np.int64)
    return dict(proposal_feats_batched=feats, proposal_center_batched=centers, proposal_bbox_batched=corners,
                proposal_batch_mask=mask, lang_feat=lang_feat, lang_len=lang_len, ref_box_corner_label=ref,
                object_cat=object_cat)


def make_cfg(Cn=4):
    ns = types.SimpleNamespace
    return ns(model=ns(num_bbox_class=18, use_lang_classifier=True, use_bidir=False, max_num_proposal=128, m=16,
                       match_type="Transformer"), data=ns(num_des_per_scene=Cn))


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType("model"); pkg.__path__ = [os.path.join(REF, "model")]; sys.modules["model"] = pkg
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    from model.listener import ListenerNet
    from lib.grounding.loss_helper import get_grounding_loss, get_lobjcls_loss

    cfg = make_cfg()
    torch.manual_seed(0)
    net = ListenerNet(cfg)
    net.load_state_dict(golden_weights(net.state_dict()))
    inp = listener_inputs()
    out = {}
    for mode in ("eval", "train"):
        net.train(mode == "train")
        net.zero_grad()
        for m_ in net.modules():          # dropout cannot be reproduced across implementations: disable it
            if isinstance(m_, torch.nn.Dropout):
                m_.p = 0.0
        d = {k: torch.from_numpy(v) for k, v in inp.items()}
        d["istrain"] = torch.tensor([1 if mode == "train" else 0])
        random.seed(3)  # random.random() -> 0.2379... < 0.5: the copy-paste branch runs in train mode
        d = net(d)
        _, d = get_grounding_loss(d, grounding=True, use_rl=False)
        _, d = get_lobjcls_loss(d, lang_cls=True, use_rl=False)
        (d["ref_loss"] + d["lang_loss"]).backward()
        for k in ("cluster_ref", "lang_scores", "lang_emb", "lang_hiddens", "lang_masks", "cluster_labels", "ref_loss",
                  "lang_loss", "ref_acc_mean", "lang_acc", "ref_iou_mean", "best_ious_mean"):
            out["%s/%s" % (mode, k)] = d[k].detach().numpy()
        out["%s/ref_iou_rate_0.25" % mode] = np.float32(d["ref_iou_rate_0.25"])
        out["%s/ref_iou_rate_0.5" % mode] = np.float32(d["ref_iou_rate_0.5"])
        out["%s/random" % mode] = np.float32(d["random"])
        if mode == "train":
            for n in ("match.self_attn.0.attention.fc_q.weight", "match.cross_attn.1.attention.fc_v.weight",
                      "match.lang_self_attn.attention.fc_k.weight", "match.match.6.weight", "lang.gru.weight_hh_l0",
                      "match.features_concat.0.weight", "match.lang_fc.0.weight"):
                out["train/grad/" + n] = dict(net.named_parameters())[n].grad.numpy()[:32].copy()   # first 32 rows only
            net.zero_grad()
    for k in list(out):
        if k.endswith(("lang_masks", "cluster_labels")):
            out[k] = out[k].astype(np.uint8)
    # inputs are NOT stored: tests rebuild them with listener_inputs() (numpy generator, seeded)
    np.savez_compressed(os.path.join(HERE, "listener_golden.npz"), **out)
    print("wrote listener_golden.npz:", {k: v.shape for k, v in out.items() if "grad" not in k})


if __name__ == "__main__":
    main()
