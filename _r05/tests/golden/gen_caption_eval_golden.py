#!/usr/bin/env python3
"""Generates tests/golden/caption_eval_golden.npz by RUNNING THE REFERENCE's own lib/captioning/eval_helper.py
(assign_dense_caption, prepare_corpus, filter / check / organize candidates) and lib/capeval/{cider,bleu,rouge} on CPU.
Run in the build container only (needs /root/reference).  Inputs are rebuilt by `caption_inputs()`."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SGN = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)


def caption_inputs(B=3, K=128, G=24, L=31, V=60, seed=17):
    rng = np.random.default_rng(seed)
    words = ["pad_", "unk", "sos", "eos"] + ["w%d" % i for i in range(V - 4)]
    vocab = {"idx2word": {str(i): w for i, w in enumerate(words)},
             "special_tokens": {"bos_token": "sos", "eos_token": "eos", "unk_token": "unk", "pad_token": "pad_"}}
    pred = np.zeros((B, K, 8, 3), np.float32)
    gt = np.zeros((B, G, 8, 3), np.float32)
    gt_mask = np.zeros((B, G), np.float32)
    gt_ids = np.zeros((B, G), np.int64)
    caps = rng.integers(4, V, (B, K, L)).astype(np.int64)
    caps[rng.random((B, K, L)) < 0.08] = 3                       # eos somewhere in most captions
    raw = []
    for b in range(B):
        ng = int(rng.integers(5, 15))
        c = rng.random((ng, 3)).astype(np.float32) * np.array([4, 3, 2], np.float32)
        s = rng.random((ng, 3)).astype(np.float32) * 0.9 + 0.3
        gt[b, :ng] = c[:, None] + SGN[None] * s[:, None] / 2
        gt_mask[b, :ng] = 1
        gt_ids[b, :ng] = rng.permutation(40)[:ng]
        # proposals: jittered copies of the GT boxes (some good, some poor) + random boxes
        pc = np.concatenate([c + rng.normal(0, 0.12, c.shape), rng.random((K - ng, 3)) * np.array([4, 3, 2])]).astype(np.float32)
        ps = np.concatenate([s * rng.uniform(0.7, 1.3, s.shape), rng.random((K - ng, 3)) * 0.9 + 0.3]).astype(np.float32)
        perm = rng.permutation(K)
        pred[b] = (pc[:, None] + SGN[None] * ps[:, None] / 2)[perm]
        for g in range(ng):
            for _ in range(int(rng.integers(1, 4))):
                raw.append({"scene_id": "scene%04d_00" % b, "object_id": str(int(gt_ids[b, g])),
                            "token": [words[i] for i in rng.integers(4, V, int(rng.integers(5, 20)))]})
    return dict(pred_captions=caps, pred_boxes=pred, gt_boxes=gt, gt_box_ids=gt_ids, gt_box_masks=gt_mask,
                scene_list=["scene%04d_00" % b for b in range(B)], vocab=vocab, raw=raw)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    import lib.captioning.eval_helper as eh
    import lib.capeval.cider.cider as capcider
    import lib.capeval.bleu.bleu as capbleu
    import lib.capeval.rouge.rouge as caprouge
    from lib.utils.bbox import generalized_box3d_iou
    inp = caption_inputs()
    t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
    gious = generalized_box3d_iou(t["pred_boxes"], t["gt_boxes"], t["gt_box_masks"].sum(1).long(), rotated_boxes=False, needs_grad=False)
    cands = eh.assign_dense_caption(t["pred_captions"], t["pred_boxes"], t["gt_boxes"], t["gt_box_ids"], t["gt_box_masks"],
                                    inp["scene_list"], inp["vocab"]["idx2word"], inp["vocab"]["special_tokens"])
    corpus = eh.prepare_corpus(inp["raw"], cands, 30)
    out = {"gious": gious.numpy(), "keys": np.array(sorted(cands.keys())),
           "ious": np.array([cands[k]["iou"] for k in sorted(cands)], np.float64),
           "captions": np.array([cands[k]["caption"] for k in sorted(cands)])}
    for thr in (0.25, 0.5):
        c = eh.filter_candidates(cands, thr)
        c = eh.check_candidates(corpus, c)
        c = eh.organize_candidates(corpus, c)
        score, scores = capcider.Cider().compute_score(corpus, c)
        out["cider_%s" % thr] = np.float64(score)
        out["cider_scores_%s" % thr] = np.asarray(scores, np.float64)
        bleu, bleu_list = capbleu.Bleu(4).compute_score(corpus, c)
        out["bleu_%s" % thr] = np.asarray(bleu, np.float64)
        out["bleu_list_%s" % thr] = np.asarray(bleu_list, np.float64)
        rouge, rouges = caprouge.Rouge().compute_score(corpus, c)
        out["rouge_%s" % thr] = np.float64(rouge)
        out["rouge_scores_%s" % thr] = np.asarray(rouges, np.float64)
        out["corpus_keys"] = np.array(list(corpus.keys()))
    np.savez_compressed(os.path.join(HERE, "caption_eval_golden.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()}, float(out["cider_0.5"]), float(out["cider_0.25"]))


if __name__ == "__main__":
    main()
