#!/usr/bin/env python3
"""Generates tests/golden/grounding_eval_golden.npz by RUNNING THE REFERENCE's own `lib/grounding/eval_helper.get_eval`
on CPU.  Run in the build container only (needs /root/reference).  Inputs are not stored: `eval_inputs()` rebuilds them."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def eval_inputs(B=3, Cn=4, K=128, seed=11):
    rng = np.random.default_rng(seed)
    sgn = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)
    mask = np.zeros((B, K), np.float32)
    corners = np.zeros((B, K, 8, 3), np.float32)
    for b in range(B):
        nv = int(rng.integers(12, 60))
        slots = rng.permutation(K)[:nv]
        mask[b, slots] = 1
        c = rng.random((nv, 3)).astype(np.float32) * np.array([4, 3, 2], np.float32)
        s = rng.random((nv, 3)).astype(np.float32) * 0.9 + 0.2
        corners[b, slots] = c[:, None] + sgn[None] * s[:, None] / 2
    N = B * Cn
    ref = np.zeros((B, Cn, 8, 3), np.float32)
    labels = np.zeros((N, K), np.float32)
    cluster_ref = rng.standard_normal((N, K)).astype(np.float32)
    for b in range(B):
        valid = np.nonzero(mask[b])[0]
        for c in range(Cn):
            o = valid[rng.integers(len(valid))]
            ref[b, c] = corners[b, o] + rng.normal(0, 0.08, (1, 3)).astype(np.float32)
            labels[b * Cn + c, o] = 1
            if rng.random() < 0.5:
                cluster_ref[b * Cn + c, o] += 4.0          # half of the descriptions are grounded correctly
    return dict(proposal_batch_mask=mask, proposal_bbox_batched=corners, cluster_ref=cluster_ref, cluster_labels=labels,
                ref_box_corner_label=ref, unique_multiple=rng.integers(0, 2, (B, Cn)).astype(np.int64),
                object_cat=rng.integers(0, 18, (B, Cn)).astype(np.int64),
                lang_scores=rng.standard_normal((N, 18)).astype(np.float32))


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    from lib.grounding.eval_helper import get_eval
    d = {k: torch.from_numpy(v) for k, v in eval_inputs().items()}
    d = get_eval(d, grounding=True, use_lang_classifier=True)
    out = {"ref_acc": np.array(d["ref_acc"], np.float32), "ref_acc_mean": d["ref_acc_mean"].numpy(), "ref_iou": d["ref_iou"].numpy(),
           "best_ious": d["best_ious"].numpy(), "ref_iou_mean": d["ref_iou_mean"].numpy(), "best_ious_mean": d["best_ious_mean"].numpy(),
           "rate25": np.float32(d["ref_iou_rate_0.25"]), "rate5": np.float32(d["ref_iou_rate_0.5"]),
           "multiple": np.array(d["ref_multiple_mask"], np.int64), "others": np.array(d["ref_others_mask"], np.int64),
           "lang_acc": d["lang_acc"].numpy(), "pred_bboxes": d["pred_bboxes"].numpy(), "cluster_ref": d["cluster_ref"].numpy()}
    np.savez_compressed(os.path.join(HERE, "grounding_eval_golden.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
