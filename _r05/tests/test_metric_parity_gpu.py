"""North-star metric parity that can fail: mAP@0.5 AND CIDEr@0.5IoU of a TRAINED detector + speaker on HELD-OUT scenes,
HIP bf16 path (what bench.py times) vs the fp32 CPU oracle on the same weights and the same scenes.

Reference path being mirrored: `PipelineNet.validation_step / validation_epoch_end` (model/pipeline.py:457-700) ->
`eval_caption_step / eval_caption_epoch` (lib/captioning/eval_helper.py:102-307) -> CIDEr (lib/capeval/cider/cider_scorer.py:11-193);
detection: `parse_predictions` + `APCalculator` (scripts/eval.py:128-166, lib/det/ap_helper.py:24-249).

Set-up (no dataset is available offline; SURVEY.md section 7 "metric parity without data"):
  * synthetic rooms with 6 hollow boxes of random class (6 object classes) whose point features carry the class NOISILY,
    so the trained semantic / offset heads make mistakes -- the operating point is not saturated;
  * every object has reference captions that are a function of what the model can see: its class, its size bucket and
    the class of its nearest neighbour ("the <class> is <size> next to the <class>") -- learnable, so CIDEr is well above 0;
  * the 7-level backbone + ScoreNet + relation graph + top-down captioner are trained jointly (mode 1, cross-entropy) with
    the bf16 MFMA executor for a few hundred AdamW steps on the training scenes;
  * evaluation in `eval()` (running BatchNorm statistics, per-proposal greedy decode) on scenes never trained on:
    HIP = the product's `validation_step` / `validation_epoch_end` + the device evaluator;
    oracle = PointGroupOracle(training=False) -> speaker_oracle.graph_module -> speaker_oracle.forward_scene_batch, scored by
    the same (golden-pinned, host-side) metric code.
Bound (BASELINE.json north_star: "mAP@0.5 / CIDEr within 0.5 % of reference"): asserted for the library's evaluation path as shipped
(eval mode runs the reference-precision kernels: d3net_amd/minkowski.py `exact_for`) on 128 held-out scenes (768 GT boxes) for three
training seeds of a model trained with the bf16 step; the bf16 kernels forced onto the evaluation are reported beside it under a
looser bound.  The oracle's mAP@0.5 is required inside (0.3, 0.95) and its CIDEr@0.5IoU > 0.2 so that equality is not 0 == 0 or 1 == 1.
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OBJ_CLASSES = [2, 3, 4, 5, 6, 7]                      # semantic ids of the six object classes (0 wall, 1 floor)
CLS_WORDS = ["chair", "table", "sofa", "bed", "shelf", "desk"]
SIZE_WORDS = ["small", "large"]
FILLER = ["the", "is", "next", "to", "a", "there", "near"]
WORDS = ["pad_", "unk", "sos", "eos"] + CLS_WORDS + SIZE_WORDS + FILLER
DIMS = (56, 44, 24)
SGN = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float32)


def make_vocab():
    return {"word2idx": {w: i for i, w in enumerate(WORDS)}, "idx2word": {str(i): w for i, w in enumerate(WORDS)},
            "special_tokens": {"bos_token": "sos", "eos_token": "eos", "unk_token": "unk", "pad_token": "pad_"}}


AMP = 3.0       # amplitude of the one-hot class evidence


def make_scene(seed, sigma):
    """one room; per-box random class; noisy class evidence in the first 20 feature channels"""
    from d3net_amd import synthetic as S
    rng = np.random.default_rng(1000 + seed)
    occ, sem, inst, boxes = S.occupancy_grid(DIMS, 6, (7, 15), (5, 13), seed=seed)
    cls = rng.choice(OBJ_CLASSES, size=len(boxes))
    for i in range(len(boxes)):
        sem[inst == i] = cls[i]
    sc = S.scene_from_grid(occ, sem, inst, seed=seed + 1, feat_seed=seed + 2)
    onehot = np.eye(20, dtype=np.float32)[np.clip(sc["sem_labels"], 0, 19)]
    sc["feats"][:, :20] = AMP * onehot + sigma * sc["feats"][:, :20]
    sc["feats"][:, 20:] = 0.0        # (the remaining channels carry nothing: fixed random values there are only something to memorise)
    sc["box_cls"] = cls
    return sc


def captions_of(batch_host, b):
    """reference captions of scene b's GT objects: {object slot: [token lists]} from class / size bucket / nearest neighbour"""
    centers, sizes = batch_host["center_label"][b].numpy(), batch_host["size_label"][b].numpy()
    mask, cls = batch_host["box_label_mask"][b].numpy() > 0, batch_host["sem_cls_label"][b].numpy()
    ids = np.nonzero(mask)[0]
    out = {}
    for o in ids:
        others = [j for j in ids if j != o]
        nb = min(others, key=lambda j: float(np.abs(centers[j] - centers[o]).sum())) if others else o
        cw = CLS_WORDS[OBJ_CLASSES.index(int(cls[o]))] if int(cls[o]) in OBJ_CLASSES else "unk"
        nw = CLS_WORDS[OBJ_CLASSES.index(int(cls[nb]))] if int(cls[nb]) in OBJ_CLASSES else "unk"
        sw = SIZE_WORDS[int(float(np.prod(sizes[o])) > 0.2 * 0.2 * 0.16)]
        out[int(o)] = [["the", cw, "is", sw, "next", "to", "the", nw], ["there", "is", "a", sw, cw, "near", "the", nw]]
    return out


def make_lang_batch(scenes, dev, chunk, seed, scene_ids):
    """device batch of `scenes` with structured captions (S.add_language picks the referred objects and fills every other key)"""
    from d3net_amd import synthetic as S
    V = len(WORDS)
    batch = S.add_language(S.make_batch(scenes, dev), dev, chunk=chunk, vocab=V, seed=seed)
    host = {k: batch[k].cpu() for k in ("center_label", "size_label", "box_label_mask", "sem_cls_label", "ref_box_label")}
    w2i = {w: i for i, w in enumerate(WORDS)}
    rng = np.random.default_rng(seed + 77)
    B = len(scenes)
    ids = np.zeros((B, chunk, batch["lang_ids"].shape[2]), np.int64)
    lens = np.zeros((B, chunk), np.int64)
    raw = []
    for b in range(B):
        caps = captions_of(host, b)
        for o, refs in caps.items():
            for r in refs:
                raw.append({"scene_id": scene_ids[b], "object_id": str(o), "token": r})
        for c in range(chunk):
            o = int(host["ref_box_label"][b, c].argmax())
            toks = caps[o][int(rng.integers(0, 2))] if o in caps else ["unk"]
            row = [w2i["sos"]] + [w2i[t] for t in toks] + [w2i["eos"]]
            ids[b, c, :len(row)] = row
            lens[b, c] = len(row)
    batch["lang_ids"] = torch.from_numpy(ids).to(dev)
    batch["lang_len"] = torch.from_numpy(lens).to(dev)
    batch["spk_lang_len"] = batch["lang_len"]
    batch["scene_id"] = list(scene_ids)
    return batch, raw


def _gt_keys(batch):
    c, s = batch["center_label"].cpu().numpy(), batch["size_label"].cpu().numpy()
    cls = batch["sem_cls_label"].cpu().numpy() - 2
    cls[cls < 0] = 17
    return dict(gt_bbox=torch.from_numpy(c[:, :, None] + SGN[None, None] * s[:, :, None] / 2),
                gt_bbox_label=batch["box_label_mask"].cpu(), sem_cls_label=torch.from_numpy(cls))


def run_parity(dev, sigma=1.0, steps=600, n_train=16, n_val=16, chunk=4, lr=4e-3, seed=0, exact_too=True, verbose=True, with_oracle=True,
               head_lr=None, train_exact=False, ablate=()):
    """train on the device, evaluate held-out scenes with the HIP path(s) and the CPU oracle -> dict of metrics
    (train_exact: the training steps run the reference-precision kernels instead of bf16 -- tools/train_precision_compare.py)"""
    from d3net_amd import synthetic as S, minkowski as ME, evaluator as ev
    from d3net_amd.caption_eval import eval_caption_step, eval_caption_epoch
    from d3net_amd.config import default_conf
    from d3net_amd.optim import FusedAdamW
    from d3net_amd.pipeline import PipelineNet
    from oracle import speaker_oracle as spo
    from oracle.pointgroup_oracle import PointGroupOracle

    V = len(WORDS)
    cfg = default_conf("pointgroup_captioning.yaml", overrides={"data": {"num_des_per_scene": chunk, "batch_size": 4}})
    assert len(cfg.model.blocks) == 7, "the shipped 7-level backbone"
    vocab = make_vocab()
    glove = np.random.default_rng(3).standard_normal((V, 300)).astype(np.float32)
    train_scenes = [make_scene(100 + i, sigma) for i in range(n_train)]
    val_scenes = [make_scene(500 + i, sigma) for i in range(n_val)]
    val_ids = ["scene%04d_00" % (900 + i) for i in range(n_val)]
    val_batches, raw_val = [], []
    for i in range(0, n_val, 4):
        b, raw = make_lang_batch(val_scenes[i:i + 4], dev, chunk, seed=50 + i, scene_ids=val_ids[i:i + 4])
        b["cluster_rand"] = torch.rand(2, 3, generator=torch.Generator().manual_seed(i))
        b["slot_perms"] = [torch.randperm(cfg.model.max_num_proposal, generator=torch.Generator().manual_seed(10 * i + j)) for j in range(4)]
        val_batches.append(b); raw_val += raw
    ds = types.SimpleNamespace(vocabulary=vocab, glove=glove, raw_data=raw_val, chunked_data=None, organized=None)
    torch.manual_seed(seed)
    net = PipelineNet(cfg, {"train": ds, "val": ds}).to(dev).train()
    params = [p for p in net.parameters() if p.requires_grad]
    if head_lr is None:
        opt = FusedAdamW(params, lr=lr, weight_decay=1e-4)
    else:       # the recurrent captioner at its own (smaller) step size
        det = {id(p) for p in net.detector.parameters()}
        opt = FusedAdamW([{"params": [p for p in params if id(p) in det], "lr": lr},
                          {"params": [p for p in params if id(p) not in det], "lr": head_lr}], lr=lr, weight_decay=1e-4)
    base_lrs = [gr["lr"] for gr in opt.param_groups]
    opt.register_step_pre_hook(lambda *a: net.detector.drop_stale_grads())
    train_batches = [make_lang_batch(train_scenes[i:i + 4], dev, chunk, seed=7 + i, scene_ids=["scene%04d_00" % (i + j) for j in range(4)])[0]
                     for i in range(0, n_train, 4)]
    # the class evidence of the TRAINING scenes is re-drawn every step (one-hot + sigma * N(0,1)): the detector has to learn to
    # denoise through its receptive field instead of memorising 16 fixed noise patterns; held-out scenes keep theirs fixed
    clean = [torch.nn.functional.one_hot(b["sem_labels"].clamp(0, 19), 20).float() for b in train_batches]
    gen = torch.Generator(device=dev).manual_seed(seed + 1)
    ME.set_exact(bool(train_exact))
    for it in range(steps):
        net.zero_grad(set_to_none=True)
        tb = dict(train_batches[it % len(train_batches)])
        f = tb["feats"].clone()
        f[:, :20] = AMP * clean[it % len(train_batches)] + sigma * torch.randn(f.shape[0], 20, device=dev, generator=gen)
        tb["feats"] = f
        if it == steps - 150:        # settle: the last steps at a quarter of the learning rate
            for gr, bl in zip(opt.param_groups, base_lrs):
                gr["lr"] = bl / 4
        loss, d = net.training_step(tb)
        loss.backward()
        opt.step()
        if verbose and (it % 100 == 0 or it == steps - 1):
            lab = tb["sem_labels"]
            om = lab > 1
            acc = float((d["semantic_scores"][0].argmax(1)[om] == lab[om]).float().mean())
            print("step %d: loss %.3f (detector %.3f, semantic %.3f, caption %.3f, cap_acc %.3f, object-point acc %.3f)" %
                  (it, float(loss.detach()), float(d["total_loss"][0].detach()), float(d["semantic_loss"][0].detach()), float(d["cap_loss"].detach()),
                   float(d["cap_acc"]), acc))
    ME.set_exact(False)
    torch.cuda.synchronize()
    if verbose:      # diagnostics: how well does the detector do on the held-out scenes, train-mode vs eval-mode BatchNorm
        saved = {k: v.clone() for k, v in net.named_buffers()}       # (a train-mode forward updates the running statistics)
        for mode in ("train", "eval"):
            net.train(mode == "train")
            with torch.no_grad():
                for k, v in net.named_buffers():
                    v.copy_(saved[k])
            acc, nobj, nraw, nkeep, calc = 0.0, 0, 0, 0, ev.APCalculator(0.5)
            with torch.no_grad():
                for b in val_batches:
                    d = net.detector.feed(dict(b), 0)
                    lab = b["sem_labels"]
                    m = lab > 1
                    acc += float((d["semantic_scores"].argmax(1)[m] == lab[m]).float().sum()); nobj += int(m.sum())
                    nraw += int(d["num_raw_proposals"]); nkeep += int(d["proposal_batch_mask"].sum())
                    d.update(_gt_keys(b))
                    calc.step(ev.parse_predictions(d, device_nms=False), ev.parse_groundtruths(d))
            print("held-out, %s-mode BatchNorm: object-point accuracy %.3f, raw clusters %d, proposals %d, mAP@0.5 %.4f"
                  % (mode, acc / max(nobj, 1), nraw, nkeep, calc.compute_metrics()["mAP"]))
        with torch.no_grad():
            for k, v in net.named_buffers():
                v.copy_(saved[k])
    net.eval()
    res = {}

    # ---- HIP: the product's validation hooks + the evaluator
    def hip_eval(bf16, exact_only=()):
        """bf16 False: the library's evaluation path as shipped (eval mode -> reference-precision kernels, minkowski.exact_for);
        True: the training step's bf16 kernels forced onto the evaluation (minkowski.set_eval_exact(False)) -- except the U-Nets named
        in exact_only (module-wise ablation: tools/bf16_ablation.py)"""
        calc = ev.APCalculator(0.5)
        outs, nprop = [], 0
        ME.set_eval_exact(not bf16)
        ME.set_eval_exact_only(exact_only)
        try:
            for b in val_batches:
                outs.append(net.validation_step(dict(b), 0))
                with torch.no_grad():
                    d = net.detector.feed(dict(b), 0)
                d.update(_gt_keys(b))
                calc.step(ev.parse_predictions(d, device_nms=False), ev.parse_groundtruths(d))
                nprop += int(d["proposal_batch_mask"].sum())
        finally:
            ME.set_eval_exact(True)
            ME.set_eval_exact_only(())
        log = net.validation_epoch_end(outs)
        cands = {}
        for o in outs:
            cands.update(o)
        return dict(mAP=calc.compute_metrics()["mAP"], cider=float(log["cider"]), bleu4=float(log["bleu-4"]), proposals=nprop, cands=cands)

    res["bf16"] = hip_eval(True)
    for names in ablate:
        res["bf16+exact:" + "+".join(names)] = hip_eval(True, names)
    if exact_too:
        res["exact"] = hip_eval(False)

    if not with_oracle:
        return res
    # ---- oracle: fp32 on the host, eval-mode BatchNorm, per-proposal greedy decode
    spo.TIE_RULE = "index"      # ties of the neighbour top-k: lower slot first, the rule csrc/proposals.hip implements (see oracle/speaker_oracle.py)
    det_sd = net.detector.state_dict()
    spk = {k: v.detach().cpu().clone() for k, v in net.speaker.state_dict().items()}
    gp = {k[len("graph."):]: v for k, v in spk.items() if k.startswith("graph.")}
    cp = {k[len("caption."):]: v for k, v in spk.items() if k.startswith("caption.")}
    orc = PointGroupOracle(cfg, det_sd, training=False)
    calc = ev.APCalculator(0.5)
    cands, nprop = {}, 0
    # the oracle evaluates the held-out batches on the host: four batches at a time on worker threads (the torch / numpy / C-oracle
    # calls release the GIL), four intra-op threads each; results are folded in batch order
    from concurrent.futures import ThreadPoolExecutor
    torch.set_num_threads(4)

    def oracle_batch(b):
        with torch.no_grad():
            host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items()}
            od = orc.feed(host, 0, rand=host["cluster_rand"], perms=host["slot_perms"])
            od.update(_gt_keys(host))
            parsed = (ev.parse_predictions(od), ev.parse_groundtruths(od))
            n = int(od["proposal_batch_mask"].sum())
            od.update(spo.graph_module(gp, od, cfg.model.num_graph_steps, cfg.model.num_locals))
            out = spo.forward_scene_batch(cp, od, cfg, cfg.model.max_num_proposal, cfg.model.num_locals, vocab["word2idx"]["sos"])
            od["lang_cap"] = out["lang_cap"]
            od["gt_bbox"] = host["gt_bbox"]          # (the language batch's gt_bbox: same corners, lib/dataset/pipeline.py:300)
            return parsed, n, eval_caption_step(od, vocab)

    with ThreadPoolExecutor(max_workers=4) as pool:
        for parsed, n, cand in pool.map(oracle_batch, val_batches):
            calc.step(*parsed)
            nprop += n
            cands.update(cand)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    spo.TIE_RULE = "topk"
    bleu, cider, rouge, _ = eval_caption_epoch(cands, raw_val, max_len=cfg.eval.max_des_len + 2, min_iou=cfg.eval.min_iou_threshold)
    res["oracle"] = dict(mAP=calc.compute_metrics()["mAP"], cider=float(cider[0]), bleu4=float(bleu[0][3]), proposals=nprop, cands=cands)
    for k in [k for k in res if k != "oracle"]:
        if k in res:
            same = sum(1 for key, v in res[k]["cands"].items() if key in cands and v["caption"] == cands[key]["caption"])
            res[k]["same_captions"] = (same, len(cands))
    if verbose:
        for k, v in res.items():
            print("%-6s mAP@0.5 %.4f  CIDEr@0.5IoU %.4f  BLEU-4 %.4f  proposals %d  %s" % (k, v["mAP"], v["cider"], v["bleu4"], v["proposals"],
                                                                                           v.get("same_captions", "")))
    return res


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_heldout_map_and_cider_parity_with_the_fp32_oracle(dev, seed):
    """128 held-out scenes (768 GT boxes: one flipped detection moves mAP@0.5 by ~0.13 %, one changed caption moves
    CIDEr@0.5IoU by ~0.15 - 0.4 %), three training seeds; the model is TRAINED with the bf16 step `bench.py` times.

    Precision policy under test (d3net_amd/minkowski.py `exact_for`): training steps run bf16 MFMA operands; evaluation -- every
    mAP / CIDEr the library reports: `validation_step`, `forward()` under `eval()` -- runs the reference-precision kernels.
      * the evaluation path AS SHIPPED must meet BASELINE.json's bound: mAP@0.5 and CIDEr@0.5IoU within 0.5 % of the fp32 CPU oracle
        on the same weights and scenes, for every seed (measured on the final tree: mAP and CIDEr identical to five digits; 766 / 768 / 768 of 768 captions);
      * the bf16 kernels forced onto the evaluation are REPORTED and held to a looser bound (mAP 1 %, CIDEr 2 % -- round 3's bound, not loosened: ADVICE r4): a bf16 forward
        perturbs the 16-dim proposal features by ~1e-2, ~1 % of the greedy captions change a token and an occasional box crosses
        IoU 0.5 -- discrete events worth 0.15 - 0.4 % of CIDEr each.  Six trained models measured +0.04 / -0.21 / -0.19 / -0.12 / -1.70 / +0.12 %
        (757 - 766 of 768 captions identical): not inside 0.5 % with any margin, which is why evaluation does not use them.
        A broken kernel (wrong neighbour table, wrong BatchNorm statistic) moves these by tens of percent.
    (the captioner trains at 1e-3, the detector at 4e-3: with ONE rate of 4e-3 the GRU captioner collapsed to the unigram
    distribution in 3 of 6 seeds -- the recipe, not the kernels)"""
    res = run_parity(dev, n_val=128, head_lr=1e-3, seed=seed, exact_too=True, verbose=False)
    o = res["oracle"]
    n_gt = len(o["cands"])
    assert n_gt >= 750, n_gt
    assert 0.3 < o["mAP"] < 0.95, ("operating point saturated or degenerate", o["mAP"])
    assert o["cider"] > 0.2, o["cider"]
    for k, name, b_map, b_cider, b_same in (("exact", "evaluation path as shipped (reference-precision kernels)", 0.005, 0.005, 0.99),
                                            ("bf16", "bf16 kernels forced onto the evaluation", 0.01, 0.02, 0.97)):
        h = res[k]
        print("seed %d, %s vs fp32 oracle: mAP@0.5 %.5f vs %.5f = %+.3f %%; CIDEr@0.5IoU %.5f vs %.5f = %+.3f %%; %d / %d captions identical"
              % (seed, name, h["mAP"], o["mAP"], 100 * (h["mAP"] - o["mAP"]) / o["mAP"], h["cider"], o["cider"],
                 100 * (h["cider"] - o["cider"]) / o["cider"], h["same_captions"][0], h["same_captions"][1]))
        assert abs(h["mAP"] - o["mAP"]) <= b_map * o["mAP"], (k, "mAP@0.5", h["mAP"], o["mAP"])
        assert abs(h["cider"] - o["cider"]) <= b_cider * o["cider"], (k, "CIDEr@0.5IoU", h["cider"], o["cider"])
        assert h["same_captions"][0] >= b_same * h["same_captions"][1], (k, h["same_captions"])
