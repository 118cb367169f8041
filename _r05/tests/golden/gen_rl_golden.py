#!/usr/bin/env python3
"""Generates tests/golden/rl_golden.npz by RUNNING THE REFERENCE's own modules on CPU for the self-critical path:
lib/capeval/cider (Cider.compute_score), lib/captioning/loss_helper.py (compute_caption_reward, compute_cap_loss with
use_rl), model/caption_module.py (beam_decode, _forward_sample_batch with use_rl), model/listener.py with use_rl
(lang_module + Transformer match module), lib/grounding/loss_helper.py (get_grounding_loss / get_lobjcls_loss with
use_rl).  model/pipeline.py (`moderator`) cannot be imported (pytorch_lightning / MinkowskiEngine / compiled ops are
absent): the moderator step between speaker and listener is oracle/rl_oracle.py's restatement, whose OUTPUT is what the
reference listener is run on.  Run in the build container only.  Inputs/weights are rebuilt by the tests from the
seeded generators here and in gen_speaker_golden.py / gen_listener_golden.py."""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_speaker_golden as S  # noqa: E402
from gen_listener_golden import golden_weights, make_cfg as listener_cfg  # noqa: E402

REF = "/root/reference"
BEAM, TOPN = 3, 3
OPT_W = dict(ref_reward_weight=1, lang_reward_weight=1, listener_reward_weight=0.1, caption_reward_weight=1)


def vocab_str():
    v = S.make_vocab()
    return {"word2idx": v["word2idx"], "idx2word": {str(i): w for i, w in v["idx2word"].items()}}


def cider_cases(seed=21):
    """hand-built CIDEr inputs: shared n-grams, repeated reference sets, a one-word candidate, an unseen word"""
    rng = np.random.default_rng(seed)
    words = ["w%d" % i for i in range(12)]
    sent = lambda n: " ".join(rng.choice(words, n)) + " eos"
    gts, res = {}, {}
    for i in range(14):
        refs = [sent(int(rng.integers(3, 9))) for _ in range(int(rng.integers(1, 5)))]
        if i % 3 == 0:   # candidate = a perturbed reference (high score)
            toks = refs[0].split()
            toks[int(rng.integers(len(toks) - 1))] = "zzz" if i == 6 else str(rng.choice(words))
            cand = " ".join(toks)
        elif i == 4:
            cand = "eos"
        else:
            cand = sent(int(rng.integers(2, 10)))
        gts[str(i)], res[str(i)] = refs, [cand]
    for i in range(14, 18):   # repeated entries, as the reward builds them (same references, different candidates)
        gts[str(i)], res[str(i)] = gts[str(i - 14)], [sent(5)]
    return gts, res


def rl_corpus(B=2, Cn=4, seed=17):
    """dataset_data[scene][chunk] -> {scene_id, object_id}; organized[scene_id][object_id] -> [{token: [...]}, ...]"""
    rng = np.random.default_rng(seed)
    words = ["w%d" % i for i in range(S.V - 4)] + ["oov_a", "oov_b"]
    organized, dataset_data = {}, []
    for b in range(B):
        sid = "scene%04d_00" % b
        organized[sid] = {}
        for o in range(6):
            organized[sid][str(o)] = [{"token": [str(w) for w in rng.choice(words, int(rng.integers(3, S.MAXLEN)))]}
                                      for _ in range(int(rng.integers(2, 6)))]
        dataset_data.append([{"scene_id": sid, "object_id": str(int(rng.integers(0, 6)))} for _ in range(Cn)])
    ids = np.arange(B, dtype=np.int64)
    chunk_ids = np.stack([rng.permutation(Cn) for _ in range(B)]).astype(np.int64)
    return dataset_data, organized, ids, chunk_ids


def sem_cls(B=2, seed=19):
    return np.random.default_rng(seed).integers(0, 20, (B, S.K)).astype(np.float32)


def pad_table(table, T, dtype):
    N, k = len(table), len(table[0])
    out = np.full((N, k, T), -1 if np.issubdtype(dtype, np.integer) else 0, dtype)
    lens = np.zeros((N, k), np.int16)
    for n in range(N):
        for j in range(k):
            a = table[n][j].detach().numpy()
            out[n, j, :len(a)] = a
            lens[n, j] = len(a)
    return out, lens


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType("model"); pkg.__path__ = [os.path.join(REF, "model")]; sys.modules["model"] = pkg
    for name in ("trimesh", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    from model.caption_module import TopDownSceneCaptionModule
    from model.listener import ListenerNet
    from lib.capeval.cider.cider import Cider
    from lib.captioning.loss_helper import compute_cap_loss, compute_caption_reward
    from lib.grounding.loss_helper import get_grounding_loss, get_lobjcls_loss
    from oracle import rl_oracle

    out = {}
    # ---- CIDEr on hand-built sentences
    gts, res = cider_cases()
    mean, scores = Cider().compute_score(gts, res)
    out["cider/mean"], out["cider/scores"] = np.float64(mean), np.asarray(scores, np.float64)

    # ---- speaker, use_rl
    cfg, vocab, emb = S.make_cfg(), vocab_str(), S.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=S.K, num_locals=S.L, use_relation=True, use_oracle=False)
    sd = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"})
    sd["embeddings"] = torch.from_numpy(emb)
    cap.load_state_dict(sd)
    g_spk = np.load(os.path.join(HERE, "speaker_golden.npz"))
    d = {k: torch.from_numpy(v) for k, v in S.speaker_inputs().items()}
    d["adjacent_mat"] = torch.from_numpy(g_spk["adjacent_mat"].astype(np.float32))
    dataset_data, organized, ids, chunk_ids = rl_corpus()
    d["id"], d["chunk_ids"] = torch.from_numpy(ids), torch.from_numpy(chunk_ids)
    d["proposal_sem_cls_batched"] = torch.from_numpy(sem_cls())

    # beam_decode alone, on the step inputs
    si = {k: torch.from_numpy(v) for k, v in S.step_inputs().items()}
    done = cap.beam_decode(si["target"], si["obj"], si["mask"], BEAM, S.MAXLEN)
    seqs = [[b["seq"] for b in s] for s in done]
    lps = [[b["logps"].gather(1, b["seq"].unsqueeze(1)).squeeze(1) for b in s] for s in done]
    out["beam/seq"], out["beam/len"] = pad_table(seqs, S.MAXLEN, np.int16)
    out["beam/logps"], _ = pad_table(lps, S.MAXLEN, np.float32)
    out["beam/p"] = np.array([[b["p"] for b in s] for s in done], np.float32)

    random.seed(5)
    dd = cap(dict(d), use_tf=True, use_rl=True, is_eval=False, beam_opt={"train_beam_size": BEAM, "train_sample_topn": TOPN})
    out["rl/lang_cap"], out["rl/lang_cap_len"] = pad_table(dd["lang_cap"], S.MAXLEN, np.int16)
    out["rl/lang_logprob"], _ = pad_table(dd["lang_logprob"], S.MAXLEN, np.float32)
    out["rl/baseline_cap"], out["rl/baseline_len"] = pad_table(dd["baseline_cap"], S.MAXLEN + 1, np.int16)
    out["rl/assigned"] = dd["assigned_bbox_id_labels"].numpy()
    out["rl/good"] = dd["good_bbox_masks"].numpy()

    # ---- moderator (oracle restatement) -> reference listener, use_rl
    T = S.MAXLEN + 2
    mod = rl_oracle.moderator(dd, torch.from_numpy(emb), T)
    lcfg = listener_cfg()
    net = ListenerNet(lcfg)
    net.load_state_dict(golden_weights(net.state_dict()))
    net.train()
    for m_ in net.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    dd.update(mod)
    dd["istrain"] = torch.tensor([1])
    dd["object_cat"] = torch.zeros(2, 4, dtype=torch.long)   # evaluated (and ignored) as the default of a dict.get (loss_helper.py:248)
    random.seed(3)
    dd = net(dd, use_rl=True)
    _, dd = get_grounding_loss(dd, grounding=True, use_rl=True)
    _, dd = get_lobjcls_loss(dd, lang_cls=True, use_rl=True)
    loss_opt = dict(use_rl=True, sample_topn=TOPN, idx2word=vocab["idx2word"], train_dataset_data=dataset_data,
                    organized_data=organized, **OPT_W)
    out["reward/sampled"] = compute_caption_reward(dd, dd["lang_cap"], TOPN, vocab["idx2word"], dataset_data, organized).numpy()
    out["reward/baseline"] = compute_caption_reward(dd, dd["baseline_cap"], TOPN, vocab["idx2word"], dataset_data, organized).numpy()
    _, dd = compute_cap_loss(dd, loss_opt)
    total = dd["cap_loss"] + dd["ref_loss"] + dd["lang_loss"]
    total.backward()
    for k in ("sampled", "baseline"):
        out["lis/cluster_ref/" + k] = dd["cluster_ref"][k].detach().numpy()
        out["lis/lang_scores/" + k] = dd["lang_scores"][k].detach().numpy()
    for k in ("ref_loss", "ref_sampled_loss", "ref_baseline_loss", "ref_acc_mean", "ref_baseline_acc", "ref_iou_mean",
              "best_ious_mean", "lang_loss", "sampled_lang_loss", "baseline_lang_loss", "lang_acc", "lang_baseline_acc",
              "cap_loss", "cap_acc", "cap_rwd", "loc_rwd", "ttl_rwd"):
        out["loss/" + k] = dd[k].detach().numpy()
    out["loss/ref_iou_rate_0.25"], out["loss/ref_iou_rate_0.5"] = np.float32(dd["ref_iou_rate_0.25"]), np.float32(dd["ref_iou_rate_0.5"])
    out["lis/cluster_labels"] = dd["cluster_labels"].numpy().argmax(-1).astype(np.int16)
    for n in ("map_topdown.weight", "recurrent_cell_1.weight_hh", "map_feat.weight", "attend.weight", "classifier.2.weight"):
        out["grad/cap/" + n] = dict(cap.named_parameters())[n].grad.numpy()[:32].copy()
    for n in ("match.self_attn.0.attention.fc_q.weight", "match.cross_attn.1.attention.fc_v.weight", "match.match.6.weight",
              "lang.gru.weight_hh_l0", "lang.lang_cls.0.weight"):
        out["grad/lis/" + n] = dict(net.named_parameters())[n].grad.numpy()[:32].copy()
    np.savez_compressed(os.path.join(HERE, "rl_golden.npz"), **out)
    print("wrote rl_golden.npz", {k: v.shape for k, v in out.items() if "grad" not in k})
    print("cider", out["cider/scores"].round(3))
    print("reward sampled", out["reward/sampled"].round(3).tolist())
    print("losses", {k: float(out["loss/" + k]) for k in ("cap_loss", "ref_loss", "lang_loss", "cap_rwd", "loc_rwd")})


if __name__ == "__main__":
    main()
