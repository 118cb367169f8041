"""Captioning losses: cross-entropy caption loss, the self-critical (CIDEr + listener reward) caption loss and the
edge-orientation loss (reference: lib/captioning/loss_helper.py:15-96, :98-224, :226-307, :309-334)."""
import numpy as np
import torch
import torch.nn.functional as F

from .cider import cider_scores


class CiderCorpus:
    """The annotation store (`organized[scene_id][object_id]` -> tokenised descriptions) as device tensors for
    csrc/cider.hip: one row of token ids per reference sentence (+ "eos", as the reference joins them, loss_helper.py:60),
    a vocabulary id where the word is in the vocabulary, a corpus-private id >= V otherwise (string equality == id
    equality).  Built once per dataset."""
    MAX_TOKENS, MAX_SET_NGRAMS, MAX_ID = 160, 2048, 65534

    def __init__(self, organized, idx2word, device):
        canon = {}
        for i, w in idx2word.items():
            canon[w] = min(int(i), canon.get(w, int(i)))
        V = max(int(i) for i in idx2word) + 1
        extra, rows, self.sets = {}, [], {}
        for sid, objs in organized.items():
            for oid, descs in objs.items():
                first = len(rows)
                for d in descs:
                    rows.append([canon[w] if w in canon else extra.setdefault(w, V + len(extra)) for w in list(d["token"]) + ["eos"]])
                self.sets[(sid, oid)] = (first, len(descs))
        self.row_len = [len(r) for r in rows]
        self.ok = bool(rows) and max(self.row_len) <= self.MAX_TOKENS and V + len(extra) <= self.MAX_ID and "eos" in canon
        if not self.ok:
            return
        ldt = max(self.row_len)
        tok = np.zeros((len(rows), ldt), np.int32)
        for i, r in enumerate(rows):
            tok[i, :len(r)] = r
        self.tokens = torch.from_numpy(tok).to(device)
        self.lens = torch.tensor(self.row_len, dtype=torch.int32, device=device)
        self.ldt, self.eos, self.device = ldt, canon["eos"], device
        lut = np.arange(V, dtype=np.int64)
        for i, w in idx2word.items():
            lut[int(i)] = canon[w]
        self.lut = torch.from_numpy(lut).to(device) if (lut != np.arange(V)).any() else None   # two ids spelling one word
        self._pinned, self._slot = [None] * 4, 0

    def staging(self, n):
        """pinned int32 staging buffer (ring of 4: an earlier asynchronous upload may still be in flight)"""
        self._slot = (self._slot + 1) % 4
        b = self._pinned[self._slot]
        if b is None or b.numel() < n:
            b = self._pinned[self._slot] = torch.empty(max(n, 4096), dtype=torch.int32).pin_memory()
        return b


_CORPORA = {}
LOGP_SUM_TENSOR = True     # (tools/ab.py py:d3net_amd.captioning_loss.LOGP_SUM_TENSOR=0,1)


def _cider_device(corpus, entry_sets, cands, sample_topn):
    """entry_sets: per valid description its (first row, #rows) in the corpus; cands: E = len(entry_sets) * sample_topn device
    token tensors -> (E,) float64 device scores, or None when the batch does not fit the kernels' fixed tables."""
    from . import _lib
    from ._lib import check
    E = len(cands)
    uniq, u_of = {}, []
    for fs in entry_sets:
        u_of.append(uniq.setdefault(fs, len(uniq)))
    U = len(uniq)
    slot_row, u_off, mult = [], [0], [0] * U
    for (first, cnt), u in uniq.items():
        if cnt < 1 or 4 * sum(corpus.row_len[first:first + cnt]) > corpus.MAX_SET_NGRAMS:
            return None
        slot_row.extend(range(first, first + cnt)); u_off.append(len(slot_row))
    for u in u_of:
        mult[u] += sample_topn
    ent_u = [u for u in u_of for _ in range(sample_topn)]
    clen = [int(c.shape[0]) for c in cands]
    if max(clen) + 1 > corpus.MAX_TOKENS:
        return None
    ldc = max(max(clen), 1)
    pos = np.concatenate([e * ldc + np.arange(l) for e, l in enumerate(clen)]) if sum(clen) else np.zeros(0, np.int64)
    SR = len(slot_row)
    ngrams = 4 * sum(corpus.row_len[r] for r in slot_row)
    hash_slots = 1024
    while hash_slots < 4 * ngrams:
        hash_slots *= 2
    meta = np.concatenate([np.asarray(a, np.int32) for a in (slot_row, u_off, mult, ent_u, clen, pos)])
    stage = corpus.staging(meta.size)
    stage[:meta.size].copy_(torch.from_numpy(meta))
    dev = corpus.device
    md = stage[:meta.size].to(dev, non_blocking=True)
    o = np.cumsum([0, SR, U + 1, U, E, E, pos.size])
    d_slot, d_uoff, d_mult, d_ent, d_clen, d_pos = (md[o[i]:o[i + 1]] for i in range(6))
    cand = torch.zeros((E, ldc), dtype=torch.int32, device=dev)
    if pos.size:
        flat = torch.cat([c.reshape(-1) for c in cands])
        if corpus.lut is not None:
            flat = corpus.lut[flat.long()]
        cand.view(-1)[d_pos.long()] = flat.to(torch.int32)
    L = _lib.lib()
    ws = torch.empty(L.d3_cider_ws_bytes(SR, E, hash_slots), dtype=torch.uint8, device=dev)
    out = torch.empty(E, dtype=torch.float64, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(L.d3_cider_scores(corpus.tokens.data_ptr(), corpus.ldt, corpus.lens.data_ptr(), d_slot.data_ptr(), d_uoff.data_ptr(),
                                d_mult.data_ptr(), d_ent.data_ptr(), U, SR, cand.data_ptr(), ldc, d_clen.data_ptr(), E, corpus.eos, 6.0,
                                hash_slots, out.data_ptr(), flag.data_ptr(), ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream), "cider_scores")
    return out


def compute_caption_reward(data_dict, cap_tables, sample_topn, idx2word, dataset_data, organized_data, device_cider=True):
    """(loss_helper.py:15-96) CIDEr of every sampled caption against ALL ground-truth descriptions of its object, one
    scorer call for the whole batch (so the idf statistics are those of the batch).  Unannotated entries score 0.
    The reference also runs BLEU-4 here and multiplies it by a hard-coded weight of 0 (:83-88): not computed.
    On a GPU the n-gram statistics and the scores are computed on the device (csrc/cider.hip) from the token tensors as they
    are: no token leaves the device; the host only looks up which reference set each description belongs to (cached in
    `data_dict` for the second call of the step).  `device_cider=False`, CPU tensors or a batch beyond the kernels' fixed
    tables use the host scorer (d3net_amd/cider.py), which is pinned bit-exact to the reference's."""
    assert len(cap_tables[0]) == sample_topn
    annotated = data_dict["annotated"].reshape(-1)
    N = annotated.shape[0]
    scores = torch.zeros(N, sample_topn, device=annotated.device)
    ent = data_dict.get("_reward_entries")
    if ent is None or ent[0] is not data_dict["annotated"] or ent[1] is not data_dict["chunk_ids"]:   # host-side ids of the batch: one transfer per step
        chunk_ids = data_dict["chunk_ids"]
        Cn = chunk_ids.shape[1]
        dataset_ids = data_dict["id"].unsqueeze(1).repeat(1, Cn).reshape(-1).tolist()
        chunk_l = chunk_ids.reshape(-1).tolist()
        valid = (annotated == 1).nonzero().view(-1)
        keys = []
        for n in valid.tolist():
            raw = dataset_data[dataset_ids[n]][chunk_l[n]]
            keys.append((raw["scene_id"], raw["object_id"]))
        ent = data_dict["_reward_entries"] = (data_dict["annotated"], data_dict["chunk_ids"], valid, valid.tolist(), keys)
    valid, valid_l, keys = ent[2:]
    if not valid_l:
        return scores
    if device_cider and annotated.is_cuda:
        # keyed by (annotation object, device) and holding a reference to the object: a bare id() can be reused by a
        # different annotation dict once the first one is freed (train / val switch) and would then return its corpus
        ck = (id(organized_data), str(annotated.device))
        entry = _CORPORA.get(ck)
        if entry is None or entry[0] is not organized_data:
            entry = _CORPORA[ck] = (organized_data, CiderCorpus(organized_data, idx2word, annotated.device))
        corpus = entry[1]
        if corpus.ok:
            out = _cider_device(corpus, [corpus.sets[k] for k in keys], [cap_tables[n][k] for n in valid_l for k in range(sample_topn)],
                                sample_topn)
            if out is not None:
                scores[valid] = out.to(scores.dtype).view(len(valid_l), sample_topn)
                return scores
    lens = [len(cap_tables[n][k]) for n in valid_l for k in range(sample_topn)]
    flat = torch.cat([cap_tables[n][k].reshape(-1) for n in valid_l for k in range(sample_topn)]).tolist() if sum(lens) else []
    refs, cands, pos = [], [], 0
    ref_cache = {}
    for n, key in zip(valid_l, keys):
        gt = ref_cache.get(key)
        if gt is None:
            gt = ref_cache[key] = [" ".join(d["token"] + ["eos"]) for d in organized_data[key[0]][key[1]]]
        for k in range(sample_topn):
            l = lens[len(cands)]
            tokens = [idx2word[str(t)] for t in flat[pos:pos + l]]
            pos += l
            if "eos" not in tokens:
                tokens.append("eos")
            refs.append(gt); cands.append(" ".join(tokens))
    _, cider = cider_scores(refs, cands)
    scores[valid] = torch.from_numpy(cider).to(scores).view(len(valid_l), sample_topn)
    return scores


def _rl_cap_loss(data_dict, loss_opt):
    """(loss_helper.py:110-176) REINFORCE with the greedy caption as baseline; reward = caption_weight * (CIDEr_sampled -
    CIDEr_greedy) + listener_weight * (the listener's loss improvement, detached).
    NOTE as in the reference, the listener losses arrive in (scene, sample, chunk) row order (the moderator moves the
    sample axis out, pipeline.py:835-838) and are `.view`ed as (scene*chunk, sample) without moving it back (:143-146)."""
    topn = loss_opt.get("sample_topn", 1)
    caps, logprobs, base_caps = data_dict["lang_cap"], data_dict["lang_logprob"], data_dict["baseline_cap"]
    good = data_dict["good_bbox_masks"].long()
    annotated = data_dict["annotated"].reshape(-1)
    logp = data_dict.get("lang_logprob_sum") if LOGP_SUM_TENSOR else None     # (native beam search: the same sums as one tensor)
    if logp is None or logp.shape[0] != sum(len(beams) for beams in logprobs):
        logp = torch.stack([lp.sum() for beams in logprobs for lp in beams])
    args = (topn, loss_opt.get("idx2word"), loss_opt.get("train_dataset_data"), loss_opt.get("organized_data"))
    sampled = compute_caption_reward(data_dict, caps, *args).type_as(logp)
    baseline = compute_caption_reward(data_dict, base_caps, *args).type_as(logp)
    good = good.unsqueeze(1).repeat(1, topn)
    annotated = annotated.unsqueeze(1).repeat(1, topn)
    cap_reward = sampled - baseline
    shape = cap_reward.shape
    ref_reward = -(data_dict["ref_sampled_loss"].detach().view(shape) - data_dict["ref_baseline_loss"].detach().view(shape))
    lang_reward = -(data_dict["sampled_lang_loss"].detach().view(shape) - data_dict["baseline_lang_loss"].detach().view(shape))
    listener_reward = loss_opt.get("ref_reward_weight", 1) * ref_reward + loss_opt.get("lang_reward_weight", 1) * lang_reward
    rewards = loss_opt.get("caption_reward_weight", 1) * cap_reward + loss_opt.get("listener_reward_weight", 1) * listener_reward
    ngood = good.sum() + 1e-8
    cap_loss = (-rewards.view(-1) * logp * good.view(-1)).sum() / ngood
    cap_acc = (sampled * good * annotated).sum() / ((good * annotated).sum() + 1e-8)
    data_dict["cap_rwd"] = (cap_reward * good).sum() / ngood
    data_dict["loc_rwd"] = (listener_reward * good).sum() / ngood
    data_dict["ttl_rwd"] = (rewards * good).sum() / ngood
    data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
    data_dict["sampled_scores"], data_dict["baseline_scores"] = sampled, baseline      # (the CIDEr-D rewards themselves: parity tests)
    return cap_loss, data_dict


class _MaskedXE(torch.autograd.Function):
    """caption cross-entropy + word accuracy + the gradient of the logits in two launches (csrc/heads.hip: d3_masked_xe)"""

    @staticmethod
    def forward(ctx, pred, target, good):
        from . import _lib
        from ._lib import check
        from .pointgroup_ops import _on, _ptr, _stream
        N, S, V = pred.shape
        dev = pred.device
        pred = pred.contiguous()
        assert target.shape == (N, S) and target.stride(1) == 1 and target.dtype == torch.int64
        good8 = good.contiguous().view(torch.uint8) if good.dtype == torch.bool else good.to(torch.uint8).contiguous()
        L = _lib.lib()
        nws = int(L.d3_masked_xe_ws_bytes(N, S))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        dpred = torch.empty_like(pred)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        with _on(dev):
            check(L.d3_masked_xe(_ptr(pred), _ptr(target), target.stride(0), _ptr(good8), N, S, V, _ptr(dpred), _ptr(out), _ptr(ws), nws,
                                 _stream()), "masked_xe")
        ctx.save_for_backward(dpred)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, _g_acc):
        dpred, = ctx.saved_tensors
        return dpred * g, None, None


def compute_cap_loss(data_dict, loss_opt={}, native=True):
    """(loss_helper.py:177-224) XE over the descriptions whose target box is good (IoU > min_iou_threshold).
    Same value as the reference's `pred[good]` selection, without its host round trip: the targets of the other
    descriptions are set to the ignored index 0, and an all-bad batch gives 0 (the reference's else branch)."""
    if loss_opt.get("use_rl", False):
        return _rl_cap_loss(data_dict, loss_opt)
    max_len = loss_opt.get("max_len", 30)
    pred = data_dict["lang_cap"]
    num_words = pred.shape[1] + 1                                   # == int(lang_len.max()) (the captioner ran num_words - 1 steps)
    target = data_dict["lang_ids"].reshape(-1, max_len)[:, 1:num_words]
    good = data_dict["good_bbox_masks"]
    V = pred.shape[2]
    if native and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.int64 and target.stride(1) == 1 \
            and target.shape[1] == pred.shape[1] and good.dim() == 1:
        cap_loss, cap_acc = _MaskedXE.apply(pred, target, good)
        z = data_dict["bbox_feature"].new_zeros(())
        data_dict["cap_rwd"], data_dict["loc_rwd"], data_dict["ttl_rwd"] = z, z, z
        data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
        return cap_loss, data_dict
    t = torch.where(good.unsqueeze(1), target, torch.zeros_like(target)).reshape(-1)
    m = t != 0
    cnt = m.sum()
    denom = cnt.clamp(min=1).to(pred.dtype)
    p = pred.reshape(-1, V)
    cap_loss = F.cross_entropy(p, t, ignore_index=0, reduction="sum") / denom
    cap_acc = ((p.argmax(-1) == t) & m).sum().to(pred.dtype) / denom
    z = data_dict["bbox_feature"].new_zeros(())
    data_dict["cap_rwd"], data_dict["loc_rwd"], data_dict["ttl_rwd"] = z, z, z
    data_dict["cap_loss"], data_dict["cap_acc"] = cap_loss, cap_acc
    return cap_loss, data_dict


def radian_to_label(radians, num_bins=6):
    """(loss_helper.py:226-242)"""
    boundaries = torch.arange(np.pi / num_bins, np.pi - 1e-8, np.pi / num_bins).type_as(radians)
    return torch.bucketize(radians, boundaries)


class _OrientationLoss(torch.autograd.Function):
    """the whole loss on the device in one launch (csrc/heads.hip: d3_orientation_loss); the gradient w.r.t. the
    orientation logits comes out of the same pass"""

    @staticmethod
    def forward(ctx, preds, edge_index, nsrc, ntar, assign, rots, rot_masks, num_bins):
        import ctypes as C
        from . import _lib
        from ._lib import check
        from .pointgroup_ops import _on, _ptr, _stream
        B, E, nb = preds.shape
        assert nb == num_bins and preds.stride(2) == 1
        bounds = torch.arange(np.pi / num_bins, np.pi - 1e-8, np.pi / num_bins).float().tolist()   # `radian_to_label`
        barr = (C.c_float * max(len(bounds), 1))(*bounds)
        dev = preds.device
        edge_index, assign = edge_index.contiguous(), assign.contiguous()
        rots, rot_masks = rots.contiguous(), rot_masks.contiguous().float()
        nsrc, ntar = nsrc.contiguous().long(), ntar.contiguous().long()
        dpreds = torch.empty((B, E, nb), dtype=torch.float32, device=dev)
        out = torch.empty(3 + 3 * 256, dtype=torch.float32, device=dev)   # [loss, accuracy, weight sum | workgroup partials]
        with _on(dev):
            check(_lib.lib().d3_orientation_loss(_ptr(preds), preds.stride(0), preds.stride(1), _ptr(edge_index), _ptr(nsrc), _ptr(ntar),
                                                 _ptr(assign), _ptr(rots), _ptr(rot_masks), B, E, assign.shape[1], rots.shape[1],
                                                 nb, C.cast(barr, C.c_void_p), len(bounds), _ptr(dpreds), _ptr(out), _stream()),
                  "orientation_loss")
        ctx.save_for_backward(dpreds, out)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, _g_acc):
        dpreds, out = ctx.saved_tensors
        return dpreds * (g / out[2]), None, None, None, None, None, None, None


def _orientation_native_ok(data_dict, num_bins):
    p, ei = data_dict["edge_orientations"], data_dict["edge_index"]
    return (p.is_cuda and p.dtype == torch.float32 and p.dim() == 3 and p.stride(2) == 1 and p.shape[2] == num_bins and num_bins <= 16
            and ei.dtype == torch.float32 and data_dict["object_assignment"].dtype == torch.int64
            and data_dict["scene_object_rotations"].dtype == torch.float32 and p.shape[1] == ei.shape[2])


def compute_node_orientation_loss(data_dict, num_bins=6, native=True):
    """(loss_helper.py:244-307) relative rotation of the GT objects assigned to the two ends of every graph edge.
    All scenes at once on the padded (B, K*L) edge tensors: the reference loops over the scenes and slices the first
    n = n_source * n_target edges of each (a host round trip per scene); here edges >= n get weight 0."""
    assign = data_dict["object_assignment"]
    edge_indices, edge_preds = data_dict["edge_index"], data_dict["edge_orientations"]
    nsrc, ntar = data_dict["num_edge_source"], data_dict["num_edge_target"]
    if native and _orientation_native_ok(data_dict, num_bins):
        return _OrientationLoss.apply(edge_preds, edge_indices, nsrc, ntar, assign, data_dict["scene_object_rotations"],
                                      data_dict["scene_object_rotation_masks"], num_bins)
    B, K = assign.shape
    E = edge_indices.shape[2]
    rots = torch.gather(data_dict["scene_object_rotations"], 1, assign.view(B, K, 1, 1).repeat(1, 1, 3, 3))
    rot_masks = torch.gather(data_dict["scene_object_rotation_masks"], 1, assign)
    n = (nsrc * ntar).view(B, 1)
    live = (torch.arange(E, device=assign.device).view(1, E) < n).to(rot_masks.dtype)
    src, tar = edge_indices[:, 0].long(), edge_indices[:, 1].long()                      # (B,E); padded entries are 0
    rs = torch.gather(rots, 1, src.view(B, E, 1, 1).expand(-1, -1, 3, 3))
    rt = torch.gather(rots, 1, tar.view(B, E, 1, 1).expand(-1, -1, 3, 3))
    rel = torch.matmul(rs, rt.transpose(3, 2))
    rel = torch.acos(torch.clamp(0.5 * (torch.diagonal(rel, dim1=-2, dim2=-1).sum(-1) - 1), -1, 1))
    labels = radian_to_label(rel, num_bins).reshape(-1)
    masks = (torch.gather(rot_masks, 1, src) * torch.gather(rot_masks, 1, tar) * live).reshape(-1)
    preds = edge_preds.reshape(B * E, -1)
    loss = (F.cross_entropy(preds, labels, reduction="none") * masks).sum() / (masks.sum() + 1e-8)
    acc = ((preds.argmax(-1) == labels).to(masks.dtype) * (masks == 1).to(masks.dtype)).sum() / (masks.sum().float() + 1e-8)
    return loss, acc


def get_captioning_loss(data_dict, caption, orientation, num_bins, loss_opt):
    """(loss_helper.py:309-334 `get_loss`)"""
    z = data_dict["bbox_feature"].new_zeros(())
    if caption:
        _, data_dict = compute_cap_loss(data_dict, loss_opt)
    else:
        data_dict["cap_loss"], data_dict["cap_acc"], data_dict["pred_ious"] = z, z, z
    if orientation:
        data_dict["ori_loss"], data_dict["ori_acc"] = compute_node_orientation_loss(data_dict, num_bins)
    else:
        data_dict["ori_loss"], data_dict["ori_acc"] = z, z
    return data_dict["cap_loss"] + 0.1 * data_dict["ori_loss"], data_dict
