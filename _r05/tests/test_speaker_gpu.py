"""GPU parity of the speaker path (d3net_amd.speaker) against golden vectors produced by the reference's own caption
module, and of the graph module against the CPU oracle.  fp32; tolerance rtol 1e-3 / atol 1e-4 on logits (summation
order of library GEMMs); index / mask outputs must be identical."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import speaker_oracle as spo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _load(dev):
    import gen_speaker_golden as G
    from gen_listener_golden import golden_weights
    from d3net_amd.speaker import TopDownSceneCaptionModule
    g = np.load(os.path.join(HERE, "golden", "speaker_golden.npz"))
    cfg, vocab, emb = G.make_cfg(), G.make_vocab(), G.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=G.K, num_locals=G.L, use_relation=True)
    sd = golden_weights({k: v for k, v in cap.state_dict().items() if k != "embeddings"})
    sd["embeddings"] = torch.from_numpy(emb)
    cap.load_state_dict(sd)
    d = {k: torch.from_numpy(v).to(dev) for k, v in G.speaker_inputs().items()}
    return G, g, cfg, vocab, cap.to(dev), d


def test_query_locals_all_matches_reference(dev):
    from d3net_amd.speaker import query_locals_all
    G, g, cfg, vocab, cap, d = _load(dev)
    adj = query_locals_all(d["proposal_bbox_batched"], d["proposal_batch_mask"], G.L, include_self=False)
    loc = query_locals_all(d["proposal_bbox_batched"], d["proposal_batch_mask"], G.L, include_self=True)
    assert np.array_equal(adj.cpu().numpy(), g["adjacent_mat"]) and np.array_equal(loc.cpu().numpy(), g["locals_incl_self"])


def test_caption_step_xe_eval_match_reference(dev):
    G, g, cfg, vocab, cap, d = _load(dev)
    from d3net_amd.speaker import query_locals_all
    d["adjacent_mat"] = query_locals_all(d["proposal_bbox_batched"], d["proposal_batch_mask"], G.L, include_self=False)
    si = {k: torch.from_numpy(v).to(dev) for k, v in G.step_inputs().items()}
    o, _, h, m = cap.step(si["word"], (si["h1"], si["h2"]), si["target"], si["obj"], si["mask"])
    for got, key in ((o, "step/out"), (h[0], "step/h1"), (h[1], "step/h2"), (m, "step/attn")):
        assert np.allclose(got.detach().cpu().numpy(), g[key], rtol=1e-3, atol=1e-4), key
    random.seed(5)
    dd = cap(dict(d), use_tf=True, use_rl=False, is_eval=False)
    assert np.array_equal(dd["assigned_bbox_id_labels"].cpu().numpy(), g["xe/assigned"])
    assert np.array_equal(dd["good_bbox_masks"].cpu().numpy(), g["xe/good"])
    assert np.array_equal(dd["valid_masks"].cpu().numpy(), g["xe/valid_masks"])
    assert np.allclose(dd["lang_cap"].detach().cpu().numpy(), g["xe/lang_cap"], rtol=1e-3, atol=1e-4)
    assert np.allclose(dd["topdown_attn"].detach().cpu().numpy(), g["xe/topdown_attn"], rtol=1e-3, atol=1e-5)
    assert abs(float(dd["pred_ious"]) - float(g["xe/pred_ious"])) < 1e-5
    logits = dd["lang_cap"]
    tgt = d["lang_ids"].reshape(-1, G.MAXLEN + 2)[:, 1:logits.shape[1] + 1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, G.V), tgt.reshape(-1), ignore_index=0)
    assert abs(float(loss) - float(g["xe/loss"])) < 1e-4
    loss.backward()
    params = dict(cap.named_parameters())
    for k in g.files:
        if k.startswith("xe/grad/"):
            ref = g[k]; got = params[k[len("xe/grad/"):]].grad.cpu().numpy()[:32]
            assert np.allclose(got, ref, rtol=5e-3, atol=1e-6 + 2e-3 * np.abs(ref).max()), k
    de = cap(dict(d), is_eval=True)
    assert np.array_equal(de["valid_masks"].cpu().numpy(), g["eval/valid_masks"])
    toks = de["lang_cap"].cpu().numpy()
    agree = (toks == g["eval/lang_cap"]).mean()
    assert agree > 0.999, agree     # greedy argmax: a near-tie may resolve differently under another GEMM summation order
    gi, gl = cap.greedy_decode(si["target"], si["obj"], si["mask"], G.MAXLEN + 1)
    assert [len(x[0]) for x in gi] == g["greedy/len"].tolist()
    assert np.array_equal(gi[0][0].cpu().numpy(), g["greedy/ids0"]) and np.allclose(gl[0][0].cpu().numpy(), g["greedy/lp0"], atol=1e-4)


def test_graph_module_vs_oracle(dev):
    import gen_speaker_golden as G
    from gen_listener_golden import golden_weights
    from d3net_amd.speaker import GraphModule
    gm = GraphModule(16, 128, 2, G.K, 128, G.L, return_edge=True, return_orientation=True)
    sd = golden_weights(gm.state_dict())
    gm.load_state_dict(sd)
    inp = G.speaker_inputs()
    dcpu = {k: torch.from_numpy(v) for k, v in inp.items()}
    ref = spo.graph_module(sd, dcpu, 2, G.L)
    out = gm.to(dev)({k: v.to(dev) for k, v in dcpu.items()})
    for k in ("adjacent_mat", "num_edge_source", "num_edge_target", "edge_index"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k].numpy()), k
    for k in ("bbox_feature", "edge_feature", "edge_orientations", "edge_distances"):
        assert np.allclose(out[k].detach().cpu().numpy(), ref[k].numpy(), rtol=1e-3, atol=1e-4), k
    assert int(ref["num_edge_source"][0]) == 37 and int(ref["num_edge_target"][0]) == G.L


@pytest.mark.parametrize("fuse_gates,N,V", [(1, 32, 3004), (0, 32, 3004), (1, 72, 600), (1, 13, 600)])
def test_native_topdown_pass_matches_step_by_step_at_config_shape(dev, fuse_gates, N, V):
    """(fuse_gates: the backward step's GRU gate kernels as epilogues of the GEMMs that complete their input -- D3_TD_FUSE_GATES, the
    default -- or as launches of their own, rounds 2-4)
    csrc/topdown.hip (one native call for the S-step teacher-forced pass, one for its backward) against the same module
    run step by step through library ops, at the shape of conf/pointgroup_captioning.yaml: batch 4 x 8 descriptions,
    K = 128 proposals, V = 3004, up to 31 steps.  fp32 both ways: logits / attention 1e-4, every parameter gradient and
    the gradients w.r.t. the object and target features 2e-3 of their scale (summation order over 31 steps)."""
    import types
    from d3net_amd.speaker import TopDownSceneCaptionModule, TopDownXEFunction, _TD_KEYS
    from d3net_amd import _lib, synthetic as S
    # (N = 72: the 8-scene batch of bench.py's strong-scaling ceiling and the joint step run more than 32 sequences -- the gate
    # epilogue then rides in the K-split kernel's two-row-tile variant; N = 13: a ragged single tile)
    torch.manual_seed(11)
    K, L = 128, 10
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(max_spk_len=30, min_iou_threshold=0.25))
    emb = np.random.default_rng(3).standard_normal((V, 300)).astype(np.float32)
    cap = TopDownSceneCaptionModule(cfg, S.make_vocabulary(V), emb, num_proposals=K, num_locals=L, use_relation=True).to(dev)
    g = torch.Generator().manual_seed(2)
    obj = torch.randn(N, K, 128, generator=g).to(dev).requires_grad_(True)
    tgt = torch.randn(N, 128, generator=g).to(dev).requires_grad_(True)
    masks = torch.zeros(N, K)
    for n in range(N):
        masks[n, torch.randperm(K, generator=g)[:L]] = 1
    masks = masks.to(dev)
    lens = torch.randint(10, 33, (N,), generator=g)
    lens[0] = 32
    words = torch.zeros(N, 32, dtype=torch.long)
    for n in range(N):
        words[n, 0] = 2; words[n, 1:lens[n] - 1] = torch.randint(4, V, (int(lens[n]) - 2,), generator=g); words[n, lens[n] - 1] = 3
    words = words.to(dev)
    Ssteps = int(lens.max()) - 1
    tgt_ids = words[:, 1:Ssteps + 1]

    def loss_of(logits):
        return torch.nn.functional.cross_entropy(logits.reshape(-1, V), tgt_ids.reshape(-1), ignore_index=0)

    # step by step (library ops)
    h = (obj.new_zeros(N, 512), obj.new_zeros(N, 512))
    proj = cap.map_feat(obj)
    outs, att = [], []
    for s in range(Ssteps):
        lo, _, h, m = cap.step(words[:, s], h, tgt, obj, masks.unsqueeze(-1), proj)
        outs.append(lo.unsqueeze(1)); att.append(m)
    ref_logits, ref_attn = torch.cat(outs, 1), torch.cat(att, -1)
    loss_of(ref_logits).backward()
    ref = {k: p.grad.clone() for k, p in cap.named_parameters()}
    ref_dobj, ref_dtgt = obj.grad.clone(), tgt.grad.clone()
    cap.zero_grad(); obj.grad = None; tgt.grad = None
    # native
    assert _lib.lib().d3_tuning_set(b"D3_TD_FUSE_GATES", fuse_gates) == 0
    sd = dict(cap.named_parameters())
    logits, attn = TopDownXEFunction.apply(cap.embeddings, words, masks, Ssteps, obj, tgt, *[sd[_TD_KEYS[k]] for k in _lib.TOPDOWN_PARAMS])
    assert logits.shape == (N, Ssteps, V) and attn.shape == (N, K, Ssteps)
    scale = float(ref_logits.abs().max())
    assert float((logits - ref_logits).abs().max()) < 1e-4 * scale, float((logits - ref_logits).abs().max())
    assert float((attn - ref_attn).abs().max()) < 1e-5
    try:
        loss_of(logits).backward()
        torch.cuda.synchronize()
    finally:
        _lib.lib().d3_tuning_set(b"D3_TD_FUSE_GATES", 1)
    for k, p in cap.named_parameters():
        err = float((p.grad - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-12)
        assert err < 2e-3, (k, err)
    assert float((obj.grad - ref_dobj).abs().max()) < 2e-3 * float(ref_dobj.abs().max())
    assert float((tgt.grad - ref_dtgt).abs().max()) < 2e-3 * float(ref_dtgt.abs().max())


def test_native_graph_module_matches_per_scene_form_with_gradients(dev):
    """csrc/edgeconv.hip (all scenes as one padded edge matrix, deterministic segmented adds) against the per-scene
    library-op form of the same module (itself checked against the CPU oracle above): every output identical / 1e-4, every
    parameter gradient and the gradient w.r.t. the proposal features 1e-3.  Scenes with 37, 3 (fewer valid proposals than
    num_locals + 1) and 0 valid proposals."""
    import gen_speaker_golden as G
    from d3net_amd.speaker import GraphModule
    torch.manual_seed(4)
    gm = GraphModule(16, 128, 2, G.K, 128, G.L, return_edge=True, return_orientation=True).to(dev)
    inp = {k: torch.from_numpy(v).to(dev) for k, v in G.speaker_inputs().items()}
    B = inp["proposal_batch_mask"].shape[0]
    # three scenes: the golden one, one with 3 valid proposals, one with none
    rep = lambda t: torch.cat([t[:1]] * 3, 0).clone()
    feats, masks, boxes = rep(inp["proposal_feats_batched"]), rep(inp["proposal_batch_mask"]), rep(inp["proposal_bbox_batched"])
    keep = masks[1].nonzero().view(-1)
    masks[1] = 0; masks[1, keep[:3]] = 1
    masks[2] = 0
    w_bbox = torch.randn(3, G.K, 128, device=dev); w_edge = torch.randn(3, G.K, G.L, 128, device=dev); w_ori = torch.randn(3, G.K * G.L, 7, device=dev)
    res = {}
    for native in (False, True):
        gm.native = native
        gm.zero_grad()
        f = feats.clone().requires_grad_(True)
        out = gm({"proposal_feats_batched": f, "proposal_batch_mask": masks, "proposal_bbox_batched": boxes})
        pred = torch.cat([out["edge_orientations"], out["edge_distances"].unsqueeze(-1)], -1)
        loss = (out["bbox_feature"] * w_bbox).sum() + (out["edge_feature"] * w_edge).sum() + (pred * w_ori).sum()
        loss.backward()
        res[native] = (out, {k: p.grad.clone() for k, p in gm.named_parameters()}, f.grad.clone())
    a, b = res[False], res[True]
    for k in ("adjacent_mat", "num_edge_source", "num_edge_target", "edge_index"):
        assert torch.equal(a[0][k].float(), b[0][k].float()), k
    assert int(a[0]["num_edge_source"][0]) == 37 and int(a[0]["num_edge_source"][1]) == 3 and int(a[0]["num_edge_source"][2]) == 0
    for k in ("bbox_feature", "edge_feature", "edge_orientations", "edge_distances"):
        assert float((a[0][k] - b[0][k]).abs().max()) < 1e-4 * (float(a[0][k].abs().max()) + 1e-6), k
    for k in a[1]:
        assert float((a[1][k] - b[1][k]).abs().max()) < 1e-3 * (float(a[1][k].abs().max()) + 1e-9), k
    assert float((a[2] - b[2]).abs().max()) < 1e-3 * float(a[2].abs().max())


def test_caption_inputs_from_per_scene_tensors_equal_the_replicated_form(dev):
    """d3_caption_select_target / d3_caption_inputs_fwd / _bwd against the reference's form (replicate every per-scene tensor per
    description, gather, masked_scatter: model/caption_module.py:416-508, :530-560, :866-885) at the config's shape: identical
    target ids / IoUs / labels / object features / masks (bit for bit), gradients of the proposal and edge features to 1e-6,
    including two descriptions of a scene that pick the same target."""
    import random
    from d3net_amd.speaker import TopDownSceneCaptionModule, query_locals_all
    import gen_speaker_golden as G
    torch.manual_seed(2)
    B, Cn, K, L, Fd = 4, 8, 256, 10, 128
    N = B * Cn
    cfg, vocab, emb = G.make_cfg(), G.make_vocab(), G.make_embeddings()
    cap = TopDownSceneCaptionModule(cfg, vocab, emb, num_proposals=K, num_locals=L, use_relation=True).to(dev)
    rng = np.random.default_rng(5)
    ctr = rng.random((B, K, 3)).astype(np.float32) * 4
    sz = (0.2 + rng.random((B, K, 3))).astype(np.float32)
    sg = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    corners = torch.from_numpy(ctr[:, :, None] + sg[None, None] * sz[:, :, None] / 2).to(dev)
    mask = torch.from_numpy((rng.random((B, K)) < 0.45).astype(np.float32)).to(dev)
    pick = rng.integers(0, K, N)
    pick[1] = pick[0]                                              # same target twice in scene 0
    refc = corners.view(B, 1, K, 8, 3).expand(B, Cn, K, 8, 3).reshape(N, K, 8, 3)[torch.arange(N), torch.from_numpy(pick).to(dev)]
    refc = refc + 0.03 * torch.randn(N, 1, 3, device=dev)
    ref_lab = torch.zeros(N, 128, device=dev)
    ref_lab[torch.arange(N), torch.from_numpy(rng.integers(0, 128, N)).to(dev)] = 1
    adj = query_locals_all(corners, mask, L, False, 0.5, "corner")
    d = dict(proposal_batch_mask=mask, proposal_center_batched=torch.from_numpy(ctr).to(dev), proposal_bbox_batched=corners,
             center_label=torch.randn(B, 128, 3, device=dev), gt_bbox=torch.randn(B, 128, 8, 3, device=dev),
             ref_box_label=ref_lab.view(B, Cn, 128), ref_box_corner_label=refc.view(B, Cn, 8, 3),
             annotated=torch.ones(B, Cn, device=dev), adjacent_mat=adj,
             lang_ids=torch.randint(1, 30, (B, Cn, cfg.data.max_spk_len + 2), device=dev),
             lang_len=torch.full((B, Cn), 6, device=dev))
    base0, edge0 = torch.randn(B, K, Fd, device=dev), torch.randn(B, K, L, Fd, device=dev)
    wobj, wtf = torch.randn(N, K, Fd, device=dev), torch.randn(N, Fd, device=dev)
    got = {}

    def hook(mode, *a):       # capture the captioner's inputs instead of running the recurrence
        got[mode] = a
        raise StopIteration

    outs = {}
    for mode, native in (("native", True), ("library", False)):
        cap.native = native
        base, edge = base0.clone().requires_grad_(True), edge0.clone().requires_grad_(True)
        dd = dict(d, bbox_feature=base, edge_feature=edge)
        import d3net_amd.speaker as SP
        orig_apply, orig_step = SP.TopDownXEFunction.apply, cap.step
        SP.TopDownXEFunction.apply = staticmethod(lambda emb_, wid, vm, S, obj, tf, *p: hook(mode, vm, obj, tf))
        cap.step = lambda word, hid, tf, obj, vm, proj: hook(mode, vm.squeeze(-1), obj, tf)
        try:
            cap._forward_sample_batch(dd, True, False)
        except StopIteration:
            pass
        finally:
            SP.TopDownXEFunction.apply, cap.step = orig_apply, orig_step
        vm, obj, tf = got[mode]
        ((obj * wobj).sum() + (tf * wtf).sum()).backward()
        outs[mode] = (dd["assigned_bbox_id_labels"], vm.reshape(N, K).float(), obj, tf, base.grad, edge.grad)
    a, b = outs["native"], outs["library"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert torch.allclose(a[4], b[4], atol=1e-5, rtol=1e-5) and torch.allclose(a[5], b[5], atol=1e-5, rtol=1e-5)
    assert float(a[5].abs().sum()) > 0
    # target ids / IoUs of the kernel against the library-op IoU chain
    cap.native = True
    ids_n, ious_n, lab_n = cap.select_target(mask, d["proposal_center_batched"], corners, d["center_label"], d["gt_bbox"], ref_lab, refc,
                                             torch.ones(N, device=dev))
    cap.native = False
    ids_l, ious_l, lab_l = cap.select_target(mask, d["proposal_center_batched"], corners, d["center_label"], d["gt_bbox"], ref_lab, refc,
                                             torch.ones(N, device=dev))
    assert torch.equal(ids_n, ids_l) and torch.equal(ious_n, ious_l) and torch.equal(lab_n, lab_l)
    assert torch.equal(ids_n.cpu(), torch.from_numpy(pick))


def test_local_context_mask_kernel_equals_topk_scatter(dev):
    """d3_query_locals_mask against torch.topk(largest=False) + scatter on the distance rows of query_locals_all, including rows
    with fewer than L valid candidates (ties among the masked 1e30 entries) and fully masked target rows"""
    import d3net_amd.speaker as SP
    rng = np.random.default_rng(8)
    B, K, L = 4, 256, 10
    ctr = rng.random((B, K, 3)).astype(np.float32) * 4
    sz = (0.2 + rng.random((B, K, 3))).astype(np.float32)
    sg = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    corners = torch.from_numpy(ctr[:, :, None] + sg[None, None] * sz[:, :, None] / 2).to(dev)
    mask = torch.from_numpy((rng.random((B, K)) < 0.45).astype(np.float32)).to(dev)
    mask[1] = 0; mask[1, :6] = 1            # a scene with 6 proposals: every row falls back on masked entries
    mask[2] = 0                              # an empty scene
    for include_self, thr in ((False, 0.5), (True, 0.5), (True, 0.05)):
        SP.NATIVE_TOPK_MASK = True
        a = SP.query_locals_all(corners, mask, L, include_self, thr, "corner")
        SP.NATIVE_TOPK_MASK = False
        b = SP.query_locals_all(corners, mask, L, include_self, thr, "corner")
        SP.NATIVE_TOPK_MASK = True
        assert torch.equal(a.sum(-1), torch.full((B, K), float(L), device=dev))
        valid_cols = mask.unsqueeze(1).expand(-1, K, -1) == 1
        assert torch.equal(a * valid_cols, b * valid_cols)            # identical wherever a valid proposal is concerned
        full = (b * valid_cols).sum(-1) == L
        assert torch.equal(a[full], b[full])                           # rows decided without ties: identical masks


def test_beam_and_greedy_selection_kernels_match_the_library_formulation(dev):
    """csrc/topdown.hip d3_beam_select / d3_greedy_select (round 4: one launch per decode step) against the ~25 library ops they
    replace (log_softmax, candidate sums, the b best of live * V best first, gathers of histories / running sums / hidden states,
    the -1000 penalty of finished beams: model/caption_module.py:176-307,367-371).  Integers identical; floats to 1e-6."""
    import ctypes as C
    import torch.nn.functional as F
    from d3net_amd import _lib
    L = _lib.lib()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, b, V, H, Tmax, eos = 7, 3, 3004, 64, 6, 3
    g = torch.Generator().manual_seed(4)
    seq_prev = torch.zeros(N, b, Tmax, dtype=torch.long, device=dev)
    sums = torch.zeros(N, b, device=dev)
    for t in range(4):
        live = 1 if t == 0 else b
        logits = (torch.randn(N * b, V, generator=g) * 3).to(dev)
        logits[:, eos] += 6.0 * (t >= 2)                                    # some beams end
        h1, h2 = torch.randn(N * b, H, generator=g).to(dev), torch.randn(N * b, H, generator=g).to(dev)
        last = int(t == 3)
        # library formulation
        logp = F.log_softmax(logits.view(N, b, V)[:, :live], dim=-1)
        cand = (sums[:, :live].unsqueeze(-1) + logp).reshape(N, live * V)
        ix = torch.sort(cand, dim=-1, descending=True, stable=True)[1][:, :b]
        beam_ix, tok = ix // V, ix % V
        chosen = logp.reshape(N, live * V).gather(1, ix)
        snap = sums[:, :live].gather(1, beam_ix) + chosen
        ended = (tok == eos) if not last else torch.ones_like(tok, dtype=torch.bool)
        seq = torch.cat([seq_prev[:, :, :t].gather(1, beam_ix.unsqueeze(-1).expand(N, b, t)), tok.unsqueeze(-1)], -1)
        base = torch.arange(N, device=dev).unsqueeze(1) * b
        sel = (base + beam_ix).reshape(-1)
        # kernel
        seq_out = torch.zeros(N, b, Tmax, dtype=torch.long, device=dev)
        tok_k = torch.empty(N * b, dtype=torch.long, device=dev)
        snap_k, sums_k = torch.empty(N, b, device=dev), torch.empty(N, b, device=dev)
        ended_k = torch.empty(N, b, dtype=torch.uint8, device=dev)
        h1o, h2o = torch.empty_like(h1), torch.empty_like(h2)
        sums_in = sums if t > 0 else torch.zeros(N, b, device=dev)
        rc = L.d3_beam_select(ptr(logits), ptr(sums_in), N, live, b, V, eos, last, t, Tmax, ptr(seq_prev) if t > 0 else None, ptr(seq_out), ptr(tok_k),
                              ptr(snap_k), ptr(ended_k), ptr(sums_k), ptr(h1), ptr(h2), ptr(h1o), ptr(h2o), H, st)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(tok_k.view(N, b), tok), t
        assert torch.equal(seq_out[:, :, :t + 1], seq), t
        assert torch.equal(ended_k.bool(), ended)
        assert torch.allclose(snap_k, snap, rtol=1e-6, atol=1e-5)
        assert torch.allclose(sums_k, snap - 1000.0 * ended.float(), rtol=1e-6, atol=1e-4)
        assert torch.equal(h1o, h1.index_select(0, sel)) and torch.equal(h2o, h2.index_select(0, sel))
        seq_prev, sums = seq_out, sums_k
    logits = (torch.randn(33, V, generator=g) * 2).to(dev)
    word, lp = torch.empty(33, dtype=torch.long, device=dev), torch.empty(33, device=dev)
    assert L.d3_greedy_select(ptr(logits), 33, V, ptr(word), ptr(lp), st) == 0
    rl, rw = F.log_softmax(logits, dim=-1).max(-1)
    assert torch.equal(word, rw) and torch.allclose(lp, rl, rtol=1e-6, atol=1e-6)


def test_selection_kernels_survive_nan_and_all_minus_inf_logits(dev):
    """NaN / all -inf logits leave every `c > best` comparison false; the arg-max sentinel must not be used as an index
    (a far out-of-bounds read = GPU memory fault).  torch.topk / .max return NaN scores there, which a non-finite-loss guard can
    skip: the kernels return in-range tokens with NaN (or -inf) scores, and healthy samples of the same launch are untouched."""
    import ctypes as C
    import torch.nn.functional as F
    from d3net_amd import _lib
    L = _lib.lib()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, b, V, H, Tmax, eos = 4, 3, 3004, 64, 6, 3
    g = torch.Generator().manual_seed(11)
    for t, live in ((0, 1), (1, b)):
        logits = (torch.randn(N * b, V, generator=g) * 3).to(dev)
        logits[0 * b:(0 + 1) * b] = float("nan")              # sample 0: NaN everywhere
        logits[1 * b:(1 + 1) * b] = float("-inf")             # sample 1: all -inf (log-softmax of it is NaN)
        h1, h2 = torch.randn(N * b, H, generator=g).to(dev), torch.randn(N * b, H, generator=g).to(dev)
        sums_in = torch.zeros(N, b, device=dev)
        seq_prev = torch.zeros(N, b, Tmax, dtype=torch.long, device=dev)
        seq_out = torch.zeros(N, b, Tmax, dtype=torch.long, device=dev)
        tok_k = torch.full((N * b,), -7, dtype=torch.long, device=dev)
        snap_k, sums_k = torch.empty(N, b, device=dev), torch.empty(N, b, device=dev)
        ended_k = torch.empty(N, b, dtype=torch.uint8, device=dev)
        h1o, h2o = torch.empty_like(h1), torch.empty_like(h2)
        rc = L.d3_beam_select(ptr(logits), ptr(sums_in), N, live, b, V, eos, 0, t, Tmax, ptr(seq_prev) if t > 0 else None, ptr(seq_out), ptr(tok_k),
                              ptr(snap_k), ptr(ended_k), ptr(sums_k), ptr(h1), ptr(h2), ptr(h1o), ptr(h2o), H, st)
        assert rc == 0
        torch.cuda.synchronize()                                # (a fault would surface here)
        tk = tok_k.view(N, b)
        assert bool(((tk >= 0) & (tk < V)).all())
        assert not bool(torch.isfinite(snap_k[:2]).any()) and bool(torch.isfinite(snap_k[2:]).all())
        # the healthy samples equal the library formulation
        logp = F.log_softmax(logits.view(N, b, V)[2:, :live], dim=-1)
        ix = torch.sort(logp.reshape(N - 2, live * V), dim=-1, descending=True, stable=True)[1][:, :b]
        assert torch.equal(tk[2:], ix % V)
    logits = (torch.randn(5, V, generator=g) * 2).to(dev)
    logits[1] = float("nan"); logits[3] = float("-inf")
    word, lp = torch.empty(5, dtype=torch.long, device=dev), torch.empty(5, device=dev)
    assert L.d3_greedy_select(ptr(logits), 5, V, ptr(word), ptr(lp), st) == 0
    torch.cuda.synchronize()
    assert bool(((word >= 0) & (word < V)).all()) and bool(torch.isnan(lp[[1, 3]]).all())
    rl, rw = F.log_softmax(logits[[0, 2, 4]], dim=-1).max(-1)
    assert torch.equal(word[[0, 2, 4]], rw) and torch.allclose(lp[[0, 2, 4]], rl, rtol=1e-6, atol=1e-6)
