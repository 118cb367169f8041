"""GPU parity of the listener path (d3net_amd.listener, HIP attention core) against the golden vectors produced by the
reference's own modules and against the pinned CPU oracle.  Tolerance: fp32 everywhere; differences come from the
summation order of library GEMMs / MIOpen GRU / the attention kernel: rtol 1e-3, atol 1e-4 on outputs."""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_attention_core_fwd_bwd_vs_torch(dev):
    from d3net_amd.listener import AttentionCoreFunction
    torch.manual_seed(0)
    for (B, h, nq, nk, dk, dv, div, use_bias, use_mask) in [(8, 4, 128, 128, 32, 32, 4, True, False), (6, 4, 128, 24, 32, 32, 1, False, True),
                                                          (6, 4, 24, 24, 16, 16, 1, False, True), (3, 2, 50, 77, 8, 24, 3, True, True)]:
        q = torch.randn(B, nq, h * dk, device=dev, requires_grad=True)
        k = torch.randn(B, nk, h * dk, device=dev, requires_grad=True)
        v = torch.randn(B, nk, h * dv, device=dev, requires_grad=True)
        bias = torch.randn(B // div, h, nq, nk, device=dev) if use_bias else None
        mask = (torch.rand(B, nk, device=dev) > 0.3).float() if use_mask else None
        if mask is not None:
            mask[:, 0] = 1
        out = AttentionCoreFunction.apply(q, k, v, bias, mask, h, div)
        g = torch.randn_like(out)
        out.backward(g)
        q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        att = torch.matmul(q2.view(B, nq, h, dk).permute(0, 2, 1, 3), k2.view(B, nk, h, dk).permute(0, 2, 3, 1)) / np.sqrt(dk)
        if bias is not None:
            att = att + bias.repeat_interleave(div, 0)
        if mask is not None:
            att = att.masked_fill(mask[:, None, None, :] == 0, -np.inf)
        ref = torch.matmul(torch.softmax(att, -1), v2.view(B, nk, h, dv).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, nq, h * dv)
        ref.backward(g)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
        for a, b in ((q.grad, q2.grad), (k.grad, k2.grad), (v.grad, v2.grad)):
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), float((a - b).abs().max())


def _load(dev):
    from gen_listener_golden import golden_weights, make_cfg, listener_inputs
    from d3net_amd.listener import ListenerNet
    g = np.load(os.path.join(HERE, "golden", "listener_golden.npz"))
    net = ListenerNet(make_cfg())
    net.load_state_dict(golden_weights(net.state_dict()))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    d = {k: torch.from_numpy(v).to(dev) for k, v in listener_inputs().items()}
    return g, net.to(dev), d


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_listener_matches_reference_golden(dev, mode):
    from d3net_amd.listener import get_grounding_loss, get_lobjcls_loss
    g, net, d = _load(dev)
    net.train(mode == "train")
    d["istrain"] = torch.tensor([1 if mode == "train" else 0])
    random.seed(3)
    d = net(d)
    assert abs(d["random"] - float(g[mode + "/random"])) < 1e-7
    _, d = get_grounding_loss(d)
    _, d = get_lobjcls_loss(d)
    for k in ("cluster_ref", "lang_scores", "lang_emb", "lang_hiddens", "lang_masks", "cluster_labels", "ref_loss", "lang_loss",
              "ref_acc_mean", "lang_acc", "ref_iou_mean", "best_ious_mean", "ref_iou_rate_0.25", "ref_iou_rate_0.5"):
        ref = g["%s/%s" % (mode, k)]
        got = d[k].detach().cpu().numpy()
        assert np.allclose(got, ref, rtol=1e-3, atol=1e-4), (mode, k, float(np.abs(got - ref).max()))
    if mode == "train":
        (d["ref_loss"] + d["lang_loss"]).backward()
        params = dict(net.named_parameters())
        for k in g.files:
            if k.startswith("train/grad/"):
                ref = g[k]
                got = params[k[len("train/grad/"):]].grad.cpu().numpy()[:32]
                assert np.allclose(got, ref, rtol=5e-3, atol=1e-5 + 2e-3 * np.abs(ref).max()), (k, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("bidir", [False, True], ids=["forward", "bidirectional"])
def test_native_packed_gru_matches_library_gru_at_config_shape(dev, bidir):
    """(bidirectional: model/lang_module.py:15-24,58-61 -- the reverse direction is the native recurrence over the descriptions read
    backwards, the two directions averaged.)
    csrc/topdown.hip's packed-sequence GRU (one GEMM for all input gates + one fused launch per step) against nn.GRU over
    pack_padded_sequence on the same parameters: batch 32, T = 128, lengths 1..128 (conf/pointgroup_grounding.yaml shape).
    Outputs 1e-5, parameter gradients 1e-3 of their scale (128 sequential steps, summation order)."""
    import types
    from d3net_amd.listener import LangModule
    torch.manual_seed(3)
    cfg = types.SimpleNamespace(model=types.SimpleNamespace(num_bbox_class=18, use_lang_classifier=True, use_bidir=bidir))
    lm = LangModule(cfg).to(dev)
    for m in lm.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B, Cn, T = 4, 8, 128
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(B, Cn, T, 300, generator=g).to(dev)
    lens = torch.randint(1, T + 1, (B, Cn), generator=g)
    lens[0, 0], lens[0, 1] = 1, T
    lens = lens.to(dev)
    w_h, w_l, w_s = torch.randn(B * Cn, T, 256, device=dev), torch.randn(B * Cn, 256, device=dev), torch.randn(B * Cn, 18, device=dev)
    res = {}
    for native in (False, True):
        lm.native = native
        lm.zero_grad()
        d = lm({"lang_feat": feat, "lang_len": lens})
        loss = (d["lang_hiddens"] * w_h).sum() + (d["lang_emb"] * w_l).sum() + (d["lang_scores"] * w_s).sum()
        loss.backward()
        res[native] = (d, {k: p.grad.clone() for k, p in lm.named_parameters()})
    a, b = res[False], res[True]
    for k in ("lang_hiddens", "lang_emb", "lang_scores"):
        assert float((a[0][k] - b[0][k]).abs().max()) < 1e-5 * (1 + float(a[0][k].abs().max())), k
    assert torch.equal(a[0]["lang_masks"], b[0]["lang_masks"])
    for k in a[1]:
        err = float((a[1][k] - b[1][k]).abs().max()) / (float(a[1][k].abs().max()) + 1e-12)
        assert err < 1e-3, (k, err)


def test_attention_mfma_form_equals_scalar_form(dev):
    """csrc/attention.hip: the fp32-MFMA kernels against the scalar-FMA kernels of round 1 (D3_ATTN_SCALAR=1) on ragged shapes
    (queries / keys not multiples of 16, head dims below 32, bias shared by groups of batch items, masked keys): the same fp32
    arithmetic in another summation order."""
    import os
    from d3net_amd.listener import AttentionCoreFunction
    torch.manual_seed(1)
    for (B, h, nq, nk, dk, dv, div, use_bias, use_mask) in [(8, 4, 128, 128, 32, 32, 4, True, True), (4, 2, 37, 101, 12, 20, 2, True, True),
                                                          (3, 4, 128, 5, 32, 32, 1, False, True), (2, 1, 1, 128, 32, 32, 1, False, False)]:
        q = torch.randn(B, nq, h * dk, device=dev)
        k = torch.randn(B, nk, h * dk, device=dev)
        v = torch.randn(B, nk, h * dv, device=dev)
        bias = torch.randn(B // div, h, nq, nk, device=dev) if use_bias else None
        mask = (torch.rand(B, nk, device=dev) > 0.4).float() if use_mask else None
        if mask is not None:
            mask[:, 0] = 1
        g = torch.randn(B, nq, h * dv, device=dev)
        res = []
        from d3net_amd import _lib
        for scalar in (False, True):
            with _lib.tuning(D3_ATTN_SCALAR=int(scalar)):
                qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
                out = AttentionCoreFunction.apply(qq, kk, vv, bias, mask, h, div)
                out.backward(g)
                torch.cuda.synchronize()
                res.append((out.detach(), qq.grad, kk.grad, vv.grad))
        for a, b in zip(*res):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * (1 + float(b.abs().max()))), float((a - b).abs().max())


def test_native_projections_layernorm_and_channel_gemms_match_the_library_ops(dev):
    """d3net_amd/nativelinear.py (fc_q / fc_k / fc_v / fc_o, lang_fc, the kernel-size-1 Conv1d stacks as channels-last GEMMs on
    csrc/hgemm.hip; LayerNorm(a + b) on csrc/layernorm.hip) against the library-op formulation of the SAME module
    (`listener._CONV1D_LIB = True`: nn.Linear / nn.Conv1d / nn.LayerNorm) at the shape of conf/pointgroup_grounding.yaml:
    4 scenes x 8 descriptions, 128 proposals, T = 128.  Reference: model/match_module.py:143-336,
    model/transformer/attention.py:134-176.  Outputs 1e-4, every parameter gradient 2e-3 of its scale."""
    import d3net_amd.listener as LI
    from gen_listener_golden import make_cfg
    torch.manual_seed(5)
    cfg = make_cfg()
    net = LI.TransformerMatchModule(cfg).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B, Cn, K, T = 4, cfg.data.num_des_per_scene, cfg.model.max_num_proposal, 128
    g = torch.Generator().manual_seed(1)
    d0 = {"proposal_center_batched": torch.rand(B, K, 3, generator=g).to(dev) * 3,
          "proposal_feats_batched": torch.randn(B, K, cfg.model.m, generator=g).to(dev),
          "proposal_batch_mask": (torch.rand(B, K, generator=g) > 0.4).float().to(dev),
          "istrain": torch.tensor([0]),
          "lang_hiddens": torch.randn(B * Cn, T, 256, generator=g).to(dev),
          "lang_masks": (torch.arange(T)[None, :] < torch.randint(5, T, (B * Cn, 1), generator=g)).float().to(dev)}
    res = {}
    for lib in (True, False):
        LI._CONV1D_LIB = lib
        try:
            net.zero_grad(set_to_none=True)
            random.seed(2)
            out = net(dict(d0))["cluster_ref"]
            w = torch.linspace(-1, 1, out.numel(), device=dev).view_as(out)
            (out * w).sum().backward()
            torch.cuda.synchronize()
            res[lib] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            LI._CONV1D_LIB = False
    (o_lib, g_lib), (o_nat, g_nat) = res[True], res[False]
    assert o_lib.shape == (B * Cn, K)
    assert torch.allclose(o_nat, o_lib, rtol=1e-4, atol=1e-4 * float(o_lib.abs().max())), float((o_nat - o_lib).abs().max())
    assert set(g_lib) == set(g_nat)
    for n in g_lib:
        scale = float(g_lib[n].abs().max()) + 1e-12
        # (+ 2e-5: a bias in front of a BatchNorm has a zero gradient up to rounding -- both sides hold ~1e-6 of noise there)
        assert float((g_nat[n] - g_lib[n]).abs().max()) <= 2e-3 * scale + 2e-5, (n, float((g_nat[n] - g_lib[n]).abs().max()), scale)


def test_add_layer_norm_kernel_vs_torch(dev):
    """csrc/layernorm.hip against torch.nn.functional.layer_norm (+ the residual add), ragged row counts and widths"""
    from d3net_amd import nativelinear as NL
    torch.manual_seed(0)
    for R, D, with_b in ((4096, 128, True), (37, 128, False), (1000, 300, True), (5, 64, True), (130, 1000, False)):
        ln = torch.nn.LayerNorm(D).to(dev)
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5); ln.bias.normal_()
        a = torch.randn(R, D, device=dev, requires_grad=True)
        b = torch.randn(R, D, device=dev, requires_grad=True) if with_b else None
        y = NL.add_layer_norm(a, b, ln)
        gy = torch.randn_like(y)
        y.backward(gy)
        got = (y.detach(), a.grad.clone(), None if b is None else b.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone())
        a2 = a.detach().clone().requires_grad_(True)
        b2 = b.detach().clone().requires_grad_(True) if with_b else None
        ln.zero_grad()
        y2 = ln(a2 if b2 is None else a2 + b2)
        y2.backward(gy)
        ref = (y2.detach(), a2.grad, None if b2 is None else b2.grad, ln.weight.grad, ln.bias.grad)
        for u, v in zip(got, ref):
            if u is not None:
                assert torch.allclose(u, v, rtol=1e-4, atol=1e-4 * (1 + float(v.abs().max()))), (R, D, float((u - v).abs().max()))
