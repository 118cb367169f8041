"""GPU parity of the point-level head kernels (csrc/heads.hip) against the library ops they replace, fp32:
tall-skinny Linear weight/bias gradients (deterministic two-stage reduction; tolerance 1e-5 relative to the result
scale: a 165k-term fp32 sum), cross entropy with ignore_index (loss 1e-6, gradient 1e-6), voxel->point gather backward
(exact up to fp32 summation order: <= 2 points per voxel here, so bit-exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("N,I,O", [(164253, 16, 20), (50000, 16, 16), (20011, 16, 3), (9000, 32, 32)])
def test_tall_linear_matches_library(dev, N, I, O):
    from d3net_amd import heads
    torch.manual_seed(N)
    lin = torch.nn.Linear(I, O).to(dev)
    x = torch.randn(N, I, device=dev)
    g = torch.randn(N, O, device=dev)
    xa = x.clone().requires_grad_(True)
    ya = heads.linear(lin, xa)
    ya.backward(g)
    gw, gb, gx = lin.weight.grad.clone(), lin.bias.grad.clone(), xa.grad.clone()
    lin.zero_grad()
    xb = x.clone().requires_grad_(True)
    yb = lin(xb)
    yb.backward(g)
    assert rel(ya, yb) < 1e-6 and rel(gx, xb.grad) < 1e-6
    ref_w = (g.double().t() @ x.double())
    assert rel(gw, ref_w) < 1e-5 and rel(gb, g.double().sum(0)) < 1e-5
    assert rel(lin.weight.grad, ref_w) < 1e-4   # the library path, for scale


def test_tall_linear_deterministic(dev):
    from d3net_amd import heads
    torch.manual_seed(0)
    lin = torch.nn.Linear(16, 20).to(dev)
    x = torch.randn(100000, 16, device=dev); g = torch.randn(100000, 20, device=dev)
    outs = []
    for _ in range(2):
        lin.zero_grad()
        heads.linear(lin, x.clone().requires_grad_(True)).backward(g)
        outs.append(lin.weight.grad.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("N,C", [(164253, 20), (10000, 20), (8192, 7)])
def test_cross_entropy_matches_library(dev, N, C):
    from d3net_amd import heads
    torch.manual_seed(C)
    z = (torch.randn(N, C, device=dev) * 3).requires_grad_(True)
    lab = torch.randint(-1, C, (N,), device=dev)       # -1 = ignore
    z2 = z.detach().clone().requires_grad_(True)
    la = heads.cross_entropy(z, lab, ignore_index=-1)
    lb = torch.nn.functional.cross_entropy(z2, lab, ignore_index=-1)
    (la * 1.7).backward(); (lb * 1.7).backward()
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    assert rel(z.grad, z2.grad) < 1e-5


def test_devoxelize_backward(dev):
    from d3net_amd import heads, pointgroup_ops as ops
    rng = np.random.default_rng(5)
    coords = torch.from_numpy(rng.integers(0, 12, size=(20000, 3))).long()
    coords = torch.cat([torch.zeros(20000, 1, dtype=torch.long), coords], 1).to(dev)
    vc, p2v, v2p = ops.voxelization_idx(coords, 1, 4)
    M = vc.size(0)
    f = torch.randn(M, 16, device=dev)
    g = torch.randn(20000, 16, device=dev)
    fa = f.clone().requires_grad_(True); fb = f.clone().requires_grad_(True)
    heads.devoxelize(fa, p2v, v2p).backward(g)
    fb[p2v.long()].backward(g)
    assert rel(fa.grad, fb.grad) < 1e-5


def test_offset_losses_match_library(dev):
    """fused offset L1 + direction loss (csrc/heads.hip) vs the reference's expression (model/pointgroup.py:397-420), fp32:
    values 1e-5 relative (165k-term sums), gradient 1e-5 of its scale"""
    from d3net_amd import heads
    torch.manual_seed(3)
    N = 120000
    pt = (torch.randn(N, 3, device=dev) * 0.2)
    pt[:50] = 0                                        # |pt| = 0 rows: the norm's subgradient is 0
    coords = torch.rand(N, 3, device=dev) * 4
    info = torch.rand(N, 12, device=dev) * 4
    ids = torch.randint(-1, 9, (N,), device=dev)
    a = pt.clone().requires_grad_(True); b = pt.clone().requires_grad_(True)
    na, da, va = heads.offset_losses(a, coords, info, ids, -1)
    heads_rows, heads.TALL_ROWS = heads.TALL_ROWS, 10 ** 9   # library path
    try:
        nb, db, vb = heads.offset_losses(b, coords, info, ids, -1)
    finally:
        heads.TALL_ROWS = heads_rows
    (1.3 * na + 0.7 * da).backward(); (1.3 * nb + 0.7 * db).backward()
    assert abs(float(na) - float(nb)) < 1e-5 * abs(float(nb)) and abs(float(da) - float(db)) < 1e-5 * abs(float(db))
    assert float(va) == float(vb)
    assert rel(a.grad, b.grad) < 1e-5


def test_gather_cluster_rows_backward(dev):
    from d3net_amd import heads
    torch.manual_seed(1)
    f = torch.randn(5000, 16, device=dev)
    idx = torch.cat([torch.randperm(5000, device=dev)[:3000], torch.randperm(5000, device=dev)[:2500]])   # <= 2 per row
    g = torch.randn(idx.numel(), 16, device=dev)
    a = f.clone().requires_grad_(True); b = f.clone().requires_grad_(True)
    heads.gather_cluster_rows(a, idx).backward(g)
    b[idx].backward(g)
    assert torch.equal(a.grad, b.grad)     # two addends per row at most: order independent, bit-exact


@pytest.mark.parametrize("P,nInst", [(21, 12), (1, 1), (700, 37)])
def test_score_loss_matches_library(dev, P, nInst):
    """Fused proposal score loss vs the reference's op sequence (ious.max(1) -> get_segmented_scores ->
    binary_cross_entropy_with_logits(...).mean(), model/pointgroup.py:436-452): loss / gradient 1e-6 relative."""
    from d3net_amd import heads
    torch.manual_seed(P)
    ious = torch.rand(P, nInst, device=dev)
    ious[::3] *= 0.2          # rows below bg, between, and above fg
    ious[1::3, 0] = 0.9
    scores = (torch.randn(P, 1, device=dev) * 3).requires_grad_(True)
    fg, bg = 0.75, 0.25
    loss, gt = heads.score_loss(scores, ious, fg, bg)
    loss.backward()
    ga = scores.grad.clone()
    s2 = scores.detach().clone().requires_grad_(True)
    gt_ref, _ = ious.max(1)
    fgm, bgm = gt_ref > fg, gt_ref < bg
    z = torch.where(~fgm & ~bgm, gt_ref * (1 / (fg - bg)) + bg / (bg - fg), fgm.float())
    ref = torch.nn.functional.binary_cross_entropy_with_logits(s2.view(-1), z, reduction="none").mean()
    ref.backward()
    assert torch.equal(gt, gt_ref)
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert rel(ga, s2.grad) < 1e-5


def test_stack_to_batch_matches_library_path(dev, monkeypatch):
    """Fused convert_stack_to_batch + object assignment vs the library-op form of the same function (reference
    model/pointgroup.py:216-263): identical tensors (pure data movement; corners in fp64 then cast), incl. scenes with
    more than K kept proposals (the overflow is dropped) and an empty scene; gradients of the two differentiable inputs."""
    from types import SimpleNamespace as NS
    from d3net_amd import heads
    from d3net_amd.pointgroup import PointGroup
    torch.manual_seed(5)
    B, K, m, P, G = 3, 16, 16, 41, 7
    bids = torch.cat([torch.zeros(22), torch.full((9,), 2.0), torch.zeros(4), torch.full((6,), 2.0)]).int().to(dev)   # scene 1 empty
    crop = torch.randn(P, 9, device=dev); crop[:, 3:6] = crop[:, 3:6].abs()
    perms = [torch.randperm(K) for _ in range(B)]
    fake = NS(cfg=NS(model=NS(max_num_proposal=K, m=m), general=NS(task="train")),
              _box_corners=PointGroup._box_corners, get_object_assignments=lambda d: PointGroup.get_object_assignments(fake, d))

    def run(fused):
        pf = torch.randn(P, m, device=dev, generator=torch.Generator(dev).manual_seed(1)).requires_grad_(True)
        sc = torch.rand(P, device=dev, generator=torch.Generator(dev).manual_seed(2)).requires_grad_(True)
        d = {"batch_offsets": list(range(B + 1)), "proposal_feats": pf * 1.0, "proposal_crop_bbox": crop,
             "proposals_batchId": bids, "proposal_objectness_scores": sc * 1.0,
             "center_label": torch.randn(B, G, 3, device=dev, generator=torch.Generator(dev).manual_seed(3))}
        if not fused:
            monkeypatch.setattr(heads, "stack_to_batch", lambda *a, **k: None)
        out = PointGroup.convert_stack_to_batch(fake, d, perms=perms)
        monkeypatch.undo()
        w1 = torch.randn(B, K, m, device=dev, generator=torch.Generator(dev).manual_seed(4))
        w2 = torch.randn(B, K, device=dev, generator=torch.Generator(dev).manual_seed(5))
        ((out["proposal_feats_batched"] * w1).sum() + (out["proposal_scores_batched"] * w2).sum()).backward()
        return out, pf.grad, sc.grad

    a, gpa, gsa = run(True)
    b, gpb, gsb = run(False)
    for k in ("proposal_feats_batched", "proposal_bbox_batched", "proposal_center_batched", "proposal_sem_cls_batched",
              "proposal_scores_batched", "proposal_batch_mask", "object_assignment"):
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    assert a["proposal_batch_mask"].sum() == K + 15 and torch.equal(gpa, gpb) and torch.equal(gsa, gsb)


def test_fused_adamw_matches_torch(dev):
    """Single-launch AdamW vs torch.optim.AdamW over 5 steps on tensors of assorted sizes (one spanning several chunks, one
    of a single element, one whose gradient tensor is replaced mid-run): parameters and moments within 1e-6 relative."""
    from d3net_amd.optim import FusedAdamW
    torch.manual_seed(11)
    shapes = [(27, 16, 32), (1,), (5000, 3), (112,), (3, 3, 3, 48, 48)]
    pa = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdamW(pa, lr=2e-3, weight_decay=0.05)
    ob = torch.optim.AdamW(pb, lr=2e-3, weight_decay=0.05)
    for it in range(5):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a)
            if a.grad is None or (it == 3 and a.numel() == 112):
                a.grad = g.clone()          # (a replaced gradient tensor: the pointer table is rebuilt)
            else:
                a.grad.copy_(g)
            b.grad = g.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert rel(a.detach(), b.detach()) < 1e-6
        assert rel(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]) < 1e-6
        assert rel(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) < 1e-6


def test_no_stale_executor_gradients_when_a_step_has_no_proposals(dev):
    """zero_grad() only marks the executors' flat gradient buffers stale; a following step whose ScoreNet backward
    never runs (no proposals) must not re-apply the previous step's ScoreNet gradient.  torch semantics: those parameters
    have grad None, the optimizer leaves them and their moments untouched."""
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.optim import FusedAdamW
    from d3net_amd.pointgroup import PointGroup
    cfg = default_conf(overrides={"model": {"blocks": [1, 2, 3]}})
    torch.manual_seed(0)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = True
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2, weight_decay=1e-2)
    opt.register_step_pre_hook(lambda *a: model.drop_stale_grads())
    scene = S.small_scene(dims=(40, 32, 20), n_boxes=2, seed=3)

    def step():
        model.zero_grad(set_to_none=True)
        loss, d = model.training_step(S.make_batch([scene], dev))
        loss.backward()
        opt.step()
        return d

    d = step()
    assert d["proposal_scores"][2].numel() - 1 > 0
    sn = {k: p.detach().clone() for k, p in model.score_net.named_parameters()}
    m1 = {k: opt.state[p]["exp_avg"].clone() for k, p in model.score_net.named_parameters()}
    bb = model.backbone[1].blocks.block0.conv_branch[2].kernel.detach().clone()
    assert any(float(v.abs().sum()) > 0 for v in m1.values())          # the ScoreNet did train in step 1
    model.cluster_npoint_thre = 10 ** 9                                  # no cluster survives: the _no_proposals path
    d = step()
    assert d["proposal_scores"][2].numel() - 1 == 0
    torch.cuda.synchronize()
    for k, p in model.score_net.named_parameters():
        assert torch.equal(p.detach(), sn[k]), k                         # untouched: no gradient, no weight decay, no moment update
        assert torch.equal(opt.state[p]["exp_avg"], m1[k]), k
        assert p.grad is None, k
    assert not torch.equal(model.backbone[1].blocks.block0.conv_branch[2].kernel.detach(), bb)   # the backbone still trains
    model.cluster_npoint_thre = cfg.cluster.cluster_npoint_thre
    d = step()                                                           # and the ScoreNet resumes afterwards
    assert d["proposal_scores"][2].numel() - 1 > 0
    assert any(not torch.equal(p.detach(), sn[k]) for k, p in model.score_net.named_parameters())


def test_fused_adamw_follows_load_state_dict(dev):
    """load_state_dict() replaces the moment tensors -- the cached device pointer table must be rebuilt, and the
    step count must come from the checkpoint (torch.optim.AdamW stores it per parameter)"""
    from d3net_amd.optim import FusedAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(33, 7, device=dev)), torch.nn.Parameter(torch.randn(5, device=dev))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a, b = FusedAdamW(ps, lr=1e-2, weight_decay=1e-2), torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    for it in range(3):
        for p, q in zip(ps, ref):
            g = torch.randn_like(p); p.grad = g.clone(); q.grad = g.clone()
        a.step(); b.step()
    # restart the fused optimizer from the library optimizer's checkpoint
    a2 = FusedAdamW(ps, lr=1e-2, weight_decay=1e-2)
    for p, q in zip(ps, ref):
        p.grad = torch.zeros_like(p)
    a2.step()                                                            # builds a table over fresh (zero) moments
    with torch.no_grad():
        for p, q in zip(ps, ref):
            p.copy_(q)
    import copy
    a2.load_state_dict(copy.deepcopy(b.state_dict()))      # (load_state_dict keeps same-device tensors by reference)
    for it in range(2):
        for p, q in zip(ps, ref):
            g = torch.randn_like(p); p.grad = g.clone(); q.grad = g.clone()
        a2.step(); b.step()
    for p, q in zip(ps, ref):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)


def test_fused_adamw_per_parameter_step_counts_and_checkpoint_round_trip(dev):
    """torch.optim.AdamW bias-corrects PER PARAMETER.  A tensor that starts receiving gradients late (ScoreNet /
    score_linear once `epoch > prepare_epochs`, model/pointgroup.py:332) must start at t = 1: its first updates are ~lr in
    magnitude, not ~lr/3.  And the fused optimizer's state_dict loads into torch.optim.AdamW (and back)."""
    from d3net_amd.optim import FusedAdamW
    torch.manual_seed(3)
    shapes = [(40, 9), (4100,), (7,), (3, 5)]
    late = {1, 3}                                              # these tensors get their first gradient at step 4
    pa = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = FusedAdamW(pa, lr=3e-3, weight_decay=0.02), torch.optim.AdamW(pb, lr=3e-3, weight_decay=0.02)

    def grads(it):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i in late and it < 4:
                a.grad = None; b.grad = None
            else:
                g = torch.randn_like(a); a.grad = g.clone(); b.grad = g.clone()

    for it in range(1, 8):
        grads(it)
        before = [p.detach().clone() for p in pa]
        oa.step(); ob.step()
        if it == 4:                                            # first update of a late tensor: |delta| ~ lr (bias correction at t = 1)
            d = (pa[1].detach() - before[1] * (1 - 3e-3 * 0.02)).abs()
            assert 0.9 * 3e-3 < float(d.median()) < 1.1 * 3e-3, float(d.median())
    for a, b in zip(pa, pb):
        assert rel(a.detach(), b.detach()) < 1e-6
        assert rel(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) < 1e-6
    sd = oa.state_dict()
    steps = [float(sd["state"][i]["step"]) for i in range(len(shapes))]
    assert steps == [7.0, 4.0, 7.0, 4.0], steps
    # fused -> torch.optim.AdamW -> two more steps on both -> still the same
    import copy
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oc = torch.optim.AdamW(pc, lr=3e-3, weight_decay=0.02)
    oc.load_state_dict(copy.deepcopy(sd))
    for it in range(8, 10):
        for a, c in zip(pa, pc):
            g = torch.randn_like(a); a.grad = g.clone(); c.grad = g.clone()
        oa.step(); oc.step()
    for a, c in zip(pa, pc):
        assert rel(a.detach(), c.detach()) < 1e-6
    # torch -> fused
    od = FusedAdamW(pa, lr=3e-3, weight_decay=0.02)
    with torch.no_grad():
        for a, c in zip(pa, pc):
            a.copy_(c)
    od.load_state_dict(copy.deepcopy(oc.state_dict()))
    for a, c in zip(pa, pc):
        g = torch.randn_like(a); a.grad = g.clone(); c.grad = g.clone()
    od.step(); oc.step()
    for a, c in zip(pa, pc):
        assert rel(a.detach(), c.detach()) < 1e-6
    sd2 = od.state_dict()["state"]
    assert [float(sd2[i]["step"]) for i in range(len(shapes))] == [10.0, 7.0, 10.0, 7.0]


def test_orientation_loss_one_launch_matches_library_form_and_reference(dev):
    """d3_orientation_loss (csrc/heads.hip) against the library-op form of compute_node_orientation_loss and the reference's own
    value (tests/golden/speaker_golden.npz ori/*, lib/captioning/loss_helper.py:244-307): loss, accuracy, gradient of the
    logits; then on rotations by exact multiples of 90 degrees (angles ON the bin boundaries), padded edge slots, a scene
    without live edges, and through the strided view of the (B, E, bins + 1) edge predictions the model hands over."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import gen_speaker_golden as G
    from d3net_amd.captioning_loss import compute_node_orientation_loss
    g = np.load(os.path.join(here, "golden", "speaker_golden.npz"))

    def both(inp):
        outs = []
        for native in (True, False):
            d = {k: v.clone() for k, v in inp.items()}
            full = d.pop("edge_preds_full").requires_grad_(True)
            d["edge_orientations"] = full[:, :, :-1]
            loss, acc = compute_node_orientation_loss(d, 6, native=native)
            (loss * 1.7).backward()
            outs.append((float(loss), float(acc), full.grad.clone()))
        return outs

    inp = {k: torch.from_numpy(v).to(dev) for k, v in G.orientation_inputs().items()}
    eo = inp.pop("edge_orientations")
    inp["edge_preds_full"] = torch.cat([eo, torch.randn(eo.shape[0], eo.shape[1], 1, device=dev)], -1)
    (ln, an, gn), (ll, al, gl) = both(inp)
    assert abs(ln - float(g["ori/loss"])) < 1e-5 and abs(an - float(g["ori/acc"])) < 1e-6
    assert abs(ln - ll) < 1e-5 and abs(an - al) < 1e-6
    assert torch.allclose(gn, gl, atol=1e-7, rtol=1e-4) and float(gn[:, :, -1].abs().max()) == 0.0
    # boundary angles: rotations about z by k * 30 degrees (cos exactly representable only for some) and exact quarter turns
    rng = np.random.default_rng(3)
    B, K, L, Gn = 3, 256, 10, 128
    E = K * L

    def rotz(a):
        c, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
        return np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)

    quarter = [np.array(m, np.float32) for m in ([[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, -1, 0], [1, 0, 0], [0, 0, 1]],
                                                 [[-1, 0, 0], [0, -1, 0], [0, 0, 1]], [[0, 1, 0], [-1, 0, 0], [0, 0, 1]])]
    rots = np.stack([np.stack([quarter[rng.integers(4)] if rng.random() < 0.5 else rotz(rng.integers(12) * np.pi / 6)
                               for _ in range(Gn)]) for _ in range(B)])
    ei = np.zeros((B, 2, E), np.float32)
    nsrc, ntar = np.array([200, 0, 37], np.int64), np.array([10, 0, 9], np.int64)
    for b in range(B):
        n = int(nsrc[b] * ntar[b])
        ei[b, :, :n] = rng.integers(0, K, (2, n))
    inp2 = dict(object_assignment=torch.from_numpy(rng.integers(0, Gn, (B, K)).astype(np.int64)).to(dev),
                edge_index=torch.from_numpy(ei).to(dev), num_edge_source=torch.from_numpy(nsrc).to(dev),
                num_edge_target=torch.from_numpy(ntar).to(dev), scene_object_rotations=torch.from_numpy(rots).to(dev),
                scene_object_rotation_masks=torch.from_numpy((rng.random((B, Gn)) > 0.3).astype(np.float32)).to(dev),
                edge_preds_full=torch.randn(B, E, 7, device=dev))
    (ln, an, gn), (ll, al, gl) = both(inp2)
    assert abs(ln - ll) < 2e-5 * max(1.0, abs(ll)) and abs(an - al) < 1e-5, (ln, ll, an, al)
    assert torch.allclose(gn, gl, atol=1e-7, rtol=1e-4)
    # no live edge anywhere: loss 0, accuracy 0, zero gradient
    inp3 = dict(inp2, num_edge_source=torch.zeros(B, dtype=torch.long, device=dev))
    (ln, an, gn), (ll, al, gl) = both(inp3)
    assert ln == 0.0 and an == 0.0 and ll == 0.0 and float(gn.abs().max()) == 0.0


def test_caption_cross_entropy_two_launches_matches_library_form(dev):
    """d3_masked_xe against the library-op form of compute_cap_loss (lib/captioning/loss_helper.py:177-224): loss, word accuracy
    and the gradient of the logits at the config's shape (32 descriptions x 29 steps x V = 3004), with descriptions whose box is
    not good, padded (0) targets, an exactly tied row maximum, and the all-bad batch (loss 0, zero gradient)."""
    from d3net_amd.captioning_loss import compute_cap_loss
    torch.manual_seed(4)
    N, S, V, ML = 32, 29, 3004, 32
    ids = torch.randint(1, V, (N, ML), device=dev)
    lens = torch.randint(3, S + 1, (N,), device=dev)
    ids = torch.where(torch.arange(ML, device=dev).view(1, -1) <= lens.view(-1, 1), ids, torch.zeros_like(ids))
    good = torch.rand(N, device=dev) > 0.3
    logits = torch.randn(N, S, V, device=dev)
    logits[0, 0, 7] = logits[0, 0, 1900] = 9.0          # tie: the first index wins in both forms
    ids[0, 1] = 7

    def run(native, good):
        p = logits.clone().requires_grad_(True)
        d = dict(lang_cap=p, lang_ids=ids, good_bbox_masks=good, bbox_feature=logits)
        loss, d = compute_cap_loss(d, {"use_rl": False, "max_len": ML}, native=native)
        (loss * 0.6).backward()
        return float(loss), float(d["cap_acc"]), p.grad

    (ln, an, gn), (ll, al, gl) = run(True, good), run(False, good)
    assert abs(ln - ll) < 2e-6 * abs(ll) and abs(an - al) < 1e-6, (ln, ll, an, al)
    assert torch.allclose(gn, gl, atol=1e-9, rtol=1e-4)
    bad = torch.zeros_like(good)
    (ln, an, gn), (ll, al, gl) = run(True, bad), run(False, bad)
    assert ln == 0.0 and ll == 0.0 and an == 0.0 and float(gn.abs().max()) == 0.0


@pytest.mark.parametrize("N,C", [(164253, 20), (20001, 7), (8200, 32)])
def test_point_heads_three_launches_match_the_module_form(dev, N, C):
    """d3_point_heads_fwd (semantic scores + arg-max + Linear-BN-ReLU-Linear offsets, model/pointgroup.py:77-85, 277-283) against
    the nn.Module form: outputs, predictions, running statistics / num_batches_tracked, gradients of every parameter and of the
    input; then eval mode (running statistics) and a no-grad call."""
    import copy
    from d3net_amd import heads
    torch.manual_seed(N)
    m = 16
    sem = torch.nn.Linear(m, C).to(dev)
    off = torch.nn.Sequential(torch.nn.Linear(m, m), torch.nn.BatchNorm1d(m, eps=1e-4, momentum=0.1), torch.nn.ReLU(),
                              torch.nn.Linear(m, 3)).to(dev)
    with torch.no_grad():
        off[1].weight.uniform_(0.5, 1.5); off[1].bias.uniform_(-0.3, 0.3)
    sem2, off2 = copy.deepcopy(sem), copy.deepcopy(off)
    x0 = torch.randn(N, m, device=dev) * 1.3 + 0.2
    gs, go = torch.randn(N, C, device=dev), torch.randn(N, 3, device=dev)
    for training in (True, False):
        for mod in (off, off2):
            mod.train(training)
        xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        s1, p1, o1 = heads.point_heads(sem, off, xa)
        s2 = sem2(xb); o2 = off2(xb); p2 = s2.max(1)[1]
        assert rel(s1, s2) < 1e-5 and rel(o1, o2) < 2e-5
        same = (p1 == p2).float().mean()
        assert float(same) > 0.9999, float(same)         # near-ties may flip with the summation order
        assert rel(off[1].running_mean, off2[1].running_mean) < 1e-5 and rel(off[1].running_var, off2[1].running_var) < 1e-5
        assert int(off[1].num_batches_tracked) == int(off2[1].num_batches_tracked)
        ((s1 * gs).sum() + (o1 * go).sum()).backward()
        ((s2 * gs).sum() + (o2 * go).sum()).backward()
        assert rel(xa.grad, xb.grad) < 1e-4
        for (n1, q1), (n2, q2) in zip(list(sem.named_parameters()) + list(off.named_parameters()),
                                      list(sem2.named_parameters()) + list(off2.named_parameters())):
            if n1 == "0.bias" and training:      # the batch norm removes the mean: this gradient is exactly 0, both are rounding noise
                assert float(q1.grad.abs().max()) < 5e-3 and float(q2.grad.abs().max()) < 5e-3
            else:
                assert rel(q1.grad, q2.grad) < 2e-4, (n1, rel(q1.grad, q2.grad))
            q1.grad = None; q2.grad = None
    with torch.no_grad():
        s3, p3, o3 = heads.point_heads(sem, off, x0)
    assert rel(s3, sem2(x0)) < 1e-5 and rel(o3, off2(x0)) < 2e-5
