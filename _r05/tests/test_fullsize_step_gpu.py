"""BASELINE configs[1] at FULL size under pytest: one canonical-scene detector step (142,920 voxels, 164 k points, the
7-level backbone) against the CPU oracle, and the native executor's fused epilogues at canonical row counts.

  * exact-fp32 HIP step == oracle step: identical proposals_idx / proposals_offset (bit-exact clustering + indexing at
    full size), total loss within 1e-3 (reference step: model/pointgroup.py:466-479,266-370,387-463);
  * the bf16 / native-executor step (what bench.py times) against the same oracle step: point-wise semantic scores and
    offsets within 3e-2 relative L2, loss within 2e-2; parameter gradients of the whole 7-level backbone agree in
    direction (cosine) -- end-to-end bf16 gradients decorrelate through ~70 ReLU layers (tests/test_sparse_gpu.py explains);
  * (every executor op with its fused epilogues is pinned separately, op by op on the executor's own activations:
    tests/test_executor_ops_gpu.py).
The oracle step costs ~15-60 s of CPU, once per module.
"""
import functools

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so

pytestmark = pytest.mark.gpu


def l2err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cos(a, b):
    a = a.detach().cpu().double().flatten(); b = b.detach().cpu().double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _degenerate(name, g):
    """parameters whose exact gradient is zero (a Linear bias directly in front of a BatchNorm: the normalisation removes
    any shift) -- what is left is rounding noise, not comparable"""
    return name == "offset_net.0.bias" or float(g.norm()) == 0.0


@pytest.fixture(scope="module")
def canonical(dev):
    """the benchmark's model + scene, and ONE oracle step (forward + loss + backward) on the host"""
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg = default_conf()
    torch.manual_seed(cfg.general.manual_seed)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = True
    occ, sem, inst, _ = S.occupancy_grid()
    scene = S.scene_from_grid(occ, sem, inst)
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in S.make_batch([scene], dev).items()}
    assert cpu["voxel_locs"].shape[0] == 142920
    torch.set_num_threads(min(16, torch.get_num_threads()))
    orc = PointGroupOracle(cfg, model.state_dict())
    orc.teacher = True
    od = orc.loss(orc.feed(cpu, 0, rand=rand, perms=perms))
    od["total_loss"].backward()
    return dict(cfg=cfg, model=model, scene=scene, rand=rand, perms=perms, orc=orc, od=od)


def _hip_step(c, dev, exact):
    from d3net_amd import synthetic as S, minkowski as ME
    model = c["model"]
    model.zero_grad(set_to_none=True)
    ME.set_exact(exact)
    try:
        batch = S.make_batch([c["scene"]], dev)
        batch["cluster_rand"], batch["slot_perms"] = c["rand"], c["perms"]
        loss, d = model.training_step(batch)
        loss.backward()
    finally:
        ME.set_exact(False)
    torch.cuda.synchronize()
    return loss, d


def test_canonical_step_exact_mode_equals_oracle(dev, canonical):
    c = canonical
    od = c["od"]
    loss, d = _hip_step(c, dev, exact=True)
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2]), "cluster offsets differ"
    assert d["proposal_scores"][2].numel() - 1 >= 16          # both clustering branches carry real load
    rel = abs(float(loss) - float(od["total_loss"])) / abs(float(od["total_loss"]))
    assert rel < 1e-3, (float(loss), float(od["total_loss"]))
    assert l2err(d["semantic_scores"][0], od["semantic_scores"]) < 1e-3
    assert l2err(d["pt_offsets"][0], od["pt_offsets"]) < 1e-3
    for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss", "score_loss"):
        a, b = float(d[k][0]), float(od[k])
        assert abs(a - b) <= 1e-3 * abs(b) + 1e-6, (k, a, b)
    # batched proposal tensors (convert_stack_to_batch with the injected permutation)
    for k in ("proposal_bbox_batched", "proposal_center_batched", "proposal_batch_mask", "proposal_sem_cls_batched"):
        assert torch.allclose(d[k].cpu().float(), od[k].float(), atol=1e-4), k
    assert torch.equal(d["object_assignment"].cpu(), od["object_assignment"])
    # parameter gradients of the exact path against autograd through the oracle (fp32 vs fp32: relative L2)
    # (fp32 vs fp32 through ~70 ReLU layers: a pre-activation within rounding of 0 flips its mask and moves single
    # gradient entries by O(1) -- tests/test_sparse_gpu.py -- so the bound is a relative L2 per tensor, measured 1.2e-2
    # median on MI355X; tensors whose true gradient is zero up to rounding -- a bias in front of a BatchNorm -- are skipped)
    errs = {}
    for n, p in c["model"].named_parameters():
        g = c["orc"].p[n].grad
        if p.grad is not None and g is not None and not _degenerate(n, g):
            errs[n] = l2err(p.grad, g)
    worst = max(errs, key=errs.get)
    vals = sorted(errs.values())
    print("exact step vs oracle: parameter-gradient rel-L2 median %.2e, 90%% %.2e, worst %.2e (%s)" %
          (vals[len(vals) // 2], vals[len(vals) * 9 // 10], errs[worst], worst))
    # (fp32 MFMA executor, MI355X: median 8.0e-3, 90 % 1.0e-2, worst 1.7e-2)
    assert vals[len(vals) // 2] < 2e-2, vals[len(vals) // 2]
    assert vals[len(vals) * 9 // 10] < 5e-2, vals[len(vals) * 9 // 10]


def test_canonical_step_exact_executor_equals_exact_module_path(dev, canonical):
    """minkowski.set_exact(True) through the NATIVE executor (fp32 program: csrc/unet.hip with the D3_CONV_F32 kernels) against the
    same step run module by module with the same kernels: one schedule, two drivers -- loss to 1e-5, parameter gradients to 2e-3 median / 2e-2 worst relative L2 (the
    fused epilogues change the summation order of the BatchNorm reductions; a mask flip at a ReLU moves single entries)"""
    c = canonical
    model = c["model"]
    loss_e, _ = _hip_step(c, dev, exact=True)
    assert model._execs.get("backbone/f32") is not None, "the reference-precision executor did not run"
    g_exec = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():      # (the executor's gradient views are only marked stale by zero_grad(): autograd would add to them)
        p.grad = None
    model.native_exact = False
    try:
        loss_m, _ = _hip_step(c, dev, exact=True)
    finally:
        model.native_exact = True
    assert abs(float(loss_e) - float(loss_m)) <= 1e-5 * abs(float(loss_m)), (float(loss_e), float(loss_m))
    errs = {n: l2err(g_exec[n], p.grad) for n, p in model.named_parameters() if p.grad is not None and not _degenerate(n, p.grad)}
    worst = max(errs, key=errs.get)
    vals = sorted(errs.values())
    print("exact executor vs exact module path: gradient rel-L2 median %.2e, worst %.2e (%s)" % (vals[len(vals) // 2], errs[worst], worst))
    # (measured on MI355X: median 3.7e-4, worst 2.0e-3 -- fp32 against fp32 with different reduction orders through ~70 ReLU layers)
    assert vals[len(vals) // 2] < 2e-3 and errs[worst] < 2e-2, (vals[len(vals) // 2], worst, errs[worst])
    # the bf16 step bench.py times, bounded against THIS path (same kernels' structure, same schedule, fp32 operands): what bf16
    # MFMA operands alone change.  Direction (cosine) and size (norm ratio) per tensor; ~70 ReLU layers decorrelate single entries,
    # so the bound is statistical: median cosine > 0.7, 10th percentile > 0.3, none pointing backwards, norms within 2x
    # (measured on MI355X: median cosine 0.81, norm ratio median 1.00, extremes 0.7 .. 1.4).
    for p in model.parameters():
        p.grad = None
    _hip_step(c, dev, exact=False)
    cs, ratio = {}, {}
    for n, p in model.named_parameters():
        if p.grad is not None and n in g_exec and not _degenerate(n, g_exec[n]):
            cs[n] = cos(p.grad, g_exec[n])
            ratio[n] = float(p.grad.double().norm() / (g_exec[n].double().norm() + 1e-30))
    cv, rv = sorted(cs.values()), sorted(ratio.values())
    print("bf16 executor vs fp32-MFMA executor: grad cosine median %.4f 10%% %.4f worst %.4f; norm ratio median %.3f min %.3f max %.3f" %
          (cv[len(cv) // 2], cv[len(cv) // 10], cv[0], rv[len(rv) // 2], rv[0], rv[-1]))
    assert cv[len(cv) // 2] > 0.7 and cv[len(cv) // 10] > 0.3 and cv[0] > 0.0, (cv[len(cv) // 2], cv[len(cv) // 10], cv[0])
    assert 0.8 < rv[len(rv) // 2] < 1.25 and rv[0] > 0.4 and rv[-1] < 2.5, (rv[0], rv[len(rv) // 2], rv[-1])


def test_canonical_step_bf16_executor_close_to_oracle(dev, canonical):
    """the step bench.py times (bf16 MFMA operands, native executor) against the fp32 oracle step"""
    c = canonical
    od = c["od"]
    assert c["model"].native_unet
    loss, d = _hip_step(c, dev, exact=False)
    assert c["model"]._execs.get("backbone") is not None, "the native executor did not run"
    # clustering is driven by the labels (teacher): the integer results must not depend on the precision of the backbone
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1])
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2])
    e_sem, e_off = l2err(d["semantic_scores"][0], od["semantic_scores"]), l2err(d["pt_offsets"][0], od["pt_offsets"])
    assert e_sem < 3e-2 and e_off < 3e-2, (e_sem, e_off)
    rel = abs(float(loss) - float(od["total_loss"])) / abs(float(od["total_loss"]))
    assert rel < 2e-2, (float(loss), float(od["total_loss"]))
    # End-to-end bf16 gradients are only statistically comparable with fp32 ones on this net: a 2 % forward difference
    # flips ~1 % of the ReLU masks per layer and the backward passes ~70 of them (the per-kernel arithmetic is pinned to 1e-4
    # in tests/test_conv_fullsize_gpu.py and every executor op, with its fused epilogues, at 1e-3 in
    # tests/test_executor_ops_gpu.py).  Measured on MI355X: median cosine 0.81.
    cs = {}
    for n, p in c["model"].named_parameters():
        g = c["orc"].p[n].grad
        if p.grad is not None and g is not None and not _degenerate(n, g):
            cs[n] = cos(p.grad, g)
    vals = sorted(cs.values())
    worst = min(cs, key=cs.get)
    print("bf16 executor vs fp32 oracle: fwd rel-L2 %.2e / %.2e, loss rel %.2e, grad cosine median %.4f 10%% %.4f worst %.4f (%s)" %
          (e_sem, e_off, rel, vals[len(vals) // 2], vals[len(vals) // 10], cs[worst], worst))
    assert vals[len(vals) // 2] > 0.7, vals[len(vals) // 2]
    assert vals[len(vals) // 10] > 0.3, vals[len(vals) // 10]
    assert cs[worst] > 0.0, (worst, cs[worst])        # no tensor points the wrong way
