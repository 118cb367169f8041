"""BASELINE configs[0] ("conf/pointgroup.yaml, 1 synthetic 50k-point scene") as a parity case: the reference runs it on its
CPU backend as a plumbing check; here the same configuration runs on the device and is compared with the CPU oracle's step.

  * exact-fp32 HIP step == oracle step: identical proposals_idx / proposals_offset, every loss term within 1e-3
    (reference step: model/pointgroup.py:466-479,266-370,387-463 driven by conf/pointgroup.yaml);
  * the default (bf16 MFMA operands, native executor) step: identical clusters, point-wise scores / offsets within 3e-2.
The product has no CPU path by design; the CPU side of this configuration is the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _l2(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def config0(dev):
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    from oracle.pointgroup_oracle import PointGroupOracle
    cfg = default_conf()                       # conf/pointgroup.yaml
    torch.manual_seed(cfg.general.manual_seed)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = True
    occ, sem, inst, _ = S.occupancy_grid((120, 90, 60), 6, (10, 34), (10, 28), 0)
    scene = S.scene_from_grid(occ, sem, inst)
    assert 45000 <= scene["locs"].shape[0] <= 60000      # the 50k-point scene of configs[0]
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in S.make_batch([scene], dev).items()}
    orc = PointGroupOracle(cfg, model.state_dict())
    orc.teacher = True
    od = orc.loss(orc.feed(cpu, 0, rand=rand, perms=perms))
    return dict(model=model, scene=scene, rand=rand, perms=perms, od=od)


def _step(c, dev, exact):
    from d3net_amd import synthetic as S, minkowski as ME
    c["model"].zero_grad(set_to_none=True)
    ME.set_exact(exact)
    try:
        batch = S.make_batch([c["scene"]], dev)
        batch["cluster_rand"], batch["slot_perms"] = c["rand"], c["perms"]
        loss, d = c["model"].training_step(batch)
        loss.backward()
    finally:
        ME.set_exact(False)
    torch.cuda.synchronize()
    return loss, d


@pytest.mark.parametrize("exact", [True, False])
def test_config0_step_against_oracle(dev, config0, exact):
    od = config0["od"]
    loss, d = _step(config0, dev, exact)
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2]), "cluster offsets differ"
    assert d["proposal_scores"][2].numel() - 1 >= 8
    tol = 1e-3 if exact else 3e-2
    assert _l2(d["semantic_scores"][0], od["semantic_scores"]) < tol
    assert _l2(d["pt_offsets"][0], od["pt_offsets"]) < tol
    rel = abs(float(loss.detach()) - float(od["total_loss"].detach())) / abs(float(od["total_loss"].detach()))
    assert rel < (1e-3 if exact else 2e-2), (float(loss.detach()), float(od["total_loss"].detach()))
    if exact:
        for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss", "score_loss"):
            a, b = float(d[k][0]), float(od[k])
            assert abs(a - b) <= 1e-3 * abs(b) + 1e-6, (k, a, b)
        assert torch.equal(d["object_assignment"].cpu(), od["object_assignment"])
    grads = [p.grad for p in config0["model"].parameters() if p.grad is not None]
    assert grads and all(bool(torch.isfinite(g).all()) for g in grads)
