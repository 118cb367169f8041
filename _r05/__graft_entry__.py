"""Driver entry points: build() compiles every native piece; smoke() runs one small detector step on cuda:0
and checks the HIP path against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """hipcc --offload-arch=gfx950 for libd3hip.so (cross-compiles without a GPU) + the C oracle (checker)."""
    from d3net_amd import build as b
    so = b.build()
    assert os.path.exists(so)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    # the reference's own CPU natives need <google/dense_hash_map> / <THC/THC.h>: unbuildable here, no oracle/_ref
    from d3net_amd import _lib
    assert _lib.lib().d3_arch() == b"gfx950"
    import d3net_amd.pointgroup  # noqa: F401  (imports the package)


def smoke():
    """One small PointGroup training step (forward + loss + backward) on cuda:0, checked against the oracle."""
    import numpy as np
    import torch
    from d3net_amd import synthetic as S, minkowski as ME
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    from oracle.pointgroup_oracle import PointGroupOracle

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda", 0)
    cfg = default_conf(overrides={"model": {"blocks": [1, 2, 3]}})
    torch.manual_seed(0)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = True
    scene = S.small_scene(dims=(40, 32, 20), n_boxes=2, seed=3)
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]

    ME.set_exact(True)  # fp32 kernels: the clustering inputs then match the oracle bit for bit
    try:
        batch = S.make_batch([scene], dev)
        batch["cluster_rand"], batch["slot_perms"] = rand, perms
        loss, d = model.training_step(batch)
        loss.backward()
    finally:
        ME.set_exact(False)
    torch.cuda.synchronize()

    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in S.make_batch([scene], dev).items()}
    orc = PointGroupOracle(cfg, model.state_dict())
    orc.teacher = True
    od = orc.loss(orc.feed(cpu, 0, rand=rand, perms=perms))
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2]), "cluster offsets differ"
    rel = abs(float(loss) - float(od["total_loss"])) / abs(float(od["total_loss"]))
    assert rel < 1e-3, ("loss mismatch", float(loss), float(od["total_loss"]))

    # and one step on the bf16-MFMA path (the one bench.py times)
    batch = S.make_batch([scene], dev)
    loss2, _ = model.training_step(batch)
    loss2.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss2)
    print("smoke ok: loss %.6f (oracle %.6f), bf16 loss %.6f, proposals %d" %
          (float(loss), float(od["total_loss"]), float(loss2), d["proposal_scores"][2].numel() - 1))


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
