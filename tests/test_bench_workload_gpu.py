"""The benchmark's OWN workload under pytest (BASELINE configs[2]: PipelineNet mode 1 on the 40-box synthetic ScanNet scenes,
V = 3004, 8 descriptions per scene -- `bench.make_scenes("speaker", 0)`), against the CPU oracle chain that `bench.py`'s
`cpu_baseline` leg times (PointGroupOracle -> speaker_oracle.graph_module -> speaker_oracle.forward_sample_batch):

  (a) ONE bench scene (~188 k points, ~160 k voxels, ~107 proposals) through `PipelineNet.training_step` with the exact-fp32
      kernels vs the oracle chain: identical `proposals_idx / proposals_offset` and `object_assignment`, batched proposal
      tensors, detector loss terms and the caption loss to 1e-3, teacher-forced `lang_cap` logits to rtol 1e-3, and the
      bf16 step bench.py times beside it (same integer outputs; losses within 2e-2);
  (b) the 4-scene batch bench.py steps (649 k voxels, batch ids 0..3 in the hash keys, the row-split plans of that size)
      against the four single-scene runs: every integer output of the detector (voxel coordinates / maps, cluster
      membership in BFS order, cluster offsets, batch ids, point counts) is bit-equal to the per-scene results put
      together the way the reference's merge does (model/pointgroup.py:299-316);
  (c) the level-0 convolution kernels (both generations: lane-table and dense-table) at the 649 k rows of that batch against oracle/sparse_oracle.py (the canonical
      143 k-row cases are in tests/test_conv_fullsize_gpu.py).
Reference step: model/pipeline.py:152-185, model/pointgroup.py:266-370,466-479, model/caption_module.py:510-687,
lib/captioning/loss_helper.py:177-224.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def l2err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def bench_setup(dev):
    """bench.py's model, dataset and scenes (rank 0, config "speaker")"""
    import bench
    from d3net_amd.config import default_conf
    from d3net_amd.pipeline import PipelineNet
    cfg = default_conf(bench.CONF["speaker"])
    torch.manual_seed(cfg.general.manual_seed)
    scenes = bench.make_scenes("speaker", 0)
    net = PipelineNet(cfg, bench.make_dataset(len(scenes), cfg.data.num_des_per_scene, False)).to(dev).train()
    net.detector.teacher = True
    return dict(cfg=cfg, net=net, scenes=scenes, vocab=bench.VOCAB)


def _lang(batch, dev, cfg, vocab):
    from d3net_amd import synthetic as S
    batch = S.add_language(batch, dev, chunk=cfg.data.num_des_per_scene, vocab=vocab)
    batch["lang_len"] = batch["spk_lang_len"]
    return batch


def test_one_bench_scene_pipeline_step_equals_oracle_chain(dev, bench_setup):
    from d3net_amd import synthetic as S, minkowski as ME
    from oracle import speaker_oracle as spo
    from oracle.pointgroup_oracle import PointGroupOracle
    c = bench_setup
    cfg, net = c["cfg"], c["net"]
    scene = c["scenes"][0]
    rand = torch.rand(2, 3)
    perms = [torch.randperm(cfg.model.max_num_proposal)]

    # ---- oracle chain (what bench.py's cpu_baseline leg runs), fp32 on the host
    host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in _lang(S.make_batch([scene], dev), dev, cfg, c["vocab"]).items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    orc = PointGroupOracle(cfg, net.detector.state_dict())
    orc.teacher = True
    od = orc.loss(orc.feed(host, 0, rand=rand, perms=perms))
    spk = {k: v.detach().cpu().clone() for k, v in net.speaker.state_dict().items()}
    spo.TIE_RULE = "index"          # (ties of the neighbour selection: see the relation-graph checks below)
    try:
        with torch.no_grad():
            g = spo.graph_module({k[len("graph."):]: v for k, v in spk.items() if k.startswith("graph.")}, od, cfg.model.num_graph_steps,
                                 cfg.model.num_locals)
            od.update(g)
            out = spo.forward_sample_batch({k[len("caption."):]: v for k, v in spk.items() if k.startswith("caption.")}, od, cfg,
                                           cfg.model.max_num_proposal, cfg.model.num_locals)
    finally:
        spo.TIE_RULE = "topk"
    ologits, good = out["lang_cap"], out["good"]
    tgt = host["lang_ids"].reshape(-1, cfg.data.max_spk_len + 2)[:, 1:ologits.shape[1] + 1]
    assert bool(good.any()), "no description refers to a detected box: the caption loss would be vacuous"
    ocap = F.cross_entropy(ologits[good].reshape(-1, ologits.shape[-1]), tgt[good].reshape(-1), ignore_index=0)
    n_prop = int(od["proposal_batch_mask"].sum())
    assert n_prop >= 30, n_prop          # ~40 boxes per scene, two clustering branches

    def hip_step(exact):
        net.zero_grad(set_to_none=True)
        ME.set_exact(exact)
        try:
            batch = _lang(S.make_batch([scene], dev), dev, cfg, c["vocab"])
            batch["cluster_rand"], batch["slot_perms"] = rand, perms
            loss, d = net.training_step(batch)
            loss.backward()
        finally:
            ME.set_exact(False)
        torch.cuda.synchronize()
        return loss, d

    # ---- exact-fp32 kernels: everything must agree with the oracle
    loss, d = hip_step(True)
    assert np.array_equal(d["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1]), "cluster membership differs"
    assert np.array_equal(d["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2]), "cluster offsets differ"
    assert torch.equal(d["object_assignment"].cpu(), od["object_assignment"])
    assert torch.equal(d["proposal_batch_mask"].cpu(), od["proposal_batch_mask"])
    for k in ("proposal_bbox_batched", "proposal_center_batched", "proposal_sem_cls_batched"):
        assert torch.allclose(d[k].cpu().float(), od[k].float(), atol=1e-4), k
    assert l2err(d["proposal_feats_batched"], od["proposal_feats_batched"]) < 1e-3
    for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss", "score_loss"):
        a, b = float(d[k][0]), float(od[k])
        assert abs(a - b) <= 1e-3 * abs(b) + 1e-6, (k, a, b)
    a, b = float(d["total_loss"][0]), float(od["total_loss"])
    assert abs(a - b) <= 1e-3 * abs(b), ("detector loss", a, b)
    assert torch.equal(d["good_bbox_masks"].cpu(), good)
    assert torch.equal(d["assigned_bbox_id_labels"].cpu(), out["assigned"])
    # relation graph.  The two clustering branches find most objects twice -- identical boxes, exactly tied distances -- and
    # padded slots all sit at 1e30: the reference's top-k choice among ties is implementation-defined
    # (model/graph_module.py:184-227); the oracle ran with TIE_RULE "index" (lower slot first), the rule csrc/proposals.hip implements
    assert torch.equal(d["adjacent_mat"].cpu(), g["adjacent_mat"])
    vm = od["proposal_batch_mask"] == 1
    assert l2err(d["bbox_feature"].cpu()[vm], g["bbox_feature"][vm]) < 1e-3
    assert l2err(d["bbox_feature"], g["bbox_feature"]) < 1e-3
    assert torch.equal(d["num_edge_source"].cpu(), g["num_edge_source"]) and torch.equal(d["num_edge_target"].cpu(), g["num_edge_target"])
    logits = d["lang_cap"].detach().cpu()
    assert logits.shape == ologits.shape, (logits.shape, ologits.shape)
    scale = float(ologits.abs().max())
    assert torch.allclose(logits, ologits, rtol=1e-3, atol=1e-3 * scale), (float((logits - ologits).abs().max()), scale)
    assert l2err(logits, ologits) < 1e-3
    a, b = float(d["cap_loss"]), float(ocap)
    assert abs(a - b) <= 1e-3 * abs(b), ("caption loss", a, b)
    print("bench scene, exact fp32 vs oracle chain: %d proposals, detector loss %.6f / %.6f, caption loss %.6f / %.6f, logits rel-L2 %.2e"
          % (n_prop, float(d["total_loss"][0]), float(od["total_loss"]), float(d["cap_loss"]), float(ocap), l2err(logits, ologits)))

    # ---- the bf16 step bench.py times: same integers (teacher clustering), losses close
    loss_b, db = hip_step(False)
    assert net.detector._execs.get("backbone") is not None, "the native executor did not run"
    assert np.array_equal(db["proposal_scores"][1].cpu().numpy(), od["proposal_scores"][1])
    assert np.array_equal(db["proposal_scores"][2].cpu().numpy(), od["proposal_scores"][2])
    # (which raw proposals pass TEST_SCORE_THRESH depends on the ScoreNet output, i.e. on the precision: a few borderline scores may flip)
    flips = int((db["proposal_thres_mask"].cpu() != od["proposal_thres_mask"]).sum())
    assert flips <= 0.05 * od["proposal_thres_mask"].numel(), flips
    a, b = float(db["total_loss"][0]), float(od["total_loss"])
    assert abs(a - b) <= 2e-2 * abs(b), ("bf16 detector loss", a, b)
    a, b = float(db["cap_loss"]), float(ocap)
    assert abs(a - b) <= 2e-2 * abs(b), ("bf16 caption loss", a, b)
    print("bench scene, bf16 executor: detector loss %.6f, caption loss %.6f, logits rel-L2 vs oracle %.2e"
          % (float(db["total_loss"][0]), float(db["cap_loss"]), l2err(db["lang_cap"], ologits)))


def _split_branches(idx, off):
    """(S,2), (P+1) of one detector run -> the two clustering branches' cluster lists [(members...)], in order.  The
    clusters of a branch are ordered by their seed = smallest member = first member in BFS order
    (src/bfs_cluster/bfs_cluster.cpp:37-52), so the second branch starts where the seed sequence drops."""
    P = off.shape[0] - 1
    seeds = idx[off[:-1], 1]
    drop = np.nonzero(np.diff(seeds) < 0)[0]
    assert len(drop) <= 1, "seed order is ascending inside a branch"
    cut = int(drop[0]) + 1 if len(drop) else P
    clusters = [idx[off[p]:off[p + 1], 1] for p in range(P)]
    assert all((idx[off[p]:off[p + 1], 0] == p).all() for p in range(P))
    return clusters[:cut], clusters[cut:]


def test_four_scene_batch_equals_four_single_scene_runs(dev, bench_setup):
    """the integer outputs of the batched detector pass (what bench.py steps) == the per-scene passes put together"""
    from d3net_amd import synthetic as S
    c = bench_setup
    cfg, det = c["cfg"], c["net"].detector
    scenes = c["scenes"]
    assert len(scenes) == 4
    rand = torch.rand(2, 3)

    def run(sc):
        b = S.make_batch(sc, dev)
        b["cluster_rand"] = rand
        with torch.no_grad():
            d = det.feed(b, 0)
        torch.cuda.synchronize()
        return b, d

    bb, db = run(scenes)
    assert bb["voxel_locs"].shape[0] > 600000        # the size bench.py reports (649,114 voxels)
    singles = [run([s]) for s in scenes]
    # input voxelisation: the batch's voxel list / maps are the scenes' lists with batch ids and row offsets applied
    v0 = p0 = 0
    for b, (sb, sd) in enumerate(singles):
        M, N = sb["voxel_locs"].shape[0], sb["locs"].shape[0]
        want = sb["voxel_locs"].clone(); want[:, 0] = b
        assert torch.equal(bb["voxel_locs"][v0:v0 + M], want), "voxel coordinates"
        assert torch.equal(bb["p2v_map"][p0:p0 + N], sb["p2v_map"] + v0), "point -> voxel map"
        rule = sb["v2p_map"].clone()
        cnt = rule[:, 0:1].long()
        col = torch.arange(rule.shape[1] - 1, device=dev).view(1, -1)
        rule[:, 1:] = torch.where(col < cnt, rule[:, 1:] + p0, rule[:, 1:])
        got = bb["v2p_map"][v0:v0 + M, :rule.shape[1]]
        assert torch.equal(got[:, 0], rule[:, 0]) and torch.equal(torch.where(col < cnt, got[:, 1:], rule[:, 1:]), rule[:, 1:]), "voxel -> point rules"
        v0 += M; p0 += N
    # clustering: branch 1 of all scenes in scene order, then branch 2 of all scenes (model/pointgroup.py:299-316)
    idx, off = db["proposal_scores"][1].cpu().numpy(), db["proposal_scores"][2].cpu().numpy()
    b1, b2 = [], []
    p0 = 0
    for sb, sd in singles:
        si, so_ = sd["proposal_scores"][1].cpu().numpy(), sd["proposal_scores"][2].cpu().numpy()
        f, s = _split_branches(si, so_)
        b1 += [m + p0 for m in f]; b2 += [m + p0 for m in s]
        p0 += sb["locs"].shape[0]
    want = b1 + b2
    P = off.shape[0] - 1
    assert P == len(want), (P, len(want))
    assert P > 300                                    # ~107 proposals x 4 scenes after thresholds, more before
    for p in range(P):
        assert np.array_equal(idx[off[p]:off[p + 1], 1], want[p]), "cluster %d: membership / BFS order" % p
        assert (idx[off[p]:off[p + 1], 0] == p).all()
    assert np.array_equal(db["proposals_npoint"].cpu().numpy(), np.array([len(m) for m in want], np.float32))
    # per-scene proposal counts after the thresholds need the scores (floating point, batch statistics differ) -- but the
    # batch id of every raw cluster is integer work
    bo = bb["batch_offsets"].cpu().numpy()
    want_bid = np.array([np.searchsorted(bo, m[0], side="right") - 1 for m in want])
    thres = db["proposal_thres_mask"].cpu().numpy()
    assert np.array_equal(db["proposals_batchId"].cpu().numpy(), want_bid[thres])
    # the level sizes of the batch's coordinate pyramid are the sums of the scenes' (no voxel is shared across batch ids)
    ex = det._exec("backbone")
    ex.debug_keep = True
    try:
        def rows_of(b):
            with torch.no_grad():
                det.feed(dict(b, cluster_rand=rand), 0)
            return np.array(ex.debug_last[1], np.int64)
        rows_b = rows_of(bb)
        acc = sum(rows_of(sb) for sb, _ in singles)
    finally:
        ex.debug_keep, ex.debug_last = False, None
    assert rows_b[0] == bb["voxel_locs"].shape[0] and np.array_equal(acc, rows_b), (acc, rows_b)


# ------------------------------------------------------------------ (c) level-0 convolutions at the batch's 649 k rows
@pytest.fixture(scope="module")
def canon(dev, bench_setup):
    """level-0 kernel maps of the 4-scene bench batch (device maps bit-checked against the oracle's), in the layout
    tests/test_conv_fullsize_gpu.py's helpers take"""
    from d3net_amd import minkowski as ME
    from oracle import sparse_oracle as so
    coords = np.concatenate([np.concatenate([np.full((s["locs_vox"].shape[0], 1), b, np.int64), s["locs_vox"]], 1)
                             for b, s in enumerate(bench_setup["scenes"])], 0)
    coords = np.unique(coords, axis=0)          # one row per voxel (raster order; the order does not matter here)
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    ocm = so.OracleCoords(coords)
    nbr = cm.k3(1)
    child, up, Mo = cm.down(1)
    parent, kidx, oMo = ocm.get_down(1)
    onbr = ocm.get_k3(1)
    assert Mo == oMo and np.array_equal(nbr.cpu().numpy(), onbr), "649 k-row kernel map differs from the oracle's"
    M = nbr.size(0)
    ref_up = np.full((M, 8), -1); ref_up[np.arange(M), kidx] = parent
    ref_child = np.full((Mo, 8), -1); ref_child[parent, kidx] = np.arange(M)
    assert np.array_equal(up.cpu().numpy(), ref_up) and np.array_equal(child.cpu().numpy(), ref_child)
    assert M > 600000
    import test_conv3_gpu as T3
    from d3net_amd import _lib
    tq, ok = T3._packq(_lib.lib(), nbr, dev)          # the lane table of the 649 k-row map (spconv_fwd3_kernel, round 6)
    assert ok == 1
    return {0: dict(M=M, Mo=Mo, nbr=nbr, child=child, up=up, onbr=onbr, parent=parent, kidx=kidx, tq=tq)}


@pytest.mark.parametrize("kind,cin,cout", [("k3", 16, 16), ("k3", 32, 16), ("k3", 136, 16), ("down", 16, 32), ("up", 32, 16)])
def test_bench_batch_level0_forward(dev, canon, kind, cin, cout):
    import test_conv_fullsize_gpu as T
    T.test_fwd2_big_kernel_forward_and_epilogues(dev, canon, 0, kind, cin, cout)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (16, 32)])
def test_bench_batch_level0_lane_table_kernel(dev, canon, cin, cout):
    """the lane-table kernel (what the bench step runs for these layers) at the batch's 649 k rows: forward with its epilogues and the
    data gradient with the BatchNorm-backward epilogue, 1e-4 against the bf16-mode oracle"""
    import test_conv3_gpu as T3
    T3.test_fwd3_forward_and_epilogues(dev, canon, 0, cin, cout)
    T3.test_fwd3_data_gradient_with_bn_backward_epilogue(dev, canon, 0, cin, cout)


@pytest.mark.parametrize("kind,cin,cout", [("k3", 16, 16), ("k3", 32, 16), ("down", 16, 32)])
def test_bench_batch_level0_data_gradient(dev, canon, kind, cin, cout):
    import test_conv_fullsize_gpu as T
    T.test_fwd2_big_kernel_data_gradient_and_bn_backward_epilogue(dev, canon, 0, kind, cin, cout)


@pytest.mark.parametrize("kind,cin,cout,xbf", [("k3", 16, 16, True), ("k3", 16, 16, False), ("k3", 32, 16, True), ("k3", 136, 16, True),
                                                ("down", 16, 32, True)])
def test_bench_batch_level0_weight_gradient(dev, canon, kind, cin, cout, xbf):
    import test_conv_fullsize_gpu as T
    T.test_wgrad2_row_splits_and_reduction(dev, canon, 0, kind, cin, cout, xbf)
