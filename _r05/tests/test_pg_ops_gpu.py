"""GPU parity tests: d3net_amd.pointgroup_ops (HIP, through the C ABI) vs the CPU oracle.

Bit-exact for every operator here (integer / index results, min / max / argmax, and the fp32 ops
whose summation order is pinned by the reference: sec_mean, voxelize fp/bp, get_iou).
"""
import numpy as np
import pytest
import torch

from oracle import pg_oracle as o

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def ragged_offsets(rng, total, nseg, with_empty=True):
    cuts = np.sort(rng.integers(0, total + 1, nseg - 1))
    off = np.concatenate([[0], cuts, [total]]).astype(np.int32)
    if with_empty and nseg > 2:
        off[2] = off[1]  # an empty segment
        off = np.sort(off).astype(np.int32)
    return off


@pytest.mark.parametrize("C", [1, 3, 16, 20])
def test_sec_ops_bit_exact(dev, C):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(10 + C)
    S = 20000
    x = rng.standard_normal((S, C)).astype(np.float32)
    off = ragged_offsets(rng, S, 37)
    off[-2] = off[-1] - 9000 if off[-1] - 9000 > off[-3] else off[-2]  # one long segment
    for name in ("sec_mean", "sec_min", "sec_max"):
        got = N(getattr(P, name)(T(x, dev), T(off, dev)))
        ref = getattr(o, name)(x, off)
        assert np.array_equal(got, ref), name


@pytest.mark.parametrize("C", [3, 16])
def test_roipool_fwd_bwd_bit_exact(dev, C):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(20 + C)
    S = 30000
    f = rng.integers(-3, 4, (S, C)).astype(np.float32)  # heavy ties: first argmax must win
    f[rng.random((S, C)) < 0.3] += rng.standard_normal()
    off = ragged_offsets(rng, S, 50)
    ft = T(f, dev).requires_grad_(True)
    out = P.roipool(ft, T(off, dev))
    ref, refidx = o.roipool(f, off)
    assert np.array_equal(N(out), ref)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    g[np.isinf(ref)] = 0
    out.backward(T(g, dev))
    assert np.array_equal(N(ft.grad), o.roipool_bp(g, off, refidx, S))


def test_get_iou_bit_exact(dev):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(30)
    Npts, nInst, S = 50000, 37, 30000
    inst = rng.integers(-1, nInst, Npts).astype(np.int64)
    pn = np.bincount(inst[inst >= 0], minlength=nInst).astype(np.int32)
    pidx = rng.integers(0, Npts, S).astype(np.int32)
    off = ragged_offsets(rng, S, 25)
    got = N(P.get_iou(T(pidx, dev), T(off, dev), T(inst, dev), T(pn, dev)))
    assert np.array_equal(got, o.get_iou(pidx, off, inst, pn))


@pytest.mark.parametrize("mode", [3, 4])
def test_voxelization_fwd_bwd_bit_exact(dev, mode):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(40 + mode)
    n, C = 20000, 19
    c = rng.integers(0, 24, (n, 4)).astype(np.int64); c[:, 0] = rng.integers(0, 2, n)
    _, p2v, v2p = o.voxelization_idx(c, 2, mode)
    f = rng.standard_normal((n, C)).astype(np.float32)
    ft = T(f, dev).requires_grad_(True)
    out = P.voxelization(ft, T(v2p, dev), mode)
    ref = o.voxelization(f, v2p, mode)
    assert np.array_equal(N(out), ref)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(T(g, dev))
    assert np.array_equal(N(ft.grad), o.voxelization_bp(g, v2p, n, mode))


@pytest.mark.parametrize("ncols", [3, 4])
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_voxelization_idx_bit_exact(dev, ncols, mode):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(50 + mode + 10 * ncols)
    n = 30000
    c = rng.integers(-5, 30, (n, ncols)).astype(np.int64)
    if ncols == 4:
        c[:, 0] = rng.integers(0, 3, n)
    # device input -> device output
    oc, p2v, v2p = P.voxelization_idx(T(c, dev), 3, mode)
    roc, rp2v, rv2p = o.voxelization_idx(c, 3, mode)
    assert np.array_equal(N(oc), roc) and np.array_equal(N(p2v), rp2v) and np.array_equal(N(v2p), rv2p)
    # CPU input (the reference's calling convention) -> CPU output
    oc2, p2v2, v2p2 = P.voxelization_idx(torch.from_numpy(c), 3, mode)
    assert not oc2.is_cuda and np.array_equal(N(oc2), roc) and np.array_equal(N(v2p2), rv2p)


def test_voxelization_idx_cluster_shaped(dev):
    """clusters_voxelization-shaped input: column 0 = cluster id, 14^3 grid, many points per voxel."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(60)
    n = 60000
    c = np.concatenate([np.sort(rng.integers(0, 40, (n, 1))), rng.integers(0, 14, (n, 3))], 1).astype(np.int64)
    got = P.voxelization_idx(T(c, dev), 40, 4)
    ref = o.voxelization_idx(c, 40, 4)
    for g, r in zip(got, ref):
        assert np.array_equal(N(g), r)


def test_voxelization_idx_range_error(dev):
    from d3net_amd import pointgroup_ops as P, _lib
    c = np.array([[0, 1, 2, 1 << 20]], np.int64)
    with pytest.raises(_lib.D3Error):
        P.voxelization_idx(T(c, dev), 1, 4)


def _scene_points(rng, n, spread):
    xyz = rng.random((n, 3)).astype(np.float32) * np.asarray(spread, np.float32)
    order = np.lexsort((xyz[:, 2], xyz[:, 1], (xyz[:, 0] * 20).astype(int)))  # coarse raster order
    return xyz[order]


def test_ballquery_bit_exact(dev):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(70)
    n1, n2 = 9000, 6000
    xyz = np.concatenate([_scene_points(rng, n1, (2, 1.5, 0.3)), _scene_points(rng, n2, (1, 1, 0.3))])
    bi = np.concatenate([np.zeros(n1, np.int32), np.ones(n2, np.int32)]); bo = np.array([0, n1, n1 + n2], np.int32)
    idx, sl = P.ballquery_batch_p(T(xyz, dev), T(bi, dev), T(bo, dev), 0.05, 50)
    ridx, rsl = o.ballquery_batch_p(xyz, bi, bo, 0.05, 50)
    assert np.array_equal(N(sl), rsl)
    assert np.array_equal(N(idx), ridx)


def test_ballquery_random_order_and_cap(dev):
    """Incoherent point order (culling useless) + a collapsed blob that hits the 1000 cap."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(71)
    n = 6000
    xyz = rng.random((n, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    xyz[rng.permutation(n)[:2500]] = np.array([0.5, 0.5, 0.1], np.float32) + rng.normal(0, 0.004, (2500, 3)).astype(np.float32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = P.ballquery_batch_p(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03, 300)
    ridx, rsl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert rsl[:, 1].max() == 1000
    assert np.array_equal(N(sl), rsl) and np.array_equal(N(idx), ridx)


def test_ballquery_padded_matches_compact_and_feeds_bfs(dev):
    """The sync-free padded form: same lists in the same order as the compact one (incl. capped lists), and
    bfs_cluster on it returns the oracle's clusters bit for bit."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(72)
    n = 7000
    xyz = rng.random((n, 3)).astype(np.float32) * np.array([1.2, 1, 0.2], np.float32)
    xyz[rng.permutation(n)[:2200]] = np.array([0.5, 0.5, 0.1], np.float32) + rng.normal(0, 0.004, (2200, 3)).astype(np.float32)
    n1 = 4000
    bi = np.concatenate([np.zeros(n1, np.int32), np.ones(n - n1, np.int32)]); bo = np.array([0, n1, n], np.int32)
    sem = (1 + (xyz[:, 0] * 3).astype(np.int32) % 2).astype(np.int32)
    ridx, rsl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert rsl[:, 1].max() == 1000
    pidx, psl = P.ballquery_batch_p_padded(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03)
    pidx, psl_h = N(pidx), N(psl)
    cap = pidx.size // n
    # (a point's list sits in its own slot, or -- members of a clique cell, csrc/ballquery.hip -- in the slot of the cell's leader)
    assert np.array_equal(psl_h[:, 1], rsl[:, 1]) and (psl_h[:, 0] % cap == 0).all() and (psl_h[:, 0] // cap < n).all()
    for q in rng.permutation(n)[:400]:
        assert np.array_equal(pidx[psl_h[q, 0]:psl_h[q, 0] + rsl[q, 1]], ridx[rsl[q, 0]:rsl[q, 0] + rsl[q, 1]])
    rci, rco = o.bfs_cluster(sem, ridx, rsl, 20)
    ci, co = P.bfs_cluster(T(sem, dev), T(pidx, dev), psl, 20)
    assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci)
    assert P.ballquery_batch_p_padded(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03, max_bytes=1024) is None


def test_padded_lists_beyond_2_gib_of_slots(dev):
    """Up to round 4 the padded form refused more than 2 GiB of slots (536 k points) although the library addresses n * cap up to the
    int range (2.1 M points): the 8-scene strong-scaling batch (852 k object points) fell back to the compact form.  600 k points
    (2.4 GB of slots): the padded lists equal the compact ones -- every list, compared on the device -- and so do the clusters."""
    from d3net_amd import pointgroup_ops as P
    g = torch.Generator().manual_seed(73)
    B, per = 4, 150_000
    n = B * per
    assert not P.ballquery_padded_fits(n, max_bytes=2 << 30) and P.ballquery_padded_fits(n)
    assert not P.ballquery_padded_fits(2_200_000) and not P.ballquery_padded_fits(0)
    xyz = (torch.rand((n, 3), generator=g) * 0.85).to(dev).contiguous()      # ~32 neighbours per 3 cm ball
    # one dense blob per scene (capped lists, clique cells) among the sparse points
    for b in range(B):
        xyz[b * per:b * per + 3000] = 0.4 + 0.004 * torch.randn((3000, 3), generator=g).to(dev)
    bi = torch.arange(B, dtype=torch.int32).repeat_interleave(per).to(dev)
    bo = torch.arange(0, n + 1, per, dtype=torch.int32).to(dev)
    sem = (1 + (xyz[:, 0] * 5).int() % 2).int().contiguous()
    cidx, csl = P.ballquery_batch_p(xyz, bi, bo, 0.03, 50)
    pidx, psl = P.ballquery_batch_p_padded(xyz, bi, bo, 0.03)
    assert pidx.numel() * 4 > (2 << 30)
    assert torch.equal(psl[:, 1], csl[:, 1]) and int(csl[:, 1].max()) == 1000
    ln = csl[:, 1].long()
    within = torch.arange(int(ln.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(ln, 0) - ln, ln)
    ppos = torch.repeat_interleave(psl[:, 0].long(), ln) + within
    assert torch.equal(pidx[ppos], cidx)
    del ppos, within
    a = P.bfs_cluster(sem, cidx, csl, 30, True)
    b_ = P.bfs_cluster(sem, pidx, psl, 30, True)
    assert a[1].numel() > 1 + B and torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1])


def test_workspaces_are_per_stream(dev):
    """One host thread driving two streams gets two workspaces: d3_bfs_cluster_run returns with its fill still in flight, and the
    next clustering on ANOTHER stream must not write into the buffer that fill reads (r05_f: the 16-scene batch faulted)."""
    from d3net_amd import pointgroup_ops as P
    side = torch.cuda.Stream(device=dev)
    a = P._workspace(1 << 20, torch.device(dev), "cl")
    with torch.cuda.stream(side):
        b = P._workspace(1 << 20, torch.device(dev), "cl")
        b2 = P._workspace(1 << 10, torch.device(dev), "cl")
    assert a.data_ptr() != b.data_ptr() and b2.data_ptr() == b.data_ptr()
    assert P._workspace(1 << 10, torch.device(dev), "cl").data_ptr() == a.data_ptr()
    # ... and the two-stream sequence itself: compact lists, branch 1 on the current stream, branch 2 on the side stream right behind
    rng = np.random.default_rng(74)
    n = 60000
    xyz = (rng.random((n, 3)) * np.array([1.5, 1.5, 0.3])).astype(np.float32)
    xyz2 = (xyz + rng.normal(0, 0.01, xyz.shape)).astype(np.float32)
    bi = T(np.zeros(n, np.int32), dev); bo = T(np.array([0, n], np.int32), dev)
    sem = T(np.ones(n, np.int32), dev)

    def branch(x):
        idx, sl = P.ballquery_batch_p(x, bi, bo, 0.03, 50)
        return P.bfs_cluster(sem, idx, sl, 10, True)
    x1, x2 = T(xyz, dev), T(xyz2, dev)
    ref1, ref2 = branch(x1), branch(x2)
    torch.cuda.synchronize()
    for _ in range(5):
        side.wait_stream(torch.cuda.current_stream())
        r1 = branch(x1)
        with torch.cuda.stream(side):
            r2 = branch(x2)
        torch.cuda.synchronize()
        assert all(torch.equal(u, v) for u, v in zip(r1 + r2, ref1 + ref2))


def _padded_lists_equal(pidx, psl, ridx, rsl, qs):
    for q in qs:
        a = pidx[psl[q, 0]:psl[q, 0] + psl[q, 1]]
        b = ridx[rsl[q, 0]:rsl[q, 0] + rsl[q, 1]]
        if not np.array_equal(a, b):
            return q
    return None


@pytest.mark.parametrize("grid", [1, 0])
def test_ballquery_padded_cell_grid_regimes(dev, grid):
    """The cell-grid search of the padded form (csrc/ballquery.hip) in every regime, against the C oracle's brute force
    (src/bfs_cluster/bfs_cluster.cu:15-60): sparse surfaces (<= 64 candidates: one bitonic pass across the lanes), a blob
    wider than a cell (hundreds to thousands of candidates, capped lists: LDS sort, the 1000-smallest cut), EXACTLY collapsed
    instances of 70 / 999 / 1000 / 1001 / 3000 points (clique cells: one list, shared by every member), a collapsed
    instance sitting on a cell corner (its members fall into up to 8 cells), negative coordinates, points exactly on cell
    boundaries, three batch items with one empty, shuffled point order.  grid = 0: the ordered chunk scan, same results."""
    from d3net_amd import _lib, pointgroup_ops as P
    rng = np.random.default_rng(90)
    r = 0.03
    edge = np.float32(r * 1.001)
    parts = []
    g = np.stack(np.meshgrid(np.arange(60), np.arange(50), indexing="ij"), -1).reshape(-1, 2).astype(np.float32) * 0.02
    parts.append(np.concatenate([g - 0.4, np.full((len(g), 1), -0.2, np.float32)], 1))                # a sheet, negative coordinates
    parts.append(rng.normal(0, 0.012, (2600, 3)).astype(np.float32) + np.array([1.0, 0.3, 0.2], np.float32))   # blob wider than a cell
    parts.append(rng.normal(0, 0.004, (2300, 3)).astype(np.float32) + np.array([0.2, 0.9, 0.1], np.float32))   # blob inside ~one cell, not a clique
    for m, c in ((70, (0.5, 0.5, 0.5)), (999, (0.7, 0.5, 0.5)), (1000, (0.9, 0.5, 0.5)), (1001, (1.1, 0.5, 0.5)), (3000, (1.3, 0.5, 0.5))):
        parts.append(np.repeat(np.array([c], np.float32), m, 0))                                       # exactly collapsed instances
    corner = np.array([edge * 20, edge * 21, edge * 22], np.float32)                                   # on a cell corner, +- 1 ulp
    jig = np.stack([np.nextafter(corner, np.float32(s), dtype=np.float32) for s in (-1, 1)])[rng.integers(0, 2, (1500, 3)), np.arange(3)]
    parts.append(jig.astype(np.float32))
    parts.append((np.arange(0, 40)[:, None] * edge * np.array([[1, 0, 0]], np.float32) + np.array([0, 2.0, 0], np.float32)).astype(np.float32))  # on cell boundaries
    xyz = np.concatenate(parts).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    n = len(xyz)
    cut1, cut2 = n // 2, n // 2                       # batch item 1 is empty
    bi = np.concatenate([np.zeros(cut1, np.int32), np.full(n - cut2, 2, np.int32)]); bo = np.array([0, cut1, cut2, n], np.int32)
    ridx, rsl = o.ballquery_batch_p(xyz, bi, bo, r, 300)
    assert rsl[:, 1].max() == 1000 and (rsl[:, 1] == 1000).sum() > 3000 and (rsl[:, 1] < 20).sum() > 1000
    with _lib.tuning(D3_BQ_GRID=grid):
        pidx, psl = P.ballquery_batch_p_padded(T(xyz, dev), T(bi, dev), T(bo, dev), r)
    pidx, psl = N(pidx), N(psl)
    assert np.array_equal(psl[:, 1], rsl[:, 1])
    bad = _padded_lists_equal(pidx, psl, ridx, rsl, range(n))
    assert bad is None, ("list of point", bad)
    shared = int((psl[:, 0] != np.arange(n) * 1000).sum())
    assert (shared > 4000) if grid else (shared == 0), shared       # the collapsed instances share their leader's list
    sem = np.ones(n, np.int32)
    rci, rco = o.bfs_cluster(sem, ridx, rsl, 50)
    ci, co = P.bfs_cluster(T(sem, dev), T(pidx, dev), T(psl, dev), 50, True)
    assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci)


def test_ballquery_padded_cell_grid_equals_chunk_scan_at_scene_size(dev):
    """one 40-box bench scene (~150 k object points), both clustering inputs (original and exactly shifted coordinates): the
    cell-grid lists == the ordered chunk scan's lists (itself pinned to the oracle above and in the tests around)"""
    from d3net_amd import _lib, pointgroup_ops as P, synthetic as S
    occ, sem, inst, _ = S.occupancy_grid((200, 150, 100), 40, (8, 30), (8, 30), seed=0)
    sc = S.scene_from_grid(occ, sem, inst, seed=1, feat_seed=2)
    keep = sc["sem_labels"] > 0
    xyz = sc["locs"][keep]
    info, _ = S.instance_info(sc["locs"], sc["instance_ids"])
    off = np.where((sc["instance_ids"] >= 0)[:, None], info[:, :3] - sc["locs"], 0).astype(np.float32)[keep]
    n = len(xyz)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    for pts in (xyz, (xyz + off).astype(np.float32)):
        res = []
        for grid in (0, 1):
            with _lib.tuning(D3_BQ_GRID=grid):
                pidx, psl = P.ballquery_batch_p_padded(T(pts, dev), T(bi, dev), T(bo, dev), 0.03)
            res.append((N(pidx), N(psl)))
        (i0, s0), (i1, s1) = res
        assert np.array_equal(s0[:, 1], s1[:, 1])
        bad = _padded_lists_equal(i1, s1, i0, s0, np.random.default_rng(3).permutation(n)[:20000])
        assert bad is None, bad


def test_ballquery_empty_batch_item_and_tiny(dev):
    from d3net_amd import pointgroup_ops as P
    xyz = np.array([[0, 0, 0], [0.01, 0, 0], [1, 1, 1]], np.float32)
    bi = np.array([0, 0, 2], np.int32); bo = np.array([0, 2, 2, 3], np.int32)
    idx, sl = P.ballquery_batch_p(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03, 50)
    ridx, rsl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 50)
    assert np.array_equal(N(sl), rsl) and np.array_equal(N(idx), ridx)


@pytest.mark.parametrize("seed", [80, 81])
def test_bfs_cluster_bit_exact(dev, seed):
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(seed)
    n = 12000
    xyz = _scene_points(rng, n, (2, 1.5, 0.2))
    sem = (1 + (xyz[:, 0] * 2).astype(np.int32) % 3 + (rng.random(n) < 0.05)).astype(np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.04, 50)
    rci, rco = o.bfs_cluster(sem, idx, sl, 20)
    assert len(rco) > 3
    # device inputs
    ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 20)
    assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci)
    # CPU inputs (reference calling convention)
    ci2, co2 = P.bfs_cluster(torch.from_numpy(sem), torch.from_numpy(idx), torch.from_numpy(sl), 20)
    assert not ci2.is_cuda and np.array_equal(N(ci2), rci) and np.array_equal(N(co2), rco)


def test_bfs_cluster_truncated_lists(dev):
    """Capped, asymmetric lists (collapsed blobs) -- the shifted-coordinate regime of PointGroup."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(82)
    n = 5000
    xyz = rng.normal(0, 0.008, (n, 3)).astype(np.float32)
    xyz[:, 0] += (rng.integers(0, 3, n) * 0.03).astype(np.float32)
    sem = rng.integers(1, 3, n).astype(np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert (sl[:, 1] >= 1000).sum() > 500
    rci, rco = o.bfs_cluster(sem, idx, sl, 10)
    ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 10)
    assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci)
    ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 10, True)     # ascending-list hint (prefix-skipping label push)
    assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci)


@pytest.mark.parametrize("star", [True, False])
def test_bfs_cluster_collapsed_instances_star_shortcut(dev, star, monkeypatch):
    """The shifted-coordinate regime with exact offsets: every point of an instance sits on the instance centre, lists are
    the instance's first 1000 members.  Clusters whose members all sit in the seed's list are written by cl_star_kernel
    (no edge records, no level loop); instances of 999 / 1000 / 1001 / 1500 points straddle the 1000-entry cap (beyond it
    the late members are reached by nobody and drop out as singletons -- the reference's behaviour).  Both code paths
    (D3_BFS_NO_STAR forces the level loop) must equal the sequential oracle, in shuffled point order."""
    from d3net_amd import pointgroup_ops as P
    from d3net_amd import _lib
    assert _lib.lib().d3_tuning_set(b"D3_BFS_NO_STAR", 0 if star else 1) == 0     # (library switches: csrc/tuning.hip)
    rng = np.random.default_rng(84)
    sizes = [60, 999, 1000, 1001, 1500, 40, 300]
    centres = rng.random((len(sizes), 3)).astype(np.float32) * 3
    xyz = np.concatenate([np.repeat(c[None], m, 0) for c, m in zip(centres, sizes)])
    # a sparse sheet as well: a cluster that needs many BFS levels next to the stars
    g = np.stack(np.meshgrid(np.arange(40), np.arange(30), indexing="ij"), -1).reshape(-1, 2).astype(np.float32) * 0.02
    sheet = np.concatenate([g, np.full((len(g), 1), 5.0, np.float32)], 1)
    xyz = np.concatenate([xyz, sheet])
    sem = np.concatenate([np.full(m, 1 + k % 3, np.int32) for k, m in enumerate(sizes)] + [np.full(len(sheet), 2, np.int32)])
    perm = rng.permutation(len(xyz))
    xyz, sem = xyz[perm], sem[perm]
    n = len(xyz)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    rci, rco = o.bfs_cluster(sem, idx, sl, 50)
    assert sorted(np.diff(rco).tolist()) == [60, 300, 999, 1000, 1000, 1000, 1200]
    try:
        for asc in (False, True):
            ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 50, asc)
            assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci), (star, asc)
    finally:
        _lib.lib().d3_tuning_set(b"D3_BFS_NO_STAR", 0)


@pytest.mark.parametrize("order", ["forward", "reverse", "shuffled"])
def test_bfs_cluster_capped_chain(dev, order):
    """A chain of dense blobs, every list capped at 1000: the labels travel blob to blob, so the label push needs
    several sweeps (worklist, paired sweeps per host round trip) -- in three index orders."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(83)
    nb, per = 7, 1300
    xyz = np.concatenate([rng.normal(0, 0.004, (per, 3)).astype(np.float32) + np.array([0.02 * k, 0, 0], np.float32) for k in range(nb)])
    n = xyz.shape[0]
    if order == "reverse":
        xyz = xyz[::-1].copy()
    elif order == "shuffled":
        xyz = xyz[rng.permutation(n)]
    sem = np.ones(n, np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert (sl[:, 1] >= 1000).sum() > n // 2
    rci, rco = o.bfs_cluster(sem, idx, sl, 10)
    for asc in (False, True):
        ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 10, asc)
        assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci), asc


@pytest.mark.parametrize("spec", [1, 0])
def test_bfs_cluster_run_speculative_fill_and_many_clusters(dev, spec):
    """d3_bfs_cluster_run: the fill enqueued behind the count kernels with its sizes read on the device (D3_CL_SPEC=1, the default)
    equals the count -> host -> fill order (D3_CL_SPEC=0) and the oracle -- also with more kept clusters than the speculative replay
    launch has slots (2048: the rest replays in a second launch), with and without the star shortcut."""
    from d3net_amd import pointgroup_ops as P, _lib
    rng = np.random.default_rng(84)
    # 3000 isolated triples (clusters of 3) + one 600-point rod (a real level loop) in index order after them
    ncl = 3000
    base = np.stack([np.arange(ncl) % 60, (np.arange(ncl) // 60) % 60, np.zeros(ncl)], 1).astype(np.float32) * 0.2
    tri = (base[:, None, :] + rng.normal(0, 0.004, (ncl, 3, 3)).astype(np.float32)).reshape(-1, 3)
    rod = np.stack([np.linspace(0, 6.0, 600), np.full(600, 20.0), np.zeros(600)], 1).astype(np.float32)
    xyz = np.concatenate([tri, rod]).astype(np.float32)
    n = xyz.shape[0]
    sem = np.ones(n, np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 50)
    rci, rco = o.bfs_cluster(sem, idx, sl, 3)
    assert len(rco) - 1 > 2048 + 500
    L = _lib.lib()
    try:
        assert L.d3_tuning_set(b"D3_CL_SPEC", spec) == 0
        for no_star in (0, 1):
            assert L.d3_tuning_set(b"D3_BFS_NO_STAR", no_star) == 0
            ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 3, True)
            assert np.array_equal(N(co), rco) and np.array_equal(N(ci), rci), (spec, no_star)
        # nothing kept at all: offsets == [0]
        ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 100000, True)
        assert N(co).tolist() == [0] and ci.shape[0] == 0
    finally:
        L.d3_tuning_set(b"D3_CL_SPEC", 1)
        L.d3_tuning_set(b"D3_BFS_NO_STAR", 0)


@pytest.mark.parametrize("spec", [1, 0])
def test_bfs_cluster_begin_end_two_in_flight(dev, spec):
    """d3_bfs_cluster_begin / _end (round 5): ONE host thread keeps two clusterings in flight on two streams -- begin, begin, end, end,
    the order PointGroup.forward uses -- and gets what the one-call form and the oracle return: padded and compact lists, a
    threshold nothing passes, an empty input; with D3_CL_SPEC=0 `begin` runs the whole blocking call and `end` only hands over."""
    from d3net_amd import pointgroup_ops as P, _lib
    rng = np.random.default_rng(86)
    n = 9000
    xyz1 = (rng.random((n, 3)) * np.array([1.0, 1.0, 0.25])).astype(np.float32)
    xyz1[rng.permutation(n)[:1500]] = np.array([0.3, 0.6, 0.1], np.float32) + rng.normal(0, 0.003, (1500, 3)).astype(np.float32)   # capped lists
    xyz2 = (xyz1 + rng.normal(0, 0.006, xyz1.shape)).astype(np.float32)
    bi = np.concatenate([np.zeros(5000, np.int32), np.ones(n - 5000, np.int32)]); bo = np.array([0, 5000, n], np.int32)
    sem = (1 + (xyz1[:, 1] * 4).astype(np.int32) % 3).astype(np.int32)
    refs = []
    for x in (xyz1, xyz2):
        idx, sl = o.ballquery_batch_p(x, bi, bo, 0.03, 50)
        refs.append(o.bfs_cluster(sem, idx, sl, 15))
    L = _lib.lib()
    side = torch.cuda.Stream(device=dev)
    semd, bid, bod = T(sem, dev), T(bi, dev), T(bo, dev)
    x1, x2 = T(xyz1, dev), T(xyz2, dev)
    try:
        assert L.d3_tuning_set(b"D3_CL_SPEC", spec) == 0
        for padded in (True, False):
            for rep in range(3):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    i2, s2 = P.ballquery_batch_p_padded(x2, bid, bod, 0.03, ws_tag="s") if padded else P.ballquery_batch_p(x2, bid, bod, 0.03, 50)
                    h2 = P.bfs_cluster_begin(semd, i2, s2, 15, True, ws_tag="s")
                i1, s1 = P.ballquery_batch_p_padded(x1, bid, bod, 0.03, ws_tag="m") if padded else P.ballquery_batch_p(x1, bid, bod, 0.03, 50)
                h1 = P.bfs_cluster_begin(semd, i1, s1, 15, True, ws_tag="m")
                r1 = P.bfs_cluster_end(h1)
                with torch.cuda.stream(side):
                    r2 = P.bfs_cluster_end(h2)
                torch.cuda.synchronize()
                for got, want in ((r1, refs[0]), (r2, refs[1])):
                    assert np.array_equal(N(got[1]), want[1]) and np.array_equal(N(got[0]), want[0]), (spec, padded, rep)
        # nothing kept / nothing there
        i1, s1 = P.ballquery_batch_p_padded(x1, bid, bod, 0.03)
        ci, co = P.bfs_cluster_end(P.bfs_cluster_begin(semd, i1, s1, 1000000, True))
        assert N(co).tolist() == [0] and ci.shape[0] == 0
        e = torch.zeros(0, dtype=torch.int32, device=dev)
        ci, co = P.bfs_cluster_end(P.bfs_cluster_begin(e, e, torch.zeros((0, 2), dtype=torch.int32, device=dev), 10, True))
        assert N(co).tolist() == [0] and ci.shape[0] == 0
    finally:
        L.d3_tuning_set(b"D3_CL_SPEC", 1)


def test_bfs_replay_forms_agree_with_the_oracle(dev):
    """Round 5: cl_bfs3_kernel (thread per frontier node, key election in LDS words; D3_BFS3=1, the default) and the edge-parallel hash
    form (cl_bfs2_kernel, D3_BFS3=0) against the sequential oracle on inputs that reach every path of the new kernel: a 190 x 190
    sheet in shuffled order (36,100 nodes: fits the discovery words; square wavefronts of > 512 nodes -> frontiers read back from the
    global queue in several batches), a 200 x 200 sheet (40,000 nodes: beyond the words -> stays on cl_bfs2_kernel inside the same
    call), a loose 3-d blob (lists of 100 - 400 entries: the beyond-registers part of a list) and a dense blob with capped lists
    (512 x 1000 entries per chunk -> the 64-nodes-at-a-time sub-batches)."""
    from d3net_amd import pointgroup_ops as P
    from d3net_amd import _lib
    rng = np.random.default_rng(85)

    def sheet(nx, ny, z):
        g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij"), -1).reshape(-1, 2).astype(np.float32) * 0.02
        return np.concatenate([g, np.full((len(g), 1), z, np.float32)], 1)
    parts = [sheet(190, 190, 0.0), sheet(200, 200, 1.0),
             rng.normal(0, 0.03, (3000, 3)).astype(np.float32) + np.array([8.0, 0, 3.0], np.float32),
             rng.normal(0, 0.006, (2500, 3)).astype(np.float32) + np.array([12.0, 0, 5.0], np.float32)]
    xyz = np.concatenate(parts)
    sem = np.concatenate([np.full(len(p), 2 + k, np.int32) for k, p in enumerate(parts)])
    perm = rng.permutation(len(xyz))
    xyz, sem = xyz[perm], sem[perm]
    n = len(xyz)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = P.ballquery_batch_p(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03, 300)
    idx, sl = N(idx), N(sl)
    rci, rco = o.bfs_cluster(sem, idx, sl, 50)
    sizes = np.diff(rco)
    assert 36100 in sizes and 40000 in sizes and (sl[:, 1] >= 1000).sum() > 500 and ((sl[:, 1] > 100) & (sl[:, 1] < 1000)).sum() > 500
    try:
        for form in (1, 0):
            assert _lib.lib().d3_tuning_set(b"D3_BFS3", form) == 0
            for asc in (False, True):
                ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 50, asc)
                assert np.array_equal(N(co), rco), (form, asc)
                assert np.array_equal(N(ci), rci), (form, asc, int((N(ci) != rci).any(1).argmax()))
    finally:
        _lib.lib().d3_tuning_set(b"D3_BFS3", 0)


def test_bfs_cluster_no_clusters(dev):
    from d3net_amd import pointgroup_ops as P
    sem = np.array([1, 2, 3], np.int32); idx = np.array([0, 1, 2], np.int32)
    sl = np.array([[0, 1], [1, 1], [2, 1]], np.int32)
    ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 50)
    assert ci.shape == (0, 2) and N(co).tolist() == [0]


def test_full_size_ballquery_and_cluster_properties(dev):
    """Canonical 164k-point scene (BASELINE config 2): size-independent properties, no oracle."""
    from d3net_amd import pointgroup_ops as P, synthetic as S
    sc = S.canonical_scene(n_feat=1)
    keep = sc["sem_labels"] > 0
    xyz = sc["locs"][keep]; sem = sc["sem_labels"][keep].astype(np.int32)
    n = xyz.shape[0]
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = P.ballquery_batch_p(T(xyz, dev), T(bi, dev), T(bo, dev), 0.03, 50)
    idx, sl = N(idx), N(sl)
    assert (sl[:, 1] >= 1).all() and sl[:, 1].max() < 1000
    assert np.array_equal(sl[:, 0], np.concatenate([[0], np.cumsum(sl[:, 1])[:-1]]))
    owner = np.repeat(np.arange(n), sl[:, 1])
    # every list is strictly ascending and contains the point itself
    same = owner[1:] == owner[:-1]
    assert (idx[1:][same] > idx[:-1][same]).all()
    assert np.isin(np.arange(n) * (1 << 20) + np.arange(n), owner.astype(np.int64) * (1 << 20) + idx).all()
    # symmetry: (i,j) listed <=> (j,i) listed (no list was capped)
    a = np.sort(owner.astype(np.int64) * (1 << 20) + idx); b = np.sort(idx.astype(np.int64) * (1 << 20) + owner)
    assert np.array_equal(a, b)
    # spot-check 200 points against brute force
    rng = np.random.default_rng(0)
    for i in rng.integers(0, n, 200):
        d = xyz[i] - xyz
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        assert np.array_equal(idx[sl[i, 0]:sl[i, 0] + sl[i, 1]], np.nonzero(d2 < np.float32(0.03) * np.float32(0.03))[0])
    ci, co = P.bfs_cluster(T(sem, dev), T(idx, dev), T(sl, dev), 50, True)
    ci, co = N(ci), N(co)
    # clusters partition a subset of the points, are label-pure, sized >= 50, seeds ascending
    assert len(np.unique(ci[:, 1])) == len(ci)
    sizes = np.diff(co)
    assert (sizes >= 50).all() and co[-1] == len(ci)
    assert (ci[:, 0] == np.repeat(np.arange(len(sizes)), sizes)).all()
    seeds = ci[co[:-1], 1]
    assert (np.diff(seeds) > 0).all()
    for c in range(len(sizes)):
        pts = ci[co[c]:co[c + 1], 1]
        assert len(np.unique(sem[pts])) == 1 and pts[0] == pts.min()
    # the full-size result equals the sequential oracle BFS (cheap: O(nActive))
    rci, rco = o.bfs_cluster(sem, idx, sl, 50)
    assert np.array_equal(ci, rci) and np.array_equal(co, rco)


def test_cluster_coordinate_stats_and_transform_equal_the_library_op_form(dev):
    """d3_cluster_coords_stats / d3_cluster_transform (the per-point passes of PointGroup.clusters_voxelization,
    model/pointgroup.py:125-178) against the chain of library ops they replace -- gather, sec_mean / sec_min / sec_max of the
    shifted copy, index_select, multiply, add, `.long()`, cat -- bit for bit, on clusters of very different sizes (one point,
    a few, tens of thousands), with the reference's scale / offset arithmetic in between."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(11)
    N = 120000
    coords = torch.from_numpy((rng.random((N, 3)) * 4).astype(np.float32)).to(dev)
    sizes = [1, 2, 7, 50, 333, 4096, 40000, 1, 25000, 19]
    pts = [rng.choice(N, s, replace=False) for s in sizes]
    cidx = np.concatenate([np.stack([np.full(s, i), p], 1) for i, (s, p) in enumerate(zip(sizes, pts))]).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    clusters_idx, clusters_offset = torch.from_numpy(cidx).to(dev), torch.from_numpy(offs).to(dev)
    fullscale, scale = 14, 50
    r01 = torch.from_numpy(rng.random((2, 3)).astype(np.float32)).to(dev)
    # library-op form (what clusters_voxelization did before)
    c_idxs, cid = clusters_idx[:, 1].long(), clusters_idx[:, 0].long()
    cc = coords[c_idxs]
    mean = P.sec_mean(cc, clusters_offset)
    cc = cc - torch.index_select(mean, 0, cid)
    cmin, cmax = P.sec_min(cc, clusters_offset), P.sec_max(cc, clusters_offset)

    def scalars(cmin, cmax):
        sc = 1 / ((cmax - cmin) / fullscale).max(1)[0] - 0.01
        sc = torch.clamp(sc, min=None, max=scale)
        min_xyz, max_xyz = cmin * sc.unsqueeze(-1), cmax * sc.unsqueeze(-1)
        rg = max_xyz - min_xyz
        off = -min_xyz + torch.clamp(fullscale - rg - 0.001, min=0) * r01[0] + torch.clamp(fullscale - rg + 0.001, max=0) * r01[1]
        return sc, off

    sc, off = scalars(cmin, cmax)
    ref = cc * torch.index_select(sc, 0, cid).unsqueeze(-1) + torch.index_select(off, 0, cid)
    ref = torch.cat([cid.view(-1, 1), ref.long()], 1).contiguous()
    # fused form
    mean2, rmin, rmax = P.cluster_coords_stats(coords, clusters_idx, clusters_offset)
    assert torch.equal(mean2, mean)
    assert torch.equal(rmin - mean2, cmin) and torch.equal(rmax - mean2, cmax)
    sc2, off2 = scalars(rmin - mean2, rmax - mean2)
    got = P.cluster_transform(coords, clusters_idx, mean2, sc2, off2)
    assert got.dtype == torch.int64 and torch.equal(got, ref)
    # the per-cluster arithmetic in one launch (d3_cluster_norm_params): bit-equal to the elementwise library chain, incl. a
    # one-point cluster (zero extent -> 1/0 = inf -> capped scale) and clusters wider than the grid
    size3, center3, sc3, off3 = P.cluster_norm_params(mean2, rmin, rmax, fullscale, scale, r01[0].cpu(), r01[1].cpu())
    assert torch.equal(sc3, sc2) and torch.equal(off3, off2)
    assert torch.equal(size3, cmax - cmin) and torch.equal(center3, (cmax + cmin) / 2 + mean2)
    # many random clusters with odd extents (tiny, huge, degenerate along one axis)
    Pn = 5000
    m = torch.from_numpy((rng.random((Pn, 3)) * 6).astype(np.float32)).to(dev)
    ext = torch.from_numpy((10.0 ** rng.uniform(-4, 1.2, (Pn, 3))).astype(np.float32)).to(dev)
    ext[::17, 1] = 0
    lo, hi = m - ext * 0.37, m + ext * 0.63
    size4, center4, sc4, off4 = P.cluster_norm_params(m, lo, hi, fullscale, scale, r01[0].cpu(), r01[1].cpu())
    sc_ref, off_ref = scalars(lo - m, hi - m)
    assert torch.equal(sc4, sc_ref) and torch.equal(off4, off_ref)
    assert torch.equal(size4, (hi - m) - (lo - m)) and torch.equal(center4, ((hi - m) + (lo - m)) / 2 + m)


def test_cluster_select_and_merge_equal_the_library_op_chain(dev):
    """d3_cluster_select / d3_cluster_merge against the library ops of model/pointgroup.py:288-316 they replace (gathers by
    object_idxs, the shifted coordinates, the in-place id / offset shifts and the three concatenations with the reference's
    one-element-short batch-id vector), incl. empty first / second cluster sets."""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(5)
    N = 50000
    locs = torch.from_numpy(rng.random((N, 3)).astype(np.float32)).to(dev)
    offs = torch.from_numpy((rng.random((N, 3)) - 0.5).astype(np.float32)).to(dev)
    sem = torch.from_numpy(rng.integers(0, 20, N)).to(dev)
    batch = torch.from_numpy(np.sort(rng.integers(0, 4, N)).astype(np.int32)).to(dev)
    obj = torch.nonzero(sem > 1).view(-1)
    b_, c_, sh_, s_ = P.cluster_select(locs, offs, sem, batch, obj)
    assert torch.equal(b_, batch[obj]) and torch.equal(c_, locs[obj]) and torch.equal(s_, sem[obj].int())
    assert torch.equal(sh_, locs[obj] + offs[obj])
    # ... and with the batch offsets of the object points (get_batch_offsets, model/pointgroup.py:110-122) from the same launch: every
    # pattern of scenes WITHOUT object points (first, middle, last, all but one), a single object point
    from d3net_amd.pointgroup import PointGroup
    for B, present in ((4, (0, 1, 2, 3)), (6, (1, 2, 4)), (6, (0, 5)), (5, (3,)), (3, (0, 1, 2))):
        bt = torch.from_numpy(np.sort(rng.choice(np.array(present), N)).astype(np.int32)).to(dev)
        for ob in (obj, obj[7:8]):
            got = P.cluster_select(locs, offs, sem, bt, ob, batch_size=B)
            assert len(got) == 5 and torch.equal(got[0], bt[ob]) and torch.equal(got[2], locs[ob] + offs[ob])
            want = PointGroup.get_batch_offsets(bt[ob], B)
            assert got[4].dtype == torch.int32 and torch.equal(got[4], want), (B, present, got[4].tolist(), want.tolist())
    e = P.cluster_select(locs, offs, sem, batch, obj[:0], batch_size=4)
    assert e[0].numel() == 0 and e[4].tolist() == [0, 0, 0, 0, 0]
    n = obj.numel()

    def fake(nc, lo, hi):
        sizes = rng.integers(lo, hi, nc) if nc else np.zeros(0, np.int64)
        idx = np.concatenate([np.stack([np.full(s, i), rng.integers(0, n, s)], 1) for i, s in enumerate(sizes)] + [np.zeros((0, 2), np.int64)])
        return torch.from_numpy(idx.astype(np.int32)).to(dev), torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)

    for nc1, nc2 in ((7, 5), (0, 4), (6, 0), (1, 1)):
        i1, o1 = fake(nc1, 50, 400)
        i2, o2 = fake(nc2, 50, 400)
        got = P.cluster_merge(i1, o1, i2, o2, obj, batch)
        # the library-op chain
        a, b = i1.clone(), i2.clone()
        a[:, 1] = obj[a[:, 1].long()].int(); b[:, 1] = obj[b[:, 1].long()].int()
        ba, bb = batch[a[:, 1].long()].int(), batch[b[:, 1].long()].int()
        b[:, 0] += (o1.size(0) - 1)
        o2s = o2 + o1[-1]
        ref = (torch.cat((a, b), 0), torch.cat((o1, o2s[1:])), torch.cat((ba, bb[1:])))
        for g, r in zip(got, ref):
            assert g.dtype == r.dtype and torch.equal(g, r), (nc1, nc2, g.shape, r.shape)


def test_voxelization_of_a_column_concatenation_without_the_copy(dev):
    """d3_voxelize_fp2 == voxelization(cat(feats, locs)) bit for bit (modes mean / sum, crowded voxels)"""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(2)
    N = 30000
    coords = torch.from_numpy(np.concatenate([np.zeros((N, 1)), rng.integers(0, 14, (N, 3))], 1).astype(np.int64)).to(dev)
    feats = torch.from_numpy(rng.standard_normal((N, 131)).astype(np.float32)).to(dev)
    locs = torch.from_numpy(rng.random((N, 3)).astype(np.float32)).to(dev)
    for mode in (4, 3):
        _, _, v2p = P.voxelization_idx(coords, 1, mode)
        ref = P.voxelization(torch.cat((feats, locs), 1).contiguous(), v2p, mode)
        got = P.voxelization_cat(feats, locs, v2p, mode)
        assert int(v2p[:, 0].max()) > 8 and torch.equal(got, ref)


def test_proposal_bookkeeping_one_launch_equals_the_library_op_chain(dev):
    """d3_proposal_prepare against the library ops of model/pointgroup.py:338-372 (points per proposal, score / size threshold
    mask, the batch id read at the cluster start from the one-element-short batch-id vector, the (P,9) crop box): bit-equal"""
    from d3net_amd import pointgroup_ops as P
    rng = np.random.default_rng(21)
    Pn, N = 300, 5000
    sizes = rng.integers(1, 200, Pn)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)
    S = int(offs[-1])
    pidx = torch.from_numpy(np.stack([np.repeat(np.arange(Pn), sizes), rng.integers(0, N, S)], 1).astype(np.int32)).to(dev)
    bid_all = torch.from_numpy(rng.integers(0, 4, S - 1).astype(np.int32)).to(dev)          # one short, like the reference's concat
    sem = torch.from_numpy(rng.integers(0, 20, N).astype(np.int64)).to(dev)
    sig = torch.sigmoid(torch.randn(Pn, device=dev))
    center, size = torch.randn(Pn, 3, device=dev), torch.rand(Pn, 3, device=dev)
    thr_s, thr_n = 0.35, 50
    npoint, mask, bid, crop = P.proposal_prepare(sig, offs, bid_all, pidx, sem, center, size, thr_s, thr_n)
    npoint_ref = (offs[1:] - offs[:-1]).float()
    mask_ref = torch.logical_and(sig > thr_s, npoint_ref > thr_n)
    starts = offs[:-1].long().clamp(max=bid_all.numel() - 1)
    crop_ref = torch.zeros(Pn, 9, device=dev)
    crop_ref[:, :3] = center; crop_ref[:, 3:6] = size
    crop_ref[:, 7] = sem[pidx[offs[:-1].long(), 1].long()].float(); crop_ref[:, 8] = sig
    assert torch.equal(npoint, npoint_ref) and torch.equal(mask, mask_ref) and mask.dtype == torch.bool
    assert torch.equal(bid, bid_all[starts]) and torch.equal(crop, crop_ref)
