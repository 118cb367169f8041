"""CPU tests of the oracle (oracle/pg_ops_oracle.c) -- known answers and self-consistency.

The reference ships no tests or vectors for these operators; the only recorded
reference outputs are the two toy results of SURVEY.md section 8(c) (produced there by compiling the
reference's voxelize.cpp / bfs_cluster.cpp with stand-in headers, which this build may not redo).
"""
import numpy as np

from oracle import pg_oracle as o
import bfs_parallel_model as model


def test_voxelize_idx_survey_toy():
    # coords -> [[0,1,1,1],[0,2,2,2],[1,1,1,1]], p2v [0,0,1,2,0], v2p [[3,0,1,4],[1,2,0,0],[1,3,0,0]]
    c = np.array([[0, 1, 1, 1], [0, 1, 1, 1], [0, 2, 2, 2], [1, 1, 1, 1], [0, 1, 1, 1]], np.int64)
    oc, p2v, v2p = o.voxelization_idx(c, 2, 4)
    assert oc.tolist() == [[0, 1, 1, 1], [0, 2, 2, 2], [1, 1, 1, 1]]
    assert p2v.tolist() == [0, 0, 1, 2, 0]
    assert v2p.tolist() == [[3, 0, 1, 4], [1, 2, 0, 0], [1, 3, 0, 0]]


def test_bfs_cluster_survey_toy():
    # 5-point toy graph -> [[0,0],[0,1],[0,2],[1,3],[1,4]], offsets [0,3,5]
    sem = np.array([1, 1, 1, 2, 2], np.int32)
    lists = [[0, 1], [0, 1, 2], [1, 2], [3, 4], [3, 4]]
    idx = np.array(sum(lists, []), np.int32)
    ln = np.array([len(l) for l in lists]); st = np.concatenate([[0], np.cumsum(ln)[:-1]])
    sl = np.stack([st, ln], 1).astype(np.int32)
    ci, co = o.bfs_cluster(sem, idx, sl, 2)
    assert ci.tolist() == [[0, 0], [0, 1], [0, 2], [1, 3], [1, 4]]
    assert co.tolist() == [0, 3, 5]


def test_voxelize_modes_and_order():
    rng = np.random.default_rng(0)
    c = rng.integers(0, 4, (200, 4)).astype(np.int64)
    c[:, 0] = rng.integers(0, 2, 200)
    for mode in (1, 2, 3, 4):
        oc, p2v, v2p = o.voxelization_idx(c, 2, mode)
        M = oc.shape[0]
        # first-occurrence order: the first point of voxel v precedes the first point of voxel v+1
        firsts = [np.nonzero(p2v == v)[0][0] for v in range(M)]
        assert firsts == sorted(firsts)
        for v in range(M):
            pts = np.nonzero(p2v == v)[0]
            if mode in (3, 4):
                assert v2p[v, 0] == len(pts) and v2p[v, 1:1 + len(pts)].tolist() == pts.tolist()
                assert (v2p[v, 1 + len(pts):] == 0).all()
            elif mode == 1:
                assert v2p[v].tolist() == [1, pts[0]]
            else:
                assert v2p[v].tolist() == [1, pts[-1]]
            assert (oc[v] == c[v2p[v, 1]]).all()


def test_voxelize_fp_bp_against_numpy():
    rng = np.random.default_rng(1)
    c = rng.integers(0, 5, (300, 3)).astype(np.int64)
    _, p2v, v2p = o.voxelization_idx(c, 1, 4)
    f = rng.standard_normal((300, 7)).astype(np.float32)
    out = o.voxelization(f, v2p, 4)
    ref = np.stack([f[p2v == v].astype(np.float64).mean(0) for v in range(v2p.shape[0])])
    assert np.allclose(out, ref, atol=1e-5)
    g = rng.standard_normal(out.shape).astype(np.float32)
    dg = o.voxelization_bp(g, v2p, 300, 4)
    cnt = np.bincount(p2v)
    assert np.allclose(dg, g[p2v] / cnt[p2v][:, None], atol=1e-6)


def test_ballquery_against_numpy():
    rng = np.random.default_rng(2)
    n = 500
    xyz = rng.random((n, 3)).astype(np.float32) * 0.3
    bi = np.concatenate([np.zeros(300, np.int32), np.ones(200, np.int32)]); bo = np.array([0, 300, 500], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.05, 20)
    for i in range(n):
        d = xyz[i] - xyz
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        hit = np.nonzero((d2 < np.float32(0.05) * np.float32(0.05)) & (bi == bi[i]))[0]
        assert idx[sl[i, 0]:sl[i, 0] + sl[i, 1]].tolist() == hit.tolist()
    assert sl[:, 0].tolist() == np.concatenate([[0], np.cumsum(sl[:, 1])[:-1]]).tolist()


def test_ballquery_cap_1000():
    xyz = np.zeros((1500, 3), np.float32)
    bi = np.zeros(1500, np.int32); bo = np.array([0, 1500], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert (sl[:, 1] == 1000).all()
    assert idx[:1000].tolist() == list(range(1000))


def test_segment_ops():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((100, 3)).astype(np.float32)
    off = np.array([0, 10, 10, 55, 100], np.int32)
    mn, mx, me = o.sec_min(x, off), o.sec_max(x, off), o.sec_mean(x, off)
    for p in range(4):
        seg = x[off[p]:off[p + 1]]
        if len(seg) == 0:
            assert np.isposinf(mn[p]).all() and np.isneginf(mx[p]).all() and (me[p] == 0).all()
        else:
            assert (mn[p] == seg.min(0)).all() and (mx[p] == seg.max(0)).all()
            assert np.allclose(me[p], seg.mean(0), atol=1e-6)
    f = rng.integers(0, 4, (100, 5)).astype(np.float32)  # many ties -> first argmax matters
    out, am = o.roipool(f, off)
    for p in (0, 2, 3):
        seg = f[off[p]:off[p + 1]]
        assert (out[p] == seg.max(0)).all() and (am[p] == off[p] + seg.argmax(0)).all()
    assert (am[1] == -1).all()
    g = rng.standard_normal(out.shape).astype(np.float32)
    df = o.roipool_bp(g, off, am, 100)
    assert np.isclose(df.sum(), g[[0, 2, 3]].sum(), atol=1e-4)


def test_get_iou():
    rng = np.random.default_rng(4)
    N = 400
    inst = rng.integers(-1, 5, N).astype(np.int64)
    pn = np.bincount(inst[inst >= 0], minlength=5).astype(np.int32)
    pidx = rng.permutation(N)[:150].astype(np.int32)
    off = np.array([0, 50, 150], np.int32)
    iou = o.get_iou(pidx, off, inst, pn)
    for p in range(2):
        pts = pidx[off[p]:off[p + 1]]
        for j in range(5):
            inter = (inst[pts] == j).sum()
            ref = np.float32(np.float32(inter) / (np.float64(np.float32(len(pts) + pn[j] - inter)) + 1e-5))
            assert iou[p, j] == ref


def test_parallel_bfs_model_matches_sequential_oracle():
    """The data-parallel formulation implemented by csrc/cluster.hip == the reference's sequential BFS."""
    rng = np.random.default_rng(5)
    for trial in range(4):
        n = 300
        xyz = rng.random((n, 3)).astype(np.float32) * np.array([0.5, 0.5, 0.1], np.float32)
        sem = rng.integers(1, 3, n).astype(np.int32)
        bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
        idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.06, 50)
        ci, co = o.bfs_cluster(sem, idx, sl, 5)
        ci2, co2 = model.bfs_cluster_parallel_model(sem, idx, sl, 5)
        assert np.array_equal(ci, ci2) and np.array_equal(co, co2)


def test_parallel_bfs_model_truncated_lists():
    """Asymmetric (capped) lists: a blob whose lists are all cut at 1000 entries."""
    rng = np.random.default_rng(6)
    n = 2200
    xyz = rng.normal(0, 0.006, (n, 3)).astype(np.float32)
    xyz[:, 0] += (rng.integers(0, 2, n) * 0.03).astype(np.float32)
    sem = rng.integers(1, 3, n).astype(np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert (sl[:, 1] >= 1000).sum() > 100
    ci, co = o.bfs_cluster(sem, idx, sl, 5)
    ci2, co2 = model.bfs_cluster_parallel_model(sem, idx, sl, 5)
    assert np.array_equal(ci, ci2) and np.array_equal(co, co2)


def test_key_election_bfs_model_equals_the_reference_order():
    """cl_bfs3_kernel's formulation (round 5: a lane group per frontier node, (batch, node position, list position) min-election in
    a per-node word) against the sequential oracle: sparse surfaces, blobs with capped (asymmetric) lists, tiny batches (several
    per level) and a tiny batch-number range (the renumbering of visited words at the wrap)."""
    rng = np.random.default_rng(16)
    n = 1500
    xyz = (rng.random((n, 3)) * np.array([1.0, 1.0, 0.05])).astype(np.float32)
    sem = rng.integers(1, 3, n).astype(np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.06, 50)
    ci, co = o.bfs_cluster(sem, idx, sl, 5)
    for kw in (dict(), dict(T=7), dict(T=5, qmax=4)):
        ci2, co2 = model.bfs_cluster_keys_model(sem, idx, sl, 5, **kw)
        assert np.array_equal(ci, ci2) and np.array_equal(co, co2), kw
    n = 2200
    xyz = rng.normal(0, 0.006, (n, 3)).astype(np.float32)
    xyz[:, 0] += (rng.integers(0, 2, n) * 0.03).astype(np.float32)
    sem = rng.integers(1, 3, n).astype(np.int32)
    bi = np.zeros(n, np.int32); bo = np.array([0, n], np.int32)
    idx, sl = o.ballquery_batch_p(xyz, bi, bo, 0.03, 300)
    assert (sl[:, 1] >= 1000).sum() > 100
    ci, co = o.bfs_cluster(sem, idx, sl, 5)
    for kw in (dict(), dict(T=100)):
        ci2, co2 = model.bfs_cluster_keys_model(sem, idx, sl, 5, **kw)
        assert np.array_equal(ci, ci2) and np.array_equal(co, co2), kw
