"""A second, INDEPENDENT pin of the oracle's operator restatements that have no golden vectors from the reference itself
(lib/pointgroup_ops needs nvcc + google::dense_hash_map, neither is in this image, so oracle/pg_ops_oracle.c cannot be
checked against a reference build).  Here the same operators are recomputed with published third-party algorithms that share no code
with the oracle -- numpy's `unique`, scipy's KD-tree and scipy's breadth-first search on a CSR graph -- and must agree BIT FOR BIT:
  voxelize_idx   src/voxelize/voxelize.cpp:66-110     first-occurrence voxel ids, ascending point lists
  ball query     src/bfs_cluster/bfs_cluster.cu:15-60  d2 < r2 (strict, float32), same scene only, ascending neighbour ids
  BFS clustering src/bfs_cluster/bfs_cluster.cpp:28-112 FIFO visitation order from ascending seeds, same-label edges, size >= threshold
This does not turn the oracle into the reference (DESIGN.md 4 keeps "parity unpinned" for these rows); it removes the possibility that
the oracle and the HIP kernels share a misreading that a textbook implementation would not."""
import numpy as np
import pytest

from oracle import pg_oracle as po

scipy_spatial = pytest.importorskip("scipy.spatial")
scipy_sparse = pytest.importorskip("scipy.sparse")
from scipy.sparse import csgraph  # noqa: E402


def _scene(seed, n=3000, nb=3, extent=1.2, nlabels=4):
    """random points of `nb` scenes (sorted by scene), a few duplicated voxels, labels in {0..3}"""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(n // (2 * nb), n // nb, nb)
    xyz = rng.random((int(sizes.sum()), 3)).astype(np.float32) * np.float32(extent)
    batch = np.repeat(np.arange(nb), sizes).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    labels = rng.integers(0, nlabels, xyz.shape[0]).astype(np.int32)
    return xyz, batch, offsets, labels


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_voxelize_idx_equals_numpy_unique_in_first_occurrence_order(seed):
    rng = np.random.default_rng(seed)
    N = 5000
    coords = np.concatenate([rng.integers(0, 3, (N, 1)), rng.integers(0, 12, (N, 3))], 1).astype(np.int64)   # (batch, x, y, z): many collisions
    out_coords, input_map, output_map = po.voxelization_idx(coords, 3, mode=4)
    uniq, first, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                    # unique rows in the order of their first occurrence
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    assert np.array_equal(out_coords, uniq[order])
    assert np.array_equal(input_map, rank[inv.reshape(-1)].astype(np.int32))
    M = len(order)
    assert output_map.shape[0] == M
    for v in range(0, M, 7):
        pts = np.nonzero(rank[inv.reshape(-1)] == v)[0]
        assert output_map[v, 0] == len(pts) and np.array_equal(output_map[v, 1:1 + len(pts)], pts.astype(np.int32))


@pytest.mark.parametrize("seed,radius", [(0, 0.05), (1, 0.08), (2, 0.03)])
def test_ball_query_equals_kdtree_with_the_strict_float32_test(seed, radius):
    xyz, batch, offsets, _ = _scene(seed, extent=0.6)
    idx, start_len = po.ballquery_batch_p(xyz, batch, offsets, radius, 50)
    r2 = np.float32(radius) * np.float32(radius)
    for b in range(len(offsets) - 1):
        lo, hi = int(offsets[b]), int(offsets[b + 1])
        tree = scipy_spatial.cKDTree(xyz[lo:hi].astype(np.float64))
        cand = tree.query_ball_point(xyz[lo:hi].astype(np.float64), radius * 1.001)     # superset; the reference's own test decides
        for i in range(lo, hi, 5):
            c = np.sort(np.asarray(cand[i - lo], np.int64)) + lo
            d = xyz[c] - xyz[i]                                                            # float32, as the kernel computes it
            d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            want = c[d2 < r2]
            if np.any(np.abs(d2 - r2) < 1e-7 * r2):
                continue                                                                   # (a tie of the last float32 bit: FMA contraction may decide)
            got = idx[start_len[i, 0]:start_len[i, 0] + start_len[i, 1]]
            assert np.array_equal(got, want.astype(np.int32)), (b, i)


@pytest.mark.parametrize("seed,radius,threshold", [(0, 0.06, 5), (1, 0.09, 20), (2, 0.07, 2)])
def test_bfs_clustering_equals_scipy_breadth_first_order(seed, radius, threshold):
    xyz, batch, offsets, labels = _scene(seed, n=2400, extent=0.45, nlabels=2)      # dense enough for components of hundreds of points
    idx, start_len = po.ballquery_batch_p(xyz, batch, offsets, radius, 50)
    ci, co = po.bfs_cluster(labels, idx, start_len, threshold)
    assert len(co) > 3 and int(np.diff(co).max()) >= 80          # (the case is not trivial: several kept components, one of them deep)
    n = xyz.shape[0]
    # CSR graph of the same-label edges, neighbours in list (= ascending) order: scipy visits a node's neighbours in CSR order
    rows = np.repeat(np.arange(n), start_len[:, 1])
    cols = idx.astype(np.int64)
    keep = labels[rows] == labels[cols]
    g = scipy_sparse.csr_matrix((np.ones(int(keep.sum()), np.int8), (rows[keep], cols[keep])), shape=(n, n))
    g.sort_indices()
    visited = np.zeros(n, bool)
    members, offs = [], [0]
    for s in range(n):
        if visited[s]:
            continue
        order = csgraph.breadth_first_order(g, s, directed=True, return_predecessors=False)
        assert not visited[order].any()              # (mutual lists: a component is closed)
        visited[order] = True
        if len(order) >= threshold:
            members.append(order); offs.append(offs[-1] + len(order))
    want = np.concatenate(members) if members else np.zeros(0, np.int64)
    assert np.array_equal(co, np.asarray(offs, np.int32))
    assert np.array_equal(ci[:, 1], want.astype(np.int32))
    assert np.array_equal(ci[:, 0], np.repeat(np.arange(len(offs) - 1), np.diff(offs)).astype(np.int32))
