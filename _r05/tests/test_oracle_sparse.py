"""Pin the sparse-conv oracle (oracle/sparse_oracle.py) against dense torch convolutions on small grids.

MinkowskiEngine is absent and unpinned in the reference ("parity unpinned"); the
semantics the build fixes are exactly: sparse conv == dense conv sampled at the active sites."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sparse_oracle as so


def _random_sparse(rng, dims, occ, C, batch=2):
    coords, feats = [], []
    for b in range(batch):
        m = rng.random(dims) < occ
        c = np.argwhere(m)
        c = c[rng.permutation(len(c))]
        coords.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    coords = np.concatenate(coords).astype(np.int64)
    feats = torch.from_numpy(rng.standard_normal((len(coords), C)).astype(np.float32))
    return coords, feats


def _dense(coords, feats, dims, batch):
    d = torch.zeros((batch, feats.shape[1]) + tuple(dims[::-1]))  # (B,C,Z,Y,X)
    c = torch.as_tensor(coords)
    d[c[:, 0], :, c[:, 3], c[:, 2], c[:, 1]] = feats
    return d


def _w_dense(W, K):
    # W (K^3, Cin, Cout) with k = ox + K*oy + K*K*oz  ->  (Cout, Cin, Z, Y, X)
    return W.view(K, K, K, W.shape[1], W.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


def test_conv_k3_matches_dense():
    rng = np.random.default_rng(0)
    dims = (7, 6, 5)
    coords, x = _random_sparse(rng, dims, 0.4, 5)
    W = torch.from_numpy(rng.standard_normal((27, 5, 4)).astype(np.float32))
    out = so.conv_k3(x, W, so.kmap_k3(coords, 1))
    dense = F.conv3d(_dense(coords, x, dims, 2), _w_dense(W, 3), padding=1)
    c = torch.as_tensor(coords)
    ref = dense[c[:, 0], :, c[:, 3], c[:, 2], c[:, 1]]
    assert torch.allclose(out, ref, atol=1e-4)


def test_conv_down_and_up_match_dense():
    rng = np.random.default_rng(1)
    dims = (8, 6, 6)
    coords, x = _random_sparse(rng, dims, 0.3, 4)
    W = torch.from_numpy(rng.standard_normal((8, 4, 3)).astype(np.float32))
    oc, parent, kidx = so.kmap_down(coords, 1)
    out = so.conv_down(x, W, parent, kidx, oc.shape[0])
    dense = F.conv3d(_dense(coords, x, dims, 2), _w_dense(W, 2), stride=2)
    c = torch.as_tensor(oc)
    ref = dense[c[:, 0], :, c[:, 3] // 2, c[:, 2] // 2, c[:, 1] // 2]
    assert torch.allclose(out, ref, atol=1e-4)
    # output coordinate set == cells with >= 1 active child, first-occurrence order
    assert len(np.unique(so._key(oc))) == oc.shape[0]
    firsts = [np.nonzero(parent == p)[0][0] for p in range(oc.shape[0])]
    assert firsts == sorted(firsts)
    # transposed conv back onto the fine coordinates
    Wt = torch.from_numpy(rng.standard_normal((8, 3, 5)).astype(np.float32))
    up = so.conv_up(out, Wt, parent, kidx)
    cdims = tuple(d // 2 for d in dims)
    coarse = torch.zeros((2, 3) + cdims[::-1])
    coarse[c[:, 0], :, c[:, 3] // 2, c[:, 2] // 2, c[:, 1] // 2] = out
    wd = Wt.view(2, 2, 2, 3, 5).permute(3, 4, 0, 1, 2).contiguous()  # (Cin, Cout, Z, Y, X)
    dense_up = F.conv_transpose3d(coarse, wd, stride=2)
    f = torch.as_tensor(coords)
    ref_up = dense_up[f[:, 0], :, f[:, 3], f[:, 2], f[:, 1]]
    assert torch.allclose(up, ref_up, atol=1e-4)


def test_second_level_stride():
    """k3 at tensor stride 2 uses offsets of +-2 and floor-anchored coarse coordinates."""
    rng = np.random.default_rng(2)
    dims = (12, 8, 8)
    coords, x = _random_sparse(rng, dims, 0.25, 3, batch=1)
    oc, parent, kidx = so.kmap_down(coords, 1)
    assert (oc[:, 1:] % 2 == 0).all()
    tbl = so.kmap_k3(oc, 2)
    assert (tbl[:, 13] == np.arange(len(oc))).all()  # centre offset = identity
    xc = torch.from_numpy(rng.standard_normal((len(oc), 3)).astype(np.float32))
    W = torch.from_numpy(rng.standard_normal((27, 3, 2)).astype(np.float32))
    out = so.conv_k3(xc, W, tbl)
    half = oc.copy(); half[:, 1:] //= 2
    cd = tuple(d // 2 for d in dims)
    dense = F.conv3d(_dense(half, xc, cd, 1), _w_dense(W, 3), padding=1)
    h = torch.as_tensor(half)
    assert torch.allclose(out, dense[h[:, 0], :, h[:, 3], h[:, 2], h[:, 1]], atol=1e-4)


def test_canonical_scene_level_sizes():
    """M_l and 27-neighbour pair counts of the canonical scene."""
    from d3net_amd import synthetic as S
    occ, _, _, _ = S.occupancy_grid()
    vox = np.argwhere(occ)
    coords = np.concatenate([np.zeros((len(vox), 1), np.int64), vox], 1)
    cm = so.OracleCoords(coords)
    Ms, Ps, ts = [], [], 1
    for _ in range(7):
        Ms.append(cm.levels[ts].shape[0])
        Ps.append(int((cm.get_k3(ts) >= 0).sum()))
        cm.get_down(ts); ts *= 2
    assert Ms == [142920, 35127, 8282, 1945, 460, 104, 22]
    assert Ps == [1332424, 355069, 88232, 22779, 5710, 1236, 212]
