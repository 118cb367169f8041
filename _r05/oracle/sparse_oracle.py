"""fp32 CPU restatement of the MinkowskiEngine subset the reference's detector uses
(TEST INFRASTRUCTURE ONLY -- never imported by d3net_amd/).

Reference call sites: model/common.py:13-15,32,36-41,64-66,88-90,96-98,114; model/pointgroup.py:65,70,73,91,176,268.
MinkowskiEngine itself is a third-party dependency that is NOT vendored and NOT version-pinned by the
reference (README.md:25-29; era-consistent release 0.5.x), and the reference holds no tests or vectors at
that boundary: PARITY UNPINNED.  The semantics are therefore fixed here from ME's published definition of
the generalized sparse convolution and pinned against dense torch.nn.functional.conv3d /
conv_transpose3d on small grids (tests/test_oracle_sparse.py):

  conv k3 s1 : out[u] = sum_{o in {-1,0,1}^3, u+o*ts active} x[u+o*ts] @ W[k(o)],  out coords = in coords
  conv k2 s2 : out coords = unique(floor(c/(2ts))*2ts) (first-occurrence order);
               out[p] = sum_{o in {0,1}^3, p+o*ts active} x[p+o*ts] @ W[k(o)]
  convT k2 s2: onto the cached fine coordinates: out[p+o*ts] = x[p] @ W[k(o)]
  k1         : out = x @ W   (W is (Cin,Cout): ME keeps a volume-1 kernel 2-D and uses a plain mm)
  kernel index k(o) = (ox-o0) + K*(oy-o0) + K*K*(oz-o0)  (x fastest; a permutation hook exists in the
  product for real ME checkpoints -- the true ME order is unverifiable offline).
Weights are (K^3, Cin, Cout), no bias.  Algorithm: gather -> mm -> index_add per kernel offset (what ME's
CPU backend does), differentiable through torch autograd, which also provides the backward oracle.
"""
import numpy as np
import torch


def _key(c):
    c = np.asarray(c, np.int64)
    return ((c[:, 0] << 48) + ((c[:, 1] + 32768) << 32) + ((c[:, 2] + 32768) << 16) + (c[:, 3] + 32768))


def _lookup(coords):
    return {int(k): i for i, k in enumerate(_key(coords))}


def kmap_k3(coords, ts=1):
    """(M,27) int64 table: row of coords[u] + o*ts or -1; k = (ox+1) + 3(oy+1) + 9(oz+1)."""
    coords = np.asarray(coords, np.int64)
    M = coords.shape[0]
    keys = _key(coords)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    tbl = np.full((M, 27), -1, np.int64)
    k = 0
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                q = coords.copy()
                q[:, 1] += ox * ts; q[:, 2] += oy * ts; q[:, 3] += oz * ts
                qk = _key(q)
                pos = np.searchsorted(skeys, qk)
                pos = np.clip(pos, 0, M - 1)
                hit = skeys[pos] == qk
                tbl[hit, k] = order[pos[hit]]
                k += 1
    return tbl


def kmap_down(coords, ts=1):
    """stride-2 map: (out_coords (Mo,4), parent (M), kidx (M)); out coords in first-occurrence order."""
    coords = np.asarray(coords, np.int64)
    s = 2 * ts
    p = coords.copy()
    p[:, 1:] = np.floor_divide(coords[:, 1:], s) * s
    keys = _key(p)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    rank = np.argsort(np.argsort(first))  # unique id (sorted by key) -> first-occurrence rank
    parent = rank[inv]
    out_coords = np.zeros((len(first), 4), np.int64)
    out_coords[rank] = p[first]
    d = (coords[:, 1:] - p[:, 1:]) // ts
    kidx = d[:, 0] + 2 * d[:, 1] + 4 * d[:, 2]
    return out_coords, parent.astype(np.int64), kidx.astype(np.int64)


# ------------------------------------------------------------------------------- arithmetic modes
# "fp32": plain fp32 matmuls (the reference's precision).
# "bf16": the arithmetic of the product's MFMA kernels, restated exactly: both matmul operands rounded to
#         bf16 (round-to-nearest-even), products exact, fp32 accumulation -- in the forward (x, W), the data
#         gradient (dy, W) and the weight gradient (x, dy).  Only the summation order differs from the kernels.
_PRECISION = "fp32"


def set_precision(p):
    global _PRECISION
    assert p in ("fp32", "bf16")
    _PRECISION = p


def _rb(t):
    return t.bfloat16().float()


class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return _rb(x) @ _rb(W)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dyr = _rb(dy)
        return dyr @ _rb(W).t(), _rb(x).t() @ dyr


def mm(x, W):
    return _MM.apply(x, W) if _PRECISION == "bf16" else x @ W


def conv_k3(x, W, tbl):
    """x (M,Cin), W (27,Cin,Cout), tbl (M,27) -> (M,Cout)"""
    out = x.new_zeros((x.shape[0], W.shape[2]))
    tbl_t = torch.as_tensor(tbl)
    for k in range(27):
        col = tbl_t[:, k]
        o = torch.nonzero(col >= 0).squeeze(1)
        if o.numel():
            out = out.index_add(0, o, mm(x[col[o]], W[k]))
    return out


def conv_down(x, W, parent, kidx, Mout):
    """x (M,Cin) fine, W (8,Cin,Cout) -> (Mout,Cout)"""
    out = x.new_zeros((Mout, W.shape[2]))
    parent_t, kidx_t = torch.as_tensor(parent), torch.as_tensor(kidx)
    for k in range(8):
        i = torch.nonzero(kidx_t == k).squeeze(1)
        if i.numel():
            out = out.index_add(0, parent_t[i], mm(x[i], W[k]))
    return out


def conv_up(x, W, parent, kidx):
    """x (Mcoarse,Cin), W (8,Cin,Cout) -> (Mfine,Cout) on the cached fine coordinates"""
    parent_t, kidx_t = torch.as_tensor(parent), torch.as_tensor(kidx)
    out = x.new_zeros((parent_t.shape[0], W.shape[2]))
    for k in range(8):
        i = torch.nonzero(kidx_t == k).squeeze(1)
        if i.numel():
            out = out.index_copy(0, i, mm(x[parent_t[i]], W[k]))
    return out


def bn_relu(x, gamma, beta, eps=1e-4, relu=True, running=None, momentum=0.1, training=True):
    """MinkowskiBatchNorm == BatchNorm1d over the rows (reference: model/pointgroup.py:65) + MinkowskiReLU."""
    rm, rv = running if running is not None else (None, None)
    y = torch.nn.functional.batch_norm(x, rm, rv, gamma, beta, training, momentum, eps)
    return torch.relu(y) if relu else y


# ----------------------------------------------------------------------------------------- modules
class OracleCoords:
    """coordinate sets + kernel maps per tensor stride (what ME's CoordinateManager caches)."""

    def __init__(self, coords):
        self.levels = {1: np.asarray(coords, np.int64)}
        self.k3 = {}
        self.down = {}

    def get_k3(self, ts):
        if ts not in self.k3:
            self.k3[ts] = kmap_k3(self.levels[ts], ts)
        return self.k3[ts]

    def get_down(self, ts):
        if ts not in self.down:
            oc, parent, kidx = kmap_down(self.levels[ts], ts)
            self.levels[2 * ts] = oc
            self.down[ts] = (parent, kidx, oc.shape[0])
        return self.down[ts]


class OracleUNet(torch.nn.Module):
    """The reference's backbone: stem conv + UBlock(ResidualBlock) + BN + ReLU, parameter-for-parameter
    (reference: model/pointgroup.py:69-74, model/common.py:22-53,73-118).  Parameters are created by the
    caller (shared with the HIP model) as a flat name->tensor dict with the reference's state-dict names."""

    def __init__(self, params, nPlanes, block_reps=2, eps=1e-4, prefix="1", training=True):
        super().__init__()
        self.bn_training = training     # False: running statistics (model.eval() of the reference's MinkowskiBatchNorm)
        self.p = params
        self.nPlanes = list(nPlanes)
        self.reps = block_reps
        self.eps = eps
        self.prefix = prefix

    def _bn(self, x, name, relu=True):
        if not self.bn_training:
            return bn_relu(x, self.p[name + ".bn.weight"], self.p[name + ".bn.bias"], self.eps, relu,
                           running=(self.p[name + ".bn.running_mean"], self.p[name + ".bn.running_var"]), training=False)
        return bn_relu(x, self.p[name + ".bn.weight"], self.p[name + ".bn.bias"], self.eps, relu)

    def _res(self, x, name, cm, ts, cin, cout):
        identity = x
        h = self._bn(x, name + ".conv_branch.0")
        h = conv_k3(h, self.p[name + ".conv_branch.2.kernel"], cm.get_k3(ts))
        h = self._bn(h, name + ".conv_branch.3")
        h = conv_k3(h, self.p[name + ".conv_branch.5.kernel"], cm.get_k3(ts))
        if cin != cout:
            identity = mm(identity, self.p[name + ".downsample.0.kernel"])  # (Cin,Cout): ME stores a 1x1 kernel 2-D
        return h + identity

    def ublock(self, x, cm, ts, planes, name):
        c = planes[0]
        for i in range(self.reps):
            x = self._res(x, "%s.blocks.block%d" % (name, i), cm, ts, c, c)
        identity = x
        if len(planes) > 1:
            parent, kidx, Mo = cm.get_down(ts)
            h = self._bn(x, name + ".conv.0")
            h = conv_down(h, self.p[name + ".conv.2.kernel"], parent, kidx, Mo)
            h = self.ublock(h, cm, 2 * ts, planes[1:], name + ".u")
            h = self._bn(h, name + ".deconv.0")
            h = conv_up(h, self.p[name + ".deconv.2.kernel"], parent, kidx)
            x = torch.cat([identity, h], 1)
            for i in range(self.reps):
                x = self._res(x, "%s.blocks_tail.block%d" % (name, i), cm, ts, c * (2 - i), c)
        return x

    def forward(self, x, cm):
        return self.ublock(x, cm, 1, self.nPlanes, self.prefix)
