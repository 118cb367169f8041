"""GPU parity of the convolution kernels the benchmark ACTUALLY runs, at the size it runs them (BASELINE configs[1]).

`conv2_plan` (csrc/spconv2.hip) sends >= 1024 16-row tiles to the persistent wave-per-tile `spconv_fwd2_kernel<NT,WLDS,XBF>`
and fewer to `spconv_fwd2_split_kernel`; tests/test_sparse_gpu.py only reaches the latter.  Here every case is driven
straight through the C ABI (d3_spconv_pack / d3_spconv_fwd2 / d3_spconv_fwd2_bnbwd / d3_spconv_wgrad2) on the level-0
(142,920 rows) and level-1 (35,127 rows) kernel maps of the canonical scene, `d3_spconv_fwd2_plan`
asserts the variant, and the result is compared with oracle/sparse_oracle.py (reference semantics:
model/common.py:32-41,88-98; model/pointgroup.py:69-74):
  * <= 1e-4 (max-norm, relative to the output scale) against the oracle's "bf16" mode -- operands rounded to bf16, exact
    products, fp32 accumulation: the kernels' arithmetic restated; only the summation order differs;
  * <= 2e-2 against the fp32 oracle (bf16 operand rounding: ~2^-9 * sqrt(K * Cin) relative).
Epilogues covered: residual add, accumulate-into, strided output (concat halves), BatchNorm statistics partials, the
BatchNorm-backward data-gradient epilogue; weight gradient: row splits + the fixed-order reduction, x- and dy-stationary,
bf16 and fp32 operands.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so

pytestmark = pytest.mark.gpu

FLIPK, TRANSW, XSTAT, ACCUM, XBF16, DYBF16 = 1, 2, 8, 16, 32, 64


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def canon(dev):
    """coordinate levels 0/1 of the canonical scene: device kernel maps (bit-checked against the oracle's) + oracle maps"""
    from d3net_amd import minkowski as ME, synthetic as S
    occ, _, _, _ = S.occupancy_grid()
    vox = np.argwhere(occ)
    coords = np.concatenate([np.zeros((len(vox), 1), np.int64), vox], 1)
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    ocm = so.OracleCoords(coords)
    lv = {}
    for level, ts in ((0, 1), (1, 2)):
        nbr = cm.k3(ts)
        child, up, Mo = cm.down(ts)
        parent, kidx, oMo = ocm.get_down(ts)
        assert Mo == oMo
        onbr = ocm.get_k3(ts)
        assert np.array_equal(nbr.cpu().numpy(), onbr), "full-size kernel map differs from the oracle's"
        M = nbr.size(0)
        ref_up = np.full((M, 8), -1); ref_up[np.arange(M), kidx] = parent
        ref_child = np.full((Mo, 8), -1); ref_child[parent, kidx] = np.arange(M)
        assert np.array_equal(up.cpu().numpy(), ref_up) and np.array_equal(child.cpu().numpy(), ref_child)
        lv[level] = dict(M=M, Mo=Mo, nbr=nbr, child=child, up=up, onbr=onbr, parent=parent, kidx=kidx)
    assert lv[0]["M"] == 142920 and lv[1]["M"] == 35127 and lv[1]["Mo"] == 8282
    return lv


def _geom(canon, level, kind):
    """-> (forward table, transposed table, Min, Mout, K, flip-on-backward, oracle conv fn(x, W))"""
    g = canon[level]
    if kind == "k3":
        return g["nbr"], g["nbr"], g["M"], g["M"], 27, FLIPK, lambda x, W: so.conv_k3(x, W, g["onbr"])
    if kind == "down":
        return g["child"], g["up"], g["M"], g["Mo"], 8, 0, lambda x, W: so.conv_down(x, W, g["parent"], g["kidx"], g["Mo"])
    if kind == "up":     # from the coarser level `level + 1` back onto `level`
        return g["up"], g["child"], g["Mo"], g["M"], 8, 0, lambda x, W: so.conv_up(x, W, g["parent"], g["kidx"])
    return None, None, g["M"], g["M"], 1, 0, lambda x, W: so.mm(x, W[0])


def _plan(L, Mout, K, Cin, Cout):
    out = (C.c_int * 6)()
    assert L.d3_spconv_fwd2_plan(Mout, K, Cin, Cout, out) == 0
    return dict(split=out[0], waves=out[1], grid=out[2], wlds=out[3], ntw=out[4], gy=out[5])


def _pack(L, W3, flags, dev):
    from d3net_amd.pointgroup_ops import _ptr, _stream
    K, a, b = W3.shape
    Cin, Cout = (b, a) if flags & TRANSW else (a, b)
    wp = torch.empty(L.d3_spconv_pack_bytes(K, Cin, Cout), dtype=torch.uint8, device=dev)
    assert L.d3_spconv_pack(_ptr(W3), _ptr(wp), K, Cin, Cout, flags, _stream()) == 0
    return wp


def _fwd2(L, x, tbl, wp, out, Mout, K, Cin, Cout, flags=0, res=None, part=None, col0=0):
    from d3net_amd.pointgroup_ops import _ptr, _stream
    o = out[:, col0:]
    rc = L.d3_spconv_fwd2(_ptr(x), x.stride(0), _ptr(tbl) if tbl is not None else None, _ptr(wp),
                          C.c_void_p(o.data_ptr()), out.stride(0), _ptr(res) if res is not None else None,
                          res.stride(0) if res is not None else 0, _ptr(part) if part is not None else None,
                          x.size(0), Mout, K, Cin, Cout, flags, _stream())
    assert rc == 0, rc


# (level, kind, Cin, Cout): the layer shapes of levels 0 and 1 of the backbone (model/common.py:32,38,41,90,98)
CASES = [(0, "k3", 16, 16), (0, "k3", 32, 16), (0, "k3", 136, 16), (0, "down", 16, 32), (0, "up", 32, 16), (0, "k1", 32, 16),
         (1, "k3", 32, 32), (1, "k3", 64, 32), (1, "up", 48, 32), (1, "k1", 64, 32)]


@pytest.mark.parametrize("level,kind,cin,cout", CASES)
def test_fwd2_big_kernel_forward_and_epilogues(dev, canon, level, kind, cin, cout):
    from d3net_amd import _lib
    L = _lib.lib()
    tbl_f, tbl_b, Min, Mout, K, _, conv = _geom(canon, level, kind)
    p = _plan(L, Mout, K, cin, cout)
    assert p["split"] == 0, p          # the persistent wave-per-tile kernel, not the few-row split kernel
    rng = np.random.default_rng(level * 100 + cin + cout)
    x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32))
    so.set_precision("fp32")
    with torch.no_grad():
        ref32 = conv(x, W)
        so.set_precision("bf16")
        try:
            refb = conv(x, W)
            xq = x.bfloat16()
            refq = conv(xq.float(), W)            # bf16-stored input (the BN->ReLU->conv operand)
        finally:
            so.set_precision("fp32")
    xd, Wd, resd = x.to(dev), W.to(dev).contiguous(), res.to(dev)
    wp = _pack(L, Wd, 0, dev)
    # (1) fp32 input, plain store
    out = torch.full((Mout, cout), float("nan"), device=dev)
    _fwd2(L, xd, tbl_f, wp, out, Mout, K, cin, cout)
    assert relerr(out, refb) < 1e-4, relerr(out, refb)
    assert relerr(out, ref32) < 2e-2, relerr(out, ref32)
    # (2) bf16 input + residual + BatchNorm partials + strided output (second half of a concatenated buffer)
    nparts = L.d3_spconv_fwd2_nparts(Mout, K, cin, cout)
    assert nparts == p["grid"]
    pw = (cout + 15) // 16 * 16
    part = torch.full((nparts, 2, pw), float("nan"), device=dev)
    wide = torch.full((Mout, 2 * cout), -7.0, device=dev)
    _fwd2(L, xq.to(dev), tbl_f, wp, wide, Mout, K, cin, cout, flags=XBF16, res=resd, part=part, col0=cout)
    got = wide[:, cout:]
    want = refq + res
    assert relerr(got, want) < 1e-4, relerr(got, want)
    assert bool((wide[:, :cout] == -7.0).all()), "strided store touched the other half of the buffer"
    s1 = part[:, 0, :cout].double().sum(0).cpu(); s2 = part[:, 1, :cout].double().sum(0).cpu()
    g64 = got.double().cpu()
    assert float((s1 - g64.sum(0)).abs().max() / g64.abs().sum(0).max()) < 1e-5
    assert float((s2 - (g64 * g64).sum(0)).abs().max() / (g64 * g64).sum(0).max()) < 1e-5
    # (3) accumulate-into (second contribution to a gradient buffer)
    acc = resd.clone()
    _fwd2(L, xd, tbl_f, wp, acc, Mout, K, cin, cout, flags=ACCUM)
    assert relerr(acc, refb + res) < 1e-4


@pytest.mark.parametrize("level,cin,cout", [(0, 16, 16), (0, 32, 16), (0, 136, 16), (0, 16, 32), (1, 32, 32), (1, 64, 32)])
def test_fwd2_offset_compaction_and_offset_split_equal_the_plain_kernel(dev, canon, level, cin, cout):
    """Round 5: spconv_fwd2_c_kernel (the offsets no row of a 16-row tile has are dropped before the reduction loop: D3_C2_COMPACT,
    1 = the stem only, 2 = every statically shaped instance) adds the same products in the same order as the plain kernel --
    outputs and BatchNorm partials are bit-equal; spconv_fwd2_ks_kernel (the stem, D3_C2_KSPLIT=1: four waves per tile, a quarter of
    the offsets each, partial sums through LDS) adds them in another order -- equal to fp32 rounding."""
    from d3net_amd import _lib
    L = _lib.lib()
    tbl_f, _, Min, Mout, K, _, _ = _geom(canon, level, "k3")
    rng = np.random.default_rng(900 + level * 100 + cin + cout)
    x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32)).bfloat16().to(dev)
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)).to(dev).contiguous()
    res = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32)).to(dev)
    wp = _pack(L, W, 0, dev)
    nparts = L.d3_spconv_fwd2_nparts(Mout, K, cin, cout)
    pw = (cout + 15) // 16 * 16

    def run():
        out = torch.full((Mout, cout), float("nan"), device=dev)
        part = torch.full((nparts, 2, pw), float("nan"), device=dev)
        _fwd2(L, x, tbl_f, wp, out, Mout, K, cin, cout, flags=XBF16, res=res, part=part)
        torch.cuda.synchronize()
        return out, part[:, :, :cout].clone()
    try:
        assert L.d3_tuning_set(b"D3_C2_KSPLIT", 0) == 0 and L.d3_tuning_set(b"D3_C2_COMPACT", 0) == 0
        o0, p0 = run()
        assert bool(torch.isfinite(o0).all())
        for mode in (1, 2):
            assert L.d3_tuning_set(b"D3_C2_COMPACT", mode) == 0
            o, pp = run()
            assert torch.equal(o, o0) and torch.equal(pp, p0), (mode, cin, cout)
        if cin == 136:
            assert L.d3_tuning_set(b"D3_C2_COMPACT", 0) == 0 and L.d3_tuning_set(b"D3_C2_KSPLIT", 1) == 0
            o, pp = run()
            assert relerr(o, o0) < 1e-5 and relerr(pp, p0) < 1e-5
    finally:
        L.d3_tuning_set(b"D3_C2_KSPLIT", 0); L.d3_tuning_set(b"D3_C2_COMPACT", 1)


@pytest.mark.parametrize("level,kind,cin,cout", [c for c in CASES if c[2] != 136])
def test_fwd2_big_kernel_data_gradient_and_bn_backward_epilogue(dev, canon, level, kind, cin, cout):
    """dgrad = the same kernel over the transposed map with W^T (packed with FLIPK | TRANSW); and, as the data gradient of
    a BN -> ReLU -> conv unit, the ReLU mask and the two BatchNorm-backward reductions in its epilogue"""
    from d3net_amd import _lib
    from d3net_amd.pointgroup_ops import _ptr, _stream
    L = _lib.lib()
    tbl_f, tbl_b, Min, Mout, K, flip, conv = _geom(canon, level, kind)
    p = _plan(L, Min, K, cout, cin)
    if p["split"]:
        pytest.skip("input level has < 16384 rows: the split kernel (tests/test_sparse_gpu.py) runs this data gradient")
    rng = np.random.default_rng(level * 100 + cin + cout + 7)
    x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32))
    so.set_precision("bf16")
    try:
        xo = x.clone().requires_grad_(True)
        conv(xo, W).backward(dy)
    finally:
        so.set_precision("fp32")
    dxb = xo.grad
    wp = _pack(L, W.to(dev).contiguous(), flip | TRANSW, dev)
    dyd = dy.to(dev)
    out = torch.full((Min, cin), float("nan"), device=dev)
    _fwd2(L, dyd, tbl_b, wp, out, Min, K, cout, cin)
    assert relerr(out, dxb) < 1e-4, relerr(out, dxb)
    # BatchNorm-backward epilogue: bnx = the BatchNorm input of the unit, (mean, var) its batch statistics
    eps = 1e-4
    bnx = torch.from_numpy((rng.standard_normal((Min, cin)) * 1.5 + 0.3).astype(np.float32))
    gamma = torch.from_numpy((rng.random(cin) + 0.5).astype(np.float32)); beta = torch.from_numpy((rng.standard_normal(cin) * 0.3).astype(np.float32))
    mean = bnx.mean(0); var = bnx.var(0, unbiased=False)
    xh = (bnx - mean) * torch.rsqrt(var + eps)
    pre = xh.double() * gamma.double() + beta.double()
    g = torch.where(pre > 0, dxb.double(), torch.zeros_like(pre))
    sure = (pre.abs() > 1e-5)                               # elements whose ReLU mask does not hinge on the last ulp
    nparts = L.d3_spconv_fwd2_nparts(Min, K, cout, cin)
    pw = (cin + 15) // 16 * 16
    part = torch.full((nparts, 2, pw), float("nan"), device=dev)
    out2 = torch.full((Min, cin), float("nan"), device=dev)
    bnxd, meand, vard, gammad, betad = (t.to(dev) for t in (bnx, mean, var, gamma, beta))   # (kept alive across the call)
    rc = L.d3_spconv_fwd2_bnbwd(_ptr(dyd), cout, _ptr(tbl_b) if tbl_b is not None else None, _ptr(wp), _ptr(out2), cin, _ptr(part),
                                _ptr(bnxd), cin, _ptr(meand), _ptr(vard), _ptr(gammad), _ptr(betad),
                                eps, 1, Mout, Min, K, cout, cin, 0, _stream())
    assert rc == 0
    o2 = out2.cpu().double()
    assert float(((o2 - g).abs() * sure).max() / g.abs().max()) < 1e-4
    assert int((~sure).sum()) < 256
    s1 = part[:, 0, :cin].double().sum(0).cpu(); s2 = part[:, 1, :cin].double().sum(0).cpu()
    assert float((s1 - o2.sum(0)).abs().max() / o2.abs().sum(0).max()) < 1e-5
    ref2 = (o2 * xh.double()).sum(0)
    assert float((s2 - ref2).abs().max() / (o2 * xh.double()).abs().sum(0).max()) < 1e-5


@pytest.mark.parametrize("xbf", [False, True])
@pytest.mark.parametrize("level,kind,cin,cout", CASES)
def test_wgrad2_row_splits_and_reduction(dev, canon, level, kind, cin, cout, xbf):
    """d3_spconv_wgrad2 at canonical row counts: row-split partials + fixed-order reduction; the executor's operand choice
    (x-stationary over the transposed map when Cin > Cout) and operand types (x bf16 from BN->ReLU, dy fp32)"""
    from d3net_amd import _lib
    from d3net_amd.pointgroup_ops import _ptr, _stream
    L = _lib.lib()
    tbl_f, tbl_b, Min, Mout, K, flip, conv = _geom(canon, level, kind)
    rng = np.random.default_rng(level * 100 + cin + cout + 13)
    x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32))
    if xbf:
        x = x.bfloat16().float()
    W = torch.zeros((K, cin, cout), requires_grad=True)
    dy = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32))
    so.set_precision("bf16")
    try:
        conv(x, W).backward(dy)
    finally:
        so.set_precision("fp32")
    ref = W.grad
    xstat = cin > cout
    flags = (XSTAT | flip) if xstat else 0
    tbl = tbl_b if xstat else tbl_f
    splits = L.d3_spconv_wgrad2_splits(Min, Mout, K, cin, cout, flags | (XBF16 if xbf else 0))
    assert splits > 1, "canonical levels 0/1 are row-split"
    ws = torch.empty(max(L.d3_spconv_wgrad2_ws_bytes(Min, Mout, K, cin, cout, flags | (XBF16 if xbf else 0)), 16), dtype=torch.uint8, device=dev)
    dW = torch.full((K, cin, cout), float("nan"), device=dev)
    xd = x.to(dev).bfloat16() if xbf else x.to(dev)
    dyd = dy.to(dev)
    rc = L.d3_spconv_wgrad2(_ptr(xd), cin, _ptr(tbl) if tbl is not None else None, _ptr(dyd), cout, _ptr(dW), Min, Mout, K, cin, cout,
                            cin, flags | (XBF16 if xbf else 0), _ptr(ws), ws.numel(), _stream())
    assert rc == 0, rc
    assert relerr(dW, ref) < 1e-4, relerr(dW, ref)
    # accumulate semantics (a second backward before zero_grad)
    rc = L.d3_spconv_wgrad2(_ptr(xd), cin, _ptr(tbl) if tbl is not None else None, _ptr(dyd), cout, _ptr(dW), Min, Mout, K, cin, cout,
                            cin, flags | ACCUM | (XBF16 if xbf else 0), _ptr(ws), ws.numel(), _stream())
    assert rc == 0
    assert relerr(dW, 2 * ref) < 1e-4
    if xbf:
        # bf16 dy (the executor's single-consumer gradient buffers): rounding at the store instead of at the load -- the
        # very same MFMA operands, hence the very same result as with the fp32 dy
        fl = flags | XBF16 | DYBF16
        assert L.d3_spconv_wgrad2_splits(Min, Mout, K, cin, cout, fl) >= 1
        wsb = torch.empty(max(L.d3_spconv_wgrad2_ws_bytes(Min, Mout, K, cin, cout, fl), 16), dtype=torch.uint8, device=dev)
        dW1 = torch.full((K, cin, cout), float("nan"), device=dev)
        dW2 = torch.full((K, cin, cout), float("nan"), device=dev)
        dyb = dyd.bfloat16()
        assert L.d3_spconv_wgrad2(_ptr(xd), cin, _ptr(tbl) if tbl is not None else None, _ptr(dyd), cout, _ptr(dW1), Min, Mout, K, cin, cout,
                                  cin, flags | XBF16, _ptr(ws), ws.numel(), _stream()) == 0
        assert L.d3_spconv_wgrad2(_ptr(xd), cin, _ptr(tbl) if tbl is not None else None, _ptr(dyb), cout, _ptr(dW2), Min, Mout, K, cin, cout,
                                  cin, fl, _ptr(wsb), wsb.numel(), _stream()) == 0
        assert torch.equal(dW1, dW2)


# ------------------------------------------------------------------------------------------- reference precision (D3_CONV_F32)
F32 = 256


@pytest.mark.parametrize("level,kind,cin,cout", CASES + [(1, "down", 32, 48)])
def test_f32_mfma_kernels_match_the_fp32_oracle(dev, canon, level, kind, cin, cout):
    """D3_CONV_F32 (csrc/spconv2.hip): fp32 operands, exact fp32 products on v_mfma_f32_16x16x4_f32, fp32 accumulation -- the
    reference's precision (MinkowskiEngine is fp32 throughout: model/common.py:32-41).  Forward, data gradient and weight
    gradient at canonical row counts against the fp32 oracle: 2e-6 of the output scale (summation order only), i.e. four
    orders of magnitude below the bf16 path's 2e-2.  (1, down, 32 -> 48) lands on 8,282 rows: the few-row split kernel."""
    from d3net_amd import _lib
    from d3net_amd.pointgroup_ops import _ptr, _stream
    L = _lib.lib()
    tbl_f, tbl_b, Min, Mout, K, flip, conv = _geom(canon, level, kind)
    rng = np.random.default_rng(level * 100 + cin + cout + 31)
    x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32)).requires_grad_(True)
    W = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((Mout, cout)).astype(np.float32))
    so.set_precision("fp32")
    ref = conv(x, W)
    ref.backward(dy)
    xd, Wd, dyd, resd = x.detach().to(dev), W.detach().to(dev).contiguous(), dy.to(dev), res.to(dev)

    def pack(flags):
        Kk, a, b = Wd.shape
        ci, co = (b, a) if flags & TRANSW else (a, b)
        wp = torch.empty(L.d3_spconv_pack_bytes_ex(Kk, ci, co, F32), dtype=torch.uint8, device=dev)
        assert L.d3_spconv_pack(_ptr(Wd), _ptr(wp), Kk, ci, co, flags | F32, _stream()) == 0
        return wp

    # forward (+ residual epilogue)
    out = torch.full((Mout, cout), float("nan"), device=dev)
    _fwd2(L, xd, tbl_f, pack(0), out, Mout, K, cin, cout, flags=F32, res=resd)
    assert relerr(out, ref.detach() + res) < 2e-6, relerr(out, ref.detach() + res)
    # the bf16-input flag is refused with D3_CONV_F32
    assert L.d3_spconv_fwd2(_ptr(xd), cin, _ptr(tbl_f) if tbl_f is not None else None, _ptr(pack(0)), _ptr(out), cout, None, 0, None,
                            Min, Mout, K, cin, cout, F32 | XBF16, _stream()) == -3
    # data gradient
    if cin % 4 == 0 and cin != 136:
        dx = torch.full((Min, cin), float("nan"), device=dev)
        _fwd2(L, dyd, tbl_b, pack(flip | TRANSW), dx, Min, K, cout, cin, flags=F32)
        assert relerr(dx, x.grad) < 2e-6, relerr(dx, x.grad)
    # weight gradient (x- or dy-stationary as the executor picks), accumulate semantics
    xstat = cin > cout
    flags = ((XSTAT | flip) if xstat else 0) | F32
    tbl = tbl_b if xstat else tbl_f
    ws = torch.empty(max(L.d3_spconv_wgrad2_ws_bytes(Min, Mout, K, cin, cout, flags), 16), dtype=torch.uint8, device=dev)
    dW = torch.full((K, cin, cout), float("nan"), device=dev)
    args = (_ptr(xd), cin, _ptr(tbl) if tbl is not None else None, _ptr(dyd), cout, _ptr(dW), Min, Mout, K, cin, cout, cin)
    assert L.d3_spconv_wgrad2(*args, flags, _ptr(ws), ws.numel(), _stream()) == 0
    assert relerr(dW, W.grad) < 5e-6, relerr(dW, W.grad)
    assert L.d3_spconv_wgrad2(*args, flags | ACCUM, _ptr(ws), ws.numel(), _stream()) == 0
    assert relerr(dW, 2 * W.grad) < 5e-6
