"""GPU parity: sparse convolution / batch-norm kernels and the whole U-Net vs the fp32 CPU oracle.

Kernel maps are integer work: bit-exact.  Convolutions: the exact-fp32 kernels within 1e-4 of the fp32 oracle
(relative to the output scale); the bf16-MFMA kernels (bf16 operands, fp32 accumulate) within 2e-2 of the fp32
oracle -- bf16 has 8 mantissa bits, so a K-term dot product of unit-scale operands carries ~2^-9*sqrt(K)
relative error -- and within 1e-3 of the oracle's "bf16" mode, which restates the kernels' arithmetic exactly
(operands rounded to bf16, exact products, fp32 accumulation; only the summation order differs).
"""
import functools

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so

pytestmark = pytest.mark.gpu


def rand_coords(rng, dims, occ, batch=2):
    cs = []
    for b in range(batch):
        c = np.argwhere(rng.random(dims) < occ)
        c = c[rng.permutation(len(c))] - np.array([3, 0, 2])  # some negative coordinates too
        cs.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    return np.concatenate(cs).astype(np.int64)


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def l2err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_kernel_maps_bit_exact(dev):
    from d3net_amd import minkowski as ME
    rng = np.random.default_rng(0)
    coords = rand_coords(rng, (40, 30, 20), 0.08)
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    ocm = so.OracleCoords(coords)
    ts = 1
    for level in range(4):
        nbr = cm.k3(ts).cpu().numpy()
        assert np.array_equal(nbr, ocm.get_k3(ts)), level
        child, up, Mo = cm.down(ts)
        parent, kidx, oMo = ocm.get_down(ts)
        assert Mo == oMo
        assert np.array_equal(cm.coords[2 * ts].cpu().numpy(), ocm.levels[2 * ts])
        child, up = child.cpu().numpy(), up.cpu().numpy()
        M = len(parent)
        ref_up = np.full((M, 8), -1); ref_up[np.arange(M), kidx] = parent
        ref_child = np.full((Mo, 8), -1); ref_child[parent, kidx] = np.arange(M)
        assert np.array_equal(up, ref_up) and np.array_equal(child, ref_child)
        ts *= 2


CONV_CASES = [("k3", 16, 16), ("k3", 134, 16), ("k3", 32, 48), ("k3", 224, 112), ("down", 16, 32), ("down", 96, 112),
              ("up", 32, 16), ("up", 112, 96), ("k1", 32, 16), ("k1", 224, 112)]


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("kind,cin,cout", CONV_CASES)
def test_conv_fwd_bwd_vs_oracle(dev, kind, cin, cout, exact):
    from d3net_amd import minkowski as ME
    ME.set_exact(exact)
    try:
        rng = np.random.default_rng(hash((kind, cin, cout)) % 1000)
        coords = rand_coords(rng, (24, 20, 12), 0.15)
        cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
        ocm = so.OracleCoords(coords)
        M = coords.shape[0]
        parent, kidx, Mo = ocm.get_down(1)
        cm.down(1)
        if kind == "k3":
            layer = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3); Min, ts_in = M, 1
        elif kind == "down":
            layer = ME.MinkowskiConvolution(cin, cout, kernel_size=2, stride=2, dimension=3); Min, ts_in = M, 1
        elif kind == "up":
            layer = ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3); Min, ts_in = Mo, 2
        else:
            layer = ME.MinkowskiConvolution(cin, cout, kernel_size=1, dimension=3); Min, ts_in = M, 1
        layer = layer.to(dev)
        x = torch.from_numpy(rng.standard_normal((Min, cin)).astype(np.float32))
        W = layer.kernel.detach().cpu().clone().requires_grad_(True)
        xo = x.clone().requires_grad_(True)
        if kind == "k3":
            ref = so.conv_k3(xo, W, ocm.get_k3(1))
        elif kind == "down":
            ref = so.conv_down(xo, W, parent, kidx, Mo)
        elif kind == "up":
            ref = so.conv_up(xo, W, parent, kidx)
        else:
            ref = xo @ W
        g = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        ref.backward(g)
        xd = x.to(dev).requires_grad_(True)
        out = layer(ME.SparseTensor(xd, coordinate_manager=cm, tensor_stride=ts_in))
        out.F.backward(g.to(dev))
        tol = 1e-4 if exact else 2e-2
        assert out.F.shape == ref.shape
        assert relerr(out.F, ref) < tol
        assert relerr(xd.grad, xo.grad) < tol
        assert relerr(layer.kernel.grad, W.grad) < tol
        if not exact:  # against the exact restatement of the kernel arithmetic
            so.set_precision("bf16")
            xb = x.clone().requires_grad_(True); Wb = W.detach().clone().requires_grad_(True)
            if kind == "k3":
                refb = so.conv_k3(xb, Wb, ocm.get_k3(1))
            elif kind == "down":
                refb = so.conv_down(xb, Wb, parent, kidx, Mo)
            elif kind == "up":
                refb = so.conv_up(xb, Wb, parent, kidx)
            else:
                refb = so.mm(xb, Wb)
            refb.backward(g)
            assert relerr(out.F, refb) < 1e-4
            assert relerr(xd.grad, xb.grad) < 1e-4
            assert relerr(layer.kernel.grad, Wb.grad) < 1e-4
    finally:
        ME.set_exact(False)
        so.set_precision("fp32")


@pytest.mark.parametrize("C,relu", [(16, True), (48, False), (112, True)])
def test_batchnorm_relu_fwd_bwd(dev, C, relu):
    from d3net_amd import minkowski as ME
    rng = np.random.default_rng(C)
    M = 5000
    x = torch.from_numpy((rng.standard_normal((M, C)) * 2 + 0.5).astype(np.float32))
    gamma = torch.from_numpy(rng.random(C).astype(np.float32) + 0.5); beta = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.3)
    xo = x.clone().requires_grad_(True); go = gamma.clone().requires_grad_(True); bo = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    ref = so.bn_relu(xo, go, bo, 1e-4, relu, (rm, rv))
    g = torch.from_numpy(rng.standard_normal((M, C)).astype(np.float32))
    ref.backward(g)
    bn = ME.MinkowskiBatchNorm(C, eps=1e-4, momentum=0.1).to(dev)
    bn.fused_relu = relu
    with torch.no_grad():
        bn.bn.weight.copy_(gamma); bn.bn.bias.copy_(beta)
    xd = x.to(dev).requires_grad_(True)
    coords = torch.zeros((M, 4), dtype=torch.int32, device=dev); coords[:, 1] = torch.arange(M, device=dev) % 16000
    out = bn(ME.SparseTensor(xd, coordinates=coords))
    out.F.backward(g.to(dev))
    assert relerr(out.F, ref) < 1e-5
    assert relerr(xd.grad, xo.grad) < 1e-4
    assert relerr(bn.bn.weight.grad, go.grad) < 1e-4 and relerr(bn.bn.bias.grad, bo.grad) < 1e-4
    assert relerr(bn.bn.running_mean, rm) < 1e-5 and relerr(bn.bn.running_var, rv) < 1e-5


def _shared_unet(dev, planes, cin):
    """the HIP backbone and the oracle backbone on the same parameters"""
    from d3net_amd import minkowski as ME, common
    torch.manual_seed(123)
    norm = functools.partial(ME.MinkowskiBatchNorm, eps=1e-4, momentum=0.1)
    net = torch.nn.Sequential(ME.MinkowskiConvolution(cin, planes[0], kernel_size=3, bias=False, dimension=3),
                              common.UBlock(planes, norm, 2, common.ResidualBlock), norm(planes[0]),
                              ME.MinkowskiReLU(inplace=True))
    ME.fuse_bn_relu(net)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bn.weight"):
                p.uniform_(0.5, 1.5)
            if n.endswith("bn.bias"):
                p.uniform_(-0.2, 0.2)
    params = {n: p.detach().clone().requires_grad_(True) for n, p in net.named_parameters()}
    return net.to(dev), params


@pytest.mark.parametrize("exact", [True, False])
def test_unet_forward_backward_vs_oracle(dev, exact):
    from d3net_amd import minkowski as ME
    ME.set_exact(exact)
    so.set_precision("fp32" if exact else "bf16")
    try:
        rng = np.random.default_rng(7)
        planes, cin = [16, 32, 48, 64], 134
        coords = rand_coords(rng, (40, 32, 20), 0.12)
        x = torch.from_numpy(rng.standard_normal((len(coords), cin)).astype(np.float32))
        net, params = _shared_unet(dev, planes, cin)
        # oracle
        ocm = so.OracleCoords(coords)
        xo = x.clone().requires_grad_(True)
        h = so.conv_k3(xo, params["0.kernel"], ocm.get_k3(1))
        h = so.OracleUNet(params, planes).forward(h, ocm)
        ref = so.bn_relu(h, params["2.bn.weight"], params["2.bn.bias"], 1e-4, True)
        g = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        ref.backward(g)
        # HIP
        xd = x.to(dev).requires_grad_(True)
        out = net(ME.SparseTensor(xd, coordinates=torch.from_numpy(coords).int().to(dev)))
        out.F.backward(g.to(dev))
        # Exact mode: relative L2 error.  The ReLU mask of a pre-activation within rounding of 0 may flip, which
        # moves single gradient entries by O(1) (each worth ~1/sqrt(numel) = 0.3 % here), so a max-norm bound is
        # meaningless for the deep net and a handful of flips is expected even between two fp32 summation orders.
        # bf16 mode: rounding operands to bf16 is itself discontinuous, so two implementations that differ only in
        # fp32 summation order decorrelate down to the bf16 noise floor after a few layers (forward ~1e-2), ~0.5 %
        # of the ReLU masks then differ and the end-to-end gradient is only statistically comparable: the bound is
        # a cosine similarity.  The kernels themselves are pinned per layer to 1e-4 against the exact restatement
        # of their arithmetic (test_conv_fwd_bwd_vs_oracle) and the plumbing is shared with the exact mode.
        def cos(a, b):
            a = a.detach().cpu().double().flatten(); b = b.detach().cpu().double().flatten()
            return float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if exact:
            assert l2err(out.F, ref) < 1e-3, l2err(out.F, ref)
            assert l2err(xd.grad, xo.grad) < 2e-2, l2err(xd.grad, xo.grad)
            errs = {n: l2err(p.grad, params[n].grad) for n, p in net.named_parameters()}
            worst = max(errs, key=errs.get)
            assert errs[worst] < 5e-2, (worst, errs[worst])
        else:
            assert l2err(out.F, ref) < 3e-2, l2err(out.F, ref)
            assert cos(xd.grad, xo.grad) > 0.9, cos(xd.grad, xo.grad)
            cs = {n: cos(p.grad, params[n].grad) for n, p in net.named_parameters()}
            worst = min(cs, key=cs.get)
            vals = sorted(cs.values())
            assert vals[len(vals) // 2] > 0.9, vals[len(vals) // 2]      # typical parameter
            assert cs[worst] > 0.5, (worst, cs[worst])                     # no parameter points the wrong way
    finally:
        ME.set_exact(False)
        so.set_precision("fp32")


@pytest.mark.parametrize("stem", [True, False])
def test_native_executor_vs_oracle_and_module_path(dev, stem):
    """csrc/unet.hip (one call per forward / backward, fused epilogues, side-stream weight gradients) against the
    oracle's bf16 restatement and against the module-by-module HIP path on the same parameters."""
    from d3net_amd import minkowski as ME, netexec
    so.set_precision("bf16")
    try:
        rng = np.random.default_rng(11)
        planes, cin = [16, 32, 48, 64], (134 if stem else 16)
        coords = rand_coords(rng, (40, 32, 20), 0.12)
        x = torch.from_numpy(rng.standard_normal((len(coords), cin)).astype(np.float32))
        net, params = _shared_unet(dev, planes, cin)
        ocm = so.OracleCoords(coords)
        xo = x.clone().requires_grad_(True)
        h = so.conv_k3(xo, params["0.kernel"], ocm.get_k3(1)) if stem else xo
        h = so.OracleUNet(params, planes).forward(h, ocm)
        ref = so.bn_relu(h, params["2.bn.weight"], params["2.bn.bias"], 1e-4, True)
        g = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        ref.backward(g)
        cd = torch.from_numpy(coords).int().to(dev)
        # module path
        xm = x.to(dev).requires_grad_(True)
        st = ME.SparseTensor(xm, coordinates=cd)
        out_m = (net(st) if stem else net[3](net[2](net[1](st)))).F
        out_m.backward(g.to(dev))
        gm = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        xm_grad = xm.grad.clone()
        rm_module = net[2].bn.running_mean.clone()
        for p in net.parameters():
            p.grad = None
        # native executor (running statistics restored first: both paths update them)
        net2, _ = _shared_unet(dev, planes, cin)
        ex = netexec.NativeUNet(net2[0] if stem else None, net2[1], net2[2], cin, not stem)
        xn = x.to(dev).requires_grad_(True)
        out_n = ex(xn, ME.CoordinateManager(cd), True)
        out_n.backward(g.to(dev))
        torch.cuda.synchronize()

        def cos(a, b):
            a = a.detach().cpu().double().flatten(); b = b.detach().cpu().double().flatten()
            return float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert l2err(out_n, ref) < 3e-2, l2err(out_n, ref)
        assert l2err(out_n, out_m) < 3e-2, l2err(out_n, out_m)
        assert relerr(net2[2].bn.running_mean, rm_module) < 1e-2
        if not stem:
            assert cos(xn.grad, xo.grad) > 0.9 and cos(xn.grad, xm_grad) > 0.9, (cos(xn.grad, xo.grad), cos(xn.grad, xm_grad))
        names = [n for n, _ in net2.named_parameters() if stem or not n.startswith("0.")]
        cs = {n: cos(dict(net2.named_parameters())[n].grad, params[n].grad) for n in names}
        cm_ = {n: cos(dict(net2.named_parameters())[n].grad, gm[n]) for n in names}
        vals = sorted(cs.values())
        assert vals[len(vals) // 2] > 0.9 and vals[0] > 0.5, (vals[len(vals) // 2], min(cs, key=cs.get), vals[0])
        vals = sorted(cm_.values())
        assert vals[len(vals) // 2] > 0.9 and vals[0] > 0.5, (vals[len(vals) // 2], min(cm_, key=cm_.get), vals[0])
        # a second backward accumulates into the same gradient views (torch semantics)
        g1 = net2[1].blocks.block0.conv_branch[2].kernel.grad.clone()
        out2 = ex(x.to(dev).requires_grad_(True), ME.CoordinateManager(cd), True)
        out2.backward(g.to(dev))
        torch.cuda.synchronize()
        g2 = net2[1].blocks.block0.conv_branch[2].kernel.grad
        assert l2err(g2, 2 * g1) < 5e-2, l2err(g2, 2 * g1)
    finally:
        so.set_precision("fp32")


def test_unet_bf16_forward_close_to_fp32_oracle(dev):
    """stated tolerance of the bf16-MFMA backbone against the fp32 reference arithmetic: 5e-2 relative L2"""
    from d3net_amd import minkowski as ME
    rng = np.random.default_rng(8)
    planes, cin = [16, 32, 48], 134
    coords = rand_coords(rng, (40, 32, 20), 0.12)
    x = torch.from_numpy(rng.standard_normal((len(coords), cin)).astype(np.float32))
    net, params = _shared_unet(dev, planes, cin)
    ocm = so.OracleCoords(coords)
    with torch.no_grad():
        h = so.conv_k3(x, params["0.kernel"], ocm.get_k3(1))
        h = so.OracleUNet(params, planes).forward(h, ocm)
        ref = so.bn_relu(h, params["2.bn.weight"], params["2.bn.bias"], 1e-4, True)
        out = net(ME.SparseTensor(x.to(dev), coordinates=torch.from_numpy(coords).int().to(dev)))
    assert l2err(out.F, ref) < 5e-2, l2err(out.F, ref)


def test_canonical_scene_maps(dev):
    """Full-size (BASELINE config 2) coordinate maps: level sizes and pair counts of SURVEY.md row A3."""
    from d3net_amd import minkowski as ME, synthetic as S
    occ, _, _, _ = S.occupancy_grid()
    vox = np.argwhere(occ)
    coords = np.concatenate([np.zeros((len(vox), 1), np.int64), vox], 1)
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    Ms, Ps, ts = [], [], 1
    for _ in range(7):
        nbr = cm.k3(ts)
        Ms.append(nbr.size(0)); Ps.append(int((nbr >= 0).sum()))
        # symmetry of the neighbour table: nbr[nbr[u,k], 26-k] == u
        u = torch.arange(nbr.size(0), device=dev)
        for k in (0, 5, 13, 20):
            v = nbr[:, k].long(); m = v >= 0
            assert torch.equal(nbr[v[m], 26 - k].long(), u[m])
        cm.down(ts); ts *= 2
    assert Ms == [142920, 35127, 8282, 1945, 460, 104, 22]
    assert Ps == [1332424, 355069, 88232, 22779, 5710, 1236, 212]


def test_native_executor_eval_mode_vgg_and_frozen_parameters(dev):
    """the executor in the configurations the training step does not exercise: eval mode (running statistics), VGG
    blocks (cfg.model.block_residual = False, model/common.py:56-70), frozen parameters (scripts/train.py:312-325
    freeze_*: no gradient may be produced for them), num_batches_tracked bookkeeping"""
    from d3net_amd import minkowski as ME, common, netexec
    rng = np.random.default_rng(21)
    planes, cin = [16, 32, 48], 16
    coords = rand_coords(rng, (32, 28, 16), 0.15)
    cd = torch.from_numpy(coords).int().to(dev)
    x = torch.from_numpy(rng.standard_normal((len(coords), cin)).astype(np.float32)).to(dev)
    norm = functools.partial(ME.MinkowskiBatchNorm, eps=1e-4, momentum=0.1)
    for block in (common.ResidualBlock, common.VGGBlock):
        torch.manual_seed(5)
        net = torch.nn.Sequential(common.UBlock(planes, norm, 2, block), norm(planes[0]), ME.MinkowskiReLU(inplace=True)).to(dev)
        ME.fuse_bn_relu(net)
        ex = netexec.NativeUNet(None, net[0], net[1], cin, True)
        # training forward: same as the module path on a twin (running statistics included)
        twin = torch.nn.Sequential(common.UBlock(planes, norm, 2, block), norm(planes[0]), ME.MinkowskiReLU(inplace=True)).to(dev)
        twin.load_state_dict(net.state_dict())
        ME.fuse_bn_relu(twin)
        net.train(); twin.train()
        out_n = ex(x.clone().requires_grad_(True), ME.CoordinateManager(cd), True)
        out_m = twin(ME.SparseTensor(x.clone(), coordinates=cd)).F
        assert l2err(out_n, out_m) < 3e-2
        sd_n, sd_m = net.state_dict(), twin.state_dict()
        for k in sd_n:
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert relerr(sd_n[k], sd_m[k]) < 2e-2, k
            if k.endswith("num_batches_tracked"):
                assert int(sd_n[k]) == int(sd_m[k]) == 1, k
        # eval forward uses the running statistics and needs no batch reduction
        net.eval(); twin.eval()
        with torch.no_grad():
            e_n = ex(x, ME.CoordinateManager(cd), False)
            e_m = twin(ME.SparseTensor(x, coordinates=cd)).F
        assert l2err(e_n, e_m) < 3e-2
        # frozen parameters get no gradient, the others do
        net.train()
        frozen = [p for n, p in net.named_parameters() if n.startswith("0.blocks.")]
        for p in frozen:
            p.requires_grad_(False)
        xin = x.clone().requires_grad_(True)
        ex(xin, ME.CoordinateManager(cd), True).sum().backward()
        torch.cuda.synchronize()
        assert all(p.grad is None for p in frozen)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters() if p.requires_grad)
        assert xin.grad is not None and torch.isfinite(xin.grad).all()


def test_coordinate_pyramid_one_round_trip_matches_per_level_build(dev):
    """d3_kmap_pyramid (all stride-2 levels, one host round trip) == d3_kmap_down_count/_fill level by level, bit for bit"""
    from d3net_amd import minkowski as ME
    rng = np.random.default_rng(3)
    coords = torch.from_numpy(rand_coords(rng, (48, 40, 24), 0.1)).int().to(dev)
    a, b = ME.CoordinateManager(coords), ME.CoordinateManager(coords)
    b.build_pyramid(5)
    ts = 1
    for _ in range(4):
        ca, ua, Ma = a.down(ts)
        cb, ub, Mb = b.down(ts)
        assert Ma == Mb and torch.equal(ca, cb) and torch.equal(ua, ub)
        assert torch.equal(a.coords[2 * ts], b.coords[2 * ts])
        assert torch.equal(a.k3(2 * ts), b.k3(2 * ts))
        ts *= 2


@pytest.mark.parametrize("kind,cin,cout,dybf", [("k3", 16, 16, False), ("k3", 32, 16, False), ("k3", 32, 32, True), ("k3", 64, 32, True),
                                                ("k3", 48, 48, False), ("down", 16, 32, False), ("up", 32, 16, True)])
def test_wgrad3_strided_operands_ragged_rows_and_absent_neighbours(dev, kind, cin, cout, dybf):
    """spconv_wgrad3_kernel's raw buffer gathers: operands that are column views of wider buffers (row pitch > channels, as the
    executor's concatenation buffers), a row count that is no multiple of the kernel's iteration, kernel maps with absent
    neighbours (index -1 -> an out-of-range buffer offset the hardware answers with zeros) and trailing rows; against the
    second-generation kernel (D3_WG3=0) on dense copies of the same operands, and the bf16-dy variant against the fp32 one."""
    import os
    from d3net_amd import _lib
    from d3net_amd.pointgroup_ops import _ptr, _stream
    L = _lib.lib()
    FLIPK, XSTAT, XBF16, DYBF16 = 1, 8, 32, 64
    rng = np.random.default_rng(cin * 100 + cout)
    if kind == "k3":
        K, Min, Mout = 27, 5003, 5003
        tbl = rng.integers(0, Min, (Mout, K)).astype(np.int32)
        tbl[rng.random((Mout, K)) < 0.4] = -1
        tbl_t = rng.integers(0, Mout, (Min, K)).astype(np.int32)
        tbl_t[rng.random((Min, K)) < 0.4] = -1
    else:
        K = 8
        Min, Mout = (9001, 2501) if kind == "down" else (2501, 9001)
        tbl = rng.integers(0, Min, (Mout, K)).astype(np.int32); tbl[rng.random((Mout, K)) < 0.5] = -1
        tbl_t = rng.integers(0, Mout, (Min, K)).astype(np.int32); tbl_t[rng.random((Min, K)) < 0.5] = -1
    xstat = cin > cout
    flags = (XSTAT | (FLIPK if kind == "k3" else 0)) if xstat else 0
    t = torch.from_numpy(tbl_t if xstat else tbl).to(dev)
    # x: columns [8, 8+cin) of a (Min, cin + 24) bf16 buffer; dy: columns [8, 8+cout) of a (Mout, cout + 16) buffer
    xw = torch.from_numpy(rng.standard_normal((Min, cin + 24)).astype(np.float32)).to(dev).bfloat16()
    dyw = torch.from_numpy(rng.standard_normal((Mout, cout + 16)).astype(np.float32)).to(dev)
    if dybf:
        dyw = dyw.bfloat16()
    xv, dyv = xw[:, 8:8 + cin], dyw[:, 8:8 + cout]
    fl = flags | XBF16 | (DYBF16 if dybf else 0)

    def run(x, ldx, dy, ldy, f):
        ws = torch.empty(max(L.d3_spconv_wgrad2_ws_bytes(Min, Mout, K, cin, cout, f), 16), dtype=torch.uint8, device=dev)
        dW = torch.full((K, cin, cout), float("nan"), device=dev)
        rc = L.d3_spconv_wgrad2(x.data_ptr(), ldx, _ptr(t), dy.data_ptr(), ldy, _ptr(dW), Min, Mout, K, cin, cout, cin, f, _ptr(ws),
                                ws.numel(), _stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
        return dW

    assert xv.data_ptr() % 16 == 0 and dyv.data_ptr() % 8 == 0
    got = run(xv, cin + 24, dyv, cout + 16, fl)                       # strided views, third-generation kernel
    from d3net_amd import _lib
    with _lib.tuning(D3_WG3=0):
        ref = run(xv.contiguous(), cin, dyv.contiguous(), cout, fl)    # dense copies, second-generation kernel
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 2e-6, relerr(got, ref)
    if dybf:                                                          # the same bf16 values handed over as fp32: identical result
        same = run(xv, cin + 24, dyv.float().contiguous(), cout, flags | XBF16)
        assert torch.equal(got, same)
