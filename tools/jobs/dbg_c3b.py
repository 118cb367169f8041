import sys, os, functools, numpy as np, torch
sys.path.insert(0, os.getcwd())
from d3net_amd import _lib, netexec, common, minkowski as ME, synthetic as S
L = _lib.lib()
dev = torch.device("cuda", 0)
netexec.SINGLE_READER_BF16 = os.environ.get("T_BF16", "0") == "1"
occ, _, _, _ = S.occupancy_grid()
vox = np.argwhere(occ)
coords = np.concatenate([np.zeros((len(vox), 1), np.int64), vox], 1)
planes = [16, 32, 48, 64, 80, 96, 112]
cin = 16
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.standard_normal((len(coords), cin)).astype(np.float32))
def run(c3, warm):
    L.d3_tuning_set(b"D3_C3", c3)
    torch.manual_seed(9)
    norm = functools.partial(ME.MinkowskiBatchNorm, eps=1e-4, momentum=0.1)
    net = torch.nn.Sequential(ME.MinkowskiConvolution(cin, planes[0], kernel_size=3, bias=False, dimension=3),
                              common.UBlock(planes, norm, 2, common.ResidualBlock), norm(planes[0]), ME.MinkowskiReLU(inplace=True))
    ME.fuse_bn_relu(net)
    net = net.to(dev)
    ex = netexec.NativeUNet(None, net[1], net[2], cin, True)
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    if warm:
        for l in range(7): cm.k3(1 << l) if l == 0 else None
        cm.build_pyramid(7)
        for l in range(3): cm.k3_q(1 << l)
        torch.cuda.synchronize()
        for l in range(3): cm.k3_q(1 << l)
    xn = x.to(dev).requires_grad_(True)
    n0 = L.d3_spconv_fwd3_launches()
    out = ex(xn, cm, True)
    g = torch.from_numpy(np.random.default_rng(6).standard_normal(tuple(out.shape)).astype(np.float32))
    out.backward(g.to(dev))
    torch.cuda.synchronize()
    names = {id(p): n for n, p in net.named_parameters()}
    return out.detach().clone(), xn.grad.clone(), {names[id(p)]: p.grad.clone() for p in net.parameters() if p.grad is not None}, L.d3_spconv_fwd3_launches() - n0
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
B = run(0, True)
for mode in (1, 2, 3, 4, 5):
    A = run(mode, True)
    worst = sorted(((rel(A[2][k], B[2][k]), k) for k in A[2]), reverse=True)
    print("mode", mode, "launches", A[3], "out rel-L2 %.2e" % rel(A[0], B[0]), "input grad %.2e" % rel(A[1], B[1]), "worst %.2e %s" % worst[0], "median %.2e" % worst[len(worst)//2][0])
