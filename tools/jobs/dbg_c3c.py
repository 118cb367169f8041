import sys, os, functools, numpy as np, torch
sys.path.insert(0, os.getcwd())
from d3net_amd import _lib, netexec, common, minkowski as ME, synthetic as S
L = _lib.lib()
dev = torch.device("cuda", 0)
netexec.SINGLE_READER_BF16 = False
occ, _, _, _ = S.occupancy_grid()
vox = np.argwhere(occ)
coords = np.concatenate([np.zeros((len(vox), 1), np.int64), vox], 1)
planes = [16, 32, 48, 64, 80, 96, 112]
cin = 16
x = torch.from_numpy(np.random.default_rng(5).standard_normal((len(coords), cin)).astype(np.float32))
def run(c3):
    L.d3_tuning_set(b"D3_C3", c3)
    torch.manual_seed(9)
    norm = functools.partial(ME.MinkowskiBatchNorm, eps=1e-4, momentum=0.1)
    net = torch.nn.Sequential(ME.MinkowskiConvolution(cin, planes[0], kernel_size=3, bias=False, dimension=3),
                              common.UBlock(planes, norm, 2, common.ResidualBlock), norm(planes[0]), ME.MinkowskiReLU(inplace=True))
    ME.fuse_bn_relu(net)
    net = net.to(dev)
    ex = netexec.NativeUNet(None, net[1], net[2], cin, True)
    ex.debug_keep = True
    cm = ME.CoordinateManager(torch.from_numpy(coords).int().to(dev))
    cm.build_pyramid(7)
    for l in range(3): cm.k3_q(1 << l)
    torch.cuda.synchronize()
    for l in range(3): cm.k3_q(1 << l)
    out = ex(x.to(dev), cm, True)
    torch.cuda.synchronize()
    arena, rows = ex.debug_last
    return ex, arena.cpu(), rows
def tensor(ex, arena, rows, i):
    level, Cc, width, coff, dtype, buf = ex.b.tensors[i]
    if buf < 0: return x
    off = L.d3_net_tensor_offset(ex._net(), i)
    es = 2 if dtype == 1 else 4
    raw = arena[off:off + ((rows[level] - 1) * width + Cc) * es]
    t = raw.view(torch.bfloat16 if dtype == 1 else torch.float32)
    return torch.as_strided(t, (rows[level], Cc), (width, 1)).float()
exA, aA, rows = run(1); exB, aB, _ = run(0)
for op in exA.b.ops:
    if op[0] in (1, 2):
        oi = op[2]
        a, b = tensor(exA, aA, rows, oi), tensor(exB, aB, rows, oi)
        d = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        mx = float((a - b).abs().max())
        lev = exA.b.tensors[oi][0]
        print("op", "CONV" if op[0] == 1 else "BN  ", "level", lev, "C", exA.b.tensors[oi][1], "K", op[7] if op[0] == 1 else "-", "res", op[3] if op[0] == 1 else "-", "rel-L2 %.2e max %.2e" % (d, mx))
