"""host-side time of the python autograd Functions' backward/forward calls in one detector step (monkeypatched timers)"""
import sys, time, collections, torch
sys.path.insert(0, '.')
from d3net_amd import synthetic as S, pointgroup_ops as ops, heads, netexec
from d3net_amd.config import default_conf
from d3net_amd.pointgroup import PointGroup
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    for meth in ("forward", "backward"):
        f = getattr(cls, meth)
        def g(*a, _f=f, _k=name + "." + meth, **k):
            t = time.perf_counter(); r = _f(*a, **k); acc[_k][0] += 1; acc[_k][1] += time.perf_counter() - t; return r
        setattr(cls, meth, staticmethod(g))
for cls, n in ((netexec._NetFunction, "net"), (ops.RoiPool, "roipool"), (ops.Voxelization, "voxelization"), (heads._Devoxelize, "devox"),
               (heads._TallLinear, "tall"), (heads._CrossEntropy, "ce"), (heads._OffsetLoss, "offloss"), (ops.BFSCluster, "bfs"),
               (ops.BallQueryBatchP, "bq"), (ops.Voxelization_Idx, "voxidx"), (ops.GetIoU, "iou"), (ops.SecMean, "secmean")):
    wrap(cls, n)
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
cfg = default_conf(); torch.manual_seed(123)
model = PointGroup(cfg).to(dev).train(); model.teacher = True
opt = torch.optim.AdamW(model.parameters(), lr=0.002, fused=True)
occ, sem, inst, _ = S.occupancy_grid(); batch = S.make_batch([S.scene_from_grid(occ, sem, inst)], dev)
tb = [0.0, 0.0, 0.0]
def step():
    d = dict(batch); model.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); loss, d = model.training_step(d); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    tb[0] += t1 - t0; tb[1] += t2 - t1; tb[2] += t3 - t2
for _ in range(4): step()
torch.cuda.synchronize(); acc.clear(); tb[:] = [0, 0, 0]
N = 10
for _ in range(N): step()
torch.cuda.synchronize()
print("host ms/step: forward+loss %.2f  backward %.2f  optimizer %.2f" % tuple(1e3 * x / N for x in tb))
for k, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print("  %-24s calls/step %4.1f  ms/step %.3f" % (k, c / N, 1e3 * t / N))
