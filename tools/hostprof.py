import cProfile, pstats, sys, io, torch
sys.path.insert(0, '.')
from d3net_amd import synthetic as S
from d3net_amd.config import default_conf
from d3net_amd.pointgroup import PointGroup
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
cfg = default_conf(); torch.manual_seed(123)
model = PointGroup(cfg).to(dev).train(); model.teacher = True
opt = torch.optim.AdamW(model.parameters(), lr=0.002, fused=True)
occ, sem, inst, _ = S.occupancy_grid(); scene = S.scene_from_grid(occ, sem, inst)
batch = S.make_batch([scene], dev)
def step():
    d = dict(batch); model.zero_grad(set_to_none=True)
    loss, d = model.training_step(d); loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
