"""world_size-2 gloo test of the N>1 path: flat gradient all-reduce == mean of the per-rank gradients,
replica broadcast, scene sharding.  Runs on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import FlatGradAllReduce, broadcast_module, shard_scenes
    torch.manual_seed(rank)  # different init per rank -> broadcast must equalise
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    broadcast_module(net)
    w0 = net[0].weight.detach().clone()
    torch.manual_seed(100 + rank)  # different data per rank (one "scene" each)
    x = torch.randn(16, 5)
    loss = net(x).pow(2).sum()
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    FlatGradAllReduce(net.parameters())()
    ret[rank] = dict(w0=w0, local=local, avg=[p.grad.clone() for p in net.parameters()],
                     shard=shard_scenes(7, rank, world), rm=net[1].running_mean.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["w0"], r1["w0"])                      # identical replicas
    for a0, a1, l0, l1 in zip(r0["avg"], r1["avg"], r0["local"], r1["local"]):
        assert torch.allclose(a0, a1)                           # every rank holds the same averaged gradient
        assert torch.allclose(a0, (l0 + l1) / 2, atol=1e-6)     # which is the mean of the local ones
    assert r0["shard"] == [0, 2, 4, 6] and r1["shard"] == [1, 3, 5]
    assert not torch.allclose(r0["rm"], r1["rm"])               # BN statistics stay per rank (no SyncBN)


def _bucket_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    torch.manual_seed(rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 7), torch.nn.Linear(7, 3))
    broadcast_module(net)
    # the first two layers keep their gradients as views of one flat buffer (what the native executors do)
    owned = list(net[0].parameters()) + list(net[1].parameters())
    flat = torch.zeros(sum(p.numel() for p in owned))
    off = 0
    for p in owned:
        p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    torch.manual_seed(100 + rank)
    net(torch.randn(16, 5)).pow(2).sum().backward()          # accumulates into the views / creates the head grads
    local = [p.grad.clone() for p in net.parameters()]
    BucketGradAllReduce(net.parameters(), lambda: ([flat], owned))()
    ret[rank] = dict(local=local, avg=[p.grad.clone() for p in net.parameters()],
                     still_views=all(p.grad.data_ptr() >= flat.data_ptr() and
                                     p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4 for p in owned))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_grad_allreduce_world2():
    """executor-owned flat gradient buffers are all-reduced in place, the remaining parameters in one packed collective"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["still_views"] and r1["still_views"]
    for a0, a1, l0, l1 in zip(r0["avg"], r1["avg"], r0["local"], r1["local"]):
        assert torch.allclose(a0, a1)
        assert torch.allclose(a0, (l0 + l1) / 2, atol=1e-6)


class _FakeExecutor:
    """host-side stand-in for netexec.NativeUNet's gradient bookkeeping (same three members the reducer uses)"""

    def __init__(self, params):
        self.params = list(params)
        self.flat = torch.zeros(sum(p.numel() for p in self.params))
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p)); off += p.numel()
        self.fresh_grads = True

    def prepare_for_allreduce(self):
        if self.fresh_grads:
            self.flat.zero_(); self.fresh_grads = False
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v


class _Owner:
    def __init__(self, ex):
        self.ex = ex

    def static_gradient_buckets(self):
        return [(self.ex.flat, self.ex.params, self.ex)]


def _ragged_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    torch.manual_seed(rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 7), torch.nn.Linear(7, 3))
    broadcast_module(net)
    ex = _FakeExecutor(net[1].parameters())          # "ScoreNet": its backward only runs on rank 0 this step
    sync = BucketGradAllReduce(net.parameters(), _Owner(ex))
    torch.manual_seed(100 + rank)
    x = torch.randn(16, 5)
    if rank == 0:
        for p, v in zip(ex.params, ex.views):
            p.grad = v
        ex.fresh_grads = False
        net(x).pow(2).sum().backward()
    else:                                            # no proposals: the middle layer and the last one get no gradient
        net[0](x).pow(2).sum().backward()
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    sync()
    ret[rank] = dict(local=local, avg=[p.grad.clone() for p in net.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_layout_is_static_when_a_rank_has_no_gradients():
    """a rank whose executor backward never ran (no proposals) still issues the same collectives: it contributes zeros and
    ends up with the average, like DDP -- no hang, no shorter packed tensor (rank-local bucket schedules)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ragged_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for a0, a1, l0, l1 in zip(r0["avg"], r1["avg"], r0["local"], r1["local"]):
        assert torch.allclose(a0, a1)
        want = (l0 + (l1 if l1 is not None else torch.zeros_like(l0))) / 2
        assert torch.allclose(a0, want, atol=1e-6)


def _early_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    torch.manual_seed(rank)
    det = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 7))     # "detector": layer 1 in an executor-style flat buffer
    head = torch.nn.Sequential(torch.nn.Linear(7, 6), torch.nn.Linear(6, 3))     # "speaker": the early bucket
    net = torch.nn.ModuleList([det, head])
    broadcast_module(net)
    ex = _FakeExecutor(det[1].parameters())
    sync = BucketGradAllReduce(net.parameters(), _Owner(ex), early=list(head.parameters()))
    out = {}
    # step 0: layout check first (late launch); step 1: one detector pass, the bucket starts inside backward();
    # step 2: two detector passes (the joint step), it starts at the second boundary only; step 3: rank 1 has no
    # proposals -> its boundary is never reached -> it issues the same collective from sync() instead
    for step, passes in enumerate((1, 1, 2, 1)):
        for p in net.parameters():
            p.grad = None
        ex.fresh_grads = True
        torch.manual_seed(1000 * step + rank)
        loss = 0
        fired_at = []
        for k in range(passes):
            x = torch.randn(16, 5)
            f = det(x)
            if step == 3 and rank == 1:
                loss = loss + f.detach().sum() * 0 + det[0](x).pow(2).sum()     # heads unused on this rank
            else:
                f, = sync.boundary(f)
                f.register_hook(lambda g, k=k: fired_at.append((k, sync._early_work is not None)))   # runs BEFORE the boundary node
                loss = loss + head(f).pow(2).sum()
        loss.backward()
        started_in_backward = sync._early_work is not None
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        sync()
        out[step] = dict(local=local, avg=[p.grad.clone() for p in net.parameters()], early=started_in_backward,
                         counters=(sync._expected, sync._fired))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_early_bucket_starts_inside_backward_and_keeps_the_collective_order():
    """the heads' bucket is all-reduced from the backward pass once it has crossed every detector boundary of the step;
    results equal the plain mean in every case, including a rank that never reaches its boundary"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_early_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for step in range(4):
        for a0, a1, l0, l1 in zip(r0[step]["avg"], r1[step]["avg"], r0[step]["local"], r1[step]["local"]):
            assert torch.allclose(a0, a1)
            z = torch.zeros_like(a0)
            assert torch.allclose(a0, ((l0 if l0 is not None else z) + (l1 if l1 is not None else z)) / 2, atol=1e-6)
        assert r0[step]["counters"] == (0, 0) and r1[step]["counters"] == (0, 0)      # re-armed for the next step
    assert not r0[0]["early"] and not r1[0]["early"]          # first step: layout comparison comes first
    assert r0[1]["early"] and r1[1]["early"] and r0[2]["early"] and r1[2]["early"]
    assert r0[3]["early"] and not r1[3]["early"]               # rank 1 launched it late; no hang, same averages


def _stale_early_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    torch.manual_seed(rank)
    det = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 7))
    head = torch.nn.Sequential(torch.nn.Linear(7, 6), torch.nn.Linear(6, 3))
    net = torch.nn.ModuleList([det, head])
    broadcast_module(net)
    ex = _FakeExecutor(det[1].parameters())
    sync = BucketGradAllReduce(net.parameters(), _Owner(ex), early=list(head.parameters()))
    out = {}
    # step 0: layout check (late path, always fine).  step 1: gradient accumulation -- two backward() calls before one
    # sync: the bucket is packed inside the first, the second adds to the head gradients afterwards.  step 2: a head
    # parameter is ALSO used below the boundary (inside the "detector"), so its gradient completes after the pack.
    for step in range(3):
        for p in net.parameters():
            p.grad = None
        ex.fresh_grads = True
        torch.manual_seed(1000 * step + rank)
        x = torch.randn(16, 5)
        if step == 2:
            h = det[0](x) * head[0].bias.sum()            # a head parameter inside the detector part of the graph
            f, = sync.boundary(det[1](h))
            head(f).pow(2).sum().backward()
        else:
            for _ in range(2 if step == 1 else 1):
                f, = sync.boundary(det(x))
                head(f).pow(2).sum().backward()
        try:
            sync()
            out[step] = "ok"
        except RuntimeError as e:
            out[step] = str(e)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_early_bucket_refuses_gradients_that_change_after_it_was_packed():
    """the early launch is gated on evidence -- post-accumulate hooks + version counters -- and fails
    loudly on every rank instead of installing a stale average"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stale_early_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in (ret[0], ret[1]):
        assert r[0] == "ok"
        assert "changed after the bucket was packed" in r[1] and "D3_EARLY_ALLREDUCE=0" in r[1]
        assert "changed after the bucket was packed" in r[2]


class _ChunkedFakeExecutor(_FakeExecutor):
    """+ the overlap interface of netexec.NativeUNet: tail chunks of the flat buffer in completion order, the backward hook"""

    def __init__(self, module):
        super().__init__(module.parameters())
        self.module = module
        self.on_backward, self.backward_done, self.backward_count, self.forward_count = None, False, 0, 0

    def set_grad_chunks(self, nchunks):
        sizes = [p.numel() for p in self.params]
        total, bounds, acc = sum(sizes), [], 0
        for i in range(len(sizes) - 1, -1, -1):      # parameters complete in reverse order
            acc += sizes[i]
            if len(bounds) < nchunks - 1 and acc >= total * (len(bounds) + 1) // nchunks:
                bounds.append(total - acc)
        bounds.append(0)
        ranges, hi = [], total
        for lo in bounds:
            if lo < hi:
                ranges.append((lo, hi)); hi = lo
        return ranges

    def chunk_wait(self, k, stream):
        raise AssertionError("no streams on the CPU")


class _FakeNativeBackward(torch.autograd.Function):
    """like netexec._NetFunction: ONE autograd node computes every parameter gradient of the sub-network into the executor's
    flat buffer, then tells the reducer"""

    @staticmethod
    def forward(ctx, x, ex):
        with torch.enable_grad():
            xin = x.detach().requires_grad_(True)
            y = ex.module(xin)
        ctx.ex, ctx.xin, ctx.y = ex, xin, y
        ex.forward_count += 1
        return y.detach()

    @staticmethod
    def backward(ctx, g):
        ex = ctx.ex
        grads = torch.autograd.grad(ctx.y, [ctx.xin] + ex.params, g)
        for p, v, gr in zip(ex.params, ex.views, grads[1:]):
            if ex.fresh_grads:
                v.copy_(gr)
            else:
                v.add_(gr)
            p.grad = v
        ex.fresh_grads = False
        ex.backward_count += 1
        ex.backward_done = True
        if ex.on_backward is not None:
            ex.on_backward(ex)
        return grads[0], None


def _chunk_worker(rank, world, port, ret, overlap, passes=1, no_heads_rank=-1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["D3_EARLY_ALLREDUCE"] = "1" if overlap else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    torch.manual_seed(rank)
    backbone = torch.nn.Sequential(*[torch.nn.Linear(9, 9) for _ in range(6)])       # "backbone": executor-owned, chunked
    point_head = torch.nn.Linear(9, 2)                                               # detector head outside the executor ("rest")
    head = torch.nn.Sequential(torch.nn.Linear(9, 6), torch.nn.Linear(6, 3))         # "speaker": the early bucket
    net = torch.nn.ModuleList([backbone, point_head, head])
    broadcast_module(net)
    ex = _ChunkedFakeExecutor(backbone)
    sync = BucketGradAllReduce(net.parameters(), _Owner(ex), early=list(head.parameters()), chunks=3)
    out = {}
    for step in range(3):
        for p in net.parameters():
            p.grad = None
        ex.fresh_grads = True
        torch.manual_seed(1000 * step + rank)
        loss = 0
        for _ in range(passes):          # (PipelineNet mode 3 runs the detector twice per step: two passes through the SAME executor)
            x = torch.randn(16, 9)
            f = _FakeNativeBackward.apply(x.requires_grad_(True), ex)
            if rank == no_heads_rank and _ == 0:
                # this rank's first detector pass produced no proposals: nothing reaches the heads, no boundary is placed --
                # the executor still ran (and owes a backward) for that pass
                loss = loss + point_head(f).pow(2).sum()
                continue
            fb, = sync.boundary(f)
            loss = loss + head(fb).pow(2).sum() + point_head(f).pow(2).sum()
        before = sync.chunk_launches
        loss.backward()
        inside = sync.chunk_launches - before
        local = [p.grad.clone() for p in net.parameters()]
        sync()
        out[step] = dict(loss=float(loss), inside=inside, local=local, avg=[p.grad.clone() for p in net.parameters()],
                         views=all(p.grad is v for p, v in zip(ex.params, ex.views)))
    ret[(rank, overlap, passes) if passes != 1 else (rank, overlap)] = out
    dist.barrier()
    dist.destroy_process_group()


def test_two_detector_passes_per_step_launch_the_chunks_after_the_second_backward():
    """PipelineNet mode 3 (speaker batch + listener batch): the executor's backward runs twice inside ONE loss.backward(); its
    chunk collectives must wait for the second run (they used to start after the first, and the reducer then raised 'backward ran
    again after its buffer had been all-reduced' -- found with `bench.py --config joint` over RCCL)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    for overlap in (True, False):
        mp.spawn(_chunk_worker, args=(world, _free_port(), ret, overlap, 2), nprocs=world, join=True)
    for step in range(3):
        a0, a1, b0, b1 = (ret[(r, o, 2)][step] for r, o in ((0, True), (1, True), (0, False), (1, False)))
        assert a0["inside"] == a1["inside"] == (0 if step == 0 else 3), (step, a0["inside"], a1["inside"])
        assert a0["loss"] == b0["loss"]
        for g0, g1, l0, l1, gb in zip(a0["avg"], a1["avg"], b0["local"], b1["local"], b0["avg"]):
            assert torch.allclose(g0, g1) and torch.allclose(g0, (l0 + l1) / 2, atol=1e-6) and torch.allclose(g0, gb, atol=1e-7)


def test_world4_one_rank_without_proposals_in_one_of_two_detector_passes():
    """VERDICT r3 item 8 / ADVICE r3: mode 3 runs the detector twice per step; on ONE of four ranks the first pass yields no
    proposals, so that rank crosses one boundary while its executor ran (and must run backward) twice.  The reducer counts the
    executor's differentiable forwards, not the boundaries: the chunk collectives start after the second backward on every
    rank, the schedule stays identical, nothing raises, and the averages equal the run without overlap."""
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    for overlap in (True, False):
        mp.spawn(_chunk_worker, args=(world, _free_port(), ret, overlap, 2, 3), nprocs=world, join=True)
    for step in range(3):
        a = [ret[(r, True, 2)][step] for r in range(world)]
        b = [ret[(r, False, 2)][step] for r in range(world)]
        assert all(x["inside"] == (0 if step == 0 else 3) for x in a), (step, [x["inside"] for x in a])
        for r in range(world):
            assert a[r]["loss"] == b[r]["loss"]
        for i in range(len(a[0]["avg"])):
            mean = sum(b[r]["local"][i] for r in range(world)) / world
            for r in range(world):
                assert torch.allclose(a[r]["avg"][i], mean, atol=1e-6) and torch.allclose(a[r]["avg"][i], b[r]["avg"][i], atol=1e-7)


def test_backbone_bucket_is_all_reduced_in_chunks_from_inside_backward():
    """the executor's flat gradient buffer is split by backward completion order and its chunks are
    all-reduced from INSIDE backward() (>= 3 collectives started there in every step after the first, which compares the
    layouts), behind the heads' bucket in the static schedule; averages and loss identical to the run without overlap"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    for overlap in (True, False):
        mp.spawn(_chunk_worker, args=(world, _free_port(), ret, overlap), nprocs=world, join=True)
    for step in range(3):
        a0, a1 = ret[(0, True)][step], ret[(1, True)][step]
        b0, b1 = ret[(0, False)][step], ret[(1, False)][step]
        assert a0["inside"] == a1["inside"] == (0 if step == 0 else 3), (step, a0["inside"], a1["inside"])
        assert b0["inside"] == 0
        assert a0["views"] and a1["views"]
        assert a0["loss"] == b0["loss"]
        # (the LOCAL gradients are read from the run without overlap -- same seeds, same data: with overlap a collective started
        # inside backward() may already have rewritten its chunk in place when backward() returns)
        for g0, g1, l0, l1, gb in zip(a0["avg"], a1["avg"], b0["local"], b1["local"], b0["avg"]):
            assert torch.allclose(g0, g1) and torch.allclose(g0, (l0 + l1) / 2, atol=1e-6) and torch.allclose(g0, gb, atol=1e-7)


def _logged_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from d3net_amd.pipeline import PipelineNet
    net = types.SimpleNamespace(logged={"train_loss/loss": torch.tensor(1.0 + rank), "train_score/cap_acc": 0.25 * (rank + 1)},
                                parameters=lambda: iter([torch.zeros(1)]))
    out = PipelineNet.reduce_logged(net)
    ret[rank] = {k: float(v) for k, v in out.items()}
    dist.barrier()
    dist.destroy_process_group()


def test_logged_scalars_one_packed_allreduce_world2():
    """the reference syncs every logged scalar separately (`sync_dist=True`, model/pipeline.py:149,182,...); here one packed
    all-reduce averages them all"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_logged_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert abs(ret[r]["train_loss/loss"] - 1.5) < 1e-6 and abs(ret[r]["train_score/cap_acc"] - 0.375) < 1e-6
