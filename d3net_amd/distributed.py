"""Scene-parallel data parallelism: one process per GPU, gradients only.

The reference trains with Lightning DDP over NCCL (scripts/train.py:265-268: `gpus=-1,
strategy="ddp_find_unused_parameters_false"`), i.e. bucketed gradient all-reduce and nothing else on the data path
(SURVEY.md section 8e).  Here: one fused all-reduce of a flat gradient buffer per step over RCCL/xGMI
(backend "nccl" on ROCm), averaged over ranks; BatchNorm statistics stay per rank, as in the reference.
Works on any backend (the CPU tests use gloo)."""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Averages the gradients of `params` across ranks with ONE collective on a persistent flat buffer."""

    def __init__(self, params, device=None):
        self.params = [p for p in params if p.requires_grad]
        dev = device if device is not None else self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat)
        self.flat.div_(dist.get_world_size())
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


_AVG_OK = None


def _avg_supported(like):
    """ReduceOp.AVG on this backend?  Probed once with a one-element collective (every rank takes the same branch)."""
    global _AVG_OK
    if _AVG_OK is None:
        _AVG_OK = False
        if dist.get_backend() == "nccl" and like is not None:
            try:
                t = torch.ones(1, device=like.device)
                dist.all_reduce(t, op=dist.ReduceOp.AVG)
                _AVG_OK = bool(abs(float(t) - 1.0) < 1e-6)
            except Exception:
                _AVG_OK = False
    return _AVG_OK


class BucketGradAllReduce:
    """Gradient averaging for models whose sub-networks already keep their gradients in flat buffers (the native U-Net
    executors, d3net_amd/netexec.py): those buffers are all-reduced in place -- one collective each, no packing -- and the
    remaining parameters (the point-level heads: a dozen small tensors) share one packed collective.
    `buckets()` -> (list of flat gradient tensors, list of the parameters they cover); called every step because the
    executors create their buffers lazily."""

    def __init__(self, params, buckets):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = buckets

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        flats, covered = self.buckets()
        cov = {id(p) for p in covered}
        # RCCL averages inside the collective (no extra pass over the 31 MB buffer); gloo (the CPU tests) has no AVG
        avg = _avg_supported(flats[0] if flats else None)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works = [dist.all_reduce(f, op=op, async_op=True) for f in flats]
        rest = [p for p in self.params if id(p) not in cov and p.grad is not None]
        if rest:
            grads = [p.grad for p in rest]
            packed = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(packed, op=op)
            if not avg:
                packed.div_(world)
            torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(packed.split([g.numel() for g in grads]), grads)])
        for w, f in zip(works, flats):
            w.wait()
            if not avg:
                f.div_(world)


def broadcast_module(module, src=0):
    """identical replicas at start (what DDP does at construction)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def shard_scenes(n_scenes, rank, world):
    """rank r gets scenes r, r+W, ... (DistributedSampler order without shuffling)"""
    return list(range(rank, n_scenes, world))
