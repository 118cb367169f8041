"""Scene-parallel data parallelism: one process per GPU, gradients only.

The reference trains with Lightning DDP over NCCL (scripts/train.py:265-268: `gpus=-1,
strategy="ddp_find_unused_parameters_false"`), i.e. bucketed gradient all-reduce and nothing else on the data path
(SURVEY.md section 8e).  Here: one fused all-reduce of a flat gradient buffer per step over RCCL/xGMI
(backend "nccl" on ROCm), averaged over ranks; BatchNorm statistics stay per rank, as in the reference.
Works on any backend (the CPU tests use gloo)."""
import os

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


def _avg_supported(device):
    """ReduceOp.AVG on this backend?  Probed once with a one-element collective (every rank takes the same branch: the
    decision depends on the backend only)."""
    global _AVG_OK
    if _AVG_OK is None:
        _AVG_OK = False
        if dist.get_backend() == "nccl":
            try:
                t = torch.ones(1, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.AVG)
                _AVG_OK = bool(abs(float(t) - 1.0) < 1e-6)
            except Exception:
                _AVG_OK = False
    return _AVG_OK


class BucketGradAllReduce:
    """Gradient averaging for models whose sub-networks keep their gradients in flat buffers (the native U-Net executors,
    d3net_amd/netexec.py): those buffers are all-reduced in place -- one collective each, no packing -- and the remaining
    parameters (point-level heads, speaker / listener) share one packed collective.

    The collective schedule is STATIC: it is derived from the `requires_grad` parameter list and the owner's executor
    set, never from which gradients happen to exist on this rank.  A rank whose step produced no proposals (its ScoreNet
    backward never ran) contributes zeros and receives the other ranks' average, exactly what DDP does for a zero
    gradient; every rank therefore issues the same collectives with the same sizes, in the same order, every step.  The
    layout signature is compared across ranks once, at the first call.

    `owner`: an object with `static_gradient_buckets()` -> [(flat tensor, [parameters], executor)] (PointGroup), or --
    legacy form used by the CPU tests -- a callable returning ([flat tensors], [covered parameters])."""

    def __init__(self, params, owner, early=()):
        self.params = [p for p in params if p.requires_grad]
        self.owner = owner
        self._checked = False
        self._rest = None
        # overlap with the backward: `early` = parameters whose gradients are complete before the detector's backward
        # starts (the speaker / listener heads: their nodes were created after every detector node, so the autograd
        # engine -- highest sequence number first -- has run all of them when it reaches a GradBoundary placed on the
        # detector's outputs).  Their bucket is then packed and all-reduced from the backward itself
        # (boundary_reached) and runs on the collective stream underneath the U-Net backward; a step in which the
        # boundary is never reached (no proposals, no gradient into the detector) issues the same collective from
        # __call__ instead, so every rank still issues the same collectives in the same order.
        ids = {id(p) for p in self.params}
        self.early = [p for p in early if id(p) in ids] if os.environ.get("D3_EARLY_ALLREDUCE", "1") != "0" else []
        self._expected = 0
        self._fired = 0
        self._early_work = None
        self.early_launches = 0     # steps whose heads bucket started inside backward()
        # Evidence, not assumption: "every head gradient is final when the last boundary fires" rests on the
        # autograd engine's ready-queue order.  Every early parameter reports its accumulation through a
        # post-accumulate-grad hook; an accumulation that arrives AFTER the bucket was packed (a head parameter also used
        # inside the detector, a node on another ready queue, a second backward() before the sync) is recorded, and
        # `_finish_early` additionally compares every gradient's identity and version counter with the snapshot taken at
        # pack time -- a mismatch raises instead of silently installing a stale average.
        self._late = []
        self._snap = None
        for i, p in enumerate(self.early):
            if hasattr(p, "register_post_accumulate_grad_hook"):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_early_grad(i))

    # ---- early bucket -------------------------------------------------------------------------------------------
    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def boundary(self, *tensors):
        """Identity on the detector's outputs that marks, in the autograd graph, the point below which no `early`
        parameter receives gradient any more.  Call once per detector pass of the step, on every output that carries
        gradient into the detector; returns the tensors to hand to the heads."""
        live = [i for i, t in enumerate(tensors) if t.requires_grad]
        if not self.early or not self._active() or not torch.is_grad_enabled() or not live:
            return tensors
        self._expected += 1
        out = list(tensors)
        for i, t in zip(live, _GradBoundary.apply(self, *[tensors[i] for i in live])):
            out[i] = t
        return tuple(out)

    def boundary_reached(self):
        self._fired += 1
        # (not before the first __call__ has compared the layout across ranks and probed ReduceOp.AVG: those are
        # collectives too, and every rank must issue them in the same position)
        if self._checked and self._fired == self._expected and self._early_work is None:   # the last pass' boundary
            self._launch_early()
            self.early_launches += 1

    def _on_early_grad(self, i):
        if self._early_work is not None:      # the bucket is already on the wire: this gradient is not in it
            self._late.append(i)

    def _launch_early(self):
        dev = self.early[0].device
        avg = _avg_supported(dev)
        self._late = []
        self._snap = [(p.grad, -1 if p.grad is None else p.grad._version) for p in self.early]
        packed = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.early])
        work = dist.all_reduce(packed, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self._early_work = (work, packed, avg)

    def _finish_early(self, world):
        work, packed, avg = self._early_work
        work.wait()
        stale = sorted(set(self._late) | {i for i, (p, (g, v)) in enumerate(zip(self.early, self._snap))
                                          if p.grad is not g or (g is not None and g._version != v)})
        if stale:
            self._early_work, self._snap, self._late = None, None, []
            self._expected = self._fired = 0
            raise RuntimeError("BucketGradAllReduce: %d gradient(s) of the early (heads) bucket changed after the bucket was packed "
                               "inside backward() (first: early[%d]) -- a head parameter receives gradient below the detector "
                               "boundary, or backward() ran more than once before the sync (gradient accumulation).  The averaged "
                               "bucket would silently drop that contribution; run with D3_EARLY_ALLREDUCE=0." % (len(stale), stale[0]))
        self._snap = None
        if not avg:
            packed.div_(world)
        for p, v in zip(self.early, packed.split([p.numel() for p in self.early])):
            if p.grad is None:
                p.grad = v.view_as(p).clone()
            else:
                p.grad.copy_(v.view_as(p))
        self._early_work = None
        self._expected = self._fired = 0

    def _buckets(self):
        if hasattr(self.owner, "static_gradient_buckets"):
            return self.owner.static_gradient_buckets()
        flats, covered = self.owner()
        return [(f, covered if i == 0 else [], None) for i, f in enumerate(flats)]

    def _check_signature(self, sizes, device):
        """every rank must run the same schedule: compare (count, sizes) once"""
        sig = torch.tensor([float(len(sizes))] + [float(n) for n in sizes], dtype=torch.float64, device=device)
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("BucketGradAllReduce: ranks disagree on the gradient bucket layout %s" % (sizes,))
        self._checked = True

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        buckets = self._buckets()
        if self._rest is None:
            cov = {id(p) for _, ps, _ in buckets for p in ps}
            self.early = [p for p in self.early if id(p) not in cov]
            cov |= {id(p) for p in self.early}
            self._rest = [p for p in self.params if id(p) not in cov]
        rest = self._rest
        dev = buckets[0][0].device if buckets else (rest[0] if rest else self.early[0]).device
        if not self._checked:
            self._check_signature([sum(p.numel() for p in self.early)] + [f.numel() for f, _, _ in buckets]
                                  + [sum(p.numel() for p in rest)], dev)
        if self.early and self._early_work is None:   # the boundary was not reached in this step's backward: same collective, now
            self._launch_early()
        # RCCL averages inside the collective (no extra pass over the 31 MB buffer); gloo (the CPU tests) has no AVG
        avg = _avg_supported(dev)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works = []
        for flat, ps, ex in buckets:
            if ex is not None:
                ex.prepare_for_allreduce()     # zero-fill if no backward wrote it this step; install the views as .grad
            works.append(dist.all_reduce(flat, op=op, async_op=True))
        if rest:
            packed = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in rest])
            dist.all_reduce(packed, op=op)
            if not avg:
                packed.div_(world)
            for p, v in zip(rest, packed.split([p.numel() for p in rest])):
                if p.grad is None:
                    p.grad = v.view_as(p).clone()
                else:
                    p.grad.copy_(v.view_as(p))
        for w, (flat, _, _) in zip(works, buckets):
            w.wait()
            if not avg:
                flat.div_(world)
        if self.early:
            self._finish_early(world)


class _GradBoundary(torch.autograd.Function):
    """identity; its backward tells the reducer that the backward pass has left the heads"""

    @staticmethod
    def forward(ctx, reducer, *tensors):
        ctx.reducer = reducer
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        ctx.reducer.boundary_reached()
        return (None,) + grads


def broadcast_module(module, src=0):
    """identical replicas at start (what DDP does at construction)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def shard_scenes(n_scenes, rank, world):
    """rank r gets scenes r, r+W, ... (DistributedSampler order without shuffling)"""
    return list(range(rank, n_scenes, world))
