"""Scene-parallel data parallelism: one process per GPU, gradients only.

The reference trains with Lightning DDP over NCCL (scripts/train.py:265-268: `gpus=-1,
strategy="ddp_find_unused_parameters_false"`), i.e. bucketed gradient all-reduce and nothing else on the data path
(SURVEY.md section 8e).  Here: one fused all-reduce of a flat gradient buffer per step over RCCL/xGMI
(backend "nccl" on ROCm), averaged over ranks; BatchNorm statistics stay per rank, as in the reference.
Works on any backend (the CPU tests use gloo)."""
import os

import torch
import torch.distributed as dist



def _dist_active():
    """a process group with more than one rank -- or with ONE rank under D3_DIST_WORLD1=1 (test switch: the RCCL plumbing of the
    reducers on a one-GPU box; averaging over one rank is the identity)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("D3_DIST_WORLD1") == "1"

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
        if not _dist_active():
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
    d3net_amd/netexec.py): those buffers are all-reduced in place -- no packing -- and the remaining parameters (point-level
    heads, speaker / listener) share packed collectives.

    The collective SCHEDULE is static and ordered: [heads' bucket (`early`), executor 1's chunks, executor 2's chunks, ...,
    the packed rest].  It is derived from the `requires_grad` parameter list and the owner's executor set, never from which
    gradients happen to exist on this rank: a rank whose step produced no proposals (its ScoreNet backward never ran)
    contributes zeros and receives the other ranks' average, exactly what DDP does for a zero gradient.  The layout
    signature is compared across ranks once, at the first call.

    OVERLAP with the backward: an item of the schedule may be started from INSIDE `backward()` as soon as its gradients are
    final, but only when every item before it has been started -- so all ranks always issue the same collectives in the
    same order, whatever each rank's step looked like; what has not been started when `__call__` runs is started there,
    in order.  Items become ready through evidence, not assumption:
      * the heads' bucket when the backward crosses the last `boundary()` placed on the detector's outputs (and
        `_finish` verifies that no gradient of it changed afterwards: post-accumulate hooks + version counters);
      * an executor's chunks when its native backward has been enqueued (`NativeUNet.on_backward`): chunk k of the flat
        buffer -- a tail range: parameter gradients complete in reverse program order -- is all-reduced on its own stream
        behind the two events `d3_net_backward` recorded for it, while the rest of that backward is still running.

    `owner`: an object with `static_gradient_buckets()` -> [(flat tensor, [parameters], executor)] (PointGroup), or --
    legacy form used by the CPU tests -- a callable returning ([flat tensors], [covered parameters])."""

    def __init__(self, params, owner, early=(), chunks=None):
        self.params = [p for p in params if p.requires_grad]
        self.owner = owner
        if chunks is None:
            # without per-chunk streams a chunk collective is ordered behind the WHOLE native backward already enqueued on the
            # caller's stream: k chunks would be k serial collectives with nothing to overlap -- one collective per executor
            # (ADVICE r3); with D3_CHUNK_STREAMS=1 three tail chunks overlap the rest of the backward
            chunks = 3 if os.environ.get("D3_CHUNK_STREAMS", "0") == "1" else 1
        self.chunks = int(os.environ.get("D3_GRAD_CHUNKS", chunks))
        self._checked = False
        self._rest = None
        ids = {id(p) for p in self.params}
        self.early = [p for p in early if id(p) in ids] if os.environ.get("D3_EARLY_ALLREDUCE", "1") != "0" else []
        self._overlap = os.environ.get("D3_EARLY_ALLREDUCE", "1") != "0"
        self._expected = 0
        self._fired = 0
        self._early_work = None
        self.early_launches = 0     # steps whose heads bucket started inside backward()
        self.chunk_launches = 0     # executor chunk collectives started inside backward() (all steps)
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
        self._exec_items = None     # [dict(flat, params, ex, ranges, works, launched, count)] in schedule order
        self._streams = {}
        # D3_CHUNK_STREAMS=1: every chunk collective waits on its OWN stream for the two events d3_net_backward recorded for
        # the chunk, i.e. it overlaps the rest of that backward.  Off by default: this build could only exercise it through
        # gloo on one shared device (where it is pathologically slow: 84 s per step, two processes' streams and gloo's
        # staging copies serialise), never over RCCL -- an unmeasured path must not be the default of the scaling run.
        # Without it the chunks are still separate collectives started inside backward(), ordered behind the caller's stream.
        self._chunk_streams = os.environ.get("D3_CHUNK_STREAMS", "0") == "1"

    # ---- schedule -------------------------------------------------------------------------------------------------
    def _active(self):
        return _dist_active()

    def _buckets(self):
        if hasattr(self.owner, "static_gradient_buckets"):
            return self.owner.static_gradient_buckets()
        flats, covered = self.owner()
        return [(f, covered if i == 0 else [], None) for i, f in enumerate(flats)]

    def _items(self):
        """executor items of the schedule (built once): chunk ranges of every flat buffer + the backward hook"""
        if self._exec_items is None:
            items = []
            for flat, ps, ex in self._buckets():
                ranges = [(0, flat.numel())]
                if ex is not None and self._overlap and hasattr(ex, "set_grad_chunks") and self.chunks > 1:
                    ranges = ex.set_grad_chunks(self.chunks)          # [(lo, hi)] in completion order (tail first)
                it = dict(flat=flat, params=ps, ex=ex, ranges=ranges, works=[], launched=False, count=-1,
                          base=getattr(ex, "backward_count", 0) if ex is not None else 0,
                          fbase=getattr(ex, "forward_count", None) if ex is not None else None)
                if ex is not None and self._overlap and hasattr(ex, "on_backward"):
                    ex.on_backward = lambda net, it=it: self._exec_ready(it)
                items.append(it)
            self._exec_items = items
        return self._exec_items

    def _stream_for(self, dev, k):
        if dev.type != "cuda":
            return None
        key = (dev.index, k)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=dev)
        return self._streams[key]

    def _launch_exec(self, it, inside_backward):
        """in-place all-reduce of an executor's flat buffer, chunk by chunk in completion order"""
        ex, flat = it["ex"], it["flat"]
        if ex is not None:
            ex.prepare_for_allreduce()     # zero-fill if no backward wrote it this step; install the views as .grad
        avg = _avg_supported(flat.device)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        for k, (lo, hi) in enumerate(it["ranges"]):
            view = flat[lo:hi]
            st = self._stream_for(flat.device, k) if (self._chunk_streams and inside_backward and ex is not None and len(it["ranges"]) > 1) else None
            if st is not None:
                # the chunk's own stream waits for the two events the native backward recorded for chunk k; the collective is
                # ordered behind that stream, not behind the caller's (which still has the rest of the backward queued)
                ex.chunk_wait(k, st)
                with torch.cuda.stream(st):
                    view.record_stream(st)
                    w = dist.all_reduce(view, op=op, async_op=True)
            else:
                w = dist.all_reduce(view, op=op, async_op=True)
            it["works"].append((w, view, avg))
        it["launched"] = True
        it["count"] = getattr(ex, "backward_count", 0) if ex is not None else 0
        if inside_backward:
            self.chunk_launches += len(it["ranges"])

    def _advance(self, inside_backward):
        """start every not-yet-started item that is ready, in schedule order, stopping at the first that is not"""
        if not self._checked or not self._overlap:
            return
        if self.early and self._early_work is None:
            if not (self._expected > 0 and self._fired == self._expected):
                return
            self._launch_early()
            self.early_launches += 1
        # a step may run the detector more than once (PipelineNet mode 3: the speaker's and the listener's batch): an executor's
        # buffer is complete when its backward has run once per differentiable FORWARD it ran since the last sync.  The executor
        # counts those itself (`forward_count`, ADVICE r3: the boundaries crossed are only a proxy -- a pass whose outputs carry
        # no gradient into the heads places none); executors without the counter (the CPU tests' stand-ins): boundaries, min 1
        for it in self._items():
            if it["launched"]:
                continue
            ex = it["ex"]
            if ex is None or not getattr(ex, "backward_done", False):
                return
            fc = getattr(ex, "forward_count", None)
            need = max(1, self._expected) if (fc is None or it["fbase"] is None) else max(1, fc - it["fbase"])
            if getattr(ex, "backward_count", 0) - it["base"] < need:
                return
            self._launch_exec(it, inside_backward)

    def _exec_ready(self, it):
        """NativeUNet.on_backward: the executor's native backward has been enqueued"""
        if self._active():
            self._advance(True)

    # ---- early bucket -------------------------------------------------------------------------------------------
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
            self._advance(True)

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
        if not _dist_active():
            return
        world = dist.get_world_size()
        items = self._items()
        if self._rest is None:
            cov = {id(p) for it in items for p in it["params"]}
            self.early = [p for p in self.early if id(p) not in cov]
            cov |= {id(p) for p in self.early}
            self._rest = [p for p in self.params if id(p) not in cov]
        rest = self._rest
        dev = items[0]["flat"].device if items else (rest[0] if rest else self.early[0]).device
        if not self._checked:
            self._check_signature([sum(p.numel() for p in self.early)] + [hi - lo for it in items for lo, hi in it["ranges"]]
                                  + [sum(p.numel() for p in rest)], dev)
        # everything that was not started from inside the backward, in schedule order
        if self.early and self._early_work is None:
            self._launch_early()
        for it in items:
            if not it["launched"]:
                self._launch_exec(it, False)
        # RCCL averages inside the collective (no extra pass over the 31 MB buffer); gloo (the CPU tests) has no AVG
        avg = _avg_supported(dev)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
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
        stale_exec = None
        for it in items:
            for w, view, wavg in it["works"]:
                w.wait()
                if not wavg:
                    view.div_(world)
            ex = it["ex"]
            if ex is not None and getattr(ex, "backward_count", it["count"]) != it["count"]:
                stale_exec = ex       # its backward ran again AFTER its buffer went on the wire
            it["works"], it["launched"], it["count"] = [], False, -1
            if ex is not None:
                it["base"] = getattr(ex, "backward_count", 0)
                it["fbase"] = getattr(ex, "forward_count", None)
            if ex is not None and hasattr(ex, "backward_done"):
                ex.backward_done = False
        if self.early:
            self._finish_early(world)
        if stale_exec is not None:
            raise RuntimeError("BucketGradAllReduce: an executor's backward ran again after its gradient buffer had been all-reduced "
                               "from inside backward() (gradient accumulation over several backward() calls); run with D3_EARLY_ALLREDUCE=0.")


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
