"""Layer-program builder and python front end of the native sparse U-Net executor (csrc/unet.hip).

The reference runs `self.backbone` / `self.score_net` (model/pointgroup.py:69-74, 88-92, 268-272, 332-333) module by
module through MinkowskiEngine.  Here the same module tree (same state-dict keys: d3net_amd/common.py,
d3net_amd/minkowski.py) is flattened once into a program of four op types -- CONV (with fused residual add, strided
concat output and BatchNorm-statistics epilogue), BNACT (finalize + normalise + ReLU, bf16 output for convolution
operands), PADCAST (stem input -> zero-padded bf16) and STATS -- and every forward / backward of the whole network is
ONE call into libd3hip.so.  Parameter gradients are written by the executor straight into one flat buffer whose views
are installed as `param.grad` (no autograd accumulation nodes: ~500 parameters, one kernel each otherwise).
"""
import ctypes as C
import struct

import torch
from torch.autograd import Function

from . import _lib, common
from . import minkowski as ME
from ._lib import check
from .pointgroup_ops import _on, _stream

OP_CONV, OP_BNACT, OP_PADCAST, OP_STATS = 1, 2, 3, 4
MAP_K1, MAP_K3, MAP_DOWN, MAP_UP = 0, 1, 2, 3
F32, BF16 = 0, 1


def _fbits(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


class _Builder:
    def __init__(self, act_dtype=BF16):
        self.tensors, self.bufs, self.ops, self.params, self.grad_params, self.bns = [], [], [], [], [], []
        self.act_dtype = act_dtype      # storage of the BN -> ReLU outputs / the padded stem input: bf16, or fp32 (reference precision)

    def buf(self, level, width, dtype):
        self.bufs.append((level, width, dtype))
        return len(self.bufs) - 1

    def view(self, buf, coff, Cc):
        level, width, dtype = self.bufs[buf]
        self.tensors.append((level, Cc, width, coff, dtype, buf))
        return len(self.tensors) - 1

    def new(self, level, Cc, dtype):
        return self.view(self.buf(level, Cc, dtype), 0, Cc)

    def external(self, Cc):
        self.tensors.append((0, Cc, Cc, 0, F32, -1))
        return len(self.tensors) - 1

    def param(self, t, grad):
        self.params.append(t)
        self.grad_params.append(bool(grad))
        return len(self.params) - 1

    def C(self, t):
        return self.tensors[t][1]

    def level(self, t):
        return self.tensors[t][0]

    def op(self, *fields):
        self.ops.append(list(fields) + [0] * (16 - len(fields)))

    # ---- op emitters
    def bnact(self, x, bn, out_dtype=None, relu=True):
        """MinkowskiBatchNorm (+ fused MinkowskiReLU) on tensor x -> new tensor."""
        if out_dtype is None:
            out_dtype = self.act_dtype
        b = bn.bn
        assert b.track_running_stats and b.affine and b.num_features == self.C(x)
        y = self.new(self.level(x), self.C(x), out_dtype)
        self.op(OP_BNACT, x, y, -1, self.param(b.weight, True), self.param(b.bias, True), self.param(b.running_mean, False),
                self.param(b.running_var, False), int(relu), _fbits(b.eps), _fbits(b.momentum))
        self.bns.append(bn)
        return y

    def conv(self, x, conv, level_out, out=None, res=-1, stats=True, cin_w=None):
        if isinstance(conv, ME.MinkowskiConvolutionTranspose):
            kind, mlevel = MAP_UP, level_out
        elif conv.kernel_size == 3:
            kind, mlevel = MAP_K3, level_out
        elif conv.kernel_size == 2:
            kind, mlevel = MAP_DOWN, self.level(x)
        else:
            kind, mlevel = MAP_K1, level_out
        if out is None:
            out = self.new(level_out, conv.out_channels, F32)
        assert self.C(out) == conv.out_channels
        self.op(OP_CONV, x, out, res, self.param(conv.kernel, True), kind, mlevel, conv.kernel_volume,
                conv.in_channels if cin_w is None else cin_w, int(stats))
        return out

    # ---- module walkers (reference: model/common.py)
    def block(self, blk, x, out=None):
        level = self.level(x)
        if isinstance(blk, common.ResidualBlock):
            skip = x if blk.downsample is None else self.conv(x, blk.downsample[0], level, stats=False)
            cb = blk.conv_branch
            t = self.conv(self.bnact(x, cb[0]), cb[2], level)
            return self.conv(self.bnact(t, cb[3]), cb[5], level, out=out, res=skip)
        cl = blk.conv_layers   # VGGBlock
        return self.conv(self.bnact(x, cl[0]), cl[2], level, out=out)

    def ublock(self, u, x, out=None):
        level = self.level(x)
        deeper = len(u.nPlanes) > 1
        c = u.nPlanes[0]
        blocks = list(u.blocks.children())
        cat = self.buf(level, 2 * c, F32) if deeper else None
        for i, blk in enumerate(blocks):
            last = i == len(blocks) - 1
            target = self.view(cat, 0, c) if (deeper and last) else (out if (last and not deeper) else None)
            x = self.block(blk, x, out=target)
        if not deeper:
            return x
        d = self.conv(self.bnact(x, u.conv[0]), u.conv[2], level + 1)
        d = self.ublock(u.u, d)
        self.conv(self.bnact(d, u.deconv[0]), u.deconv[2], level, out=self.view(cat, c, c))
        x = self.view(cat, 0, 2 * c)
        tail = list(u.blocks_tail.children())
        for i, blk in enumerate(tail):
            x = self.block(blk, x, out=out if i == len(tail) - 1 else None)
        return x


class NativeUNet:
    """`stem` (MinkowskiConvolution or None) -> UBlock -> MinkowskiBatchNorm -> ReLU as one native program."""

    def __init__(self, stem, ublock, final_bn, in_channels, input_needs_grad, exact=False):
        """exact: the reference's precision -- every buffer fp32; csrc/unet.hip then runs the D3_CONV_F32 kernels (fp32 weight
        fragments, exact fp32 products on v_mfma_f32_16x16x4_f32) and keeps every gradient in fp32"""
        b = _Builder(F32 if exact else BF16)
        self.exact = bool(exact)
        x = b.external(in_channels)
        if stem is not None:
            cpad = (in_channels + 7) // 8 * 8
            xp = b.new(0, cpad, b.act_dtype)
            b.op(OP_PADCAST, x, xp)
            x = b.conv(xp, stem, 0, cin_w=in_channels)
        else:
            b.op(OP_STATS, x)
        x = b.ublock(ublock, x)
        out = b.bnact(x, final_bn, out_dtype=F32, relu=final_bn.fused_relu)
        self.b = b
        self.out_tensor, self.out_channels = out, b.C(out)
        self.nlevels = len(ublock.nPlanes)
        self.input_needs_grad = bool(input_needs_grad)
        self.in_channels = in_channels
        self.handle = None
        self._ptr_key = None
        self._plan_key, self._plan = None, None
        self._flat_grad, self._grad_views = None, None
        self.fresh_grads = False     # set by the owner's zero_grad(): the next backward writes instead of accumulating
        self.debug_keep = False      # tests: keep the last forward's activation arena + level rows in `debug_last`
        self.debug_last = None
        self.debug_pairs = None
        # data-parallel overlap (d3net_amd/distributed.py): called with `self` once the native backward has been enqueued
        self.on_backward = None
        self.backward_done = False   # a native backward ran since the reducer last reset it
        self.backward_count = 0
        self.forward_count = 0       # differentiable forwards (each one owes a backward before the flat buffer is complete)
        self._chunk_ranges = None

    # ------------------------------------------------------------------ lazily created native state
    def _net(self):
        if self.handle is None:
            b = self.b
            L = _lib.lib()
            prog = (C.c_int64 * (16 * len(b.ops)))(*[v for op in b.ops for v in op])
            tens = (C.c_int64 * (6 * len(b.tensors)))(*[v for t in b.tensors for v in t])
            bufs = (C.c_int64 * (3 * len(b.bufs)))(*[v for t in b.bufs for v in t])
            self.handle = C.c_void_p(L.d3_net_create(prog, len(b.ops), tens, len(b.tensors), bufs, len(b.bufs), self.nlevels,
                                                     len(b.params), int(self.input_needs_grad), self.out_tensor))
            assert self.handle.value
        return self.handle

    def __deepcopy__(self, memo):
        return None     # native state is per object: the owner rebuilds its executor lazily

    def __del__(self):
        try:
            if self.handle is not None and self.handle.value:
                _lib.lib().d3_net_destroy(self.handle)
        except Exception:
            pass

    def _param_ptrs(self):
        ps = self.b.params
        key = (ps[0].data_ptr(), ps[-1].data_ptr(), ps[len(ps) // 2].data_ptr())
        if key != self._ptr_key:
            assert all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps)
            self._pp = (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
            self._ptr_key = key
            self._flat_grad = None
        return self._pp

    def _plan_for(self, rows):
        key = tuple(rows)
        if key != self._plan_key:
            a, g = C.c_size_t(0), C.c_size_t(0)
            check(_lib.lib().d3_net_plan(self._net(), (C.c_int * len(rows))(*rows), C.byref(a), C.byref(g)), "net_plan")
            self._plan_key = key
            self._plan = (a.value, g.value, _lib.lib().d3_net_tensor_offset(self._net(), self.out_tensor))
        return self._plan

    def _grads(self, device):
        """flat gradient buffer + one view per trainable parameter"""
        if self._flat_grad is None:
            ps, gp = self.b.params, self.b.grad_params
            total = sum(p.numel() for p, g in zip(ps, gp) if g)
            self._flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
            views, off = [], 0
            for p, g in zip(ps, gp):
                if g:
                    views.append(self._flat_grad[off:off + p.numel()].view_as(p)); off += p.numel()
                else:
                    views.append(None)
            self._grad_views = views
        return self._grad_views

    def set_grad_chunks(self, nchunks):
        """Split the flat gradient buffer into `nchunks` tail ranges of about equal size that complete one after the other during
        the backward (parameters are registered in program order, the backward runs the program in reverse) and tell the
        native executor where each one ends (d3_net_set_chunks: it flushes the weight-gradient reductions there and records
        the chunk's events).  -> [(lo, hi)] offsets into the flat buffer, in completion order."""
        ps, gp, ops = self.b.params, self.b.grad_params, self.b.ops
        self._grads(ps[0].device)
        offs, off = [], 0
        for p, g in zip(ps, gp):
            offs.append(off)
            off += p.numel() if g else 0
        total = off
        nchunks = max(1, min(int(nchunks), 14))      # (d3_net_set_chunks: the staging ring covers 2 * (nchunks + 2) <= 32 flushes)
        if total < (1 << 20):                # (a 0.1 MB ScoreNet buffer: one collective)
            nchunks = 1
        first_param = []                     # per op: its first parameter index (ops without parameters: None)
        for op in ops:
            idx = [int(op[4])] if op[0] == OP_CONV else ([int(op[4]), int(op[5])] if op[0] == OP_BNACT else [])
            first_param.append(min(idx) if idx else None)
        bounds, want = [], [total * (nchunks - 1 - k) // nchunks for k in range(nchunks)]     # descending lower bounds, last = 0
        k = 0
        for i in range(len(ops) - 1, -1, -1):
            if first_param[i] is None or k >= nchunks - 1:
                continue
            if offs[first_param[i]] <= want[k]:
                bounds.append((i, offs[first_param[i]]))
                k += 1
        first_op = next(i for i in range(len(ops)) if first_param[i] is not None)
        bounds.append((first_op, 0))
        ranges, hi, op_idx = [], total, []
        for i, lo in bounds:
            if lo < hi:
                ranges.append((lo, hi)); op_idx.append(i); hi = lo
        check(_lib.lib().d3_net_set_chunks(self._net(), (C.c_int * len(op_idx))(*op_idx), len(op_idx)), "net_set_chunks")
        self._chunk_ranges = ranges
        return ranges

    def chunk_wait(self, k, stream):
        """make `stream` wait for chunk k of the last native backward"""
        check(_lib.lib().d3_net_chunk_wait(self._net(), k, C.c_void_p(stream.cuda_stream)), "net_chunk_wait")

    def owned_params(self):
        """trainable parameters whose gradient lives in this executor's flat buffer"""
        return [p for p, g in zip(self.b.params, self.b.grad_params) if g and p.requires_grad]

    def drop_stale_grads(self):
        """torch semantics for a step in which this executor's backward never ran (no proposals -> no ScoreNet pass): the
        owner's zero_grad(set_to_none=True) only marked the flat buffer stale, so the parameters still point at LAST
        step's gradients -- detach them (grad None: the optimizer skips the tensor, like torch after zero_grad)."""
        if self.fresh_grads and self._grad_views is not None:
            for p, v in zip(self.b.params, self._grad_views):
                if v is not None and p.grad is v:
                    p.grad = None

    def prepare_for_allreduce(self):
        """before the in-place all-reduce of the flat buffer: a rank whose backward did not run this step contributes
        zeros (and then holds the other ranks' average like everybody else)"""
        views = self._grads(self.b.params[0].device)
        if self.fresh_grads:
            self._flat_grad.zero_()
            self.fresh_grads = False
        for p, v, g in zip(self.b.params, views, self.b.grad_params):
            if g and p.requires_grad and p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def pad_input(self, feats):
        """the stem's zero-padded bf16 operand of `feats` (M, in_channels) fp32, prepared outside the forward (the input prefetch: the
        launch the forward would issue first), or None when this executor has no such operand (no stem / reference precision)"""
        L = _lib.lib()
        cp = L.d3_net_padded_channels(self._net())
        if cp <= 0 or not (feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.size(1) == self.in_channels):
            return None
        xp = torch.empty((feats.size(0), cp), dtype=torch.bfloat16, device=feats.device)
        with _on(feats.device):
            check(L.d3_net_padcast(self._net(), C.c_void_p(feats.data_ptr()), C.c_void_p(xp.data_ptr()), feats.size(0), _stream()), "net_padcast")
        return xp

    def maps(self, cm):
        k3, child, up, rows, keep = [], [], [], [], []
        k16, ok16 = [], []
        cm.build_pyramid(self.nlevels)
        ts = 1
        for lev in range(self.nlevels):
            nbr = cm.k3(ts)
            keep.append(nbr); k3.append(nbr.data_ptr()); rows.append(nbr.size(0))
            t16 = cm.k3_16(ts) if (not self.exact and hasattr(cm, "k3_16")) else None    # (validated; the bf16 wave-per-tile kernels read it)
            if t16 is not None:
                keep.append(t16); k16.append(t16.data_ptr()); ok16.append(t16.data_ptr())
            else:
                k16.append(0); ok16.append(0)
            if lev + 1 < self.nlevels:
                ch, u, _ = cm.down(ts)
                keep += [ch, u]; child.append(ch.data_ptr()); up.append(u.data_ptr())
            else:
                child.append(0); up.append(0)
            ts *= 2
        n = self.nlevels
        keep.append(((C.c_void_p * n)(*k16), (C.c_void_p * n)(*ok16)))     # (last element: the 16-bit tables' pointer arrays)
        return ((C.c_void_p * n)(*k3), (C.c_void_p * n)(*child), (C.c_void_p * n)(*up), rows, keep)

    def __call__(self, feats, cm, training):
        ps = self.b.params
        # ONE trainable parameter rides along as an input so that autograd runs the backward even when `feats` needs no
        # gradient (the executor writes every parameter gradient itself and returns None for it; passing all ~250
        # parameters made the engine visit 250 AccumulateGrad nodes with nothing to accumulate: 0.3 ms of host time right
        # before the optimizer)
        anchor = next((p for p, g in zip(ps, self.b.grad_params) if g and p.requires_grad), None)
        if training and torch.is_grad_enabled() and (anchor is not None or feats.requires_grad):
            self.forward_count += 1
        return _NetFunction.apply(feats, self, cm, bool(training), *(() if anchor is None else (anchor,)))


class _NetFunction(Function):
    @staticmethod
    def forward(ctx, feats, net, cm, training, *trainable):
        feats = feats.contiguous()
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.size(1) == net.in_channels
        dev = feats.device
        L = _lib.lib()
        k3, child, up, rows, keep = net.maps(cm)
        assert rows[0] == feats.size(0)
        arena_bytes, grad_bytes, out_off = net._plan_for(rows)
        arena = torch.empty(arena_bytes, dtype=torch.uint8, device=dev)
        pp = net._param_ptrs()
        # (input prefetch: the stem's padded bf16 operand was prepared with the coordinate maps -- NativeUNet.pad_input)
        xp = getattr(cm, "padded_input", None)
        if xp is not None and not (torch.is_tensor(xp) and xp.is_cuda and xp.dtype == torch.bfloat16 and xp.is_contiguous() and
                                   xp.size(0) == feats.size(0) and xp.size(1) == L.d3_net_padded_channels(net._net())):
            xp = None
        with _on(dev):
            check(L.d3_net_set_k3_16(net._net(), keep[-1][0], keep[-1][1]), "net_set_k3_16")
            if xp is not None:
                check(L.d3_net_set_padded_input(net._net(), C.c_void_p(xp.data_ptr())), "net_set_padded_input")
            check(L.d3_net_forward(net._net(), pp, k3, child, up, C.c_void_p(feats.data_ptr()), C.c_void_p(arena.data_ptr()),
                                   int(training), _stream()), "net_forward")
        if training:
            for bn in net.b.bns:
                bn._steps += 1
        M, Co = rows[0], net.out_channels
        out = arena[out_off:out_off + M * Co * 4].view(torch.float32).view(M, Co)
        if net.debug_keep:
            net.debug_last = (arena, list(rows))
            # kernel-3 pairs per level (entries of the dense neighbour table that exist): bench.py's compulsory-byte count
            net.debug_pairs = [int((t >= 0).sum()) for t in keep if torch.is_tensor(t) and t.dim() == 2 and t.size(1) == 27]
        ctx.net, ctx.maps, ctx.arena, ctx.feats, ctx.grad_bytes = net, (k3, child, up, keep), arena, feats, grad_bytes
        ctx.rows = rows
        ctx.xp = xp
        ctx.cm = cm if hasattr(cm, "k3_16") else None
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        assert ctx.training, "backward through the native U-Net needs a training-mode forward (batch statistics)"
        k3, child, up, keep = ctx.maps
        dev = gout.device
        gout = gout.contiguous()
        L = _lib.lib()
        ps, gp = net.b.params, net.b.grad_params
        views = net._grads(dev)
        n = len(ps)
        pg = (C.c_void_p * n)()
        acc = (C.c_int * n)()
        fresh = net.fresh_grads
        for i in range(n):
            if gp[i] and ps[i].requires_grad:
                v = views[i]
                pg[i] = v.data_ptr()
                if fresh:
                    if ps[i].grad is not v:
                        ps[i].grad = v
                else:
                    g = ps[i].grad
                    if g is None or g.data_ptr() != v.data_ptr():
                        ps[i].grad = v
                    else:
                        acc[i] = 1
        net.fresh_grads = False
        net._plan_for(ctx.rows)   # the arena layout belongs to the forward's level sizes (another forward may have re-planned)
        garena = torch.empty(ctx.grad_bytes, dtype=torch.uint8, device=dev)
        gin = torch.empty_like(ctx.feats) if net.input_needs_grad else None
        k16 = keep[-1]
        if ctx.cm is not None and not net.exact:      # the 16-bit tables whose validity flag has landed since the forward
            n_ = net.nlevels
            ptrs = [(t.data_ptr() if t is not None else 0) for t in (ctx.cm.k3_16(1 << l) for l in range(n_))]
            k16 = ((C.c_void_p * n_)(*ptrs), (C.c_void_p * n_)(*ptrs))
        with _on(dev):
            check(L.d3_net_set_k3_16(net._net(), k16[0], k16[1]), "net_set_k3_16")
            if ctx.xp is not None:
                check(L.d3_net_set_padded_input(net._net(), C.c_void_p(ctx.xp.data_ptr())), "net_set_padded_input")
            check(L.d3_net_backward(net._net(), net._param_ptrs(), k3, child, up, C.c_void_p(ctx.feats.data_ptr()),
                                    C.c_void_p(ctx.arena.data_ptr()), C.c_void_p(garena.data_ptr()), C.c_void_p(gout.data_ptr()),
                                    pg, acc, C.c_void_p(gin.data_ptr()) if gin is not None else None, _stream()), "net_backward")
        # garena / arena are only touched by work already enqueued on this stream and on the executor's side stream,
        # which this stream has joined: the caching allocator may reuse them for later work on this stream
        net.backward_count += 1
        net.backward_done = True
        if net.on_backward is not None:      # data-parallel: the reducer may start this buffer's chunk collectives now
            net.on_backward(net)
        return (gin, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)
