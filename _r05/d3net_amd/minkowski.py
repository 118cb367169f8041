"""The MinkowskiEngine API subset the reference's detector calls, on MI355X.

Same constructor signatures, attribute names and state-dict keys as the calls in the reference
(reference: model/common.py:13-15,32,36-41,64-66,88-90,96-98,114; model/pointgroup.py:65,70,73,91,176,268):

    ME.SparseTensor(features=, coordinates=)      .features / .F / .coordinates / .C, `x += y`
    ME.MinkowskiConvolution(in, out, kernel_size=, stride=, bias=False, dimension=3)   -> .kernel
    ME.MinkowskiConvolutionTranspose(in, out, kernel_size=2, stride=2, bias=False, dimension=3)
    ME.MinkowskiBatchNorm(C, eps=, momentum=)     -> .bn.{weight,bias,running_mean,running_var,num_batches_tracked}
    ME.MinkowskiReLU(inplace=True)
    ME.cat(a, b)

All arithmetic runs in libd3hip.so (csrc/coordmap.hip, spconv.hip, bn.hip).  MinkowskiEngine is an
unpinned third-party dependency of the reference; the semantics implemented here are those of
oracle/sparse_oracle.py (pinned against dense conv3d).  Kernel offset k = ox + Kd*oy + Kd^2*oz (x fastest).  Which
order a real MinkowskiEngine checkpoint uses cannot be verified offline (third party, unpinned, absent): `set_kernel_order`
installs a load-time permutation of the `kernel` tensors for checkpoints written in another offset order.
"""
import ctypes as C
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._lib import check
from .pointgroup_ops import _on, _ptr, _stream, _workspace

D3_CONV_FLIPK, D3_CONV_TRANSW, D3_CONV_EXACT, D3_CONV_XSTAT, D3_CONV_ACCUM, D3_CONV_XBF16, D3_CONV_DYBF16 = 1, 2, 4, 8, 16, 32, 64
D3_CONV_F32 = 256   # reference precision on the matrix cores: fp32 operands, v_mfma_f32_16x16x4_f32 (csrc/spconv2.hip)

_EXACT = False  # True: the reference's precision (fp32 storage, fp32 products, fp32 accumulate); False: bf16 MFMA operands
_EXACT_FMA = False   # with _EXACT: the one-thread-per-output FMA validation kernels instead of the fp32 MFMA kernels


def set_exact(flag):
    """Select the reference-precision kernels: exact fp32 convolutions (fp32 MFMA) instead of bf16 MFMA, and the exact fp32 MFMA
    form of the heads' tall GEMMs instead of the bf16 x 3 split (csrc/hgemm.hip)."""
    global _EXACT, _HG_X3_DEFAULT
    _EXACT = bool(flag)
    try:        # (the library may not be built yet when a CPU-only test flips the flag)
        L = _lib.lib()
        if _HG_X3_DEFAULT is None:
            v = C.c_int(0)
            L.d3_tuning_get(b"D3_HG_BF16X3", C.byref(v))
            _HG_X3_DEFAULT = int(v.value)
        L.d3_tuning_set(b"D3_HG_BF16X3", 0 if _EXACT else _HG_X3_DEFAULT)
    except Exception:
        pass


_HG_X3_DEFAULT = None

# Precision policy (round 4).  TRAINING steps run bf16 MFMA operands (what bench.py times; BASELINE.json's configs name bf16 / fp16);
# EVALUATION (module.eval(): validation_step, forward(), every reported mAP / CIDEr / Acc) runs the reference-precision kernels by
# default.  Why: a bf16 forward perturbs the 16-dim proposal features by ~1e-2, which flips ~1 % of the greedy caption tokens and an
# occasional IoU-0.5 decision -- discrete events whose effect on CIDEr@0.5IoU over 768 held-out captions was measured at +0.04 /
# -0.21 / -0.19 / -0.12 / -1.70 / +0.12 % for six trained models: not inside the north star's 0.5 % with any margin, while the fp32 kernels are
# (identical metrics on the final tree).  An inference pass has no backward, so fp32 costs ~1.6x a bf16 forward there and nothing in the training step.
# D3_EVAL_BF16=1 (or set_eval_exact(False)) evaluates with the bf16 kernels.
import os as _os
_EVAL_EXACT = _os.environ.get("D3_EVAL_BF16", "0") != "1"


def set_eval_exact(flag):
    """evaluation-mode forwards on the reference-precision kernels (default) or on the training step's bf16 kernels"""
    global _EVAL_EXACT
    _EVAL_EXACT = bool(flag)


_EVAL_EXACT_ONLY = frozenset()      # with set_eval_exact(False): the U-Nets ("backbone" / "score_net") that keep the reference precision anyway


def set_eval_exact_only(names):
    """module-wise ablation of the evaluation precision (tools/bf16_ablation.py): with the bf16 kernels forced onto the evaluation
    (set_eval_exact(False)), the named U-Nets still run their reference-precision twins"""
    global _EVAL_EXACT_ONLY
    _EVAL_EXACT_ONLY = frozenset(names or ())


def exact_for(training, name=None):
    """does a forward in this mode (of the U-Net `name`) run the reference-precision program?"""
    return _EXACT or (not training and (_EVAL_EXACT or (name is not None and name in _EVAL_EXACT_ONLY)))


class heads_exact_for:
    """`with heads_exact_for(training):` -- the heads' GEMM mode follows the same policy as the U-Nets: an evaluation forward that
    runs the fp32 twin executors must not run the heads on the bf16 x 3 split either (set_exact() switched
    D3_HG_BF16X3 off, exact_for() did not).  A no-op in training mode and when the split is off anyway (the default)."""

    def __init__(self, training):
        self.on = (not training) and _EVAL_EXACT and not _EXACT
        self.prev = None

    def __enter__(self):
        if self.on:
            try:
                L = _lib.lib()
                v = C.c_int(0)
                L.d3_tuning_get(b"D3_HG_BF16X3", C.byref(v))
                if v.value:
                    self.prev = int(v.value)
                    L.d3_tuning_set(b"D3_HG_BF16X3", 0)
            except Exception:
                self.prev = None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            _lib.lib().d3_tuning_set(b"D3_HG_BF16X3", self.prev)
        return False


def _mode_flag():
    return D3_CONV_EXACT if _EXACT else 0


def _kmap16_enabled():
    try:
        v = C.c_int(1)
        _lib.lib().d3_tuning_get(b"D3_KMAP16", C.byref(v))
        return bool(v.value)
    except Exception:
        return True


# ------------------------------------------------------------------------------ coordinate manager
class CoordinateManager:
    """Coordinate sets and kernel maps per tensor stride (cached, shared by all layers of a forward)."""

    def __init__(self, coordinates):
        assert coordinates.is_cuda and coordinates.dtype == torch.int32 and coordinates.dim() == 2 and coordinates.size(1) == 4
        self.device = coordinates.device
        self.coords = {1: coordinates.contiguous()}
        self._k3 = {}
        self._k3_16 = {}
        self._down = {}
        self._pending = None      # begin_pyramid() without its build_pyramid() yet
        # build the 16-bit form of the big levels' 27-offset tables?  Only a bf16 executor reads it, and only with D3_KMAP16 on
        # (the fp32 / evaluation executors and D3_KMAP16=0 paid 54 B per row, a pinned tensor, a copy and an event per
        # level for a table nobody read); NativeUNet's caller says so through `want16`
        self.want16 = _kmap16_enabled()

    def __del__(self):
        # a begin_pyramid() whose build_pyramid() never ran (an exception in between): hand the ticket back to the library's pool
        pend = getattr(self, "_pending", None)
        if pend is not None:
            self._pending = None
            try:
                rows = (C.c_int * pend[0])()
                _lib.lib().d3_kmap_pyramid_end(pend[1], rows, pend[0])
            except Exception:
                pass

    def _ws(self, M):
        return _workspace(_lib.lib().d3_coordmap_ws_bytes(M), self.device, "cm")

    def k3(self, ts):
        if ts not in self._k3:
            c = self.coords[ts]
            M = c.size(0)
            nbr = torch.empty((M, 27), dtype=torch.int32, device=self.device)
            ws = self._ws(M)
            with _on(self.device):
                if M >= self.K3_16_MIN_ROWS and self.want16:      # big level: the 16-bit form and its validity flag in the same pass
                    n16 = torch.empty(M * 27 + 2, dtype=torch.int16, device=self.device)
                    ok = torch.empty(1, dtype=torch.int32, device=self.device)
                    check(_lib.lib().d3_kmap_k3_16(_ptr(c), M, ts, _ptr(ws), ws.numel(), _ptr(nbr), _ptr(n16), _ptr(ok), _stream()), "kmap_k3_16")
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(ok, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._k3_16[ts] = {"tbl": n16, "ok": ok, "host": host, "ev": ev, "valid": None}
                else:
                    check(_lib.lib().d3_kmap_k3(_ptr(c), M, ts, _ptr(ws), ws.numel(), _ptr(nbr), _stream()), "kmap_k3")
                    self._k3_16[ts] = {"tbl": None, "valid": False}
            self._k3[ts] = nbr
        return self._k3[ts]

    K3_16_MIN_ROWS = 32768     # levels below run the workgroup-per-tile kernels, which read the dense table

    def k3_16(self, ts):
        """The int16-delta form of k3(ts) (d3_kmap_k3_pack16) IF it is known to be valid, else None.  The validity flag is computed
        on the device and copied to pinned memory behind the table build; nobody waits for it: a consumer that asks before the
        copy has landed gets None and reads the dense table (the backbone's level 0 is built inside begin_pyramid(), so its
        flag arrives with the pyramid's row counts -- before the forward; the deeper levels' flags are there for the backward).
        Levels below K3_16_MIN_ROWS never build one."""
        if ts not in self._k3:
            self.k3(ts)
        st = self._k3_16[ts]
        if st["valid"] is None and st["ev"].query():
            st["valid"] = bool(int(st["host"][0]) == 1)
        return st["tbl"] if st["valid"] else None

    def begin_pyramid(self, nlevels):
        """Enqueue the coordinate pyramid of levels 1..nlevels-1 and the copy of its row counts WITHOUT waiting for them
        (d3_kmap_pyramid_begin): device work enqueued by the caller before `build_pyramid` runs while the host reads the counts."""
        if self._pending is not None or 1 in self._down:
            return
        c0 = self.coords[1]
        M0 = c0.size(0)
        if M0 == 0 or nlevels < 2:
            return
        self.k3_16(1)      # level 0's table + its 16-bit form go first: the validity flag then lands with the row counts below
        dev = self.device
        n1 = nlevels - 1
        cout = torch.empty((n1, M0, 4), dtype=torch.int32, device=dev)
        par = torch.empty((n1, M0), dtype=torch.int32, device=dev)
        kid = torch.empty((n1, M0), dtype=torch.int32, device=dev)
        flg = torch.empty((n1, M0), dtype=torch.int32, device=dev)
        rdev = torch.empty(nlevels, dtype=torch.int32, device=dev)
        ws = self._ws(M0)
        ticket = C.c_void_p()
        with _on(dev):
            check(_lib.lib().d3_kmap_pyramid_begin(_ptr(c0), M0, nlevels, _ptr(ws), ws.numel(), _ptr(cout), _ptr(par), _ptr(kid), _ptr(flg),
                                                  _ptr(rdev), C.byref(ticket), _stream()), "kmap_pyramid_begin")
        self._pending = (nlevels, ticket, cout, par, kid, flg, rdev, ws)

    def build_pyramid(self, nlevels):
        """Coordinates and stride-2 maps of levels 1..nlevels-1 with one host round trip (d3_kmap_pyramid_begin / _end) instead
        of one per level; afterwards `down(ts)` / `coords[ts]` are cache hits."""
        if self._pending is None:
            if all((1 << l) in self._down for l in range(nlevels - 1)) or 1 in self._down:
                return
            self.begin_pyramid(nlevels)
            if self._pending is None:
                return
        nl, ticket, cout, par, kid, flg, rdev, ws = self._pending
        self._pending = None
        dev = self.device
        L = _lib.lib()
        rows = (C.c_int * nl)()
        check(L.d3_kmap_pyramid_end(ticket, rows, nl), "kmap_pyramid_end")
        with _on(dev):
            ts = 1
            for l in range(nl - 1):
                M, Mo = rows[l], rows[l + 1]
                child = torch.empty((Mo, 8), dtype=torch.int32, device=dev)
                up = torch.empty((M, 8), dtype=torch.int32, device=dev)
                check(L.d3_kmap_down_fill2(M, Mo, _ptr(par[l]), _ptr(kid[l]), _ptr(child), _ptr(up), _stream()), "kmap_down_fill2")
                self.coords[2 * ts] = cout[l, :Mo]
                self._down[ts] = (child, up, Mo, par[l, :M], kid[l, :M])
                ts *= 2

    def down(self, ts):
        """-> (child (Mout,8), up (M,8), Mout); registers the coordinates of stride 2*ts."""
        if self._pending is not None:
            self.build_pyramid(self._pending[0])
        if ts not in self._down:
            c = self.coords[ts]
            M = c.size(0)
            parent = torch.empty(M, dtype=torch.int32, device=self.device)
            kidx = torch.empty(M, dtype=torch.int32, device=self.device)
            ws = self._ws(M)
            L = _lib.lib()
            with _on(self.device):
                Mo = C.c_int(0)
                check(L.d3_kmap_down_count(_ptr(c), M, ts, _ptr(ws), ws.numel(), _ptr(parent), _ptr(kidx),
                                           C.byref(Mo), _stream()), "kmap_down_count")
                Mo = Mo.value
                oc = torch.empty((Mo, 4), dtype=torch.int32, device=self.device)
                child = torch.empty((Mo, 8), dtype=torch.int32, device=self.device)
                up = torch.empty((M, 8), dtype=torch.int32, device=self.device)
                check(L.d3_kmap_down_fill(_ptr(c), M, ts, _ptr(ws), ws.numel(), _ptr(parent), _ptr(kidx), _ptr(oc),
                                          _ptr(child), _ptr(up), Mo, _stream()), "kmap_down_fill")
            self.coords[2 * ts] = oc
            self._down[ts] = (child, up, Mo, parent, kidx)
        return self._down[ts][:3]


class SparseTensor:
    def __init__(self, features, coordinates=None, coordinate_manager=None, tensor_stride=1):
        assert features.is_cuda, "d3net_amd.minkowski runs on the GPU only (no CPU fallback)"
        if coordinate_manager is None:
            coordinate_manager = CoordinateManager(coordinates.int().contiguous())
        self.F = features
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = tensor_stride
        self._relu_done = False
        self._conv_done = None

    @property
    def features(self):
        return self.F

    @property
    def C(self):
        return self.coordinate_manager.coords[self.tensor_stride]

    coordinates = C

    def _like(self, feats, stride=None):
        return SparseTensor(feats, coordinate_manager=self.coordinate_manager,
                            tensor_stride=self.tensor_stride if stride is None else stride)

    def __iadd__(self, other):
        assert other.tensor_stride == self.tensor_stride
        self.F = self.F + other.F
        self._relu_done = False
        return self

    def __add__(self, other):
        return self._like(self.F + other.F)


def cat(*tensors):
    """ME.cat: channel concatenation of tensors on the same coordinate map (reference: model/common.py:114)."""
    s = tensors[0].tensor_stride
    assert all(t.tensor_stride == s and t.coordinate_manager is tensors[0].coordinate_manager for t in tensors)
    return tensors[0]._like(torch.cat([t.F for t in tensors], 1))


# ------------------------------------------------------------------------------------- autograd ops
_GEN2 = True   # second-generation kernels (csrc/spconv2.hip) whenever the channel counts allow


def _f32():
    """exact mode runs the fp32-MFMA kernels (D3_CONV_F32) unless the FMA validation kernels are asked for"""
    return D3_CONV_F32 if (_EXACT and not _EXACT_FMA) else 0


def _conv_call(x, tbl, W3, Mout, K, Cin, Cout, flags):
    if _f32() and _GEN2 and Cin % 8 != 0 and Cout % 4 == 0 and not flags & D3_CONV_TRANSW:
        # (the 134-channel stem in exact mode: zero-padded to a multiple of 8 channels, like the executor's PADCAST)
        pad = (-Cin) % 8
        x = torch.nn.functional.pad(x, (0, pad))
        W3 = torch.nn.functional.pad(W3, (0, 0, 0, pad))
        Cin += pad
    out = torch.empty((Mout, Cout), dtype=torch.float32, device=x.device)
    if _GEN2 and (not _EXACT or _f32()) and Cin % 8 == 0 and Cout % 4 == 0:
        L = _lib.lib()
        f32 = _f32()
        wp = _workspace(L.d3_spconv_pack_bytes_ex(K, Cin, Cout, f32), x.device, "wpack")
        with _on(x.device):
            check(L.d3_spconv_pack(_ptr(W3), _ptr(wp), K, Cin, Cout, flags & (D3_CONV_FLIPK | D3_CONV_TRANSW) | f32, _stream()),
                  "spconv_pack")
            check(L.d3_spconv_fwd2(_ptr(x), Cin, _ptr(tbl) if tbl is not None else None, _ptr(wp), _ptr(out), Cout,
                                   None, 0, None, x.size(0), Mout, K, Cin, Cout, flags & D3_CONV_XBF16 | f32, _stream()),
                  "spconv_fwd2")
        return out
    with _on(x.device):
        check(_lib.lib().d3_spconv_fwd(_ptr(x), _ptr(tbl) if tbl is not None else None, _ptr(W3), _ptr(out),
                                       x.size(0), Mout, K, Cin, Cout, flags | _mode_flag(), _stream()), "spconv_fwd")
    return out


class SparseConvFunction(Function):
    """out = sum_k x[tbl_f[:,k]] @ W[k];  backward through tbl_b (the transposed kernel map)."""

    @staticmethod
    def forward(ctx, x, W, tbl_f, tbl_b, Mout, bwd_flags):
        x = x.contiguous()
        W3 = W if W.dim() == 3 else W.unsqueeze(0)
        K, Cin, Cout = W3.shape
        assert x.size(1) == Cin and x.dtype == torch.float32 and W3.is_contiguous()
        out = _conv_call(x, tbl_f, W3, Mout, K, Cin, Cout, 0)
        ctx.save_for_backward(x, W)
        ctx.maps = (tbl_f, tbl_b, Mout, bwd_flags)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        tbl_f, tbl_b, Mout, bwd_flags = ctx.maps
        dy = dy.contiguous()
        W3 = W if W.dim() == 3 else W.unsqueeze(0)
        K, Cin, Cout = W3.shape
        dx = dW = None
        if ctx.needs_input_grad[0]:
            # data gradient: the same contraction over the transposed map with W^T (Cout -> Cin)
            dx = _conv_call(dy, tbl_b, W3, x.size(0), K, Cout, Cin, bwd_flags | D3_CONV_TRANSW)
        if ctx.needs_input_grad[1]:
            dW = _conv_wgrad(x, tbl_f, tbl_b, dy, W3, Mout, bwd_flags).view_as(W)
        return dx, dW, None, None, None, None


def _conv_wgrad(x, tbl_f, tbl_b, dy, W3, Mout, bwd_flags, xflag=0):
    """dW of out = sum_k x[tbl_f[:,k]] @ W[k]; reads the wider operand contiguously (x-stationary over the transposed map)"""
    K, Cin, Cout = W3.shape
    dW = torch.empty_like(W3)   # cleared inside d3_spconv_wgrad
    if Cin > Cout and (tbl_b is not None or tbl_f is None):
        tbl, wflags = tbl_b, D3_CONV_XSTAT | (bwd_flags & D3_CONV_FLIPK)
    else:
        tbl, wflags = tbl_f, 0
    CinP = Cin
    if _f32() and _GEN2 and Cin % 8 != 0 and Cout % 8 == 0:     # (the stem in exact mode: zero-padded channels, dW keeps Cin rows)
        CinP = Cin + (-Cin) % 8
        x = torch.nn.functional.pad(x, (0, CinP - Cin))
    if _GEN2 and (not _EXACT or _f32()) and CinP % 8 == 0 and Cout % 8 == 0:
        L = _lib.lib()
        wflags |= _f32()
        ws = _workspace(max(L.d3_spconv_wgrad2_ws_bytes(x.size(0), Mout, K, CinP, Cout, wflags | xflag), 16), x.device, "wgrad")
        with _on(x.device):
            check(L.d3_spconv_wgrad2(_ptr(x), CinP, _ptr(tbl) if tbl is not None else None, _ptr(dy), Cout, _ptr(dW),
                                     x.size(0), Mout, K, CinP, Cout, Cin, wflags | xflag, _ptr(ws), ws.numel(), _stream()),
                  "spconv_wgrad2")
        return dW
    with _on(x.device):
        check(_lib.lib().d3_spconv_wgrad(_ptr(x), _ptr(tbl) if tbl is not None else None, _ptr(dy), _ptr(dW), x.size(0),
                                         Mout, K, Cin, Cout, wflags | xflag | _mode_flag(), _stream()), "spconv_wgrad")
    return dW


class PreActConvFunction(Function):
    """The pre-activation unit of every U-Net block as ONE autograd node: BatchNorm (batch statistics) -> ReLU -> conv.
    The normalised activations are materialised once, as bf16 -- the convolution's MFMA operands are bf16 anyway, so
    this is numerically identical to an fp32 intermediate -- which halves the bytes of the gather that bounds the
    large levels and of the tensor kept for the weight gradient.  (Exact mode keeps the intermediate in fp32.)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, W, tbl_f, tbl_b, Mout, bwd_flags, eps, relu, running_mean, running_var, momentum):
        x = x.contiguous()
        M, Cc = x.shape
        W3 = W if W.dim() == 3 else W.unsqueeze(0)
        K, Cin, Cout = W3.shape
        assert Cin == Cc
        L = _lib.lib()
        stats = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        ws = _workspace(L.d3_bn_ws_bytes(Cc), x.device, "bn")
        use_bf16 = (not _EXACT) and Cc % 8 == 0
        y = torch.empty((M, Cc), dtype=torch.bfloat16 if use_bf16 else torch.float32, device=x.device)
        with _on(x.device):
            check(L.d3_bn_stats(_ptr(x), M, Cc, _ptr(stats[0]), _ptr(stats[1]),
                                _ptr(running_mean) if running_mean is not None else None,
                                _ptr(running_var) if running_var is not None else None, float(momentum or 0.0),
                                _ptr(ws), ws.numel(), _stream()), "bn_stats")
            fwd = L.d3_bn_relu_fwd_bf16 if use_bf16 else L.d3_bn_relu_fwd
            check(fwd(_ptr(x), _ptr(stats[0]), _ptr(stats[1]), _ptr(gamma), _ptr(beta), _ptr(y), M, Cc, eps, int(relu),
                      _stream()), "bn_relu_fwd")
        xflag = D3_CONV_XBF16 if use_bf16 else 0
        out = _conv_call(y, tbl_f, W3, Mout, K, Cin, Cout, xflag)
        ctx.save_for_backward(x, gamma, beta, stats, y, W)
        ctx.cfg = (tbl_f, tbl_b, Mout, bwd_flags, eps, relu, xflag)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, beta, stats, y, W = ctx.saved_tensors
        tbl_f, tbl_b, Mout, bwd_flags, eps, relu, xflag = ctx.cfg
        dout = dout.contiguous()
        W3 = W if W.dim() == 3 else W.unsqueeze(0)
        K, Cin, Cout = W3.shape
        M = x.shape[0]
        dy = _conv_call(dout, tbl_b, W3, M, K, Cout, Cin, bwd_flags | D3_CONV_TRANSW)          # grad w.r.t. the conv input
        dW = _conv_wgrad(y, tbl_f, tbl_b, dout, W3, Mout, bwd_flags, xflag).view_as(W) if ctx.needs_input_grad[3] else None
        dx, dgamma, dbeta = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        ws = _workspace(_lib.lib().d3_bn_ws_bytes(Cin), x.device, "bn")
        with _on(x.device):
            check(_lib.lib().d3_bn_relu_bwd(_ptr(x), _ptr(dy), _ptr(stats[0]), _ptr(stats[1]), _ptr(gamma), _ptr(beta),
                                            _ptr(dx), _ptr(dgamma), _ptr(dbeta), M, Cin, eps, int(relu), _ptr(ws),
                                            ws.numel(), _stream()), "bn_relu_bwd")
        return (dx, dgamma, dbeta, dW) + (None,) * 9


class BatchNormReLUFunction(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, running_mean, running_var, momentum):
        x = x.contiguous()
        M, Cc = x.shape
        stats = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        mean, var = stats[0], stats[1]
        y = torch.empty_like(x)
        L = _lib.lib()
        ws = _workspace(L.d3_bn_ws_bytes(Cc), x.device, "bn")
        with _on(x.device):
            check(L.d3_bn_stats(_ptr(x), M, Cc, _ptr(mean), _ptr(var),
                                _ptr(running_mean) if running_mean is not None else None,
                                _ptr(running_var) if running_var is not None else None, float(momentum or 0.0),
                                _ptr(ws), ws.numel(), _stream()), "bn_stats")
            check(L.d3_bn_relu_fwd(_ptr(x), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), _ptr(y), M, Cc, eps,
                                   int(relu), _stream()), "bn_relu_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (eps, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        mean, var = stats[0], stats[1]
        eps, relu = ctx.cfg
        dy = dy.contiguous()
        M, Cc = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        ws = _workspace(_lib.lib().d3_bn_ws_bytes(Cc), x.device, "bn")
        with _on(x.device):
            check(_lib.lib().d3_bn_relu_bwd(_ptr(x), _ptr(dy), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta),
                                            _ptr(dx), _ptr(dgamma), _ptr(dbeta), M, Cc, eps, int(relu), _ptr(ws),
                                            ws.numel(), _stream()), "bn_relu_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None


class BatchNormEvalFunction(Function):
    """eval mode: normalise with the running statistics (no batch reduction)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, relu):
        x = x.contiguous()
        y = torch.empty_like(x)
        with _on(x.device):
            check(_lib.lib().d3_bn_relu_fwd(_ptr(x), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), _ptr(y),
                                            x.size(0), x.size(1), eps, int(relu), _stream()), "bn_relu_fwd")
        return y


# ------------------------------------------------------------------------------------------ modules
_KERNEL_PERM = {}   # kernel volume -> index tensor: kernel_here[k] = kernel_checkpoint[perm[k]]


def kernel_permutation(kernel_size, order):
    """Permutation that converts a (K^3, Cin, Cout) kernel stored with offset order `order` into this module's x-fastest
    order.  order: "xyz" (identity: x fastest) or "zyx" (z fastest, x slowest -- a row-major (x, y, z) region iterator)."""
    K = kernel_size
    if order == "xyz":
        return torch.arange(K ** 3)
    if order != "zyx":
        raise ValueError("order must be 'xyz' or 'zyx'")
    perm = torch.empty(K ** 3, dtype=torch.long)
    for oz in range(K):
        for oy in range(K):
            for ox in range(K):
                perm[ox + K * oy + K * K * oz] = oz + K * oy + K * K * ox
    return perm


def set_kernel_order(order="xyz"):
    """Checkpoints whose convolution kernels are in `order` are permuted to x-fastest while they are loaded
    (`load_state_dict` of any module containing MinkowskiConvolution[Transpose]); "xyz" removes the hook's effect."""
    _KERNEL_PERM.clear()
    if order != "xyz":
        for ks in (2, 3):
            _KERNEL_PERM[ks ** 3] = kernel_permutation(ks, order)


def _permute_kernel_on_load(module, state_dict, prefix, *args):
    key = prefix + "kernel"
    perm = _KERNEL_PERM.get(module.kernel_volume)
    if perm is not None and key in state_dict and state_dict[key].dim() == 3:
        state_dict[key] = state_dict[key].index_select(0, perm.to(state_dict[key].device))


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not bias, "only what the reference uses is implemented"
        assert (kernel_size, stride) in ((3, 1), (2, 2), (1, 1)), (kernel_size, stride)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.kernel_volume = kernel_size ** 3
        shape = (in_channels, out_channels) if self.kernel_volume == 1 else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = None
        self.reset_parameters()
        self._register_load_state_dict_pre_hook(_permute_kernel_on_load, with_module=True)

    def reset_parameters(self, is_transpose=False):
        n = (self.out_channels if is_transpose else self.in_channels) * self.kernel_volume
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def maps(self, x):
        """-> (forward table, transposed table, output rows, backward flags, output tensor stride)"""
        cm, ts = x.coordinate_manager, x.tensor_stride
        if self.kernel_size == 3:
            nbr = cm.k3(ts)
            return nbr, nbr, nbr.size(0), D3_CONV_FLIPK, ts
        if self.kernel_size == 1:
            return None, None, x.F.size(0), 0, ts
        child, up, Mo = cm.down(ts)
        return child, up, Mo, 0, 2 * ts

    def forward(self, x):
        if x._conv_done is self:      # already applied inside the fused BN -> ReLU -> conv unit
            x._conv_done = None
            return x
        tbl_f, tbl_b, Mout, bflags, ts_out = self.maps(x)
        return x._like(SparseConvFunction.apply(x.F, self.kernel, tbl_f, tbl_b, Mout, bflags), ts_out)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d" % (self.in_channels, self.out_channels, self.kernel_size, self.stride)


class MinkowskiConvolutionTranspose(MinkowskiConvolution):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        assert (kernel_size, stride) == (2, 2)
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, dimension)
        self.reset_parameters(True)

    def maps(self, x):
        cm, ts = x.coordinate_manager, x.tensor_stride
        assert ts % 2 == 0, "transposed conv lands on the cached finer coordinates"
        child, up, Mo = cm.down(ts // 2)
        assert Mo == x.F.size(0)
        return up, child, up.size(0), 0, ts // 2


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over the rows; parameters live in `.bn` exactly as in MinkowskiEngine."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self.fused_relu = False  # set by fuse_bn_relu(): the following MinkowskiReLU becomes a no-op
        self._steps = 0
        self._register_state_dict_hook(MinkowskiBatchNorm._sync_counter)

    @staticmethod
    def _sync_counter(module, state_dict, prefix, local_metadata):
        # num_batches_tracked is only bookkeeping (momentum is not None): keep it exact without a launch per step
        if module._steps:
            module.bn.num_batches_tracked += module._steps
            module._steps = 0
            state_dict[prefix + "bn.num_batches_tracked"] = module.bn.num_batches_tracked

    def forward(self, x):
        bn = self.bn
        conv = self.__dict__.get("fused_conv")
        if conv is not None and self.fused_relu and (self.training or not bn.track_running_stats):
            track = bn.track_running_stats
            tbl_f, tbl_b, Mout, bflags, ts_out = conv.maps(x)
            f = PreActConvFunction.apply(x.F, bn.weight, bn.bias, conv.kernel, tbl_f, tbl_b, Mout, bflags, bn.eps, True,
                                         bn.running_mean if track else None, bn.running_var if track else None, bn.momentum)
            if track:
                self._steps += 1
            out = x._like(f, ts_out)
            out._relu_done, out._conv_done = True, conv
            return out
        if self.training or not bn.track_running_stats:
            track = bn.track_running_stats
            y = BatchNormReLUFunction.apply(x.F, bn.weight, bn.bias, bn.eps, self.fused_relu,
                                            bn.running_mean if track else None, bn.running_var if track else None,
                                            bn.momentum)   # running statistics are updated inside the stats kernel
            if track:
                self._steps += 1   # num_batches_tracked is synchronised lazily (state_dict / eval)
        else:
            y = BatchNormEvalFunction.apply(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                            self.fused_relu)
        out = x._like(y)
        out._relu_done = self.fused_relu
        return out


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        if x._relu_done:
            return x
        return x._like(torch.relu(x.F))


def fuse_bn_relu(module):
    """Mark every MinkowskiBatchNorm that is directly followed by a MinkowskiReLU inside an nn.Sequential so
    that one kernel applies both (the module tree and its state-dict keys are unchanged)."""
    for m in module.modules():
        if isinstance(m, nn.Sequential):
            kids = list(m.children())
            for a, b in zip(kids[:-1], kids[1:]):
                if isinstance(a, MinkowskiBatchNorm) and isinstance(b, MinkowskiReLU):
                    a.fused_relu = True
            # [BN, ReLU, conv] triples additionally run as one fused unit (PreActConvFunction); the conv is kept out
            # of the BN module's children (plain __dict__ entry) so the module tree and state-dict keys are unchanged
            for a, b, c in zip(kids[:-2], kids[1:-1], kids[2:]):
                if isinstance(a, MinkowskiBatchNorm) and isinstance(b, MinkowskiReLU) and isinstance(c, MinkowskiConvolution):
                    a.__dict__["fused_conv"] = c
    return module
