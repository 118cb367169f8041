"""`nn.Linear` / kernel-size-1 `nn.Conv1d` / residual `nn.LayerNorm` of the transformer match module on the hand-written kernels
(reference: model/transformer/attention.py:20-23,61-75,149-176 `fc_q/fc_k/fc_v/fc_o` + `layer_norm`; model/match_module.py:160-187
`features_concat`, `match`, `lang_fc`).  Same parameters, same state-dict keys -- only the arithmetic moves:

  * `linear(x, W, b)`: y = x W^T + b as ONE `d3_hgemm` problem (csrc/hgemm.hip: exact fp32 products on v_mfma_f32_16x16x4_f32);
    backward dx = dy W and dW = dy^T x are two problems of ONE launch (the k-major operand forms), db a fixed-order column sum;
  * `linear_multi([...])`: several projections (fc_q / fc_k / fc_v of one attention layer) share a launch, forward and backward;
  * `add_layer_norm(a, b, ln)`: LayerNorm(a + b) in one pass (csrc/layernorm.hip).
CPU tensors, other dtypes and double backward fall through to the library ops."""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib
from ._lib import GemmProb, GemmSeg, check
from .pointgroup_ops import _on, _ptr, _stream, _workspace


def _seg(A, lda, a_km, B, ldb, b_km, K):
    s = GemmSeg()
    s.A, s.lda, s.a_kmajor = A.data_ptr(), lda, int(a_km)
    s.B, s.ldb, s.b_kmajor = B.data_ptr(), ldb, int(b_km)
    s.ia = None
    s.K = K
    return s


def _prob(seg, M, N, Cmat, ldc, bias=None, relu=False):
    p = GemmProb()
    p.seg[0] = seg
    p.nseg, p.M, p.N = 1, M, N
    p.C, p.ldc = Cmat.data_ptr(), ldc
    p.bias = bias.data_ptr() if bias is not None else None
    p.add, p.ldadd = None, 0
    p.relu, p.accum, p.perm_nb, p.perm_s = int(relu), 0, 0, 0
    return p


def _launch(probs, dev):
    L = _lib.lib()
    with _on(dev):
        for i in range(0, len(probs), 4):      # up to 4 problems per launch
            chunk = probs[i:i + 4]
            arr = (GemmProb * len(chunk))(*chunk)
            check(L.d3_hgemm(arr, len(chunk), _stream()), "hgemm")


def _native_ok(*ts):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in ts)


class _LinearMulti(Function):
    """(x_i, W_i, b_i) triples flattened into the argument list; relu: apply max(., 0) to every output"""

    @staticmethod
    def forward(ctx, relu, n, *args):
        xs, Ws, bs = args[0:n], args[n:2 * n], args[2 * n:3 * n]
        xs = [x.contiguous() for x in xs]
        Ws = [W.contiguous() for W in Ws]
        dev = xs[0].device
        ys, probs = [], []
        for x, W, b in zip(xs, Ws, bs):
            M, K = x.shape
            N = W.shape[0]
            y = torch.empty((M, N), dtype=torch.float32, device=dev)
            probs.append(_prob(_seg(x, K, False, W, K, False, K), M, N, y, N, bias=b, relu=relu))
            ys.append(y)
        _launch(probs, dev)
        ctx.n, ctx.relu = n, relu
        ctx.has_b = [b is not None for b in bs]
        ctx.save_for_backward(*xs, *Ws, *(ys if relu else ()))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, relu = ctx.n, ctx.relu
        saved = ctx.saved_tensors
        xs, Ws = saved[0:n], saved[n:2 * n]
        ys = saved[2 * n:3 * n] if relu else (None,) * n
        dev = xs[0].device
        L = _lib.lib()
        probs, dxs, dWs, dbs = [], [], [], []
        keep = []       # operands referenced by `probs` stay alive until the launch is enqueued
        for i in range(n):
            x, W = xs[i], Ws[i]
            M, K = x.shape
            N = W.shape[0]
            dy = dys[i]
            if dy is None:
                dxs.append(None); dWs.append(None); dbs.append(None)
                continue
            dy = dy.contiguous()
            if relu:
                dy = dy * (ys[i] > 0).to(dy.dtype)
            dx = dW = db = None
            if ctx.needs_input_grad[2 + i]:            # dx = dy W : B k-major
                dx = torch.empty((M, K), dtype=torch.float32, device=dev)
                probs.append(_prob(_seg(dy, N, False, W, K, True, N), M, K, dx, K))
            if ctx.needs_input_grad[2 + n + i]:        # dW = dy^T x : both operands k-major
                dW = torch.empty((N, K), dtype=torch.float32, device=dev)
                probs.append(_prob(_seg(dy, N, True, x, K, True, M), N, K, dW, K))
            if ctx.has_b[i] and ctx.needs_input_grad[2 + 2 * n + i]:
                db = torch.empty(N, dtype=torch.float32, device=dev)
                ws = _workspace(L.d3_colsum_ws_bytes(N), dev, "colsum")
                with _on(dev):
                    check(L.d3_colsum(_ptr(dy), N, M, N, _ptr(db), 0, _ptr(ws), ws.numel(), _stream()), "colsum")
            dxs.append(dx); dWs.append(dW); dbs.append(db)
            keep.append(dy)
        if probs:
            _launch(probs, dev)
        del keep
        return (None, None) + tuple(dxs) + tuple(dWs) + tuple(dbs)


def linear_multi(items, relu=False):
    """[(x (..., K), W (N, K), b (N) | None), ...] -> [x W^T + b, ...]; all problems of a call share launches"""
    xs = [x for x, _, _ in items]
    if not _native_ok(*[t for it in items for t in it]):
        out = []
        for x, W, b in items:
            y = torch.nn.functional.linear(x, W, b)
            out.append(torch.relu(y) if relu else y)
        return out
    flat = [x.reshape(-1, x.shape[-1]) for x in xs]
    ys = _LinearMulti.apply(relu, len(items), *flat, *[W for _, W, _ in items], *[b for _, _, b in items])
    return [y.view(*x.shape[:-1], y.shape[-1]) for x, y in zip(xs, ys)]


def linear(x, W, b=None, relu=False):
    return linear_multi([(x, W, b)], relu)[0]


class _AddLayerNorm(Function):
    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        D = a.shape[-1]
        R = a.numel() // D
        y = torch.empty_like(a)
        mean = torch.empty(R, dtype=torch.float32, device=a.device)
        rstd = torch.empty(R, dtype=torch.float32, device=a.device)
        with _on(a.device):
            check(_lib.lib().d3_layernorm_fwd(_ptr(a), _ptr(b) if b is not None else None, _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                              _ptr(rstd), R, D, float(eps), _stream()), "layernorm_fwd")
        ctx.save_for_backward(a, b if b is not None else a.new_empty(0), gamma, mean, rstd)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        D = a.shape[-1]
        R = a.numel() // D
        L = _lib.lib()
        dx = torch.empty_like(a)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = _workspace(L.d3_layernorm_ws_bytes(R, D), a.device, "lnbwd")
        with _on(a.device):
            check(L.d3_layernorm_bwd(_ptr(a), _ptr(b) if ctx.has_b else None, _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dy), _ptr(dx),
                                     _ptr(dgamma), _ptr(dbeta), R, D, _ptr(ws), ws.numel(), _stream()), "layernorm_bwd")
        return dx, (dx if ctx.has_b else None), dgamma, dbeta, None


def add_layer_norm(a, b, ln):
    """ln(a + b) (b may be None) for an nn.LayerNorm over the last dimension"""
    # the kernels index `b` with a's (R, D) shape: a broadcastable b or a LayerNorm over another width takes the library path
    if (not _native_ok(a, b, ln.weight, ln.bias) or ln.weight is None or len(ln.normalized_shape) != 1 or a.shape[-1] > 1024
            or ln.normalized_shape[0] != a.shape[-1] or (b is not None and b.shape != a.shape)):
        return ln(a if b is None else a + b)
    return _AddLayerNorm.apply(a, b, ln.weight, ln.bias, ln.eps)
