"""PointGroup operators on MI355X -- same names, arguments and return values as the reference's
`lib.pointgroup_ops.functions.pointgroup_ops` (reference:
lib/pointgroup_ops/functions/pointgroup_ops.py:39,75,112,150,182,221,253,281,309,337), each one a
thin autograd wrapper over the C ABI of libd3hip.so (include/d3hip.h).

Differences from the reference, all at the edges:
  * `voxelization_idx` and `bfs_cluster` run on the device (the reference runs them on the host).
    CPU inputs are accepted as in the reference -- they are moved to the current device, processed
    there, and CPU tensors are returned; device inputs stay on the device (the fast path used by
    d3net_amd.pointgroup).
  * `ballquery_batch_p` is count -> allocate -> fill instead of guess-and-retry; `start` values
    are the exclusive prefix sum of `len` (the reference's depend on thread scheduling).
There is no CPU implementation here: without a GPU these functions raise.
"""
import ctypes as C
import threading

import torch
from torch.autograd import Function

from . import _lib
from ._lib import check

_ws_cache = {}


# the calling thread's current stream / device straight from the runtime bindings: `torch.cuda.current_stream()` builds a
# Stream object (~5 us) and `torch.cuda.current_device()` walks the lazy-init checks; at ~45 library calls per step inside
# the host-bound stretches of the step (after a count phase the host has no lead over the GPU) that is time the GPU waits for
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _workspace(nbytes, device, tag):
    """A cached, growing device scratch buffer per (device, tag, host thread, current stream)."""
    # per host thread AND per stream: the two clustering branches of PointGroup.forward run concurrently on their own streams, from
    # two threads (the helper-thread fallback beyond the padded lists' budget) or from one -- and a call may return with kernels that read its workspace still in flight
    # (d3_bfs_cluster_run's speculative fill), so one thread driving two streams must never hand both the same buffer (r05_f: the
    # 16-scene batch, whose lists go the compact way, faulted exactly so)
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag, threading.get_ident(),
           (_stream().value or 0) if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def _on(device):
    """device guard for the raw library calls: a no-op (sub-microsecond) when `device` is already current --
    `torch.cuda.device(...)` costs ~10 us per use, which at ~600 operator calls per step is host time the GPU waits for"""
    if device.index is None:
        return _NOGUARD
    cur = _RAW_DEVICE() if _RAW_DEVICE is not None else torch.cuda.current_device()
    return _NOGUARD if device.index == cur else torch.cuda.device(device)


def _device_of(*tensors):
    for t in tensors:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise _lib.D3Error("d3net_amd.pointgroup_ops needs a GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class Voxelization_Idx(Function):
    @staticmethod
    def forward(ctx, coords, batchsize, mode=4):
        """
        :param coords:  long (N, dimension + 1) or (N, dimension), dimension = 3
        :param batchsize: int (unused by the algorithm, as in the reference beyond pre-sizing)
        :param mode: int 4=mean
        :return: output_coords long (M, dimension + 1); input_map int (N,); output_map int (M, maxActive + 1)
        """
        assert coords.is_contiguous()
        assert coords.dtype == torch.int64
        on_cpu = not coords.is_cuda
        dev = _device_of(coords)
        c = coords.to(dev) if on_cpu else coords
        N, ncols = c.shape
        L = _lib.lib()
        with _on(dev):
            input_map = torch.empty(N, dtype=torch.int32, device=dev)
            ws = _workspace(L.d3_voxelize_idx_ws_bytes(N), dev, "vi")
            M, mA = C.c_int(0), C.c_int(1)
            check(L.d3_voxelize_idx_count(_ptr(c), N, ncols, int(mode), _ptr(input_map), _ptr(ws), ws.numel(),
                                          C.byref(M), C.byref(mA), _stream()), "voxelize_idx_count")
            M, mA = M.value, mA.value
            output_coords = torch.empty((M, ncols), dtype=torch.int64, device=dev)
            output_map = torch.empty((M, mA + 1), dtype=torch.int32, device=dev)
            check(L.d3_voxelize_idx_fill(_ptr(c), N, ncols, int(mode), _ptr(input_map), _ptr(ws), ws.numel(),
                                         _ptr(output_coords), _ptr(output_map), M, mA, _stream()),
                  "voxelize_idx_fill")
        if on_cpu:
            return output_coords.cpu(), input_map.cpu(), output_map.cpu()
        return output_coords, input_map, output_map

    @staticmethod
    def backward(ctx, a=None, b=None, c=None):
        return None


voxelization_idx = Voxelization_Idx.apply


class Voxelization(Function):
    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        """
        :param map_rule: cuda int (M, maxActive + 1)
        :param feats: cuda float (N, C)
        :return: output_feats: cuda float (M, C)
        """
        assert map_rule.is_contiguous() and map_rule.is_cuda and map_rule.dtype == torch.int32
        assert feats.is_contiguous() and feats.is_cuda and feats.dtype == torch.float32
        N, Cc = feats.size()
        M = map_rule.size(0)
        maxActive = map_rule.size(1) - 1
        output_feats = torch.zeros((M, Cc), dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, mode, maxActive, N)
        with _on(feats.device):
            check(_lib.lib().d3_voxelize_fp(_ptr(feats), _ptr(output_feats), _ptr(map_rule), int(mode), M, maxActive,
                                            Cc, _stream()), "voxelize_fp")
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, mode, maxActive, N = ctx.for_backwards
        M, Cc = d_output_feats.size()
        d_output_feats = d_output_feats.contiguous()
        d_feats = torch.zeros((N, Cc), dtype=torch.float32, device=d_output_feats.device)
        with _on(d_feats.device):
            check(_lib.lib().d3_voxelize_bp(_ptr(d_output_feats), _ptr(d_feats), _ptr(map_rule), int(mode), M,
                                            maxActive, Cc, _stream()), "voxelize_bp")
        return d_feats, None, None


voxelization = Voxelization.apply


class PointRecover(Function):
    @staticmethod
    def forward(ctx, feats, map_rule, nPoint):
        """
        :param feats: cuda float M * C
        :param map_rule: cuda int M * (maxActive + 1)
        :param nPoint: int
        :return: output_feats: cuda float N * C
        """
        assert map_rule.is_contiguous() and map_rule.is_cuda
        assert feats.is_contiguous() and feats.is_cuda
        M, Cc = feats.size()
        maxActive = map_rule.size(1) - 1
        output_feats = torch.zeros((nPoint, Cc), dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, maxActive, M)
        with _on(feats.device):
            check(_lib.lib().d3_point_recover_fp(_ptr(feats), _ptr(output_feats), _ptr(map_rule), M, maxActive, Cc,
                                                 _stream()), "point_recover_fp")
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, maxActive, M = ctx.for_backwards
        N, Cc = d_output_feats.size()
        d_output_feats = d_output_feats.contiguous()
        d_feats = torch.zeros((M, Cc), dtype=torch.float32, device=d_output_feats.device)
        with _on(d_feats.device):
            check(_lib.lib().d3_point_recover_bp(_ptr(d_output_feats), _ptr(d_feats), _ptr(map_rule), M, maxActive,
                                                 Cc, _stream()), "point_recover_bp")
        return d_feats, None, None


point_recover = PointRecover.apply


class BallQueryBatchP(Function):
    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, radius, meanActive):
        """
        :param coords: (n, 3) float
        :param batch_idxs: (n) int
        :param batch_offsets: (B+1) int
        :param radius: float
        :param meanActive: int (only a sizing hint in the reference; unused here)
        :return: idx (nActive), int
        :return: start_len (n, 2), int
        """
        n = coords.size(0)
        assert coords.is_contiguous() and coords.is_cuda and coords.dtype == torch.float32
        assert batch_idxs.is_contiguous() and batch_idxs.is_cuda and batch_idxs.dtype == torch.int32
        assert batch_offsets.is_contiguous() and batch_offsets.is_cuda and batch_offsets.dtype == torch.int32
        dev = coords.device
        L = _lib.lib()
        with _on(dev):
            start_len = torch.empty((n, 2), dtype=torch.int32, device=dev)
            # single-pass ball query (hits stashed by the count phase) when the stash fits 2 GB, else count + second search
            big = L.d3_ballquery_ws_bytes_single_pass(n)
            ws = _workspace(big if big <= (2 << 30) else L.d3_ballquery_ws_bytes(n), dev, "bq")
            nActive = C.c_int(0)
            check(L.d3_ballquery_count(_ptr(coords), _ptr(batch_idxs), _ptr(batch_offsets), n, float(radius),
                                       _ptr(start_len), _ptr(ws), ws.numel(), C.byref(nActive), _stream()),
                  "ballquery_count")
            nActive = nActive.value
            idx = torch.empty(max(nActive, 1), dtype=torch.int32, device=dev)
            check(L.d3_ballquery_fill(_ptr(coords), _ptr(batch_idxs), _ptr(batch_offsets), n, float(radius),
                                      _ptr(start_len), _ptr(ws), ws.numel(), _ptr(idx), nActive, _stream()),
                  "ballquery_fill")
        return idx[:nActive], start_len

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


ballquery_batch_p = BallQueryBatchP.apply


def ballquery_padded_fits(n, max_bytes=None):
    """does `ballquery_batch_p_padded` take n points?  (n * cap slots must stay inside the int range of start_len -- the library's own
    bound, d3_ballquery_padded -- and, when given, inside `max_bytes`)"""
    cap = _lib.lib().d3_ballquery_cap()
    return n > 0 and n * cap <= 0x7FFFFFFF and (max_bytes is None or n * cap * 4 <= max_bytes)


def padded_clustering_bytes(n, branches=2):
    """device bytes the padded clustering of n object points holds at its peak: per branch the n * cap neighbour slots (4 B) and the
    BFS replay's edge records sized from the same capacity (d3_bfs_cluster_erec_bytes: 16 B per slot) -- 20 KB per point and branch,
    whatever the real nActive is.  PointGroup.forward keeps this under its `padded_list_budget` and falls back to the
    compact lists (`ballquery_batch_p`) beyond it"""
    return int(branches) * int(n) * _lib.lib().d3_ballquery_cap() * 20


def ballquery_batch_p_padded(coords, batch_idxs, batch_offsets, radius, max_bytes=None, ws_tag=""):
    """Sync-free ball query for callers that hand the result straight to `bfs_cluster`: every point owns a fixed slot
    of `cap` entries (start_len[q] = (s * cap, len) with s = q, or the leader of q's clique cell whose list q shares:
    csrc/ballquery.hip), so there is no nActive to fetch, no scan and no compaction.  Same
    neighbours in the same order as `ballquery_batch_p` (lib/pointgroup_ops/functions/pointgroup_ops.py:143-180);
    returns None when the padded buffer cannot be addressed (`ballquery_padded_fits`: n * cap beyond the int range, ~2.1 M points -- up
    to round 4 the bound was 2 GiB of slots, a quarter of that, and the 8-scene strong-scaling batch fell back to the compact form with
    its host round trip) or would exceed `max_bytes` (the caller then uses `ballquery_batch_p`)."""
    n = coords.size(0)
    L = _lib.lib()
    cap = L.d3_ballquery_cap()
    if not ballquery_padded_fits(n, max_bytes):
        return None
    assert coords.is_contiguous() and coords.is_cuda and coords.dtype == torch.float32
    assert batch_idxs.is_contiguous() and batch_idxs.is_cuda and batch_idxs.dtype == torch.int32
    assert batch_offsets.is_contiguous() and batch_offsets.is_cuda and batch_offsets.dtype == torch.int32
    dev = coords.device
    with _on(dev):
        start_len = torch.empty((n, 2), dtype=torch.int32, device=dev)
        idx = torch.empty(n * cap, dtype=torch.int32, device=dev)
        ws = _workspace(L.d3_ballquery_ws_bytes(n), dev, "bqp" + ws_tag)      # (ws_tag: two clusterings driven by ONE thread need two workspaces)
        check(L.d3_ballquery_padded(_ptr(coords), _ptr(batch_idxs), _ptr(batch_offsets), n, float(radius), _ptr(start_len),
                                    _ptr(ws), ws.numel(), _ptr(idx), _stream()), "ballquery_padded")
    return idx, start_len


class BFSCluster(Function):
    @staticmethod
    def forward(ctx, semantic_label, ball_query_idxs, start_len, threshold, ascending=False):
        """
        :param semantic_label: (N), int
        :param ball_query_idxs: (nActive), int
        :param start_len: (N, 2), int
        :param ascending: (not in the reference) the caller guarantees ascending neighbour lists -- ballquery_batch_p's order;
            lets the label propagation skip the useless prefix of every list.  Same results.
        :return: cluster_idxs:  int (sumNPoint, 2), dim 0 for cluster_id, dim 1 for corresponding point idxs in N
        :return: cluster_offsets: int (nCluster + 1)
        """
        N = start_len.size(0)
        assert semantic_label.is_contiguous() and semantic_label.dtype == torch.int32
        assert ball_query_idxs.is_contiguous() and ball_query_idxs.dtype == torch.int32
        assert start_len.is_contiguous() and start_len.dtype == torch.int32
        on_cpu = not semantic_label.is_cuda
        dev = _device_of(semantic_label, ball_query_idxs, start_len)
        sem, idx, sl = (t.to(dev) for t in (semantic_label, ball_query_idxs, start_len))
        if idx.numel() == 0:
            idx = torch.zeros(1, dtype=torch.int32, device=dev)
        L = _lib.lib()
        with _on(dev):
            ws = _workspace(L.d3_bfs_cluster_ws_bytes(N), dev, "cl")
            S, P = C.c_int(0), C.c_int(0)
            nact = int(idx.numel())
            rec = _workspace(L.d3_bfs_cluster_erec_bytes(nact), dev, "clrec")
            # ONE native call for count + fill: outputs at their upper bounds (N points, N / threshold + 1 clusters), sliced below --
            # going back to the interpreter between the phases costs the interpreter lock when the other clustering branch is busy
            capP, capC = max(N, 1), N // max(int(threshold), 1) + 1
            cluster_idxs = torch.empty((capP, 2), dtype=torch.int32, device=dev)
            cluster_offsets = torch.empty(capC + 1, dtype=torch.int32, device=dev)
            check(L.d3_bfs_cluster_run(_ptr(sem), _ptr(idx), _ptr(sl), N, int(threshold), _ptr(ws), ws.numel(), _ptr(rec), rec.numel(), nact,
                                       1 if ascending else 0, _ptr(cluster_idxs), capP, _ptr(cluster_offsets), capC,
                                       C.byref(S), C.byref(P), _stream()), "bfs_cluster_run")
            cluster_idxs, cluster_offsets = cluster_idxs[:S.value], cluster_offsets[:P.value + 1]
            if N == 0:
                cluster_offsets.zero_()
        if on_cpu:
            return cluster_idxs.cpu(), cluster_offsets.cpu()
        return cluster_idxs, cluster_offsets

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


bfs_cluster = BFSCluster.apply


class _ClusterRun:
    """an enqueued clustering (d3_bfs_cluster_begin): its buffers stay referenced until bfs_cluster_end"""
    __slots__ = ("ticket", "keep", "cluster_idxs", "cluster_offsets", "N", "dev")


def bfs_cluster_begin(semantic_label, ball_query_idxs, start_len, threshold, ascending=False, ws_tag=""):
    """`bfs_cluster` cut at its one host wait (d3_bfs_cluster_begin / _end): everything is enqueued on the current stream here, the
    cluster counts are read by `bfs_cluster_end`.  One thread keeps several clusterings in flight on different streams (begin, begin,
    end, end); ws_tag separates their cached workspaces.  Device tensors only."""
    N = start_len.size(0)
    assert semantic_label.is_cuda and semantic_label.is_contiguous() and semantic_label.dtype == torch.int32
    assert ball_query_idxs.is_contiguous() and ball_query_idxs.dtype == torch.int32
    assert start_len.is_contiguous() and start_len.dtype == torch.int32
    dev = semantic_label.device
    idx = ball_query_idxs if ball_query_idxs.numel() > 0 else torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    r = _ClusterRun()
    with _on(dev):
        ws = _workspace(L.d3_bfs_cluster_ws_bytes(N), dev, "cl" + ws_tag)
        nact = int(idx.numel())
        rec = _workspace(L.d3_bfs_cluster_erec_bytes(nact), dev, "clrec" + ws_tag)
        capP, capC = max(N, 1), N // max(int(threshold), 1) + 1
        r.cluster_idxs = torch.empty((capP, 2), dtype=torch.int32, device=dev)
        r.cluster_offsets = torch.empty(capC + 1, dtype=torch.int32, device=dev)
        tk = C.c_void_p()
        check(L.d3_bfs_cluster_begin(_ptr(semantic_label), _ptr(idx), _ptr(start_len), N, int(threshold), _ptr(ws), ws.numel(), _ptr(rec),
                                     rec.numel(), nact, 1 if ascending else 0, _ptr(r.cluster_idxs), capP, _ptr(r.cluster_offsets), capC,
                                     C.byref(tk), _stream()), "bfs_cluster_begin")
    r.ticket, r.keep, r.N, r.dev = tk, (semantic_label, idx, start_len, ws, rec), N, dev
    return r


def bfs_cluster_end(r):
    """-> (cluster_idxs (sumNPoint, 2), cluster_offsets (nCluster + 1)) of a `bfs_cluster_begin`"""
    S, P = C.c_int(0), C.c_int(0)
    tk, r.ticket = r.ticket, None
    with _on(r.dev):
        check(_lib.lib().d3_bfs_cluster_end(tk, C.byref(S), C.byref(P)), "bfs_cluster_end")
    ci, co = r.cluster_idxs[:S.value], r.cluster_offsets[:P.value + 1]
    if r.N == 0:
        co.zero_()
    r.keep = None
    return ci, co


class RoiPool(Function):
    @staticmethod
    def forward(ctx, feats, proposals_offset):
        """
        :param feats: (sumNPoint, C) float
        :param proposals_offset: (nProposal + 1) int
        :return: output_feats (nProposal, C) float
        """
        nProposal = proposals_offset.size(0) - 1
        sumNPoint, Cc = feats.size()
        assert feats.is_contiguous() and feats.is_cuda and feats.dtype == torch.float32
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda and proposals_offset.dtype == torch.int32
        output_feats = torch.empty((nProposal, Cc), dtype=torch.float32, device=feats.device)
        output_maxidx = torch.empty((nProposal, Cc), dtype=torch.int32, device=feats.device)
        with _on(feats.device):
            check(_lib.lib().d3_roipool_fp(_ptr(feats), _ptr(proposals_offset), _ptr(output_feats),
                                           _ptr(output_maxidx), nProposal, Cc, _stream()), "roipool_fp")
        ctx.for_backwards = (output_maxidx, proposals_offset, sumNPoint)
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        nProposal, Cc = d_output_feats.size()
        output_maxidx, proposals_offset, sumNPoint = ctx.for_backwards
        d_output_feats = d_output_feats.contiguous()
        d_feats = torch.zeros((sumNPoint, Cc), dtype=torch.float32, device=d_output_feats.device)
        with _on(d_feats.device):
            check(_lib.lib().d3_roipool_bp(_ptr(d_feats), _ptr(proposals_offset), _ptr(output_maxidx),
                                           _ptr(d_output_feats), nProposal, Cc, _stream()), "roipool_bp")
        return d_feats, None


roipool = RoiPool.apply


class GetIoU(Function):
    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum):
        """
        :param proposals_idx: (sumNPoint), int
        :param proposals_offset: (nProposal + 1), int
        :param instance_labels: (N), long, 0~total_nInst-1, -1
        :param instance_pointnum: (total_nInst), int
        :return: proposals_iou: (nProposal, total_nInst), float
        """
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        assert proposals_idx.is_contiguous() and proposals_idx.is_cuda and proposals_idx.dtype == torch.int32
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda and proposals_offset.dtype == torch.int32
        assert instance_labels.is_contiguous() and instance_labels.is_cuda and instance_labels.dtype == torch.int64
        assert instance_pointnum.is_contiguous() and instance_pointnum.is_cuda and instance_pointnum.dtype == torch.int32
        proposals_iou = torch.empty((nProposal, nInstance), dtype=torch.float32, device=proposals_idx.device)
        with _on(proposals_idx.device):
            check(_lib.lib().d3_get_iou(_ptr(proposals_idx), _ptr(proposals_offset), _ptr(instance_labels),
                                        _ptr(instance_pointnum), _ptr(proposals_iou), nInstance, nProposal,
                                        _stream()), "get_iou")
        return proposals_iou

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


get_iou = GetIoU.apply


def _sec(name, inp, offsets):
    nProposal = offsets.size(0) - 1
    Cc = inp.size(1)
    assert inp.is_contiguous() and inp.is_cuda and inp.dtype == torch.float32
    assert offsets.is_contiguous() and offsets.is_cuda and offsets.dtype == torch.int32
    out = torch.empty((nProposal, Cc), dtype=torch.float32, device=inp.device)
    with _on(inp.device):
        check(getattr(_lib.lib(), name)(_ptr(inp), _ptr(offsets), _ptr(out), nProposal, Cc, _stream()), name)
    return out


class SecMean(Function):
    @staticmethod
    def forward(ctx, inp, offsets):
        """:param inp: (N, C) float  :param offsets: (nProposal + 1) int  :return: out (nProposal, C) float"""
        return _sec("d3_sec_mean", inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


sec_mean = SecMean.apply


class SecMin(Function):
    @staticmethod
    def forward(ctx, inp, offsets):
        """:param inp: (N, C) float  :param offsets: (nProposal + 1) int  :return: out (nProposal, C) float"""
        return _sec("d3_sec_min", inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


sec_min = SecMin.apply


class SecMax(Function):
    @staticmethod
    def forward(ctx, inp, offsets):
        """:param inp: (N, C) float  :param offsets: (nProposal + 1) int  :return: out (nProposal, C) float"""
        return _sec("d3_sec_max", inp, offsets)

    @staticmethod
    def backward(ctx, a=None):
        return None, None


sec_max = SecMax.apply


def voxelization_cat(feats_a, feats_b, map_rule, mode=4):
    """voxelization(torch.cat((feats_a, feats_b), 1), map_rule, mode) for inputs that need no gradient (the network input:
    model/pointgroup.py:468-471), without the concatenated copy (d3_voxelize_fp2)"""
    assert map_rule.is_contiguous() and map_rule.is_cuda and map_rule.dtype == torch.int32
    assert feats_a.is_cuda and feats_a.dtype == torch.float32 and feats_b.dtype == torch.float32 and feats_a.size(0) == feats_b.size(0)
    assert not feats_a.requires_grad and not feats_b.requires_grad
    feats_a, feats_b = feats_a.contiguous(), feats_b.contiguous()
    M, maxActive = map_rule.size(0), map_rule.size(1) - 1
    out = torch.empty((M, feats_a.size(1) + feats_b.size(1)), dtype=torch.float32, device=feats_a.device)
    with _on(feats_a.device):
        check(_lib.lib().d3_voxelize_fp2(_ptr(feats_a), feats_a.size(1), _ptr(feats_b), feats_b.size(1), _ptr(out), _ptr(map_rule),
                                         int(mode), M, maxActive, _stream()), "voxelize_fp2")
    return out


def cluster_select(locs, pt_offsets, semantic_preds, batch_idxs, object_idxs, batch_size=None):
    """-> batch_idxs_ (n) int32, coords_ (n,3), shifted (n,3) = coords_ + offsets_, semantic_preds_ (n) int32 of the object points
    (d3_cluster_select: model/pointgroup.py:288-296 in one pass); with batch_size also their (batch_size + 1) batch offsets
    (:296 get_batch_offsets, for the SORTED id column the collate function builds: d3_cluster_select2) as a fifth value"""
    locs, pt_offsets = locs.contiguous(), pt_offsets.contiguous()
    assert locs.dtype == torch.float32 and pt_offsets.dtype == torch.float32 and semantic_preds.dtype == torch.int64
    assert batch_idxs.dtype == torch.int32 and object_idxs.dtype == torch.int64
    n, dev = object_idxs.numel(), locs.device
    b = torch.empty(n, dtype=torch.int32, device=dev)
    sem = torch.empty(n, dtype=torch.int32, device=dev)
    xyz = torch.empty((2, n, 3), dtype=torch.float32, device=dev)
    if batch_size is not None and n > 0:
        boff = torch.empty(int(batch_size) + 1, dtype=torch.int32, device=dev)
        with _on(dev):
            check(_lib.lib().d3_cluster_select2(_ptr(locs), _ptr(pt_offsets), _ptr(semantic_preds.contiguous()), _ptr(batch_idxs.contiguous()),
                                                _ptr(object_idxs.contiguous()), n, int(batch_size), _ptr(b), _ptr(xyz[0]), _ptr(xyz[1]), _ptr(sem),
                                                _ptr(boff), _stream()), "cluster_select2")
        return b, xyz[0], xyz[1], sem, boff
    with _on(dev):
        check(_lib.lib().d3_cluster_select(_ptr(locs), _ptr(pt_offsets), _ptr(semantic_preds.contiguous()), _ptr(batch_idxs.contiguous()),
                                           _ptr(object_idxs.contiguous()), n, _ptr(b), _ptr(xyz[0]), _ptr(xyz[1]), _ptr(sem), _stream()),
              "cluster_select")
    if batch_size is not None:
        return b, xyz[0], xyz[1], sem, torch.zeros(int(batch_size) + 1, dtype=torch.int32, device=dev)
    return b, xyz[0], xyz[1], sem


def cluster_merge(idx1, off1, idx2, off2, object_idxs, batch_idxs):
    """the two clusterings' (cluster, compact point) pairs -> proposals_idx (S1+S2,2), proposals_offset (P1+P2+1),
    proposals_batchId_all (S1+S2-1) as model/pointgroup.py:299-316 builds them (d3_cluster_merge)"""
    S1, S2, P1, P2 = idx1.shape[0], idx2.shape[0], off1.numel() - 1, off2.numel() - 1
    dev = idx1.device
    out_idx = torch.empty((S1 + S2, 2), dtype=torch.int32, device=dev)
    out_off = torch.empty(P1 + P2 + 1, dtype=torch.int32, device=dev)
    out_bid = torch.empty(max(S1 + S2 - 1, 0) if S2 > 0 else S1, dtype=torch.int32, device=dev)
    with _on(dev):
        check(_lib.lib().d3_cluster_merge(_ptr(idx1.contiguous()), S1, _ptr(off1.contiguous()), P1, _ptr(idx2.contiguous()), S2,
                                          _ptr(off2.contiguous()), P2, _ptr(object_idxs), _ptr(batch_idxs), _ptr(out_idx), _ptr(out_off),
                                          _ptr(out_bid), _stream()), "cluster_merge")
    return out_idx, out_off, out_bid


def cluster_coords_stats(coords, clusters_idx, clusters_offset):
    """mean / min / max (P,3) of the clusters' member coordinates from the (S,2) [cluster, point] pairs: what
    `sec_mean(coords[idx])`, `sec_min(coords[idx])`, `sec_max(coords[idx])` return, without the gathered copy (d3_cluster_coords_stats)"""
    assert coords.is_cuda and coords.dtype == torch.float32 and coords.is_contiguous() and coords.shape[1] == 3
    assert clusters_idx.dtype == torch.int32 and clusters_idx.is_contiguous() and clusters_offset.dtype == torch.int32
    P = clusters_offset.numel() - 1
    out = torch.empty((3, P, 3), dtype=torch.float32, device=coords.device)
    L = _lib.lib()
    S = int(clusters_idx.shape[0])
    with _on(coords.device):
        ws = _workspace(L.d3_cluster_coords_stats_ws_bytes(S), coords.device, "ccs")     # the mean chains' staged addends
        check(L.d3_cluster_coords_stats2(_ptr(coords), _ptr(clusters_idx), _ptr(clusters_offset.contiguous()), S, _ptr(out[0]), _ptr(out[1]),
                                         _ptr(out[2]), P, _ptr(ws), ws.numel(), _stream()), "cluster_coords_stats2")
    return out[0], out[1], out[2]


def proposal_prepare(sig, proposals_offset, batch_id_all, proposals_idx, semantic_preds, center, size, score_thr, npoint_thr):
    """(npoint (P) float, mask (P) bool, batch id at the cluster start (P) int32, crop box (P,9)) of model/pointgroup.py:338-372 in
    one launch (d3_proposal_prepare)"""
    P = proposals_offset.numel() - 1
    dev = sig.device
    npoint = torch.empty(P, dtype=torch.float32, device=dev)
    mask = torch.empty(P, dtype=torch.bool, device=dev)
    bid = torch.empty(P, dtype=torch.int32, device=dev)
    crop = torch.empty((P, 9), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.lib().d3_proposal_prepare(_ptr(sig.contiguous()), _ptr(proposals_offset), _ptr(batch_id_all), batch_id_all.numel(),
                                             _ptr(proposals_idx), _ptr(semantic_preds), _ptr(center.contiguous()), _ptr(size.contiguous()),
                                             float(score_thr), float(npoint_thr), P, _ptr(npoint), _ptr(mask), _ptr(bid), _ptr(crop),
                                             _stream()), "proposal_prepare")
    return npoint, mask, bid, crop


def cluster_norm_params(mean, raw_min, raw_max, fullscale, scale_cap, r0, r1):
    """per-cluster size (P,3), centre (P,3), grid scale (P,) and placement offset (P,3) of `clusters_voxelization`
    (model/pointgroup.py:146-165) in one launch (d3_cluster_norm_params); r0, r1: the two host-side `torch.rand(3)` draws"""
    P = mean.shape[0]
    dev = mean.device
    mean, raw_min, raw_max = mean.contiguous(), raw_min.contiguous(), raw_max.contiguous()
    size = torch.empty((P, 3), dtype=torch.float32, device=dev)
    center, offset = torch.empty_like(size), torch.empty_like(size)
    cscale = torch.empty((P,), dtype=torch.float32, device=dev)
    rand6 = (C.c_float * 6)(*[float(v) for v in r0.tolist()], *[float(v) for v in r1.tolist()])
    with _on(dev):
        check(_lib.lib().d3_cluster_norm_params(_ptr(mean), _ptr(raw_min), _ptr(raw_max), P, float(fullscale), float(scale_cap),
                                                C.cast(rand6, C.c_void_p), _ptr(size), _ptr(center), _ptr(cscale),
                                                _ptr(offset), _stream()), "cluster_norm_params")
    return size, center, cscale, offset


def cluster_transform(coords, clusters_idx, mean, scale, offset):
    """(S,4) int64 [cluster, trunc((coords[point] - mean[cluster]) * scale[cluster] + offset[cluster])] (d3_cluster_transform)"""
    S = clusters_idx.shape[0]
    mean, scale, offset = mean.contiguous(), scale.contiguous(), offset.contiguous()
    out = torch.empty((S, 4), dtype=torch.int64, device=coords.device)
    with _on(coords.device):
        check(_lib.lib().d3_cluster_transform(_ptr(coords), _ptr(clusters_idx), _ptr(mean), _ptr(scale), _ptr(offset), _ptr(out), S,
                                              _stream()), "cluster_transform")
    return out
