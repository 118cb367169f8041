"""ctypes/numpy front-end of oracle/pg_ops_oracle.c (TEST INFRASTRUCTURE ONLY).

Function names and argument meaning mirror the reference's python operator layer
(reference: lib/pointgroup_ops/functions/pointgroup_ops.py:39,75,150,182,221,253,281,309,337)
but take and return numpy arrays on the host.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpgoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pg_ops_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_voxelize_idx.restype = C.c_int
        _lib.orc_bfs_cluster.restype = C.c_int
        _lib.orc_ballquery_batch_p.restype = C.c_int
        _lib.orc_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def voxelization_idx(coords, batchsize, mode=4):
    """reference: functions/pointgroup_ops.py:13-32 -> (output_coords i64 (M,ncols), input_map i32 (N), output_map i32 (M,maxActive+1))"""
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    n, ncols = coords.shape
    input_map = np.zeros(n, np.int32)
    oc, om, ma = C.c_void_p(), C.c_void_p(), C.c_int()
    M = lib().orc_voxelize_idx(_p(coords), n, ncols, int(mode), _p(input_map), C.byref(oc), C.byref(om), C.byref(ma))
    ma = ma.value
    out_coords = np.ctypeslib.as_array(C.cast(oc, C.POINTER(C.c_int64)), shape=(max(M, 1), ncols))[:M].copy()
    out_map = np.ctypeslib.as_array(C.cast(om, C.POINTER(C.c_int32)), shape=(max(M, 1), ma + 1))[:M].copy()
    lib().orc_free(oc); lib().orc_free(om)
    return out_coords, input_map, out_map


def voxelization(feats, map_rule, mode=4):
    """reference: functions/pointgroup_ops.py:44-60"""
    feats = np.ascontiguousarray(feats, np.float32); map_rule = np.ascontiguousarray(map_rule, np.int32)
    M, w = map_rule.shape
    out = np.zeros((M, feats.shape[1]), np.float32)
    lib().orc_voxelize_fp(_p(feats), _p(out), _p(map_rule), M, w - 1, feats.shape[1], int(mode == 4))
    return out


def voxelization_bp(d_out, map_rule, N, mode=4):
    """reference: functions/pointgroup_ops.py:63-71"""
    d_out = np.ascontiguousarray(d_out, np.float32); map_rule = np.ascontiguousarray(map_rule, np.int32)
    M, w = map_rule.shape
    d_feats = np.zeros((N, d_out.shape[1]), np.float32)
    lib().orc_voxelize_bp(_p(d_out), _p(d_feats), _p(map_rule), M, w - 1, d_out.shape[1], int(mode == 4))
    return d_feats


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """reference: functions/pointgroup_ops.py:117-146 (including the retry loop)"""
    coords = np.ascontiguousarray(coords, np.float32)
    batch_idxs = np.ascontiguousarray(batch_idxs, np.int32); batch_offsets = np.ascontiguousarray(batch_offsets, np.int32)
    n = coords.shape[0]
    while True:
        idx = np.zeros(max(n * meanActive, 1), np.int32)
        start_len = np.zeros((n, 2), np.int32)
        nActive = lib().orc_ballquery_batch_p(n, int(meanActive), C.c_float(radius), _p(coords), _p(batch_idxs),
                                              _p(batch_offsets), _p(idx), _p(start_len))
        if nActive <= n * meanActive:
            break
        meanActive = int(nActive // n + 1)
    return idx[:nActive], start_len


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """reference: functions/pointgroup_ops.py:155-178"""
    semantic_label = np.ascontiguousarray(semantic_label, np.int32)
    ball_query_idxs = np.ascontiguousarray(ball_query_idxs, np.int32)
    start_len = np.ascontiguousarray(start_len, np.int32)
    N = start_len.shape[0]
    ci, co, nc = C.c_void_p(), C.c_void_p(), C.c_int()
    S = lib().orc_bfs_cluster(_p(semantic_label), _p(ball_query_idxs), _p(start_len), N, int(threshold),
                              C.byref(ci), C.byref(co), C.byref(nc))
    P = nc.value
    cluster_idxs = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), shape=(max(S, 1), 2))[:S].copy()
    cluster_offsets = np.ctypeslib.as_array(C.cast(co, C.POINTER(C.c_int32)), shape=(P + 1,)).copy()
    lib().orc_free(ci); lib().orc_free(co)
    return cluster_idxs, cluster_offsets


def roipool(feats, proposals_offset):
    """reference: functions/pointgroup_ops.py:187-206 -> (output_feats, output_maxidx)"""
    feats = np.ascontiguousarray(feats, np.float32); proposals_offset = np.ascontiguousarray(proposals_offset, np.int32)
    P = proposals_offset.shape[0] - 1; Cc = feats.shape[1]
    out = np.zeros((P, Cc), np.float32); mx = np.zeros((P, Cc), np.int32)
    lib().orc_roipool_fp(P, Cc, _p(feats), _p(proposals_offset), _p(out), _p(mx))
    return out, mx


def roipool_bp(d_out, proposals_offset, maxidx, sumNPoint):
    """reference: functions/pointgroup_ops.py:209-219"""
    d_out = np.ascontiguousarray(d_out, np.float32); maxidx = np.ascontiguousarray(maxidx, np.int32)
    proposals_offset = np.ascontiguousarray(proposals_offset, np.int32)
    P, Cc = d_out.shape
    d_feats = np.zeros((sumNPoint, Cc), np.float32)
    lib().orc_roipool_bp(P, Cc, _p(d_feats), _p(proposals_offset), _p(maxidx), _p(d_out))
    return d_feats


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    """reference: functions/pointgroup_ops.py:226-248"""
    proposals_idx = np.ascontiguousarray(proposals_idx, np.int32); proposals_offset = np.ascontiguousarray(proposals_offset, np.int32)
    instance_labels = np.ascontiguousarray(instance_labels, np.int64); instance_pointnum = np.ascontiguousarray(instance_pointnum, np.int32)
    nInst = instance_pointnum.shape[0]; P = proposals_offset.shape[0] - 1
    out = np.zeros((P, nInst), np.float32)
    lib().orc_get_iou(nInst, P, _p(proposals_idx), _p(proposals_offset), _p(instance_labels), _p(instance_pointnum), _p(out))
    return out


def _sec(fn, inp, offsets):
    inp = np.ascontiguousarray(inp, np.float32); offsets = np.ascontiguousarray(offsets, np.int32)
    P = offsets.shape[0] - 1; Cc = inp.shape[1]
    out = np.zeros((P, Cc), np.float32)
    getattr(lib(), fn)(P, Cc, _p(inp), _p(offsets), _p(out))
    return out


def sec_mean(inp, offsets):
    """reference: functions/pointgroup_ops.py:258-276"""
    return _sec("orc_sec_mean", inp, offsets)


def sec_min(inp, offsets):
    """reference: functions/pointgroup_ops.py:286-304"""
    return _sec("orc_sec_min", inp, offsets)


def sec_max(inp, offsets):
    """reference: functions/pointgroup_ops.py:314-332"""
    return _sec("orc_sec_max", inp, offsets)
