"""`PG_OP` -- the reference's native module name (reference: lib/pointgroup_ops/src/pointgroup_ops_api.cpp:6-24,
imported by lib/pointgroup_ops/functions/pointgroup_ops.py:9), bound to libd3hip.so.

Same 13 function names and argument lists (torch tensors, outputs passed in by the caller).  Put this module on
the path as `PG_OP` (e.g. `sys.modules["PG_OP"] = d3net_amd.compat.PG_OP`) and the reference's python operator layer
runs unchanged on MI355X.  `voxelize_idx` and `bfs_cluster` receive CPU tensors from the reference
(model/pointgroup.py:166-169,297): they are processed on the device and copied back into the caller's tensors,
which are resized exactly as the reference natives do (src/voxelize/voxelize.cpp:22-26,
src/bfs_cluster/bfs_cluster.cpp:103-106)."""
import ctypes as C

import torch

from .. import _lib
from .. import pointgroup_ops as _ops
from .._lib import check

_P = lambda t: C.c_void_p(t.data_ptr())
_S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def voxelize_idx(coords, output_coords, input_map, output_map, batchSize, mode):
    oc, im, om = _ops.voxelization_idx(coords, batchSize, mode)
    output_coords.resize_(oc.shape).copy_(oc)
    input_map.copy_(im)
    output_map.resize_(om.shape).copy_(om)


def voxelize_fp(feats, output_feats, output_map, mode, nActive, maxActive, nPlane):
    check(_lib.lib().d3_voxelize_fp(_P(feats), _P(output_feats), _P(output_map), mode, nActive, maxActive, nPlane, _S()),
          "voxelize_fp")


def voxelize_bp(d_output_feats, d_feats, output_map, mode, nActive, maxActive, nPlane):
    check(_lib.lib().d3_voxelize_bp(_P(d_output_feats), _P(d_feats), _P(output_map), mode, nActive, maxActive, nPlane,
                                    _S()), "voxelize_bp")


def point_recover_fp(feats, output_feats, idx_map, nActive, maxActive, nPlane):
    check(_lib.lib().d3_point_recover_fp(_P(feats), _P(output_feats), _P(idx_map), nActive, maxActive, nPlane, _S()),
          "point_recover_fp")


def point_recover_bp(d_output_feats, d_feats, idx_map, nActive, maxActive, nPlane):
    check(_lib.lib().d3_point_recover_bp(_P(d_output_feats), _P(d_feats), _P(idx_map), nActive, maxActive, nPlane,
                                         _S()), "point_recover_bp")


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, meanActive, radius):
    """returns the total hit count; > n*meanActive makes the reference wrapper retry with a bigger `idx`
    (functions/pointgroup_ops.py:135-142); hits beyond idx's capacity are dropped as in bfs_cluster.cu:51-59."""
    L = _lib.lib()
    ws = _ops._workspace(L.d3_ballquery_ws_bytes(n), xyz.device, "bq")
    tot = C.c_int(0)
    check(L.d3_ballquery_count(_P(xyz), _P(batch_idxs), _P(batch_offsets), n, float(radius), _P(start_len), _P(ws),
                               ws.numel(), C.byref(tot), _S()), "ballquery_count")
    check(L.d3_ballquery_fill(_P(xyz), _P(batch_idxs), _P(batch_offsets), n, float(radius), _P(start_len), _P(ws),
                              ws.numel(), _P(idx), min(idx.numel(), n * meanActive), _S()), "ballquery_fill")
    return tot.value


def bfs_cluster(semantic_label, ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N, threshold):
    ci, co = _ops.bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold)
    cluster_idxs.resize_(ci.shape).copy_(ci)
    cluster_offsets.resize_(co.shape).copy_(co)


def roipool_fp(feats, proposals_offset, output_feats, output_maxidx, nProposal, Cc):
    check(_lib.lib().d3_roipool_fp(_P(feats), _P(proposals_offset), _P(output_feats), _P(output_maxidx), nProposal, Cc,
                                   _S()), "roipool_fp")


def roipool_bp(d_feats, proposals_offset, output_maxidx, d_output_feats, nProposal, Cc):
    check(_lib.lib().d3_roipool_bp(_P(d_feats), _P(proposals_offset), _P(output_maxidx), _P(d_output_feats), nProposal,
                                   Cc, _S()), "roipool_bp")


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance, nProposal):
    check(_lib.lib().d3_get_iou(_P(proposals_idx), _P(proposals_offset), _P(instance_labels), _P(instance_pointnum),
                                _P(proposals_iou), nInstance, nProposal, _S()), "get_iou")


def sec_mean(inp, offsets, out, nProposal, Cc):
    check(_lib.lib().d3_sec_mean(_P(inp), _P(offsets), _P(out), nProposal, Cc, _S()), "sec_mean")


def sec_min(inp, offsets, out, nProposal, Cc):
    check(_lib.lib().d3_sec_min(_P(inp), _P(offsets), _P(out), nProposal, Cc, _S()), "sec_min")


def sec_max(inp, offsets, out, nProposal, Cc):
    check(_lib.lib().d3_sec_max(_P(inp), _P(offsets), _P(out), nProposal, Cc, _S()), "sec_max")
