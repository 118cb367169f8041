"""ctypes binding of libd3hip.so (declared in include/d3hip.h).  Fails loudly when missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libd3hip.so")

ERRORS = {-1: "D3_ERR_WORKSPACE", -2: "D3_ERR_RANGE (coordinate/batch outside key range)",
          -3: "D3_ERR_ARG", -4: "D3_ERR_OVERFLOW"}


class D3Error(RuntimeError):
    pass


_lib = None
vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
f64 = C.c_double
pi = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/d3hip.h declares
SIGNATURES = {
    "d3_version": (i32, []),
    "d3_arch": (C.c_char_p, []),
    "d3_sec_mean": (i32, [vp, vp, vp, i32, i32, vp]),
    "d3_sec_min": (i32, [vp, vp, vp, i32, i32, vp]),
    "d3_sec_max": (i32, [vp, vp, vp, i32, i32, vp]),
    "d3_cluster_select": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "d3_cluster_select2": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "d3_cluster_merge": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]),
    "d3_proposal_prepare": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]),
    "d3_cluster_coords_stats": (i32, [vp, vp, vp, vp, vp, vp, i32, vp]),
    "d3_cluster_coords_stats_ws_bytes": (sz, [i64]),
    "d3_cluster_coords_stats2": (i32, [vp, vp, vp, i64, vp, vp, vp, i32, vp, sz, vp]),
    "d3_cluster_transform": (i32, [vp, vp, vp, vp, vp, vp, i64, vp]),
    "d3_cluster_norm_params": (i32, [vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp, vp]),
    "d3_roipool_fp": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "d3_roipool_bp": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "d3_get_iou": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    "d3_voxelize_fp": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "d3_voxelize_fp2": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp]),
    "d3_voxelize_bp": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "d3_point_recover_fp": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "d3_point_recover_bp": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "d3_voxelize_idx_ws_bytes": (sz, [i32]),
    "d3_voxelize_idx_count": (i32, [vp, i32, i32, i32, vp, vp, sz, pi, pi, vp]),
    "d3_voxelize_idx_fill": (i32, [vp, i32, i32, i32, vp, vp, sz, vp, vp, i32, i32, vp]),
    "d3_ballquery_ws_bytes": (sz, [i32]),
    "d3_ballquery_ws_bytes_single_pass": (sz, [i32]),
    "d3_ballquery_count": (i32, [vp, vp, vp, i32, f32, vp, vp, sz, pi, vp]),
    "d3_ballquery_fill": (i32, [vp, vp, vp, i32, f32, vp, vp, sz, vp, i64, vp]),
    "d3_ballquery_cap": (i32, []),
    "d3_ballquery_padded": (i32, [vp, vp, vp, i32, f32, vp, vp, sz, vp, vp]),
    "d3_bfs_cluster_ws_bytes": (sz, [i32]),
    "d3_bfs_cluster_count": (i32, [vp, vp, vp, i32, i32, vp, sz, pi, pi, vp]),
    "d3_bfs_cluster_count_ex": (i32, [vp, vp, vp, i32, i32, vp, sz, pi, pi, i32, vp]),
    "d3_bfs_cluster_fill": (i32, [vp, vp, vp, i32, vp, sz, vp, vp, i32, i32, vp]),
    "d3_bfs_cluster_erec_bytes": (sz, [i64]),
    "d3_bfs_cluster_fill2": (i32, [vp, vp, vp, i32, vp, sz, vp, sz, i64, vp, vp, i32, i32, vp]),
    "d3_bfs_cluster_run": (i32, [vp, vp, vp, i32, i32, vp, sz, vp, sz, i64, i32, vp, i64, vp, i64, pi, pi, vp]),
    "d3_bfs_cluster_begin": (i32, [vp, vp, vp, i32, i32, vp, sz, vp, sz, i64, i32, vp, i64, vp, i64, C.POINTER(vp), vp]),
    "d3_bfs_cluster_end": (i32, [vp, pi, pi]),
    "d3_coordmap_ws_bytes": (sz, [i32]),
    "d3_kmap_k3": (i32, [vp, i32, i32, vp, sz, vp, vp]),
    "d3_kmap_k3_pack16": (i32, [vp, i32, vp, vp, vp]),
    "d3_kmap_k3_16": (i32, [vp, i32, i32, vp, sz, vp, vp, vp, vp]),
    "d3_net_set_k3_16": (i32, [vp, vp, vp]),
    "d3_net_padded_channels": (i32, [vp]),
    "d3_net_padcast": (i32, [vp, vp, vp, i64, vp]),
    "d3_net_set_padded_input": (i32, [vp, vp]),
    "d3_spconv_t16_launches": (C.c_longlong, []),
    "d3_kmap_down_count": (i32, [vp, i32, i32, vp, sz, vp, vp, pi, vp]),
    "d3_kmap_down_fill": (i32, [vp, i32, i32, vp, sz, vp, vp, vp, vp, vp, i32, vp]),
    "d3_kmap_pyramid": (i32, [vp, i32, i32, vp, sz, vp, vp, vp, vp, vp, pi, vp]),
    "d3_kmap_pyramid_begin": (i32, [vp, i32, i32, vp, sz, vp, vp, vp, vp, vp, C.POINTER(C.c_void_p), vp]),
    "d3_kmap_pyramid_end": (i32, [vp, pi, i32]),
    "d3_kmap_down_fill2": (i32, [i32, i32, vp, vp, vp, vp, vp]),
    "d3_spconv_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_pack_bytes": (sz, [i32, i32, i32]),
    "d3_spconv_pack_bytes_ex": (sz, [i32, i32, i32, i32]),
    "d3_spconv_pack": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "d3_spconv_fwd2_nparts": (i32, [i32, i32, i32, i32]),
    "d3_spconv_fwd2_nparts_ex": (i32, [i32, i32, i32, i32, i32]),
    "d3_spconv_fwd2_plan": (i32, [i32, i32, i32, i32, pi]),
    "d3_spconv_fwd2": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_fwd2_bnbwd": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_fwd2_fin": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_fwd2_bnbwd_fin": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, i32,
                                       i32, i32, i32, i32, i32, i32, vp]),
    "d3_spconv_wgrad2_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "d3_spconv_wgrad2_splits": (i32, [i32, i32, i32, i32, i32, i32]),
    "d3_spconv_wgrad2": (i32, [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "d3_net_create": (vp, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32]),
    "d3_net_destroy": (None, [vp]),
    "d3_net_plan": (i32, [vp, vp, C.POINTER(sz), C.POINTER(sz)]),
    "d3_net_tensor_offset": (i64, [vp, i32]),
    "d3_net_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "d3_net_backward": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "d3_tall_wgrad_ws_bytes": (sz, [i32, i32]),
    "d3_tall_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "d3_stack_to_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "d3_adamw_chunk": (i32, []),
    "d3_adamw": (i32, [vp, vp, vp, i32, f64, f64, f64, f64, f64, f64, f64, vp]),
    "d3_gather_rows_pad": (i32, [vp, i64, vp, vp, i64, i32, i32, vp]),
    "d3_point_heads_dy": (i32, [vp, vp, vp, i64, vp, vp]),
    "d3_point_heads_dx": (i32, [vp, vp, vp, vp, i64, i32, vp, vp]),
    "d3_point_heads_ws_bytes": (sz, []),
    "d3_point_heads_fwd": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                 vp, sz, vp]),
    "d3_score_loss": (i32, [vp, vp, i32, i32, f32, f32, vp, vp, vp, vp]),
    "d3_query_locals_mask": (i32, [vp, vp, i32, i32, i32, vp]),
    "d3_caption_select_target": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "d3_caption_inputs_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "d3_caption_inputs_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "d3_masked_xe_ws_bytes": (sz, [i32, i32]),
    "d3_masked_xe": (i32, [vp, vp, i64, vp, i32, i32, i32, vp, vp, vp, sz, vp]),
    "d3_orientation_loss": (i32, [vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp]),
    "d3_offset_loss_ws_bytes": (sz, []),
    "d3_offset_loss": (i32, [vp, vp, vp, i32, vp, i64, vp, vp, vp, i32, vp, sz, vp]),
    "d3_scatter_add_rows": (i32, [vp, vp, vp, i64, i32, vp]),
    "d3_gather_rows": (i32, [vp, vp, vp, i64, i32, vp]),
    "d3_cross_entropy_ws_bytes": (sz, []),
    "d3_cross_entropy": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "d3_attn_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "d3_attn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "d3_hgemm": (i32, [vp, i32, vp]),
    "d3_colsum_ws_bytes": (sz, [i32]),
    "d3_colsum": (i32, [vp, i64, i32, i32, vp, i32, vp, sz, vp]),
    "d3_topdown_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "d3_topdown_bwd_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "d3_topdown_xe_forward": (i32, [vp, vp]),
    "d3_topdown_xe_backward": (i32, [vp, vp, vp]),
    "d3_topdown_xe_backward_ex": (i32, [vp, vp, vp, vp]),
    "d3_topdown_step_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "d3_topdown_feat_proj": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "d3_topdown_step": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "d3_beam_select": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "d3_greedy_select": (i32, [vp, i32, i32, vp, vp, vp]),
    "d3_topdown_greedy": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp, i32, vp, vp, vp]),
    "d3_topdown_beam": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, sz, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "d3_topdown_beam_greedy": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, sz, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "d3_gru_seq_ws_bytes": (sz, [i32, i32, i32, i32]),
    "d3_gru_seq_bwd_ws_bytes": (sz, [i32, i32, i32, i32]),
    "d3_gru_seq_forward": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "d3_gru_seq_backward": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "d3_nms3d_samecls": (i32, [vp, vp, vp, i32, i32, f64, i32, vp, vp]),
    "d3_instance_cross_iou": (i32, [vp, vp, i64, i32, i32, vp, vp, vp, vp]),
    "d3_nms_matrix": (i32, [vp, vp, vp, i32, f32, vp, vp, vp, vp]),
    "d3_cider_ws_bytes": (sz, [i32, i32, i32]),
    "d3_cider_scores": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, f64, i32, vp, vp, vp, sz, vp]),
    "d3_graph_edges": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "d3_edgeconv_ws_bytes": (sz, [i32, i32, i32]),
    "d3_edgeconv_bwd_ws_bytes": (sz, [i32, i32, i32]),
    "d3_edgeconv_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "d3_edgeconv_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "d3_query_locals_dist": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, vp]),
    "d3_prof_enable": (i32, [i32]),
    "d3_net_set_chunks": (i32, [vp, vp, i32]),
    "d3_net_chunk_wait": (i32, [vp, i32, vp]),
    "d3_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "d3_layernorm_ws_bytes": (sz, [i32, i32]),
    "d3_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
    "d3_tuning_set": (i32, [C.c_char_p, i32]),
    "d3_tuning_get": (i32, [C.c_char_p, vp]),
    "d3_tuning_count": (i32, []),
    "d3_tuning_name": (C.c_char_p, [i32]),
    "d3_prof_collect": (i32, [i32, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "d3_prof_dump": (i32, [i32, C.POINTER(C.c_double), i32, C.POINTER(i32)]),
    "d3_bn_ws_bytes": (sz, [i32]),
    "d3_bn_stats": (i32, [vp, i32, i32, vp, vp, vp, vp, f32, vp, sz, vp]),
    "d3_bn_relu_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]),
    "d3_bn_relu_fwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]),
    "d3_bn_relu_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp, sz, vp]),
}


class GemmSeg(C.Structure):
    """d3_gemm_seg (include/d3hip.h)"""
    _fields_ = [("A", vp), ("ia", vp), ("lda", i64), ("a_kmajor", i32), ("B", vp), ("ldb", i64), ("b_kmajor", i32), ("K", i32)]


class GemmProb(C.Structure):
    """d3_gemm_prob (include/d3hip.h)"""
    _fields_ = [("seg", GemmSeg * 3), ("nseg", i32), ("M", i32), ("N", i32), ("C", vp), ("ldc", i64), ("bias", vp), ("add", vp),
                ("ldadd", i64), ("relu", i32), ("accum", i32), ("perm_nb", i32), ("perm_s", i32),
                ("gru", i32), ("gru_H", i32), ("g_d0", vp), ("g_ld0", i64), ("g_d1", vp), ("g_ld1", i64),
                ("g_r", vp), ("g_z", vp), ("g_n", vp), ("g_ghn", vp), ("g_hp", vp), ("g_ldh", i64),
                ("g_dgi", vp), ("g_lddgi", i64), ("g_dgh", vp), ("g_dhp", vp)]


TOPDOWN_PARAMS = ("W_td", "b_td", "Wih1", "Whh1", "bih1", "bhh1", "W_feat", "W_hidd", "w_att", "W_lang", "b_lang",
                  "Wih2", "Whh2", "bih2", "bhh2", "Wc0", "bc0", "Wc2", "bc2")


class TopdownArgs(C.Structure):
    """d3_topdown_args (include/d3hip.h)"""
    _fields_ = ([(k, i32) for k in ("N", "K", "S", "V", "H", "E", "F", "Tw")] +
                [(k, vp) for k in ("word_ids", "emb", "target", "obj", "mask")] + [(k, vp) for k in TOPDOWN_PARAMS] +
                [("logits", vp), ("attn", vp), ("ws", vp), ("ws_bytes", sz)])


class TopdownGrads(C.Structure):
    """d3_topdown_grads (include/d3hip.h)"""
    _fields_ = ([("dlogits", vp)] + [("d" + k, vp) for k in TOPDOWN_PARAMS] +
                [("dobj", vp), ("dtarget", vp), ("ws", vp), ("ws_bytes", sz)])


class tuning:
    """`with _lib.tuning(D3_WG3=0): ...` -- flip measurement / test switches of the library for a block (csrc/tuning.hip: the
    library parses its environment once; this is how tests and the A/B tools change a switch afterwards)."""

    def __init__(self, **switches):
        self.switches, self.saved = switches, {}

    def __enter__(self):
        l = lib()
        for k, v in self.switches.items():
            old = C.c_int(0)
            check(l.d3_tuning_get(k.encode(), C.byref(old)), "tuning_get(%s)" % k)
            self.saved[k] = old.value
            check(l.d3_tuning_set(k.encode(), int(v)), "tuning_set(%s)" % k)
        return self

    def __exit__(self, *exc):
        l = lib()
        for k, v in self.saved.items():
            l.d3_tuning_set(k.encode(), v)
        return False


def lib():
    """Load libd3hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise D3Error("libd3hip.so not built (%s); run `python -m d3net_amd.build` -- "
                          "d3net_amd has no CPU/eager fallback" % SO_PATH)
        # torch first: libd3hip.so and torch must share ONE HIP runtime (torch bundles libamdhip64.so.7; loading
        # /opt/rocm's copy first gives the process two runtimes and HIP reports "no device" to one of them)
        import torch  # noqa: F401
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise D3Error("%s failed: %s" % (what, ERRORS.get(rc, "hipError_t %d" % rc)))
