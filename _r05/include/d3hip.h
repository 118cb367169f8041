/*
 * d3hip.h -- C ABI of libd3hip.so, the MI355X (gfx950) implementation of D3Net's
 * PointGroup hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     stream-ordered on it;
 *   - return value: 0 = success, >0 = hipError_t, <0 = D3_ERR_* below;
 *   - no hidden allocation: ops that need scratch take `ws`/`ws_bytes` and have a
 *     `*_ws_bytes()` query (the one exception: the network object of d3_net_create owns two
 *     small grow-only job tables, allocated lazily inside d3_net_forward / d3_net_backward and
 *     freed by d3_net_destroy -- see there); data-dependent output sizes use a two-phase
 *     `*_count` (writes sizes to `*_host`, synchronises the stream) / `*_fill` pair so
 *     the caller allocates, exactly as the reference's python layer does
 *     (reference: lib/pointgroup_ops/functions/pointgroup_ops.py).
 *
 * Each entry point cites the reference interface it replaces.  Paths are relative to
 * the reference root; `PG_OP.x` is the pybind symbol of
 * lib/pointgroup_ops/src/pointgroup_ops_api.cpp:6-24.
 */
#ifndef D3HIP_H
#define D3HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3_ERR_WORKSPACE (-1) /* ws_bytes too small                          */
#define D3_ERR_RANGE     (-2) /* coordinate / batch index outside key range  */
#define D3_ERR_ARG       (-3) /* unsupported argument (mode, channel count)  */
#define D3_ERR_OVERFLOW  (-4) /* hash table / queue overflow                 */

int d3_version(void);
const char *d3_arch(void); /* "gfx950" */

/* ---- segment ops ------------------------------------------------------------------ */
/* PG_OP.sec_mean / sec_min / sec_max  (lib/pointgroup_ops/src/sec_mean/sec_mean.cu:12-86) */
int d3_sec_mean(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream);
int d3_sec_min(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream);
int d3_sec_max(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream);
/* Index plumbing around the two clusterings of PointGroup.forward (model/pointgroup.py:288-316; csrc/clusterprep.hip):
 *   select: the object points' batch ids, coordinates, shifted coordinates (coords + offsets) and semantic ids, compacted by
 *           object_idxs (n int64 scene point ids) -- four gathers, a cast and an add of the reference in one pass;
 *   merge : the (cluster, compact point) pairs of both clusterings mapped back to scene point ids, their batch ids, the second
 *           set's cluster ids / offsets shifted behind the first's, concatenated as the reference does (out_bid has S1+S2-1
 *           entries: the reference drops the first pair of the second set, :316). */
int d3_cluster_select(const float *locs, const float *pt_offsets, const int64_t *semantic_preds, const int *batch_idxs,
                      const int64_t *object_idxs, int n, int *batch_out, float *coords_out, float *shifted_out, int *semantic_out,
                      void *stream);
/* select + the batch offsets of the object points (model/pointgroup.py:110-122 get_batch_offsets, :296): batch_offsets_out[b] = object
 * points with a batch id below b, b = 0..batch_size, read off the boundaries of the (sorted) id column in the same pass; n >= 1 */
int d3_cluster_select2(const float *locs, const float *pt_offsets, const int64_t *semantic_preds, const int *batch_idxs,
                       const int64_t *object_idxs, int n, int batch_size, int *batch_out, float *coords_out, float *shifted_out,
                       int *semantic_out, int *batch_offsets_out, void *stream);
int d3_cluster_merge(const int *idx1, int S1, const int *off1, int P1, const int *idx2, int S2, const int *off2, int P2,
                     const int64_t *object_idxs, const int *batch_idxs, int *out_idx, int *out_off, int *out_bid, void *stream);
/* Per-proposal bookkeeping between the score head and the proposal selection (model/pointgroup.py:338-372): npoint (P) = points
 * per proposal, mask (P) bytes = sig > score_thr && npoint > npoint_thr, batch_id (P) = batch_id_all[min(offsets[p],
 * n_batch_id - 1)] (the reference's one-element-short batch-id vector), crop (P,9) = [center | size | 0 | semantic_preds of the
 * proposal's first point | sig].  sig = sigmoid of the proposal scores. */
int d3_proposal_prepare(const float *sig, const int *offsets, const int *batch_id_all, int n_batch_id, const int *proposals_idx,
                        const int64_t *semantic_preds, const float *center, const float *size, float score_thr, float npoint_thr,
                        int P, float *npoint, unsigned char *mask, int *batch_id, float *crop, void *stream);
/* The per-point passes of PointGroup.clusters_voxelization (model/pointgroup.py:125-178) over the S (cluster, point) pairs of
 * clusters_idx (S,2) without the gathered / shifted / scaled (S,3) temporaries of the library-op form:
 *   coords_stats: mean (P,3) = sec_mean of the clusters' point coordinates (same serial x/count chain, bit-exact), cmin / cmax
 *                 (P,3) = their per-cluster extrema (raw coordinates: min(x - m) == min(x) - m under monotone rounding);
 *   transform   : out (S,4) int64 = [cluster, trunc((coords[point] - mean[cluster]) * scale[cluster] + offset[cluster])],
 *                 each fp32 operation rounded separately (:141-166). */
int d3_cluster_coords_stats(const float *coords, const int *clusters_idx, const int *offsets, float *mean, float *cmin, float *cmax,
                            int nProposal, void *stream);
/* the same with S = number of (cluster, point) pairs and d3_cluster_coords_stats_ws_bytes(S) bytes of scratch: the addends of the
 * clusters' mean chains (coords / count, IEEE) are gathered by a chip-wide pass first and the serial chains stream them
 * (bit-identical results; 258 -> ~130 us for the 4-scene batch, whose 33 k-point floors set the launch time). */
size_t d3_cluster_coords_stats_ws_bytes(long long S);
int d3_cluster_coords_stats2(const float *coords, const int *clusters_idx, const int *offsets, long long S, float *mean, float *cmin,
                             float *cmax, int nProposal, void *ws, size_t ws_bytes, void *stream);
int d3_cluster_transform(const float *coords, const int *clusters_idx, const float *mean, const float *scale, const float *offset,
                         long long *out, long long S, void *stream);
/* The per-cluster arithmetic between the two (model/pointgroup.py:146-165): size = cmax - cmin, center = (cmax + cmin) / 2 + mean
 * (cmin / cmax relative to the mean), cscale = min(1 / max_k((cmax - cmin) / fullscale) - 0.01, scale_cap), and the placement
 * offset = -cmin * cscale + clamp(fullscale - range - 0.001, min 0) * r0 + clamp(fullscale - range + 0.001, max 0) * r1 with
 * range = (cmax - cmin) * cscale.  rand6 = HOST pointer to the six floats [r0 | r1] (the reference's two `torch.rand(3)` draws,
 * :161), passed as kernel arguments.  Bit-equal to the ~30 elementwise library launches it replaces. */
int d3_cluster_norm_params(const float *mean, const float *raw_min, const float *raw_max, int P, float fullscale, float scale_cap,
                           const float *rand6, float *size, float *center, float *cscale, float *offset, void *stream);
/* PG_OP.roipool_fp / roipool_bp  (src/roipool/roipool.cu:12-57) */
int d3_roipool_fp(const float *feats, const int *proposals_offset, float *output_feats, int *output_maxidx,
                  int nProposal, int C, void *stream);
int d3_roipool_bp(float *d_feats, const int *proposals_offset, const int *output_maxidx,
                  const float *d_output_feats, int nProposal, int C, void *stream);
/* PG_OP.get_iou  (src/get_iou/get_iou.cu:12-38) */
int d3_get_iou(const int *proposals_idx, const int *proposals_offset, const int64_t *instance_labels,
               const int *instance_pointnum, float *proposals_iou, int nInstance, int nProposal, void *stream);

/* ---- voxelize --------------------------------------------------------------------- */
/* PG_OP.voxelize_fp / voxelize_bp / point_recover_fp / point_recover_bp
 * (src/voxelize/voxelize.cu:10-53, src/voxelize/voxelize.cpp:155-202).  Outputs are
 * accumulated into (the caller zero-fills them, as functions/pointgroup_ops.py:57,70 do). */
int d3_voxelize_fp(const float *feats, float *output_feats, const int *output_map, int mode, int nActive,
                   int maxActive, int nPlane, void *stream);
/* voxelize_fp of the column concatenation [feats_a | feats_b] without materialising it; writes (does not accumulate into)
 * output_feats (M, Ca+Cb): PointGroup.feed's `voxelization(cat(feats, locs), v2p_map)` (model/pointgroup.py:468-471) */
int d3_voxelize_fp2(const float *feats_a, int Ca, const float *feats_b, int Cb, float *output_feats, const int *output_map, int mode,
                    int nActive, int maxActive, void *stream);
int d3_voxelize_bp(const float *d_output_feats, float *d_feats, const int *output_map, int mode, int nActive,
                   int maxActive, int nPlane, void *stream);
int d3_point_recover_fp(const float *feats, float *output_feats, const int *idx_map, int nActive,
                        int maxActive, int nPlane, void *stream);
int d3_point_recover_bp(const float *d_output_feats, float *d_feats, const int *idx_map, int nActive,
                        int maxActive, int nPlane, void *stream);

/* PG_OP.voxelize_idx  (src/voxelize/voxelize.cpp:10-152), on the device, two-phase.
 * coords (n, ncols) int64, ncols in {3,4} (column 0 = batch index when 4).
 * count: fills input_map (n) and writes M and maxActive to the host.
 * fill : writes output_coords (M, ncols) int64 and output_map (M, maxActive+1) int32
 *        (voxels in first-occurrence order, point ids ascending, zero padded).
 * Key range: batch in [0, 2^19), x/y/z in [-2^14, 2^14) after the reference's
 * int64->int32 truncation, else D3_ERR_RANGE. */
size_t d3_voxelize_idx_ws_bytes(int n);
int d3_voxelize_idx_count(const int64_t *coords, int n, int ncols, int mode, int *input_map, void *ws,
                          size_t ws_bytes, int *M_host, int *maxActive_host, void *stream);
int d3_voxelize_idx_fill(const int64_t *coords, int n, int ncols, int mode, const int *input_map, void *ws,
                         size_t ws_bytes, int64_t *output_coords, int *output_map, int M, int maxActive,
                         void *stream);

/* ---- ball query + clustering -------------------------------------------------------- */
/* PG_OP.ballquery_batch_p  (src/bfs_cluster/bfs_cluster.cu:15-90), two-phase, no retry loop.
 * count: per-point hit count (strict d2<r2, capped at 1000, same batch item only) ->
 *        start_len (n,2) with start = exclusive prefix sum of len in point order (the
 *        reference's starts come from atomicAdd and are scheduling dependent);
 *        *nActive_host = total.
 * fill : neighbour indices in ascending order; entries at positions >= idx_capacity are
 *        dropped exactly as the reference truncates at n*meanActive (bfs_cluster.cu:51-59). */
size_t d3_ballquery_ws_bytes(int n);
/* with a workspace of this size (adds n*1000 ints) the count phase also stashes the hits and the fill phase only
 * compacts them: one neighbour search instead of two */
size_t d3_ballquery_ws_bytes_single_pass(int n);
int d3_ballquery_count(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n, float radius,
                       int *start_len, void *ws, size_t ws_bytes, int *nActive_host, void *stream);
int d3_ballquery_fill(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n, float radius,
                      const int *start_len, const void *ws, size_t ws_bytes, int *idx, long long idx_capacity,
                      void *stream);
/* Padded, sync-free ball query: idx_padded holds n slots of d3_ballquery_cap() entries and start_len[q] = (s * cap, len)
 * with s = q, or -- when every point in the 27 search cells around q's cell lies within one ball (a collapsed instance:
 * all those queries have the same list) -- the smallest point index of q's cell, whose slot then holds the one shared copy.
 * Same neighbours in the same order as d3_ballquery_count/fill (uniform cell grid instead of the ordered scan:
 * csrc/ballquery.hip); no nActive, no host round trip.  Valid input of d3_bfs_cluster_* (they only index idx[start + e]).
 * ws: d3_ballquery_ws_bytes(n).  Replaces the same reference call as d3_ballquery_count (bfs_cluster.cu:13-63). */
int d3_ballquery_cap(void);
int d3_ballquery_padded(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n, float radius,
                        int *start_len, void *ws, size_t ws_bytes, int *idx_padded, void *stream);

/* PG_OP.bfs_cluster  (src/bfs_cluster/bfs_cluster.cpp:28-112), on the device, two-phase.
 * count: connected components (same semantic label, directed ball-query lists, seeds in
 *        ascending index) with size >= threshold -> *sumNPoint_host, *nCluster_host.
 * fill : cluster_idxs (sumNPoint,2) = (cluster_id, point_idx) in the reference's FIFO-BFS
 *        visitation order, cluster_offsets (nCluster+1). */
size_t d3_bfs_cluster_ws_bytes(int n);
int d3_bfs_cluster_count(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n,
                         int threshold, void *ws, size_t ws_bytes, int *sumNPoint_host, int *nCluster_host,
                         void *stream);
/* d3_bfs_cluster_count with flags.  D3_BFS_ASCENDING: the caller guarantees that every neighbour list is in ascending index
 * order (ballquery_batch_p's order, src/bfs_cluster/bfs_cluster.cu:27-47): the label propagation then skips, per node, the
 * prefix of neighbours whose label cannot change (two 64-way probes).  Same results. */
#define D3_BFS_ASCENDING 1
int d3_bfs_cluster_count_ex(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold,
                            void *ws, size_t ws_bytes, int *sumNPoint_host, int *nCluster_host, int flags, void *stream);
int d3_bfs_cluster_fill(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n,
                        void *ws, size_t ws_bytes, int *cluster_idxs, int *cluster_offsets, int sumNPoint,
                        int nCluster, void *stream);

/* d3_bfs_cluster_fill with the level loop in "record form" (csrc/cluster.hip): a parallel pre-pass rewrites the lists of
 * the kept clusters as (node, dense id, list start, list length) records and the BFS keeps visited bits, frontier and
 * first-discoverer arbitration in LDS -- one global round trip per batch of 3072 edges instead of four per level.
 * erec: d3_bfs_cluster_erec_bytes(nActive) bytes of scratch, nActive = length of ball_query_idxs.  Outputs are
 * bit-identical to d3_bfs_cluster_fill. */
size_t d3_bfs_cluster_erec_bytes(long long nActive);
int d3_bfs_cluster_fill2(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, void *ws,
                         size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive, int *cluster_idxs,
                         int *cluster_offsets, int sumNPoint, int nCluster, void *stream);

/* d3_bfs_cluster_count_ex + d3_bfs_cluster_fill2 as one call: outputs at their upper bounds (cluster_idxs: cap_points x 2
 * ints, cap_points >= sumNPoint -- n always suffices; cluster_offsets: cap_clusters + 1 ints -- n / max(threshold, 1) + 1
 * suffices), the used sizes come back in *sumNPoint_host / *nCluster_host.  D3_ERR_WORKSPACE when a bound is too small.
 * Replaces the pair bfs_cluster.cpp:28-112 is called through (functions/pointgroup_ops.py:203-231) without the return to
 * the caller between the phases (a Python caller re-acquires its interpreter lock there: idle device time). */
int d3_bfs_cluster_run(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold,
                       void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive, int flags,
                       int *cluster_idxs, long long cap_points, int *cluster_offsets, long long cap_clusters,
                       int *sumNPoint_host, int *nCluster_host, void *stream);
/* d3_bfs_cluster_run cut at its one host wait: `begin` enqueues everything (count kernels, the copy of their scalars + an event,
 * the fill with its sizes read on the device) and returns a ticket; `end` waits for the event, finishes the rare cases and
 * returns the sizes (the ticket is consumed whatever it returns).  One host thread keeps several clusterings in flight on
 * different streams: begin, begin, end, end.  The buffers handed to `begin` must stay alive until `end`. */
int d3_bfs_cluster_begin(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold,
                         void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive, int flags,
                         int *cluster_idxs, long long cap_points, int *cluster_offsets, long long cap_clusters,
                         void **ticket, void *stream);
int d3_bfs_cluster_end(void *ticket, int *sumNPoint_host, int *nCluster_host);

/* ---- sparse 3-D convolution (MinkowskiEngine subset) ---------------------------------- */
/* Coordinates are (M,4) int32 rows [batch, x, y, z] (ME.SparseTensor(coordinates=...),
 * reference: model/pointgroup.py:176,268).  Key range as for voxelize_idx.
 * A kernel map is a dense table tbl (Mout, K) int32: tbl[u][k] = input row feeding output row u
 * through kernel offset k, or -1.  k = ox + Kd*oy + Kd*Kd*oz (x fastest).
 *
 * d3_kmap_k3      : K=27 neighbour table of a kernel-3 stride-1 conv at tensor stride ts
 *                   (MinkowskiConvolution(kernel_size=3): model/common.py:38,41,66; model/pointgroup.py:70).
 * d3_kmap_down_*  : kernel-2 stride-2 maps (MinkowskiConvolution(kernel_size=2, stride=2) and its
 *                   MinkowskiConvolutionTranspose: model/common.py:90,98): output coordinates
 *                   floor(c/(2ts))*2ts in first-occurrence order, parent/kidx per input row,
 *                   child (Mout,8) for the strided conv, up (M,8) for the transposed conv. */
size_t d3_coordmap_ws_bytes(int M);
int d3_kmap_k3(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *nbr, void *stream);
/* 16-bit form of a d3_kmap_k3 table: nbr16 (M*27 + 2 int16) = nbr - row, -32768 = absent; *ok16 (device int) = 1 when every
 * delta fits, else 0 (the consumers then read the dense table).  MinkowskiEngine keeps one int32 pair list per kernel map
 * (no counterpart); the executor hands both forms to the convolutions of a level (d3_net_set_k3_16). */
int d3_kmap_k3_pack16(const int *nbr, int M, void *nbr16, int *ok16, void *stream);
/* d3_kmap_k3 that writes the 16-bit form and its flag in the same pass (what the coordinate manager calls for big levels) */
int d3_kmap_k3_16(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *nbr, void *nbr16, int *ok16, void *stream);
int d3_kmap_down_count(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *parent, int *kidx,
                       int *Mout_host, void *stream);
int d3_kmap_down_fill(const int *coords, int M, int ts, void *ws, size_t ws_bytes, const int *parent,
                      const int *kidx, int *out_coords, int *child, int *up, int Mout, void *stream);

/* All stride-2 levels with ONE host round trip (d3_kmap_down_count costs one per level): the coordinate pyramid is
 * built with device-side row counts, rows_host[l] returns them all, and the tables are then filled with exact sizes by
 * d3_kmap_k3 / d3_kmap_down_fill2 without further synchronisation.  coords_out: (nlevels-1, M0, 4); parent / kidx / flag:
 * (nlevels-1, M0) (level l at [l*M0], rows of level l); rows_dev: nlevels ints.  ws >= d3_coordmap_ws_bytes(M0). */
int d3_kmap_pyramid(const int *coords0, int M0, int nlevels, void *ws, size_t ws_bytes, int *coords_out, int *parent,
                    int *kidx, int *flag, int *rows_dev, int *rows_host, void *stream);
/* The same with the host round trip split in two: _begin enqueues the level kernels and the copy of the row counts and returns
 * a ticket; _end waits for that copy only (not for work enqueued on the stream in between -- the caller puts independent
 * device work there: PointGroup.feed the input voxelisation) and returns rows_host.  *ticket == NULL when M0 == 0. */
int d3_kmap_pyramid_begin(const int *coords0, int M0, int nlevels, void *ws, size_t ws_bytes, int *coords_out, int *parent,
                          int *kidx, int *flag, int *rows_dev, void **ticket, void *stream);
int d3_kmap_pyramid_end(void *ticket, int *rows_host, int nlevels);
int d3_kmap_down_fill2(int M, int Mout, const int *parent, const int *kidx, int *child, int *up, void *stream);

/* Gather-GEMM convolution  out[u,:] = sum_k x[tbl[u,k],:] @ Wk   (tbl == NULL: identity map, K = 1).
 * flags: D3_CONV_FLIPK  -> Wk = W[K-1-k]      (data gradient of a kernel-3 conv)
 *        D3_CONV_TRANSW -> W is laid out (K, Cout, Cin) and used transposed (data gradients)
 *        D3_CONV_EXACT  -> fp32 FMA kernel instead of the bf16-MFMA kernel (fp32 accumulate in both)
 * x (Min,Cin) f32 (Min = rows of x, used for bounds / traffic accounting), W (K,Cin,Cout) f32 [or (K,Cout,Cin) with TRANSW], out (Mout,Cout) f32.
 * MFMA path needs Cin % 2 == 0 and Cout <= 224, else D3_ERR_ARG. */
#define D3_CONV_FLIPK 1
#define D3_CONV_TRANSW 2
#define D3_CONV_EXACT 4
#define D3_CONV_XSTAT 8
#define D3_CONV_ACCUM 16
#define D3_CONV_XBF16 32   /* x is stored as bf16 (ushort), Cin % 8 == 0; not with D3_CONV_EXACT */
#define D3_CONV_DYBF16 64  /* dy is stored as bf16 (d3_spconv_wgrad2 only) */
#define D3_CONV_OUTBF16 512 /* d3_spconv_fwd2*: `out` is stored as bf16 (ushort, ldo in elements); not with D3_CONV_ACCUM, a residual or
                            * D3_CONV_F32.  BatchNorm partials are taken from the unrounded values. */
#define D3_CONV_F32 256    /* d3_spconv_pack / d3_spconv_fwd2* / d3_spconv_wgrad2: the REFERENCE'S PRECISION on the matrix cores -- fp32
                            * operands (x, dy fp32; weights packed as fp32 fragments: d3_spconv_pack_bytes_ex), exact fp32 products on
                            * v_mfma_f32_16x16x4_f32, fp32 accumulation.  Not with D3_CONV_XBF16 / D3_CONV_DYBF16. */
#define D3_CONV_NOREDUCE 128 /* d3_spconv_wgrad2: leave the row-split partials in ws (d3_spconv_wgrad2_splits() of them, or one
                              * when accumulating); the caller sums them (the executor does it for all layers in one launch) */
int d3_spconv_fwd(const float *x, const int *tbl, const float *W, float *out, int Min, int Mout, int K, int Cin,
                  int Cout, int flags, void *stream);
/* Weight gradient  dW[k] = sum_u x[tbl[u,k],:]^T dy[u,:]   (dW (K,Cin,Cout) f32; cleared here unless
 * D3_CONV_ACCUM is set, then accumulated into).
 * With D3_CONV_XSTAT, tbl is the TRANSPOSED map (one row per x row, entries = dy rows):
 * dW[k'] += sum_v x[v,:]^T dy[tbl[v,k],:], k' = K-1-k with D3_CONV_FLIPK else k -- same result, but the
 * wide operand (x) is read contiguously and the narrow one gathered. */
int d3_spconv_wgrad(const float *x, const int *tbl, const float *dy, float *dW, int Min, int Mout, int K, int Cin,
                    int Cout, int flags, void *stream);

/* Second-generation MFMA kernels (csrc/spconv2.hip): wave-autonomous register gather, v_mfma_f32_16x16x32_bf16,
 * weights pre-packed into bf16 MFMA fragment order.  Same contraction and flags as d3_spconv_fwd / d3_spconv_wgrad
 * (D3_CONV_FLIPK / D3_CONV_TRANSW are applied by the pack step); Cin % 8 == 0 (wgrad2: Cout % 8 == 0 too).
 *   pack   : W (K,Cin,Cout) f32 [(K,Cout,Cin) with TRANSW] -> Wp, d3_spconv_pack_bytes() bytes.
 *   fwd2   : out[u, 0:Cout] (row stride ldo) = sum_k x[tbl[u,k]] @ Wk (+ res[u] (row stride ldr)) (+ out with
 *            D3_CONV_ACCUM); x has row stride ldx (fp32, or bf16 with D3_CONV_XBF16).  part != NULL: per-workgroup
 *            per-channel sum / sum of squares of the stored values, [d3_spconv_fwd2_nparts()][2][ceil16(Cout)] f32
 *            -- the batch statistics of the following MinkowskiBatchNorm (consumed by d3_bn_finalize_parts).
 *   wgrad2 : dW (K,CinW,Cout) f32 (CinW <= Cin: x may carry zero-padded channels) written (accumulated into with D3_CONV_ACCUM); ws >= d3_spconv_wgrad2_ws_bytes()
 *            holds row-split partials that are summed in fixed order (deterministic, no atomics). */
size_t d3_spconv_pack_bytes(int K, int Cin, int Cout);
size_t d3_spconv_pack_bytes_ex(int K, int Cin, int Cout, int flags);   /* flags & D3_CONV_F32: fp32 fragments (2x) */
int d3_spconv_pack(const float *W, void *Wp, int K, int Cin, int Cout, int flags, void *stream);
int d3_spconv_fwd2_nparts(int Mout, int K, int Cin, int Cout);
int d3_spconv_fwd2_nparts_ex(int Mout, int K, int Cin, int Cout, int flags);   /* flags & D3_CONV_F32: that call's partial rows */
/* which kernel fwd2 runs for a shape (tests assert the variant they mean to cover): out[6] = {split (1 = the few-row
 * spconv_fwd2_split_kernel, 0 = the persistent wave-per-tile spconv_fwd2_kernel), waves per workgroup, grid.x,
 * weights resident in LDS, column tiles per workgroup, grid.y} */
int d3_spconv_fwd2_plan(int Mout, int K, int Cin, int Cout, int *out6);
int d3_spconv_fwd2(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, const float *res,
                   int ldr, float *part, int Min, int Mout, int K, int Cin, int Cout, int flags, void *stream);
/* fwd2 as the data gradient of a BatchNorm -> ReLU -> conv unit, with the BatchNorm-backward reductions in the epilogue:
 * out = (sum_k x[tbl[u,k]] @ Wk) * relu'(bn(bnx[u])), part = per-workgroup (sum out, sum out * xhat) per channel. */
int d3_spconv_fwd2_bnbwd(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, float *part,
                         const float *bnx, int ldbx, const float *mean, const float *var, const float *gamma,
                         const float *beta, float eps, int relu, int Min, int Mout, int K, int Cin, int Cout, int flags,
                         void *stream);
/* the two above with the partials reduced by the last workgroup to finish (device-scope ticket + fences) instead of a
 * separate finalize launch.  counter: one zero-initialised int (left at zero).  _fin: mean / var (+ running statistics,
 * d3_bn_stats semantics) of the stored values; _bnbwd_fin: sums (2C) = (sum g, sum g*xhat), dgamma / dbeta written
 * (accumulated with accum != 0). */
int d3_spconv_fwd2_fin(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, const float *res, int ldr,
                       float *part, int *counter, float *mean, float *var, float *running_mean, float *running_var,
                       float momentum, int Min, int Mout, int K, int Cin, int Cout, int flags, void *stream);
int d3_spconv_fwd2_bnbwd_fin(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, float *part,
                             const float *bnx, int ldbx, const float *mean, const float *var, const float *gamma,
                             const float *beta, float eps, int relu, int *counter, float *sums, float *dgamma, float *dbeta,
                             int accum, int Min, int Mout, int K, int Cin, int Cout, int flags, void *stream);
/* flags of the two queries: the D3_CONV_XSTAT / D3_CONV_XBF16 / D3_CONV_DYBF16 bits of the d3_spconv_wgrad2 call they size
 * (the kernel, and with it the number of row splits, depends on the operand types) */
size_t d3_spconv_wgrad2_ws_bytes(int Min, int Mout, int K, int Cin, int Cout, int flags);
int d3_spconv_wgrad2_splits(int Min, int Mout, int K, int Cin, int Cout, int flags);
int d3_spconv_wgrad2(const void *x, int ldx, const int *tbl, const void *dy, int ldy, float *dW, int Min, int Mout,
                     int K, int Cin, int Cout, int CinW, int flags, void *ws, size_t ws_bytes, void *stream);

/* Launch timing for bench.py: with profiling on, each MFMA convolution launch is bracketed by HIP events on
 * its stream.  family 0 = forward/data-gradient kernel (spconv_fwd2_kernel, or spconv_fwd_mfma_kernel for the first-
 * generation entry points), 1 = weight-gradient kernels, 2 = spconv_fwd2_split_kernel (few-row levels).
 * collect() synchronises. */
int d3_prof_enable(int on);
int d3_prof_collect(int family, long long *launches, double *total_ms, double *total_bytes, double *total_flops);
/* every sampled launch of a family as rows of 15 doubles {ms, bytes, flops, 12 tags}; families 3 = hg_gemm* (tags maxM, maxN,
 * K, problems, kernel 0 tiled / 1 split, row tiles, waves), 4 = td_gru4_fwd (1, N, H, I), 5 = cl_bfs2 (n, clusters), 6 = un_bn_*
 * (kernel, M, C); convolutions: Min, Mout, K, Cin, Cout + the template arguments of the instance (rocprofv3's kernel name).
 * *n = records of the family; at most `cap` rows are written.  No reference counterpart (measurement only). */
int d3_prof_dump(int family, double *rows, int cap, int *n);

/* Measurement / test switches.  The library reads its environment ONCE (csrc/tuning.hip: one
 * table, one parse at first use); these entry points let tests and the A/B tools flip a switch at run time instead of
 * mutating the environment.  name = the switch's environment name ("D3_WG3", "D3_BFS_NO_STAR", ...); unknown name ->
 * D3_ERR_ARG.  No reference counterpart: the reference's only switch on this path is CUDA_LAUNCH_BLOCKING
 * (scripts/train.py), read by the CUDA runtime. */
int d3_tuning_set(const char *name, int value);
int d3_tuning_get(const char *name, int *value);
int d3_tuning_count(void);
const char *d3_tuning_name(int i);

/* MinkowskiBatchNorm (+ MinkowskiReLU) over the rows of an (M,C) feature matrix
 * (reference: model/pointgroup.py:65,72-73; model/common.py:36-40).  Training-mode batch statistics.
 * stats : mean (C) and biased var (C) in fp32 (deterministic fp64 two-stage reduction); when running_mean /
 *         running_var are non-NULL they are updated like nn.BatchNorm1d in training mode
 *         (running = (1-momentum)*running + momentum*stat, unbiased variance).  ws >= d3_bn_ws_bytes(C).
 * fwd   : y = [relu]((x-mean)*rsqrt(var+eps)*gamma+beta)
 * bwd   : dx from dy (the relu mask is recomputed from x); dgamma / dbeta are WRITTEN. */
size_t d3_bn_ws_bytes(int C);
int d3_bn_stats(const float *x, int M, int C, float *mean, float *var, float *running_mean, float *running_var,
                float momentum, void *ws, size_t ws_bytes, void *stream);
int d3_bn_relu_fwd(const float *x, const float *mean, const float *var, const float *gamma, const float *beta,
                   float *y, int M, int C, float eps, int relu, void *stream);
/* same as d3_bn_relu_fwd with y stored as bf16 (RNE): for outputs consumed by a convolution (D3_CONV_XBF16) */
int d3_bn_relu_fwd_bf16(const float *x, const float *mean, const float *var, const float *gamma, const float *beta,
                        void *y_bf16, int M, int C, float eps, int relu, void *stream);
int d3_bn_relu_bwd(const float *x, const float *dy, const float *mean, const float *var, const float *gamma,
                   const float *beta, float *dx, float *dgamma, float *dbeta, int M, int C, float eps, int relu,
                   void *ws, size_t ws_bytes, void *stream);

/* ---- native sparse U-Net executor (csrc/unet.hip) ---------------------------------------------------
 * Runs the whole backbone / ScoreNet of the detector (model/pointgroup.py:69-74,88-92: stem conv, UBlock of
 * ResidualBlock / VGGBlock units (model/common.py:22-118), final BN + ReLU) from a layer program: one call for the
 * forward, one for the backward.  The program is three int64 tables built by d3net_amd/netexec.py:
 *   tensors (ntensors x 6): level, C, ld, coff, dtype (0 f32, 1 bf16), buffer id (-1 = the external input)
 *   bufs    (nbufs x 3)   : level, width, dtype
 *   prog    (nops x 16)   : [0] type 1 CONV {in, out, res|-1, weight param, map 0 k1 / 1 k3 / 2 down / 3 up, map level,
 *                           K, CinW (rows of the weight), stats 0/1}; 2 BNACT {in, out, -, gamma, beta, running_mean,
 *                           running_var, relu, eps bits, momentum bits}; 3 PADCAST {in, out} (fp32 -> zero-padded
 *                           bf16); 4 STATS {in} (batch statistics of the external input)
 * plan(rows per level) fixes the arena layout; forward writes activations / BN state / packed weights into the
 * caller's arena (kept for the backward); backward needs a gradient arena of grad_bytes.  params[i] / pgrads[i] are
 * device pointers of parameter i and of its gradient (NULL = frozen; paccum[i] != 0: accumulate).  k3 / child / up:
 * the kernel-map tables of d3_kmap_* per level.  Data gradients run on `stream`, weight gradients on an internal
 * side stream that `stream` joins before the call returns.
 * Memory owned by the network object (the exception to "no hidden allocation"): forward keeps a device copy of its
 * weight-packing job table (hipMalloc on first use / when the table grows; refreshing it synchronises `stream` once),
 * backward a pinned-host + device pair for the batched weight-gradient reduction jobs (hipHostMalloc + hipMalloc,
 * grow-only); a few KB each, released by d3_net_destroy. */
void *d3_net_create(const int64_t *prog, int nops, const int64_t *tensors, int ntensors, const int64_t *bufs, int nbufs,
                    int nlevels, int nparams, int input_needs_grad, int out_tensor);
void d3_net_destroy(void *net);
int d3_net_plan(void *net, const int *rows, size_t *arena_bytes, size_t *grad_bytes);
long long d3_net_tensor_offset(void *net, int tensor);
int d3_net_forward(void *net, const void *const *params, const int *const *k3, const int *const *child,
                   const int *const *up, const void *input, void *arena, int training, void *stream);
int d3_net_backward(void *net, const void *const *params, const int *const *k3, const int *const *child,
                    const int *const *up, const void *input, void *arena, void *grad_arena, const float *gout,
                    float *const *pgrads, const int *paccum, float *gin, void *stream);
/* Data-parallel overlap (the reference's DDP buckets: scripts/train.py:265-268 through Lightning).  The parameter gradients
 * of a backward complete in reverse program order, so a contiguous TAIL range of the flat gradient buffer is final long
 * before the call's last kernel.  d3_net_set_chunks: op_idx[k] (strictly descending) = the op after which chunk k is
 * complete; d3_net_backward then flushes the pending weight-gradient reductions there and records two events per chunk.
 * d3_net_chunk_wait makes `stream` wait for chunk k of the last backward -- the caller starts that chunk's all-reduce on
 * it while the rest of the backward is still running.  2 * (nchunks + 2) <= 32, i.e. nchunks <= 14 (else D3_ERR_ARG); 0 switches the feature off. */
int d3_net_set_chunks(void *net, const int *op_idx, int nchunks);
/* per level: the 16-bit form of the k3 table handed to the next d3_net_forward / d3_net_backward call (NULL entries, or a NULL
 * array, = dense tables only).  The caller passes only tables whose d3_kmap_k3_pack16 flag it has READ as 1 (ok16[l]: any
 * non-NULL pointer, unused by the kernels).  The arrays are copied; the tables must stay alive like the dense ones.
 * d3_spconv_t16_launches: launches so far that read a 16-bit table (tests). */
int d3_net_set_k3_16(void *net, const void *const *k3_16, const int *const *ok16);
/* Round 5 (input prefetch): the stem's zero-padded bf16 input prepared outside the forward.  d3_net_padded_channels: its width (0: this
 * executor has no such operand -- no stem, or the reference-precision program).  d3_net_padcast: (M, C_in) fp32 voxel features -> (M,
 * padded) bf16, the launch d3_net_forward would issue first.  d3_net_set_padded_input: hands the prepared buffer to the NEXT
 * d3_net_forward / d3_net_backward call (one call, then dropped), which reads the stem's operand from it.  No reference counterpart
 * (MinkowskiEngine convolves the fp32 features directly, model/pointgroup.py:70). */
int d3_net_padded_channels(void *net);
int d3_net_padcast(void *net, const void *input, void *out, long long M, void *stream);
int d3_net_set_padded_input(void *net, const void *xp);
long long d3_spconv_t16_launches(void);
int d3_net_chunk_wait(void *net, int k, void *stream);

/* ---- point-level heads (csrc/heads.hip) -------------------------------------------------------------
 * sem_seg / offset_net (model/pointgroup.py:77-85,274-279) and the semantic loss (:389-390) at N ~ 165k rows:
 *   tall_wgrad   : weight (and bias) gradient of y = x W^T + b for a tall-skinny x: dW (O,I) = dy^T x, db (O) = column
 *                  sums of dy (NULL to skip); I, O <= 32; deterministic two-stage reduction.
 *   cross_entropy: nn.functional.cross_entropy(z (N,C), label (N) int64, ignore_index), mean over counted rows:
 *                  out[0] = loss, out[1] = counted rows, grad (N,C) = softmax - onehot (0 on ignored rows). */
size_t d3_tall_wgrad_ws_bytes(int I, int O);
int d3_tall_wgrad(const float *x, const float *dy, float *dW, float *db, int N, int I, int O, void *ws, size_t ws_bytes,
                  void *stream);
/*   offset_loss  : the offset L1 and direction losses of PointGroup.loss (model/pointgroup.py:397-420) and their unscaled
 *                  gradients w.r.t. pt_offsets in one pass: out[0] = offset_norm_loss, out[1] = offset_dir_loss, out[2] =
 *                  sum(valid); d loss / d pt = (w_norm*g1 + w_dir*g2) / (out[2] + 1e-6). */
size_t d3_offset_loss_ws_bytes(void);
int d3_offset_loss(const float *pt, const float *coords, const float *info, int ldi, const int64_t *ids, long long ignore,
                   float *g1, float *g2, float *out, int N, void *ws, size_t ws_bytes, void *stream);
/* The two point-level heads of PointGroup (model/pointgroup.py:77-85, 277-283) on x (N, m = 16): scores (N,C) = x Ws^T + bs
 * (C <= 32), preds (N) int64 = first row arg-max, h (N,16) = x W0^T + b0, y (N,16) = ReLU(BatchNorm1d(h)) with batch statistics
 * (training != 0: running_mean / running_var / num_batches_tracked updated like nn.BatchNorm1d when given) or the running
 * statistics (training == 0), offsets (N,3) = y W3^T + b3.  stat (32) = [mean | 1/sqrt(var + eps)] (for the backward).
 * Three launches; x is read once.  ws: d3_point_heads_ws_bytes(). */
size_t d3_point_heads_ws_bytes(void);
int d3_point_heads_fwd(const float *x, long long N, int m, int C, const float *Ws, const float *bs, const float *W0, const float *b0,
                       const float *gamma, const float *beta, const float *W3, const float *b3, float eps, float momentum,
                       int training, float *running_mean, float *running_var, long long *num_batches_tracked, float *scores,
                       long long *preds, float *h, float *y, float *offsets, float *stat, void *ws, size_t ws_bytes, void *stream);
/* Backward pieces of the point heads: dy (N,16) = (g_off (N,3) W3 (3,16)) * (y > 0);  dx (N,16) = dh (N,16) W0 (16,16) +
 * g_scores (N,C) Ws (C,16) (either term may be NULL) -- the data gradients of the three tall Linear layers in one pass each
 * (the weight gradients: d3_tall_wgrad). */
int d3_point_heads_dy(const float *g_off, const float *W3, const float *y, long long N, float *dy, void *stream);
int d3_point_heads_dx(const float *dh, const float *W0, const float *g_scores, const float *Ws, long long N, int C, float *dx,
                      void *stream);
/* Proposal score loss of PointGroup.loss (reference model/pointgroup.py:436-452: ious.max(1), get_segmented_scores,
 * binary_cross_entropy_with_logits(...).mean()) in one launch.  ious: (P, nInst) row-major.  gt_iou: (P) row maxima;
 * dscore: (P) d loss / d score; out: 1 + P floats, out[0] = loss, out[1 + p] = proposal p's term (summed in proposal order). */
int d3_score_loss(const float *scores, const float *ious, int P, int nInst, float fg, float bg, float *gt_iou,
                  float *dscore, float *out, void *stream);
/* compute_cap_loss (lib/captioning/loss_helper.py:177-224) in two launches: pred (N,S,V) logits, target (N, S) int64 with row
 * pitch ld_target (a view of lang_ids[:, 1:S+1]), good (N) bool = descriptions whose target box passed the IoU threshold (the
 * others count as ignored, like target 0).  out2 = [sum of the word losses / count, word accuracy], count = max(#counted
 * words, 1); dpred (N,S,V) = d loss / d logits.  ws: d3_masked_xe_ws_bytes(N, S).  Deterministic. */
size_t d3_masked_xe_ws_bytes(int N, int S);
int d3_masked_xe(const float *pred, const long long *target, long long ld_target, const unsigned char *good, int N, int S, int V,
                 float *dpred, float *out2, void *ws, size_t ws_bytes, void *stream);
/* compute_node_orientation_loss (lib/captioning/loss_helper.py:244-307) in one launch.  preds: the num_bins orientation logits of
 * edge e of scene b at preds[b * ld_batch + e * ld_edge + 0..num_bins) (a view of the (B, E, num_bins + 1) edge predictions);
 * edge_index (B,2,E) float (compacted node ids; padded entries 0), num_src / num_tar (B) int64 (edges >= num_src * num_tar of a
 * scene get weight 0), assign (B,K) int64 = GT object of every proposal slot, rotations (B,G,3,3), rot_masks (B,G) fp32;
 * bounds_host = HOST pointer to the nbounds = num_bins - 1 bin boundaries (`radian_to_label`, :226-242), passed as kernel
 * arguments.  out2: 3 + 768 floats, out2[0..2] = [loss, accuracy, W = sum of the edge weights + 1e-8] (the rest: per-workgroup
 * partial sums); dpreds (B*E, num_bins) = W * d loss / d logits (the caller divides by W).  Deterministic (fixed-order reduction). */
int d3_orientation_loss(const float *preds, long long ld_batch, long long ld_edge, const float *edge_index,
                        const long long *num_src, const long long *num_tar, const long long *assign, const float *rotations,
                        const float *rot_masks, int B, int E, int K, int G, int num_bins, const float *bounds_host,
                        int nbounds, float *dpreds, float *out2, void *stream);
/* PointGroup.convert_stack_to_batch + get_object_assignments (reference model/pointgroup.py:216-263).  Kept proposals
 * (feats (P,m), crop (P,9): centre, size, -, semantic class, -; scores (P); bids (P) scene of each) are scattered to the
 * padded, per-scene shuffled (B,K,.) tensors, which the caller has zeroed: slot = b*K + inv_perm[b][rank of p in b]
 * for rank < K.  perm: (B,K) int64 permutations of 0..K-1.  slot: (P) out (-1: dropped).  assign (B,K) (optional):
 * index of the L1-nearest row of center_label (B,G,3) for every slot.  P <= 4096, B*K <= 8192. */
int d3_stack_to_batch(const float *feats, const float *crop, const float *scores, const int *bids, const long long *perm,
                      const float *center_label, int G, int P, int m, int B, int K, float *feats_b, float *bbox_b,
                      float *center_b, float *sem_b, float *scores_b, float *mask_b, long long *slot, long long *assign,
                      void *stream);
/* AdamW step (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments, no amsgrad) over a list of
 * fp32 tensors in one launch.  ptrs: device table, 4 pointers per tensor (param, grad, exp_avg, exp_avg_sq); numel:
 * elements per tensor; blocks: (tensor, chunk) int pairs, one per workgroup, chunk = d3_adamw_chunk() elements.
 * The reference trains with torch.optim.Adam/AdamW through Lightning (model/pipeline.py:738-757). */
int d3_adamw_chunk(void);
int d3_adamw(const long long *ptrs, const int *numel, const void *blocks, int nblocks, double lr, double beta1, double beta2,
             double eps, double weight_decay, double bias_correction1, double bias_correction2_sqrt, void *stream);
/* out[idx[s], :] += g[s, :] (out zero-filled by the caller): backward of the cluster feature gather
 * (model/pointgroup.py:130); deterministic when every output row receives at most two addends, as it does there */
int d3_scatter_add_rows(const float *g, const int64_t *idx, float *out, long long S, int C, void *stream);
/* out (S,C) = feats[idx]: the forward of the same gathers (C % 4 == 0) */
int d3_gather_rows(const float *feats, const int64_t *idx, float *out, long long S, int C, void *stream);
/* out[r] = idx[r] in [0, rows) ? feats[idx[r]] : 0 (scatter == 0; S rows of C floats), or its transpose for UNIQUE indices
 * (scatter != 0: out[idx[r]] = feats[r] for the in-range entries, out (rows, C) zero-filled by the caller): the padded
 * placement of the relation graph's edge messages / predictions (model/graph_module.py:291-308) without the zero-row copy. */
int d3_gather_rows_pad(const float *feats, long long rows, const int64_t *idx, float *out, long long S, int C, int scatter,
                       void *stream);
size_t d3_cross_entropy_ws_bytes(void);
int d3_cross_entropy(const float *z, const int64_t *label, float *grad, float *out, int N, int C, int ignore_index,
                     void *ws, size_t ws_bytes, void *stream);

/* ---- proposal-level attention (listener) ------------------------------------------------ */
/* Core of ScaledDotProductAttention.forward between the projections (model/transformer/attention.py:61-75):
 * softmax(q k^T / sqrt(dk) + bias, masked where mask == 0) v, per (batch item, head), fp32.
 * q (B,nq,h*dk), k (B,nk,h*dk), v (B,nk,h*dv), out (B,nq,h*dv) -- the layouts nn.Linear produces;
 * bias (B/bias_div, h, nq, nk) or NULL: additive attention weights, shared by bias_div consecutive batch items
 * (the reference replicates them per description chunk, model/match_module.py:324-326);
 * mask (B, nk) or NULL: 0 = masked key (the reference replicates it to (B,h,nq,nk), match_module.py:191-197);
 * P (B,h,nq,nk): softmax probabilities, kept for the backward.  nq,nk <= 128, dk,dv <= 32.
 * bwd: dS (B,h,nq,nk) scratch; dq, dk, dv written. */
int d3_attn_fwd(const float *q, const float *k, const float *v, const float *bias, const float *mask, float *out,
                float *P, int B, int h, int nq, int nk, int dk, int dv, int bias_div, void *stream);
int d3_attn_bwd(const float *q, const float *k, const float *v, const float *P, const float *dout, float *dS,
                float *dq, float *dk, float *dv, int B, int h, int nq, int nk, int dkdim, int dvdim, void *stream);

/* Fused residual add + LayerNorm (csrc/layernorm.hip): y = LayerNorm(a + b) * gamma + beta over the last dimension D of R
 * rows (b may be NULL), torch.nn.LayerNorm semantics (biased variance, eps inside the root).  Replaces
 * `self.layer_norm(queries + out)` of MultiHeadAttention (model/transformer/attention.py:170-176) and the LayerNorm of
 * `lang_fc` (model/match_module.py:170-173).  mean / rstd (R each) are kept for the backward; bwd: dx = d(a) = d(b),
 * dgamma / dbeta (D each, written); ws >= d3_layernorm_ws_bytes(R, D).  D <= 1024. */
int d3_layernorm_fwd(const float *a, const float *b, const float *gamma, const float *beta, float *y, float *mean,
                     float *rstd, int R, int D, float eps, void *stream);
size_t d3_layernorm_ws_bytes(int R, int D);
int d3_layernorm_bwd(const float *a, const float *b, const float *gamma, const float *mean, const float *rstd,
                     const float *dy, float *dx, float *dgamma, float *dbeta, int R, int D, void *ws, size_t ws_bytes,
                     void *stream);

/* ---- small-batch fp32 GEMMs of the proposal-level heads (csrc/hgemm.hip) ---------------------------
 * Every nn.Linear / nn.GRUCell product of the speaker and listener heads (model/caption_module.py:72-133,
 * model/graph_module.py:101-108, model/lang_module.py:51-55):
 *     C (M,N) [+]= act( sum_seg A_seg (M,K_seg) . B_seg (N,K_seg)^T + bias[N] + add (M,N) )
 * fp32 operands, exact fp32 products and accumulation on the matrix cores (v_mfma_f32_16x16x4_f32).  Up to three K
 * segments (torch.cat of inputs against one weight matrix is never materialised); A rows of a segment may be gathered
 * through `ia` (embedding lookup); an operand is row-major (element (r,k) at base[r*ld + k]) or k-major (base[k*ld + r]):
 * y = x W^T, dx = dy W (B k-major) and dW = dy^T x (both k-major) are the same kernel.  perm_nb > 0 stores row r at row
 * (r % perm_nb) * perm_s + r / perm_nb (time-major rows -> batch-major logits).  Up to 4 problems per call share a launch. */
typedef struct {
    const float *A; const int *ia; long long lda; int a_kmajor;
    const float *B; long long ldb; int b_kmajor;
    int K;
} d3_gemm_seg;
typedef struct {
    d3_gemm_seg seg[3]; int nseg;
    int M, N;
    float *C; long long ldc;
    const float *bias; const float *add; long long ldadd;
    int relu, accum, perm_nb, perm_s;
    /* Optional epilogue, gru != 0 (round 5; N == gru_H; the wave-per-tile / K-split kernels only: fewer than 2048 16 x 16 output tiles): the finished
     * element v (after bias / add / accumulate) is the last contribution to dh', the gradient of a GRUCell's NEW state, and the
     * cell's gate backward (torch.nn.GRUCell's autograd as model/caption_module.py:72-133 uses it) runs on it in place of a
     * launch of its own:   dh' = g_d0 + g_d1 + v  (NULL = absent);  dn = dh'(1-z), dz = dh'(hp-n), dn_pre = dn(1-n^2),
     *   g_dgi[row] = [dn_pre ghn r(1-r), dz z(1-z), dn_pre],  g_dgh[row] = [same, same, dn_pre r],  g_dhp = dh' z.
     * r / z / n / ghn: the cell's saved gates (M, gru_H); hp: its previous state (row stride g_ldh).  C is read (accumulate) but NOT
     * written in this mode: the carried gradient goes to g_dhp, which may alias C. */
    int gru, gru_H;
    const float *g_d0; long long g_ld0; const float *g_d1; long long g_ld1;
    const float *g_r, *g_z, *g_n, *g_ghn, *g_hp; long long g_ldh;
    float *g_dgi; long long g_lddgi; float *g_dgh; float *g_dhp;
} d3_gemm_prob;
int d3_hgemm(const d3_gemm_prob *probs, int nprobs, void *stream);
/* out[c] (+)= sum_r x[r*ld + c], r < R, c < C (bias gradients; two-stage, fixed summation order); ws >= d3_colsum_ws_bytes(C) */
size_t d3_colsum_ws_bytes(int C);
int d3_colsum(const float *x, long long ld, int R, int C, float *out, int accum, void *ws, size_t ws_bytes, void *stream);

/* ---- top-down captioner, native (csrc/topdown.hip) -------------------------------------------------------
 * TopDownSceneCaptionModule (model/caption_module.py:13-62 parameters, :72-133 step, :510-687 teacher-forced driver):
 * x1 = map_topdown([emb[word] | h2 | target]); h1 = GRUCell1(x1, h1); a = softmax_k(attend . tanh(map_feat(obj)[k] +
 * map_hidd(h1)), masked scores := 0); att = sum_k a[k] obj[k]; x2 = map_lang([att | h1]); h2 = GRUCell2(x2, h2);
 * logits = classifier(h2), for S steps with teacher forcing (step t reads word_ids[n, t]).  N samples, K proposals,
 * H hidden (512), E embedding (300), F feature (128), V vocabulary.  All matrices row-major fp32, nn.Linear layout
 * (out, in); GRU weights (3H, in) in torch's r, z, n gate order.  ws: d3_topdown_ws_bytes() bytes, filled by the forward
 * and read by the backward (saved activations).  H % 16 == 0, E % 4 == 0, F % 4 == 0, F <= 128. */
typedef struct {
    int N, K, S, V, H, E, F, Tw;          /* Tw: row stride of word_ids */
    const long long *word_ids;            /* (N, Tw) */
    const float *emb;                     /* (V, E) */
    const float *target;                  /* (N, F) */
    const float *obj;                     /* (N, K, F) */
    const float *mask;                    /* (N, K): 0 = masked proposal */
    const float *W_td, *b_td;             /* map_topdown (E, E+H+F), (E) */
    const float *Wih1, *Whh1, *bih1, *bhh1;   /* recurrent_cell_1 */
    const float *W_feat, *W_hidd, *w_att;     /* map_feat (H,F), map_hidd (H,H), attend (1,H) */
    const float *W_lang, *b_lang;         /* map_lang (E, F+H), (E) */
    const float *Wih2, *Whh2, *bih2, *bhh2;   /* recurrent_cell_2 */
    const float *Wc0, *bc0, *Wc2, *bc2;   /* classifier.0 (H,H), classifier.2 (V,H) */
    float *logits;                        /* out (N, S, V) */
    float *attn;                          /* out (N, K, S) = topdown_attn, or NULL */
    void *ws; size_t ws_bytes;
} d3_topdown_args;
typedef struct {
    const float *dlogits;                 /* (N, S, V) */
    float *dW_td, *db_td, *dWih1, *dWhh1, *dbih1, *dbhh1, *dW_feat, *dW_hidd, *dw_att, *dW_lang, *db_lang;
    float *dWih2, *dWhh2, *dbih2, *dbhh2, *dWc0, *dbc0, *dWc2, *dbc2;   /* written (same shapes as the parameters) */
    float *dobj, *dtarget;                /* (N,K,F), (N,F): written */
    void *ws; size_t ws_bytes;            /* d3_topdown_bwd_ws_bytes() of scratch */
} d3_topdown_grads;
size_t d3_topdown_ws_bytes(int N, int K, int S, int H, int E, int F);
size_t d3_topdown_bwd_ws_bytes(int N, int K, int S, int V, int H, int E, int F);
int d3_topdown_xe_forward(const d3_topdown_args *a, void *stream);
int d3_topdown_xe_backward(const d3_topdown_args *a, const d3_topdown_grads *g, void *stream);
/* the same with the parameter-gradient work (weight-gradient GEMMs batched over time, bias column sums: nothing the rest of the
 * backward waits for) on a second stream `side` (NULL: everything on `stream`).  The call forks `side` off `stream`; the caller joins:
 * `side` must be waited for before a parameter gradient is read, and every buffer of `a` / `g` must stay alive until then. */
int d3_topdown_xe_backward_ex(const d3_topdown_args *a, const d3_topdown_grads *g, void *stream, void *side);
/* One decode step for the greedy / evaluation decodes (model/caption_module.py:350-383, 689-770), inference only: uses N, K,
 * V, H, E, F, emb, target, obj, mask and the parameters of `a`.  word (N) int64 -> logits (N,V), attn (N,K); hidden states
 * h1_in / h2_in (N,H) -> h1_out / h2_out (distinct buffers).  fp = map_feat(obj) from d3_topdown_feat_proj (rows = number of
 * (K,F) object blocks * K).  obj_div: consecutive samples sharing one object block (1: one block per sample). */
size_t d3_topdown_step_ws_bytes(int N, int K, int H, int E, int F);
int d3_topdown_feat_proj(const float *obj, const float *W_feat, float *fp, int rows, int H, int F, void *stream);
int d3_topdown_step(const d3_topdown_args *a, const long long *word, const float *fp, int obj_div, const float *h1_in,
                    const float *h2_in, float *h1_out, float *h2_out, float *logits, float *attn, void *ws, size_t ws_bytes,
                    void *stream);

/* Selection step of the sampling loops around d3_topdown_step (beam search: model/caption_module.py:136-349; greedy: :350-383) in
 * one launch each: log_softmax, candidate scores sums + logp, the b best of live * V candidates best first (ties: lower flat
 * index), beam_ix / tok / chosen log-prob / running sums (snapshot and the -1000 penalised continuation, :300) / ended flags, the
 * token histories of the chosen beams (seq_out[n][r][:t] = seq_prev[n][beam_ix][:t], seq_out[n][r][t] = tok; rows of Tmax int64)
 * and the re-ordering of the two hidden states (h*_out row n*b + r = h*_in row n*b + beam_ix; h1_in NULL: skipped).
 * logits (N*b, V): row n*b + j = live beam j of sample n (live = 1 at t = 0); sums_in (N, live).  b <= 8.
 * d3_greedy_select: word[n] = first arg-max of logits[n], lp[n] = its log-softmax value. */
int d3_beam_select(const float *logits, const float *sums_in, int N, int live, int b, int V, int eos, int last, int t, int Tmax,
                   const long long *seq_prev, long long *seq_out, long long *tok_out, float *snap_out, unsigned char *ended_out,
                   float *sums_out, const float *h1_in, const float *h2_in, float *h1_out, float *h2_out, int H, void *stream);
int d3_greedy_select(const float *logits, int N, int V, long long *word, float *lp, void *stream);
/* The two decodes as single calls (same launches as the loops over d3_topdown_step + d3_*_select, issued inside the library).
 * d3_topdown_greedy (model/caption_module.py:350-383): h1_a / h2_a (N,H) = initial (zero) states, h1_b / h2_b scratch, logits (N,V)
 * and attn (N,K) scratch, first_word (N) = sos; words / lps (max_len, N) written.
 * d3_topdown_beam (:136-349): a->N = samples * b rows (row n*b + j = beam j of sample n, obj_div = b); h1 / h2: three (N,H) buffers
 * each, [0] = initial (zero) states; allseq (max_len, samples, b, max_len) zero-filled by the caller, snap_all / ended_all (max_len,
 * samples, b), sums0 (samples, b) zero-filled, sums1 (samples, b) and tok (a->N) scratch.  Every step's beams are kept (a beam that
 * ended at step t: ended_all[t] != 0, its score snap_all[t], its tokens allseq[t][..][:t+1]). */
int d3_topdown_greedy(const d3_topdown_args *a, const float *fp, int obj_div, float *h1_a, float *h2_a, float *h1_b, float *h2_b,
                      float *logits, float *attn, void *ws, size_t ws_bytes, const long long *first_word, int max_len,
                      long long *words, float *lps, void *stream);
int d3_topdown_beam(const d3_topdown_args *a, const float *fp, int b, float *const *h1, float *const *h2, float *logits, float *attn,
                    void *ws, size_t ws_bytes, const long long *first_word, int eos, int max_len, long long *allseq, float *snap_all,
                    unsigned char *ended_all, float *sums0, float *sums1, long long *tok, void *stream);
/* Both decodes of one self-critical step (model/caption_module.py:588-633) as one chain: a->N = samples * (b + 1) rows, row
 * n*(b+1) + j = beam j of sample n (j < b) / its greedy row (j = b), obj_div = b + 1; beam outputs as d3_topdown_beam, greedy
 * outputs g_words / g_lps (glen, samples), glen >= max_len.  tok (a->N) scratch.  Row for row the arithmetic of the separate calls. */
int d3_topdown_beam_greedy(const d3_topdown_args *a, const float *fp, int b, float *const *h1, float *const *h2, float *logits,
                           float *attn, void *ws, size_t ws_bytes, const long long *first_word, int eos, int max_len,
                           long long *allseq, float *snap_all, unsigned char *ended_all, float *sums0, float *sums1, long long *tok,
                           int glen, long long *g_words, float *g_lps, void *stream);

/* ---- packed-sequence GRU of the language encoder (csrc/topdown.hip) ---------------------------------------
 * nn.GRU(I -> H, batch_first=True) over pack_padded_sequence(x (N,T,I), lens (N)) as LangModule runs it
 * (model/lang_module.py:51-55, 146-150; torch gate order r, z, n): hiddens (N,T,H) zero beyond a sample's length, last (N,H)
 * the final state of every sample.  ws (d3_gru_seq_ws_bytes) keeps the input-side gates, states and gate values for the
 * backward.  backward: d_hiddens / d_last (either NULL) -> dWih (3H,I), dWhh (3H,H), dbih, dbhh (3H) written; dx (N,T,I)
 * written when non-NULL.  H % 16 == 0, I % 4 == 0. */
size_t d3_gru_seq_ws_bytes(int N, int T, int I, int H);
size_t d3_gru_seq_bwd_ws_bytes(int N, int T, int I, int H);
int d3_gru_seq_forward(const float *x, const int *lens, const float *Wih, const float *Whh, const float *bih, const float *bhh, int N,
                       int T, int I, int H, float *hiddens, float *last, void *ws, size_t ws_bytes, void *stream);
int d3_gru_seq_backward(const float *x, const int *lens, const float *Wih, const float *Whh, int N, int T, int I, int H,
                        const float *d_hiddens, const float *d_last, const void *ws, float *dWih, float *dWhh, float *dbih, float *dbhh,
                        float *dx, void *ws2, size_t ws2_bytes, void *stream);

/* ---- relation graph (csrc/edgeconv.hip) -----------------------------------------------------------------
 * GraphModule / EdgeConv (model/graph_module.py:21-114, 252-324) for all B scenes at once, fixed-size outputs, no host
 * round trip.  adj (B,K,K) 0/1 adjacency (rows = _query_locals of every proposal, L ones each), mask (B,K) valid proposals.
 * Edges = row-major non-zeros of adj restricted to valid x valid (the reference's scipy COO order), stored per scene in a
 * padded block of K*L slots:
 *   src / dst (B,K*L) int32   global node ids b*K + slot of the adjacency row (x_j) / column (x_i, aggregation target), -1 pad
 *   edge_index (B,2,K*L) f32  the reference's `edge_index` output: compacted (valid-only) node ids of the first n edges
 *   cnt (B,4) int32           E, n_source (rows with an edge), n_target = E / n_source, number of valid nodes
 *   in_ptr (B,K+1), in_list (B,K*L): incoming edges (scene-local edge ids) of every node in edge order
 *   out_start / out_cnt (B,K): the contiguous outgoing range of every node
 *   feat_src / pred_src (B,K*L) int64: row of the (B*K*L [+1 zero row], C) message / prediction matrix that lands in slot
 *       (r, k) of `edge_feature` (message r*n_target + k) resp. row j of `edge_orientations` (only when E == n); B*K*L = none
 * edgeconv_fwd: message = W2 relu(W0 [x_i | x_j - x_i] + b0) + b2 per edge (fp32 matrix cores), node = sum of incoming
 * messages in edge order.  ws (d3_edgeconv_ws_bytes) keeps [edge inputs | hidden] for the backward. */
int d3_graph_edges(const float *adj, const float *mask, int B, int K, int L, int *src, int *dst, float *edge_index, int *cnt,
                   int *in_ptr, int *in_list, int *out_start, int *out_cnt, long long *feat_src, long long *pred_src,
                   void *stream);
size_t d3_edgeconv_ws_bytes(int Emax, int Cin, int Cout);
size_t d3_edgeconv_bwd_ws_bytes(int Emax, int Cin, int Cout);
int d3_edgeconv_fwd(const float *x, const float *W0, const float *b0, const float *W2, const float *b2, const int *src,
                    const int *dst, const int *in_ptr, const int *in_list, int B, int K, int L, int Cin, int Cout, float *node,
                    float *msg, void *ws, size_t ws_bytes, void *stream);
int d3_edgeconv_bwd(const float *W0, const float *W2, const int *src, const int *dst, const int *in_ptr, const int *in_list,
                    const int *out_start, const int *out_cnt, int B, int K, int L, int Cin, int Cout, const float *d_node,
                    const float *d_msg, const void *ws, float *dx, float *dW0, float *db0, float *dW2, float *db2, void *ws2,
                    size_t ws2_bytes, void *stream);

/* ---- evaluation-path non-maximum suppressions (csrc/nms.hip) ------------------------------------------------
 * nms3d_samecls: class-aware greedy 3D box NMS of parse_predictions (lib/det/ap_helper.py:80-108, lib/det/nms.py:110-150),
 *   all scenes in one launch.  boxes (B,K,8) = [x1,y1,z1,x2,y2,z2,score,class], valid (B,K) != 0 -> pick (B,K) 0/1;
 *   float64 arithmetic like the numpy original; K <= 256.  visit: NULL (descending score, exact ties: later index first) or
 *   (B,K) int32 candidate indices in visiting order, -1 padded (numpy's argsort leaves the order of tied scores to its sort
 *   implementation; pass its order to reproduce it).
 * instance_cross_iou: point-mask IoU between all pairs of clusters (model/pointgroup.py:577-589) from the (cluster, point)
 *   lists instead of a dense (P,N) mask product; ious (P,P) f32; member: 2*N ints scratch; *overflow_dev = 1 when a point is
 *   in more than two clusters (then the result is invalid).
 * nms_matrix: get_nms_instances (lib/utils/eval.py:75-97) over a dense IoU matrix: candidates keep[i] != 0 in descending
 *   score; picked[0..*npicked) in pick order.  order_scratch / picked: n ints each.  n <= 12288. */
int d3_nms3d_samecls(const float *boxes, const float *valid, const int *visit, int B, int K, double iou_thr, int old_type, float *pick,
                     void *stream);
int d3_instance_cross_iou(const int *cluster_idxs, const int *offsets, long long S, int P, int N, float *ious, int *member,
                          int *overflow_dev, void *stream);
int d3_nms_matrix(const float *ious, const float *scores, const unsigned char *keep, int n, float thr, int *order_scratch, int *picked,
                  int *npicked, void *stream);

/* ---- CIDEr-D reward of the self-critical speaker update (csrc/cider.hip) -----------------------------
 * Replaces lib/capeval/cider/cider_scorer.py:11-193 (precook / compute_doc_freq / counts2vec / sim) as called per RL step by
 * lib/captioning/loss_helper.py:15-96 (host python over word tuples, twice per step).  Sentences are int32 token ids (< 65535;
 * reference words outside the vocabulary get corpus-private ids): tokens (R, ldt) / lens (R) = the reference corpus.  One call
 * scores E entries: entry e = candidate cand[e, :clen[e]] ("eos" appended when absent, like the reference) against the reference
 * set ent_u[e] in [0, U); set u = corpus rows slot_row[u_off[u] .. u_off[u+1]) (SR rows in total), used by mult[u] entries (the
 * document frequency counts a set once per entry).  hash_slots: power of two >= 2 x the distinct n-grams of the used sets.
 * scores (E) float64 == Cider().compute_score per-entry scores (x10, sigma 6).  *overflow_dev != 0: a hash table overflowed or
 * nothing was written -- the caller falls back to its host scorer.  Sentences are cut at 160 tokens. */
size_t d3_cider_ws_bytes(int SR, int E, int hash_slots);
int d3_cider_scores(const int *tokens, int ldt, const int *lens, const int *slot_row, const int *u_off, const int *mult,
                    const int *ent_u, int U, int SR, const int *cand, int ldc, const int *clen, int E, int eos, double sigma,
                    int hash_slots, double *scores, int *overflow_dev, void *ws, size_t ws_bytes, void *stream);

/* ---- proposal geometry (speaker / graph heads) ------------------------------------------ */
/* Distance matrix of `_query_locals` (model/graph_module.py:184-227 == model/caption_module.py:800-842) for all
 * target proposals at once: corners (B,K,8,3), masks (B,K) -> dist (B,K,K), dist[b,t,j] as the reference's pc_dist
 * for target id t before its top-k (invalid / overlaid (IoU >= overlay_threshold) -> 1e30, self -> 0 or 1e30). */
int d3_query_locals_dist(const float *corners, const float *masks, float *dist, int B, int K, int include_self,
                         float overlay_threshold, int center_mode, void *stream);
/* mask (rows, K) = 1 at the L smallest entries of every row of dist (rows, K), ties by ascending index -- torch.topk(largest =
 * False) + scatter of ones (model/graph_module.py:218-227 / caption_module.py:833-842) in one launch. */
int d3_query_locals_mask(const float *dist, float *mask, int rows, int K, int L, void *stream);
/* The captioner's per-description inputs straight from the per-scene tensors (model/caption_module.py:416-508 `select_target`,
 * :530-560, :866-885 `_add_relation_feat`); description n belongs to scene n / per_scene.
 *   select_target: target_ids[n] = first arg-max over the scene's K proposals of the AABB IoU (lib/utils/bbox.py:247-271, fp32,
 *                  operation by operation as the library ops) between corners (B,K,8,3) and ref_corners (N,8,3); target_ious[n]
 *                  that IoU; labels[n] = first arg-max of ref_labels (N,G).
 *   inputs_fwd   : obj (N,K,F) = base[b] with edge[b][t][j] (edge: (B,K,L,F), NULL: none) added at the j-th one of the target's
 *                  adjacency row adj[b][t] (B,K,K); target_feats (N,F) = base[b][t]; valid (N,K) = locals[b][t] (NULL: skipped);
 *                  nbr (N,L) int32 = those slots (saved for the backward).
 *   inputs_bwd   : d_base (B,K,F), d_edge (B,K,L,F; zero-filled by the caller, NULL: none) from g_obj (N,K,F), g_target (N,F) or
 *                  NULL; a scene's descriptions are summed in ascending order (deterministic, no atomics). */
int d3_caption_select_target(const float *corners, const float *ref_corners, const float *ref_labels, int N, int per_scene, int K,
                             int G, long long *target_ids, float *target_ious, long long *labels, void *stream);
int d3_caption_inputs_fwd(const float *base, const float *edge, const float *adj, const float *locals, const long long *target_ids,
                          int N, int per_scene, int K, int L, int F, float *obj, float *target_feats, float *valid, int *nbr,
                          void *stream);
int d3_caption_inputs_bwd(const float *g_obj, const float *g_target, const long long *target_ids, const int *nbr, int N, int per_scene,
                          int K, int L, int F, float *d_base, float *d_edge, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* D3HIP_H */
