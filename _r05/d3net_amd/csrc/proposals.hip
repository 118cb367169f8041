// proposals.hip -- proposal-level geometry for the speaker / graph heads (gfx950).
//
// d3_query_locals_dist: the distance matrix behind `_query_locals` (reference: model/graph_module.py:184-227 and the
// identical model/caption_module.py:800-842), for EVERY target proposal of every scene in one launch.  The reference
// calls `_query_locals` once per target id (128 sequential calls per forward in GraphModule._create_adjacent_mat,
// :229-238), each with a GPU -> CPU -> GPU round trip for the numpy AABB IoU (:206-210).
//   dist[b,t,j] = min over the 8 corners c of box t of sqrt(|c - centre_j|^2 + 1e-8)     ("corner" query mode)
//               = 1e30 if proposal j is invalid, or IoU(box t, box j) >= overlay_threshold
//               = 0 (include_self) / 1e30 for j == t
// centre = (min + max) / 2 of the corners; IoU as lib/utils/bbox.py:247-271 in fp32.  The k-smallest selection stays
// a library top-k on the host side.  Bytes: 96*B*K in, 4*B*K*K out -- launch bound.
#include "common.h"

__global__ void query_locals_dist_kernel(const float *__restrict__ corners, const float *__restrict__ masks,
                                         float *__restrict__ dist, int B, int K, int include_self, float thr,
                                         int center_mode) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * K * K) return;
    const int j = e % K, t = (e / K) % K, b = e / (K * K);
    const float *ct = corners + ((size_t)b * K + t) * 24, *cj = corners + ((size_t)b * K + j) * 24;
    float mn_t[3], mx_t[3], mn_j[3], mx_j[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        mn_t[a] = mx_t[a] = ct[a]; mn_j[a] = mx_j[a] = cj[a];
        for (int c = 1; c < 8; c++) {
            mn_t[a] = fminf(mn_t[a], ct[c * 3 + a]); mx_t[a] = fmaxf(mx_t[a], ct[c * 3 + a]);
            mn_j[a] = fminf(mn_j[a], cj[c * 3 + a]); mx_j[a] = fmaxf(mx_j[a], cj[c * 3 + a]);
        }
    }
    const float cen[3] = {(mn_j[0] + mx_j[0]) / 2, (mn_j[1] + mx_j[1]) / 2, (mn_j[2] + mx_j[2]) / 2};
    float d;
    if (center_mode) {
        const float tc[3] = {(mn_t[0] + mx_t[0]) / 2, (mn_t[1] + mx_t[1]) / 2, (mn_t[2] + mx_t[2]) / 2};
        const float dx = tc[0] - cen[0], dy = tc[1] - cen[1], dz = tc[2] - cen[2];
        d = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-8f));
    } else {
        d = INFINITY;
        for (int c = 0; c < 8; c++) {
            const float dx = ct[c * 3] - cen[0], dy = ct[c * 3 + 1] - cen[1], dz = ct[c * 3 + 2] - cen[2];
            d = fminf(d, sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-8f)));
        }
    }
    if (masks[(size_t)b * K + j] == 0.f) d = 1e30f;
    // AABB IoU (fp32, numpy evaluation order)
    const float ix = fmaxf(fminf(mx_t[0], mx_j[0]) - fmaxf(mn_t[0], mn_j[0]), 0.f);
    const float iy = fmaxf(fminf(mx_t[1], mx_j[1]) - fmaxf(mn_t[1], mn_j[1]), 0.f);
    const float iz = fmaxf(fminf(mx_t[2], mx_j[2]) - fmaxf(mn_t[2], mn_j[2]), 0.f);
    const float inter = __fmul_rn(__fmul_rn(ix, iy), iz);
    const float v1 = __fmul_rn(__fmul_rn(mx_t[0] - mn_t[0], mx_t[1] - mn_t[1]), mx_t[2] - mn_t[2]);
    const float v2 = __fmul_rn(__fmul_rn(mx_j[0] - mn_j[0], mx_j[1] - mn_j[1]), mx_j[2] - mn_j[2]);
    const float iou = __fdiv_rn(inter, __fadd_rn(__fsub_rn(__fadd_rn(v1, v2), inter), 1e-8f));
    if (iou >= thr) d = 1e30f;
    if (j == t) d = include_self ? 0.f : 1e30f;
    dist[e] = d;
}

extern "C" int d3_query_locals_dist(const float *corners, const float *masks, float *dist, int B, int K,
                                    int include_self, float overlay_threshold, int center_mode, void *stream) {
    D3_CLEAR();
    const long long total = (long long)B * K * K;
    if (total <= 0) return 0;
    query_locals_dist_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(corners, masks, dist, B, K,
                                                                                     include_self, overlay_threshold,
                                                                                     center_mode);
    D3_LAUNCH_CHECK();
    return 0;
}

// mask[row][j] = 1 for the L smallest entries of dist[row][0..K), 0 elsewhere -- `torch.topk(dist, L, largest=False)` followed by
// a scatter of ones (graph_module.py:218-227): the library takes everything below the L-th value and fills up with entries
// equal to it in ascending index order, i.e. the order (value, index); an entry's rank in that order is counted directly
// (K <= 1024: K^2 comparisons per row out of LDS).  One wave per row; replaces the top-k gather, its sort, a fill and a scatter.
__global__ __launch_bounds__(256) void query_locals_mask_kernel(const float *__restrict__ dist, float *__restrict__ mask, int rows,
                                                                int K, int L) {
    extern __shared__ float qm_sm[];      // 4 waves x K
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float *d = qm_sm + wave * K;
    if (row < rows)
        for (int j = lane; j < K; j += 64) d[j] = dist[(size_t)row * K + j];
    __syncthreads();
    if (row >= rows) return;
    for (int j0 = lane; j0 < K; j0 += 64) {
        const float v = d[j0];
        int rank = 0;
        for (int j = 0; j < K; j++) {
            const float o = d[j];
            rank += (o < v || (o == v && j < j0)) ? 1 : 0;
        }
        mask[(size_t)row * K + j0] = rank < L ? 1.f : 0.f;
    }
}
extern "C" int d3_query_locals_mask(const float *dist, float *mask, int rows, int K, int L, void *stream) {
    D3_CLEAR();
    if (rows <= 0) return 0;
    if (K < 1 || K > 4096 || L < 0) return D3_ERR_ARG;
    query_locals_mask_kernel<<<(rows + 3) / 4, 256, (size_t)4 * K * sizeof(float), d3_stream(stream)>>>(dist, mask, rows, K, L);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- captioner inputs
// Reference: model/caption_module.py:416-508 (`select_target`: per description the proposal with the best IoU against the
// referred box), :530-560 (the target's feature row, its local-context mask), :866-885 (`_add_relation_feat`: the target's L edge
// features added onto its L adjacency-row neighbours in ascending slot order).  The reference replicates every per-scene
// tensor once per description (`repeat`) and gathers from the copies: (N,K,L,F) = 42 MB for the edge features of the
// benchmark step, plus a masked_scatter (scan) and the matching backward passes -- ~60 library launches.  Here every description
// n reads its scene b = n / per_scene directly.
//
// select: one wave per description; IoU in fp32 with every operation rounded as the library-op chain does
// (min / max over the 8 corners, clamp(min(mx) - max(mn), 0), the library's three-element product order (x*z)*y -- its
// reduction combines lanes 0 and 2 first --, inter / ((v1 + v2) - inter + 1e-8)); first maximum wins.
__global__ __launch_bounds__(256) void cap_select_kernel(const float *__restrict__ corners, const float *__restrict__ ref_corners,
                                                         const float *__restrict__ ref_labels, int N, int per_scene, int K, int G,
                                                         long long *__restrict__ target_ids, float *__restrict__ target_ious,
                                                         long long *__restrict__ labels) {
    const int n = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int b = n / per_scene;
    const float *rc = ref_corners + (size_t)n * 24;
    float mn2[3], mx2[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        mn2[a] = mx2[a] = rc[a];
        for (int c = 1; c < 8; c++) { mn2[a] = fminf(mn2[a], rc[c * 3 + a]); mx2[a] = fmaxf(mx2[a], rc[c * 3 + a]); }
    }
    const float v2 = __fmul_rn(__fmul_rn(__fsub_rn(mx2[0], mn2[0]), __fsub_rn(mx2[2], mn2[2])), __fsub_rn(mx2[1], mn2[1]));
    float best = -INFINITY;
    int bk = 0x7FFFFFFF;
    for (int k = lane; k < K; k += 64) {
        const float *ck = corners + ((size_t)b * K + k) * 24;
        float mn1[3], mx1[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn1[a] = mx1[a] = ck[a];
            for (int c = 1; c < 8; c++) { mn1[a] = fminf(mn1[a], ck[c * 3 + a]); mx1[a] = fmaxf(mx1[a], ck[c * 3 + a]); }
        }
        float d[3];
#pragma unroll
        for (int a = 0; a < 3; a++) d[a] = fmaxf(__fsub_rn(fminf(mx1[a], mx2[a]), fmaxf(mn1[a], mn2[a])), 0.f);
        const float inter = __fmul_rn(__fmul_rn(d[0], d[2]), d[1]);
        const float v1 = __fmul_rn(__fmul_rn(__fsub_rn(mx1[0], mn1[0]), __fsub_rn(mx1[2], mn1[2])), __fsub_rn(mx1[1], mn1[1]));
        const float iou = __fdiv_rn(inter, __fadd_rn(__fsub_rn(__fadd_rn(v1, v2), inter), 1e-8f));
        if (iou > best) { best = iou; bk = k; }        // ascending k per lane: strict '>' keeps the first
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int ok = __shfl_xor(bk, o);
        if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
    }
    float lb = -INFINITY;
    int lk = 0x7FFFFFFF;
    for (int g = lane; g < G; g += 64) { const float v = ref_labels[(size_t)n * G + g]; if (v > lb) { lb = v; lk = g; } }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(lb, o);
        const int ok = __shfl_xor(lk, o);
        if (ob > lb || (ob == lb && ok < lk)) { lb = ob; lk = ok; }
    }
    if (lane == 0) { target_ids[n] = bk == 0x7FFFFFFF ? 0 : bk; target_ious[n] = best; labels[n] = lk == 0x7FFFFFFF ? 0 : lk; }
}
extern "C" int d3_caption_select_target(const float *corners, const float *ref_corners, const float *ref_labels, int N, int per_scene,
                                        int K, int G, long long *target_ids, float *target_ious, long long *labels, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (per_scene < 1 || K < 1 || G < 1) return D3_ERR_ARG;
    cap_select_kernel<<<(N + 3) / 4, 256, 0, d3_stream(stream)>>>(corners, ref_corners, ref_labels, N, per_scene, K, G, target_ids,
                                                                 target_ious, labels);
    D3_LAUNCH_CHECK();
    return 0;
}

// inputs: one workgroup per description.  nbr[n][j] = slot of the j-th one in the target's adjacency row (-1 past the end),
// obj[n][k] = base[b][k] (+ edge[b][t][j] when k = nbr[n][j], if `edge`), target_feats[n] = base[b][t],
// valid[n][k] = locals[b][t][k] (the local-context mask row; NULL: skipped)
__global__ __launch_bounds__(256) void cap_inputs_fwd_kernel(const float *__restrict__ base, const float *__restrict__ edge,
                                                             const float *__restrict__ adj, const float *__restrict__ locals,
                                                             const long long *__restrict__ target_ids, int per_scene, int K, int L,
                                                             int F, float *__restrict__ obj, float *__restrict__ target_feats,
                                                             float *__restrict__ valid, int *__restrict__ nbr) {
    extern __shared__ int slot_of[];          // K: rank of slot k among the ones of the row, or -1
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int b = n / per_scene;
    long long tg = target_ids[n];
    tg = tg < 0 ? 0 : (tg >= K ? K - 1 : tg);
    if (t < 64) {
        int base_cnt = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const bool on = edge != nullptr && k < K && adj[((size_t)b * K + tg) * K + k] == 1.f;
            const unsigned long long bal = __ballot(on);
            const int rank = base_cnt + (int)__popcll(bal & ((1ull << lane) - 1ull));
            if (k < K) slot_of[k] = (on && rank < L) ? rank : -1;
            if (on && rank < L) nbr[(size_t)n * L + rank] = k;
            base_cnt += (int)__popcll(bal);
        }
        for (int j = base_cnt + lane; j < L; j += 64) nbr[(size_t)n * L + j] = -1;
    }
    __syncthreads();
    const int f4 = F >> 2;
    for (int i = t; i < K * f4; i += blockDim.x) {
        const int k = i / f4, c = (i - k * f4) * 4;
        float4 v = *(const float4 *)(base + ((size_t)b * K + k) * F + c);
        const int j = slot_of[k];
        if (j >= 0) {
            const float4 e = *(const float4 *)(edge + (((size_t)b * K + tg) * L + j) * F + c);
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        *(float4 *)(obj + ((size_t)n * K + k) * F + c) = v;
    }
    for (int c = t; c < F; c += blockDim.x) target_feats[(size_t)n * F + c] = base[((size_t)b * K + tg) * F + c];
    if (valid != nullptr)
        for (int k = t; k < K; k += blockDim.x) valid[(size_t)n * K + k] = locals[((size_t)b * K + tg) * K + k];
}
// backward, deterministic: d_base[b][k] = sum over the scene's descriptions (ascending n) of g_obj[n][k] (+ g_target[n] at its
// target slot); d_edge[b][t][j] = sum over the descriptions with target t of g_obj[n][nbr[n][j]]; rows of other slots stay 0
// (d_edge is zero-filled by the caller).
__global__ __launch_bounds__(256) void cap_inputs_bwd_kernel(const float *__restrict__ g_obj, const float *__restrict__ g_target,
                                                             const long long *__restrict__ target_ids, const int *__restrict__ nbr,
                                                             int per_scene, int K, int L, int F, float *__restrict__ d_base,
                                                             float *__restrict__ d_edge) {
    const int b = blockIdx.y, t = threadIdx.x;
    const int f4 = F >> 2;
    const int i = blockIdx.x * blockDim.x + t;
    if (i < K * f4) {
        const int k = i / f4, c = (i - k * f4) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < per_scene; q++) {
            const int n = b * per_scene + q;
            const float4 g = *(const float4 *)(g_obj + ((size_t)n * K + k) * F + c);
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            if (g_target != nullptr && target_ids[n] == k) {
                const float4 h = *(const float4 *)(g_target + (size_t)n * F + c);
                s.x += h.x; s.y += h.y; s.z += h.z; s.w += h.w;
            }
        }
        *(float4 *)(d_base + ((size_t)b * K + k) * F + c) = s;
    }
    if (d_edge != nullptr && i < per_scene * L * f4) {     // the first description with a given target owns that target's rows
        const int q0 = i / (L * f4), r = i - q0 * (L * f4), j = r / f4, c = (r - j * f4) * 4;
        const int n0 = b * per_scene + q0;
        const long long tg = target_ids[n0];
        bool first = true;
        for (int q = 0; q < q0; q++) first = first && target_ids[b * per_scene + q] != tg;
        if (first && tg >= 0 && tg < K) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = q0; q < per_scene; q++) {
                const int n = b * per_scene + q;
                if (target_ids[n] != tg) continue;
                const int k = nbr[(size_t)n * L + j];
                if (k < 0) continue;
                const float4 g = *(const float4 *)(g_obj + ((size_t)n * K + k) * F + c);
                s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            }
            *(float4 *)(d_edge + (((size_t)b * K + tg) * L + j) * F + c) = s;
        }
    }
}
extern "C" int d3_caption_inputs_fwd(const float *base, const float *edge, const float *adj, const float *locals,
                                     const long long *target_ids, int N, int per_scene, int K, int L, int F, float *obj,
                                     float *target_feats, float *valid, int *nbr, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (per_scene < 1 || K < 1 || L < 1 || (F & 3) || (edge != nullptr && adj == nullptr) || (valid != nullptr && locals == nullptr))
        return D3_ERR_ARG;
    cap_inputs_fwd_kernel<<<N, 256, (size_t)K * sizeof(int), d3_stream(stream)>>>(base, edge, adj, locals, target_ids, per_scene, K, L, F,
                                                                                   obj, target_feats, valid, nbr);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_caption_inputs_bwd(const float *g_obj, const float *g_target, const long long *target_ids, const int *nbr, int N,
                                     int per_scene, int K, int L, int F, float *d_base, float *d_edge, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (per_scene < 1 || N % per_scene || K < 1 || L < 1 || (F & 3)) return D3_ERR_ARG;
    const int f4 = F >> 2;
    const int work = K * f4 > per_scene * L * f4 ? K * f4 : per_scene * L * f4;
    dim3 grid((work + 255) / 256, N / per_scene);
    cap_inputs_bwd_kernel<<<grid, 256, 0, d3_stream(stream)>>>(g_obj, g_target, target_ids, nbr, per_scene, K, L, F, d_base, d_edge);
    D3_LAUNCH_CHECK();
    return 0;
}
