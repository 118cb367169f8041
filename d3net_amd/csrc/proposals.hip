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
