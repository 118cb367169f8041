// nms.hip -- the non-maximum suppressions of the evaluation paths on the device (gfx950).
//
//   * class-aware 3D box NMS of `parse_predictions` (lib/det/ap_helper.py:80-108 -> lib/det/nms.py:110-150
//     `nms_3d_faster_samecls`): per scene, K <= 256 axis-aligned proposal boxes [x1,y1,z1,x2,y2,z2,score,class]; greedy in
//     descending score, a picked box removes the boxes of ITS class whose IoU with it exceeds the threshold.  The reference
//     does it per scene in numpy on the host (float64); here one workgroup per scene, boxes and the alive bitmap in LDS,
//     float64 arithmetic like numpy, one barrier per picked box.
//   * instance-mask NMS of `PointGroup.test` (model/pointgroup.py:577-601 + lib/utils/eval.py:75-97 `get_nms_instances`):
//     the reference materialises a dense (nProposal, N) 0/1 mask matrix, multiplies it with its transpose (nProposal^2 * N
//     MACs, ~0.2 GFLOP-equivalents per 100 k points... and 400 MB at 600 proposals x 165 k points) and copies the IoU matrix
//     to the host.  A point belongs to at most one cluster of each of the two clusterings, so the intersections are counted
//     directly from the (cluster, point) lists: two membership slots per point, one atomic per shared point.
// Integer / comparison work: results are index sets, bit-exact with the host restatement (d3net_amd/evaluator.py, pinned to
// the reference's own functions by tests/golden/evaluator_golden.npz) up to the order of exactly tied scores, which numpy's
// argsort leaves unspecified.
#include "common.h"

#define NMS_MAXK 256

// boxes (B,K,8) double-convertible floats: [x1,y1,z1,x2,y2,z2,score,cls]; valid (B,K) != 0 -> pick (B,K) float 0/1
// visit: optional (B,K) int32 visiting order (candidate indices, best first, -1 padded) -- numpy's argsort leaves the order of
// exactly tied scores to its sort implementation; a caller that must reproduce it passes the order, otherwise ties go to the
// later index first
__global__ __launch_bounds__(256) void det_nms3d_kernel(const float *__restrict__ boxes, const float *__restrict__ valid,
                                                        const int *__restrict__ visit, int K, double thr, int old_type,
                                                        float *__restrict__ pick) {
    __shared__ double bx[NMS_MAXK][7];    // x1 y1 z1 x2 y2 z2 area
    __shared__ float sc[NMS_MAXK], cl[NMS_MAXK];
    __shared__ int order[NMS_MAXK], alive[NMS_MAXK], nv_s;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) nv_s = 0;
    __syncthreads();
    if (t < K) {
        const float *r = boxes + ((long long)b * K + t) * 8;
        for (int q = 0; q < 6; q++) bx[t][q] = (double)r[q];
        bx[t][6] = (bx[t][3] - bx[t][0]) * (bx[t][4] - bx[t][1]) * (bx[t][5] - bx[t][2]);
        sc[t] = r[6]; cl[t] = r[7];
        alive[t] = valid[(long long)b * K + t] != 0.f ? 1 : 0;
        pick[(long long)b * K + t] = 0.f;
    }
    __syncthreads();
    if (visit) {
        if (t < K) {
            const int v = visit[(long long)b * K + t];
            order[t] = v;
            if (v >= 0) atomicAdd(&nv_s, 1);
        }
    } else if (t < K && alive[t]) {
        // rank in descending score; exact ties: the later index first (an ascending stable sort read from its end)
        int rk = 0;
        for (int j = 0; j < K; j++)
            if (alive[j] && (sc[j] > sc[t] || (sc[j] == sc[t] && j > t))) rk++;
        order[rk] = t;
        atomicAdd(&nv_s, 1);
    }
    __syncthreads();
    const int nv = nv_s;
    for (int r = 0; r < nv; r++) {
        const int i = order[r];
        if (alive[i]) {                       // uniform: alive[] is only written behind the barrier below
            if (t == 0) pick[(long long)b * K + i] = 1.f;
            const int rr = r + 1 + t;
            bool kill = false;
            if (rr < nv) {
                const int j = order[rr];
                if (alive[j]) {
                    const double l = fmax(0., fmin(bx[i][3], bx[j][3]) - fmax(bx[i][0], bx[j][0]));
                    const double w = fmax(0., fmin(bx[i][4], bx[j][4]) - fmax(bx[i][1], bx[j][1]));
                    const double h = fmax(0., fmin(bx[i][5], bx[j][5]) - fmax(bx[i][2], bx[j][2]));
                    const double inter = l * w * h;
                    double o = old_type ? inter / bx[j][6] : inter / (bx[i][6] + bx[j][6] - inter + 1e-8);
                    if (cl[i] != cl[j]) o = 0.;
                    kill = o > thr;   // (numpy's `o > thr`: a NaN overlap -- zero-volume boxes with old_type -- keeps the box)
                    if (kill) alive[j] = 0;
                }
            }
            (void)kill;
        }
        __syncthreads();
    }
}

extern "C" int d3_nms3d_samecls(const float *boxes, const float *valid, const int *visit, int B, int K, double iou_thr, int old_type,
                                float *pick, void *stream) {
    D3_CLEAR();
    if (B < 1 || K < 1 || K > NMS_MAXK) return D3_ERR_ARG;
    det_nms3d_kernel<<<B, 256, 0, d3_stream(stream)>>>(boxes, valid, visit, K, iou_thr, old_type, pick);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ instance masks
__global__ void inm_member_kernel(const int *__restrict__ cidx, long long S, int N, int *__restrict__ member, int *__restrict__ overflow) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S) return;
    const int c = cidx[e * 2], p = cidx[e * 2 + 1];
    if (p < 0 || p >= N) return;
    if (atomicCAS(&member[(long long)p * 2], -1, c) != -1)
        if (atomicCAS(&member[(long long)p * 2 + 1], -1, c) != -1) *overflow = 1;   // a third membership: caller falls back
}
__global__ void inm_count_kernel(const int *__restrict__ member, int N, int P, float *__restrict__ inter) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int a = member[(long long)p * 2], b = member[(long long)p * 2 + 1];
    if (a >= 0 && b >= 0 && a != b) { atomicAdd(&inter[(long long)a * P + b], 1.f); atomicAdd(&inter[(long long)b * P + a], 1.f); }
}
// cross_ious = inter / (n_i + n_j - inter), diagonal inter = n_i  (float32 like the reference's torch expression)
__global__ void inm_iou_kernel(float *__restrict__ inter, const int *__restrict__ offsets, int P) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)P * P) return;
    const int i = (int)(e / P), j = (int)(e - (long long)i * P);
    const float ni = (float)(offsets[i + 1] - offsets[i]), nj = (float)(offsets[j + 1] - offsets[j]);
    const float in = i == j ? ni : inter[e];
    inter[e] = in / (ni + nj - in);
}

// cluster_idxs (S,2) [cluster, point], offsets (P+1), N points -> ious (P,P) float32.  member: 2*N ints of scratch.
// *overflow_host != 0: some point sits in more than two clusters (not producible by PointGroup's two clusterings).
extern "C" int d3_instance_cross_iou(const int *cluster_idxs, const int *offsets, long long S, int P, int N, float *ious, int *member,
                                     int *overflow_dev, void *stream) {
    D3_CLEAR();
    if (P < 1 || N < 1) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    D3_CHECK(hipMemsetAsync(member, 0xFF, (size_t)N * 2 * sizeof(int), s));
    D3_CHECK(hipMemsetAsync(ious, 0, (size_t)P * P * sizeof(float), s));
    D3_CHECK(hipMemsetAsync(overflow_dev, 0, sizeof(int), s));
    if (S > 0) inm_member_kernel<<<(int)((S + 255) / 256), 256, 0, s>>>(cluster_idxs, S, N, member, overflow_dev);
    inm_count_kernel<<<(N + 255) / 256, 256, 0, s>>>(member, N, P, ious);
    const long long pp = (long long)P * P;
    inm_iou_kernel<<<(int)((pp + 255) / 256), 256, 0, s>>>(ious, offsets, P);
    D3_LAUNCH_CHECK();
    return 0;
}

// greedy NMS over a dense IoU matrix (lib/utils/eval.py:75-97): candidates = entries with keep[i] != 0, visited in descending
// score (ties: lower index first, a stable argsort of -score); picked[] receives their indices in pick order, *npicked the count
#define NMSM_T 1024
__global__ __launch_bounds__(NMSM_T) void nms_matrix_kernel(const float *__restrict__ ious, const float *__restrict__ scores,
                                                            const unsigned char *__restrict__ keep, int n, float thr,
                                                            int *__restrict__ order, int *__restrict__ picked, int *__restrict__ npicked) {
    extern __shared__ int alive[];     // n
    __shared__ int nv_s, np_s;
    const int t = threadIdx.x;
    if (t == 0) { nv_s = 0; np_s = 0; }
    __syncthreads();
    for (int i = t; i < n; i += NMSM_T) alive[i] = keep[i] ? 1 : 0;
    __syncthreads();
    for (int i = t; i < n; i += NMSM_T) {
        if (!alive[i]) continue;
        int rk = 0;
        const float si = scores[i];
        for (int j = 0; j < n; j++)
            if (alive[j] && (scores[j] > si || (scores[j] == si && j < i))) rk++;
        order[rk] = i;
        atomicAdd(&nv_s, 1);
    }
    __syncthreads();
    const int nv = nv_s;
    for (int r = 0; r < nv; r++) {
        const int i = order[r];
        if (alive[i]) {
            if (t == 0) picked[np_s++] = i;
            for (int rr = r + 1 + t; rr < nv; rr += NMSM_T) {
                const int j = order[rr];
                if (alive[j] && ious[(long long)i * n + j] > thr) alive[j] = 0;
            }
        }
        __syncthreads();
    }
    if (t == 0) *npicked = np_s;
}

extern "C" int d3_nms_matrix(const float *ious, const float *scores, const unsigned char *keep, int n, float thr, int *order_scratch,
                             int *picked, int *npicked, void *stream) {
    D3_CLEAR();
    if (n < 1 || n > 12288) return D3_ERR_ARG;
    nms_matrix_kernel<<<1, NMSM_T, (size_t)n * sizeof(int), d3_stream(stream)>>>(ious, scores, keep, n, thr, order_scratch, picked, npicked);
    D3_LAUNCH_CHECK();
    return 0;
}
