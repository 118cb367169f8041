// heads.hip -- point-level heads of the PointGroup detector (gfx950): the pieces of `sem_seg`, `offset_net` and the
// semantic loss (reference: model/pointgroup.py:77-85, 274-279, 387-395) whose library kernels collapse at
// N = 165k rows x 16..20 channels:
//   * weight gradient of a tall-skinny nn.Linear, dW (O,I) = dy^T x with a 165k-long reduction: the BLAS library
//     picks a single-workgroup kernel (0.25-0.5 ms per layer, profiles/r01_i); here 512 workgroups reduce row ranges
//     and a second kernel adds the partials in fixed order (deterministic), bias gradient included;
//   * softmax cross-entropy with ignore_index over (N, 20) logits: forward loss and the gradient softmax - onehot in
//     one pass (the library's log_softmax + nll_loss forward/backward are four passes).
// HBM bound: bytes = 4*N*(I+O) (wgrad), 8*N*C (cross entropy: logits in, gradient out).
#include "common.h"

#define TW_ROWS 64
#define TW_GRID 512
// part[block][O][I+1]: columns 0..I-1 = dW, column I = bias gradient
__global__ __launch_bounds__(256) void tall_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                        float *__restrict__ part, int N, int I, int O) {
    __shared__ float xs[TW_ROWS * 33], ds[TW_ROWS * 33];
    const int t = threadIdx.x;
    const int I1 = I + 1, npair = O * I1;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};    // O, I <= 32: at most 32*33/256 = 4.1 pairs per thread
    const int per = ((N + gridDim.x - 1) / gridDim.x + TW_ROWS - 1) / TW_ROWS * TW_ROWS;
    const int r0 = blockIdx.x * per, r1 = min(N, r0 + per);
    for (int rb = r0; rb < r1; rb += TW_ROWS) {
        const int rows = min(TW_ROWS, r1 - rb);
        for (int e = t; e < rows * I; e += 256) { const int r = e / I, c = e - r * I; xs[r * 33 + c] = x[(long long)rb * I + e]; }
        for (int e = t; e < rows * O; e += 256) { const int r = e / O, c = e - r * O; ds[r * 33 + c] = dy[(long long)rb * O + e]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const int pr = t + q * 256;
            if (pr < npair) {
                const int o = pr / I1, i = pr - o * I1;
                float a = acc[q];
                if (i < I) { for (int r = 0; r < rows; r++) a = fmaf(ds[r * 33 + o], xs[r * 33 + i], a); }
                else { for (int r = 0; r < rows; r++) a += ds[r * 33 + o]; }
                acc[q] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int pr = t + q * 256;
        if (pr < npair) part[(long long)blockIdx.x * npair + pr] = acc[q];
    }
}
// The same partial sums on v_mfma_f32_16x16x4_f32 (exact fp32 products): the scalar kernel above reads LDS twice per FMA and
// runs at 1.1 TB/s (96 us at 746 k rows x (16 + 20) channels); here a wave owns a slice of its workgroup's rows and feeds the
// MFMA straight from memory -- A[o][r] = dy[r][o] and B[r][i] = x[r][i] are both "16 consecutive channels of one row" per
// 16-lane group -- the bias gradient rides along as a constant-one column of x; the four waves are summed through LDS.
// part layout as above.  I <= 31 (one column is the bias), O <= 32.
typedef float tw_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void tall_wgrad_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                             float *__restrict__ part, int N, int I, int O) {
    __shared__ float red[4][2][2][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int I1 = I + 1, npair = O * I1;
    const int per = ((N + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
    const int r0 = blockIdx.x * per, r1 = min(N, r0 + per);
    tw_f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (tw_f32x4){0.f, 0.f, 0.f, 0.f};
    const int ot = (O + 15) >> 4, it = (I1 + 15) >> 4;
    for (int rb = r0 + wave * 16; rb < r1; rb += 64) {           // 16 rows per wave and step
        float av[2][4], bv[2][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int r = rb + g * 4 + c;
            const bool ok = r < r1;
#pragma unroll
            for (int a = 0; a < 2; a++) { const int o = a * 16 + i; av[a][c] = (ok && o < O) ? dy[(long long)r * O + o] : 0.f; }
#pragma unroll
            for (int b = 0; b < 2; b++) { const int col = b * 16 + i; bv[b][c] = !ok ? 0.f : (col < I ? x[(long long)r * I + col] : (col == I ? 1.f : 0.f)); }
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            if (a >= ot) break;
#pragma unroll
            for (int b = 0; b < 2; b++) {
                if (b >= it) break;
#pragma unroll
                for (int c = 0; c < 4; c++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[wave][a][b][q * 64 + lane] = acc[a][b][q];
    __syncthreads();
    // D layout: row (o) = g*4 + q, col (i) = lane & 15
    for (int e = t; e < 2 * 2 * 256; e += 256) {
        const int a = e >> 9, b = (e >> 8) & 1, q = (e >> 6) & 3, ln = e & 63;
        const int o = a * 16 + (ln >> 4) * 4 + q, col = b * 16 + (ln & 15);
        if (o < O && col < I1) part[(long long)blockIdx.x * npair + o * I1 + col] = (red[0][a][b][q * 64 + ln] + red[1][a][b][q * 64 + ln]) + (red[2][a][b][q * 64 + ln] + red[3][a][b][q * 64 + ln]);
    }
}
// one wave per (o, i) pair; lanes stride over the workgroup partials, fp64 wave reduction in fixed order
__global__ void tall_wgrad_reduce_kernel(const float *__restrict__ part, int nblocks, int I, int O, float *dW, float *db) {
    const int pr = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int I1 = I + 1, npair = O * I1;
    if (pr >= npair) return;
    double s = 0.;
    for (int b = lane; b < nblocks; b += 64) s += (double)part[(long long)b * npair + pr];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    const int o = pr / I1, i = pr - o * I1;
    if (i < I) dW[o * I + i] = (float)s;
    else if (db) db[o] = (float)s;
}

extern "C" size_t d3_tall_wgrad_ws_bytes(int I, int O) { return (size_t)TW_GRID * O * (I + 1) * sizeof(float); }

// dW (O,I) = dy (N,O)^T x (N,I), db (O) = column sums of dy (db may be NULL); I, O <= 32
extern "C" int d3_tall_wgrad(const float *x, const float *dy, float *dW, float *db, int N, int I, int O, void *ws,
                             size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (I < 1 || O < 1 || I > 32 || O > 32) return D3_ERR_ARG;
    if (ws_bytes < d3_tall_wgrad_ws_bytes(I, O)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int grid = TW_GRID;
    if (N < grid * TW_ROWS) grid = (N + TW_ROWS - 1) / TW_ROWS;
    if (grid < 1) grid = 1;
    if (I <= 31) tall_wgrad_mfma_kernel<<<grid, 256, 0, s>>>(x, dy, (float *)ws, N, I, O);
    else tall_wgrad_kernel<<<grid, 256, 0, s>>>(x, dy, (float *)ws, N, I, O);
    const int npair = O * (I + 1);
    tall_wgrad_reduce_kernel<<<(npair + 3) / 4, 256, 0, s>>>((const float *)ws, grid, I, O, dW, db);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ point heads, forward
// The two point-level heads of PointGroup on the (N, m = 16) backbone output (reference model/pointgroup.py:77-85, 277-283):
//   semantic_scores = sem_seg(x) (m -> C <= 32), semantic_preds = row arg-max, and
//   pt_offsets = Linear(m,3)(ReLU(BatchNorm1d(Linear(m,m)(x)))).
// As library calls: three GEMMs with K = 16 (77 / 40 / 41 us at 746 k rows: the BLAS picks generic tiles), a 76 us row-max
// reduction, batch-norm statistics + transform + clamp -- 440 us for ~340 MB of compulsory traffic.  Here:
//   pth_fwd_kernel : x read ONCE; a wave owns 16 rows per step: the three 16x16 output tiles (scores 0..15, scores 16..31,
//                    hidden) as 12 v_mfma_f32_16x16x4f32 (exact fp32 products) from one float4 of x per lane with the weights
//                    resident in registers; row arg-max across the 16-lane groups (first maximum); per-channel sum / sum of
//                    squares of the hidden layer for the batch norm (per-workgroup partials, fixed order);
//   pth_bn_kernel  : statistics -> mean / 1/sqrt(var + eps) (fp64, fixed order) + the running-statistics update of
//                    nn.BatchNorm1d (or, in eval mode, the running statistics themselves);
//   pth_out_kernel : y = ReLU(BN(h)) (kept for the backward) and the (N,3) offsets, one thread per row.
typedef float ph_f32x4 __attribute__((ext_vector_type(4)));
#define PH_GRID 1024
__global__ __launch_bounds__(256) void pth_fwd_kernel(const float *__restrict__ x, const float *__restrict__ Ws, const float *__restrict__ bs,
                                                     const float *__restrict__ W0, const float *__restrict__ b0, int N, int C,
                                                     float *__restrict__ scores, long long *__restrict__ preds, float *__restrict__ h,
                                                     float *__restrict__ part) {
    __shared__ float red[4][2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, q = lane >> 4;
    float ws0[4], ws1[4], wh[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int k = 4 * q + c;
        ws0[c] = j < C ? Ws[j * 16 + k] : 0.f;
        ws1[c] = 16 + j < C ? Ws[(16 + j) * 16 + k] : 0.f;
        wh[c] = W0[j * 16 + k];
    }
    const float bias0 = j < C ? bs[j] : 0.f, bias1 = 16 + j < C ? bs[16 + j] : 0.f, biash = b0[j];
    float s1 = 0.f, s2 = 0.f;
    const long long ntiles = ((long long)N + 15) >> 4;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long row0 = tile << 4;
        const long long ra = row0 + j;
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ra < N) xa = *(const float4 *)(x + ra * 16 + 4 * q);
        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
        ph_f32x4 a0 = {bias0, bias0, bias0, bias0}, a1 = {bias1, bias1, bias1, bias1}, ah = {biash, biash, biash, biash};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c], ws0[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c], ws1[c], a1, 0, 0, 0);
            ah = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c], wh[c], ah, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long row = row0 + 4 * q + r;
            const bool ok = row < N;
            if (ok) {
                if (j < C) scores[row * C + j] = a0[r];
                if (16 + j < C) scores[row * C + 16 + j] = a1[r];
                h[row * 16 + j] = ah[r];
                s1 += ah[r]; s2 += ah[r] * ah[r];
            }
            float bv = j < C ? a0[r] : -INFINITY;
            int bi = j;
            if (16 + j < C && a1[r] > bv) { bv = a1[r]; bi = 16 + j; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (ok && j == 0) preds[row] = bi;
        }
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lane < 16) { red[wave][0][lane] = s1; red[wave][1][lane] = s2; }
    __syncthreads();
    if (t < 32) {
        const int w = t >> 4, c = t & 15;
        part[(long long)blockIdx.x * 32 + w * 16 + c] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
    }
}
// stat[0..15] = mean, stat[16..31] = 1 / sqrt(var + eps)
__global__ __launch_bounds__(1024) void pth_bn_kernel(const float *__restrict__ part, int nblocks, long long N, float eps, float momentum,
                                                     int training, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                     long long *__restrict__ tracked, float *__restrict__ stat) {
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;     // one wave per channel
    if (!training) {
        if (lane == 0) { stat[c] = running_mean[c]; stat[16 + c] = 1.f / sqrtf(running_var[c] + eps); }
        return;
    }
    double sa = 0., sb = 0.;
    for (int b0 = lane; b0 < nblocks; b0 += 256) {       // four partial rows per lane in flight; same order every run
        float pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = b0 + u * 64;
            const long long o = (long long)(b < nblocks ? b : 0) * 32;
            pa[u] = part[o + c]; pb[u] = part[o + 16 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (b0 + u * 64 < nblocks) { sa += (double)pa[u]; sb += (double)pb[u]; }
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if (lane != 0) return;
    const double m = sa / (double)N;
    double v = sb / (double)N - m * m;
    if (v < 0.) v = 0.;
    stat[c] = (float)m; stat[16 + c] = (float)(1.0 / sqrt(v + (double)eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * ((double)N / (double)(N > 1 ? N - 1 : 1)));
    }
    if (c == 0 && tracked) tracked[0] += 1;
}
__global__ __launch_bounds__(256) void pth_out_kernel(const float *__restrict__ h, const float *__restrict__ stat,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     const float *__restrict__ W3, const float *__restrict__ b3, long long N,
                                                     float *__restrict__ y, float *__restrict__ off) {
    __shared__ float sm[16 * 4 + 48 + 3];
    float *mean = sm, *inv = sm + 16, *g = sm + 32, *bt = sm + 48, *w = sm + 64, *bb = sm + 112;
    const int t = threadIdx.x;
    if (t < 16) { mean[t] = stat[t]; inv[t] = stat[16 + t]; g[t] = gamma[t]; bt[t] = beta[t]; }
    if (t < 48) w[t] = W3[t];
    if (t < 3) bb[t] = b3[t];
    __syncthreads();
    const long long r = (long long)blockIdx.x * blockDim.x + t;
    if (r >= N) return;
    float o0 = bb[0], o1 = bb[1], o2 = bb[2];
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
        const float4 v = *(const float4 *)(h + r * 16 + c4 * 4);
        const float in[4] = {v.x, v.y, v.z, v.w};
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = c4 * 4 + e;
            const float z = fmaf((in[e] - mean[c]) * inv[c], g[c], bt[c]);
            out[e] = z > 0.f ? z : 0.f;
            o0 = fmaf(out[e], w[c], o0); o1 = fmaf(out[e], w[16 + c], o1); o2 = fmaf(out[e], w[32 + c], o2);
        }
        *(float4 *)(y + r * 16 + c4 * 4) = make_float4(out[0], out[1], out[2], out[3]);
    }
    off[r * 3] = o0; off[r * 3 + 1] = o1; off[r * 3 + 2] = o2;
}
// backward pieces of the point heads.  dy = (g_off W3) * (y > 0): the data gradient of the last Linear with the ReLU mask, one
// thread per row (the library: a K = 3 GEMM, a compare and a multiply).
__global__ __launch_bounds__(256) void pth_dy_kernel(const float *__restrict__ g_off, const float *__restrict__ W3,
                                                    const float *__restrict__ y, long long N, float *__restrict__ dy) {
    __shared__ float w[48];
    if (threadIdx.x < 48) w[threadIdx.x] = W3[threadIdx.x];
    __syncthreads();
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const float g0 = g_off[r * 3], g1 = g_off[r * 3 + 1], g2 = g_off[r * 3 + 2];
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
        const float4 yv = *(const float4 *)(y + r * 16 + c4 * 4);
        const float ys[4] = {yv.x, yv.y, yv.z, yv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = c4 * 4 + e;
            const float v = fmaf(g2, w[32 + c], fmaf(g1, w[16 + c], g0 * w[c]));
            o[e] = ys[e] > 0.f ? v : 0.f;
        }
        *(float4 *)(dy + r * 16 + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// dx = dh W0 + g_s Ws  (N,16): both data gradients that reach the backbone output in one pass.  Computed transposed on
// v_mfma_f32_16x16x4f32 -- A = the weights (constant registers; M = input channel), B = 16 rows of dh / g_s as one float4 per
// lane (N = row) -- so that a lane ends with four consecutive channels of one row: one float4 store.
__global__ __launch_bounds__(256) void pth_dx_kernel(const float *__restrict__ dh, const float *__restrict__ W0,
                                                    const float *__restrict__ gs, const float *__restrict__ Ws, long long N, int C,
                                                    float *__restrict__ dx) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, kq = lane >> 4;
    float a0[4], a1[4], a2[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int k = 4 * kq + c;
        a0[c] = W0[k * 16 + i];
        a1[c] = k < C ? Ws[k * 16 + i] : 0.f;
        a2[c] = 16 + k < C ? Ws[(16 + k) * 16 + i] : 0.f;
    }
    const bool vec = (C & 3) == 0;
    const long long ntiles = (N + 15) >> 4;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long r = (tile << 4) + i;
        const bool ok = r < N;
        float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            if (dh != nullptr) { const float4 v = *(const float4 *)(dh + r * 16 + 4 * kq); b0[0] = v.x; b0[1] = v.y; b0[2] = v.z; b0[3] = v.w; }
            if (gs != nullptr) {
                if (vec) {
                    if (4 * kq + 3 < C) { const float4 v = *(const float4 *)(gs + r * C + 4 * kq); b1[0] = v.x; b1[1] = v.y; b1[2] = v.z; b1[3] = v.w; }
                    if (16 + 4 * kq + 3 < C) { const float4 v = *(const float4 *)(gs + r * C + 16 + 4 * kq); b2[0] = v.x; b2[1] = v.y; b2[2] = v.z; b2[3] = v.w; }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int k = 4 * kq + c;
                        if (k < C) b1[c] = gs[r * C + k];
                        if (16 + k < C) b2[c] = gs[r * C + 16 + k];
                    }
                }
            }
        }
        ph_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b0[c], acc, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b1[c], acc, 0, 0, 0);
        if (C > 16) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[c], b2[c], acc, 0, 0, 0);
        }
        // D[m = 4*kq + q][n = i]: channels 4*kq .. 4*kq+3 of row i
        if (ok) *(float4 *)(dx + r * 16 + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
extern "C" int d3_point_heads_dy(const float *g_off, const float *W3, const float *y, long long N, float *dy, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    pth_dy_kernel<<<(int)((N + 255) / 256), 256, 0, d3_stream(stream)>>>(g_off, W3, y, N, dy);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_point_heads_dx(const float *dh, const float *W0, const float *g_scores, const float *Ws, long long N, int C, float *dx,
                                 void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (C < 1 || C > 32 || (dh == nullptr && g_scores == nullptr)) return D3_ERR_ARG;
    const long long ntiles = (N + 15) >> 4;
    const int grid = (int)((ntiles + 3) / 4 < 2048 ? (ntiles + 3) / 4 : 2048);
    pth_dx_kernel<<<grid, 256, 0, d3_stream(stream)>>>(dh, W0, g_scores, Ws, N, C, dx);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" size_t d3_point_heads_ws_bytes(void) { return (size_t)PH_GRID * 32 * sizeof(float); }
extern "C" int d3_point_heads_fwd(const float *x, long long N, int m, int C, const float *Ws, const float *bs, const float *W0,
                                  const float *b0, const float *gamma, const float *beta, const float *W3, const float *b3, float eps,
                                  float momentum, int training, float *running_mean, float *running_var, long long *num_batches_tracked,
                                  float *scores, long long *preds, float *h, float *y, float *offsets, float *stat, void *ws,
                                  size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (m != 16 || C < 1 || C > 32) return D3_ERR_ARG;
    if (ws == nullptr || ws_bytes < d3_point_heads_ws_bytes()) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const long long ntiles = (N + 15) >> 4;
    int grid = (int)((ntiles + 3) / 4 < PH_GRID ? (ntiles + 3) / 4 : PH_GRID);
    pth_fwd_kernel<<<grid, 256, 0, s>>>(x, Ws, bs, W0, b0, (int)N, C, scores, preds, h, (float *)ws);
    pth_bn_kernel<<<1, 1024, 0, s>>>((const float *)ws, grid, N, eps, momentum, training, running_mean, running_var, num_batches_tracked,
                                    stat);
    pth_out_kernel<<<(int)((N + 255) / 256), 256, 0, s>>>(h, stat, gamma, beta, W3, b3, N, y, offsets);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ softmax cross entropy
// per row: loss = logsumexp(z) - z[label] (label != ignore), grad = softmax(z) - onehot (0 for ignored rows).
// part[block][2] = (sum of losses, number of counted rows) in fp32; the reduce kernel produces loss_sum, count.
#define CE_GRID 1024
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float *__restrict__ z, const long long *__restrict__ label,
                                                    float *__restrict__ grad, float *__restrict__ part, int N, int C,
                                                    int ignore) {
    __shared__ float s1[256], s2[256];
    float ls = 0.f, cnt = 0.f;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (long long)gridDim.x * blockDim.x) {
        const float *zr = z + r * C;
        float *gr = grad + r * C;
        const long long lb = label[r];
        float m = -INFINITY;
        for (int c = 0; c < C; c++) m = fmaxf(m, zr[c]);
        float se = 0.f;
        for (int c = 0; c < C; c++) se += expf(zr[c] - m);
        const float lse = m + logf(se);
        if (lb == ignore || lb < 0 || lb >= C) {
            for (int c = 0; c < C; c++) gr[c] = 0.f;
        } else {
            const float inv = 1.f / se;
            for (int c = 0; c < C; c++) gr[c] = expf(zr[c] - m) * inv - (c == lb ? 1.f : 0.f);
            ls += lse - zr[lb]; cnt += 1.f;
        }
    }
    s1[threadIdx.x] = ls; s2[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = s1[0]; part[blockIdx.x * 2 + 1] = s2[0]; }
}
// Same arithmetic, rows staged through LDS: a thread walking its own 80-byte row reads and writes 64 rows x 4 bytes per
// wave instruction (0.7 TB/s on the 746 k x 20 logits of the 4-scene step: 170 us); tiles of 256 rows are loaded and stored as
// contiguous float streams and the per-row passes run on LDS (row pitch C | 1: conflict-free).  C <= 32.
__global__ __launch_bounds__(256) void ce_fwd_lds_kernel(const float *__restrict__ z, const long long *__restrict__ label,
                                                        float *__restrict__ grad, float *__restrict__ part, int N, int C,
                                                        int ignore) {
    extern __shared__ float ce_t[];          // 256 x (C | 1)
    __shared__ float s1[256], s2[256];
    const int P = C | 1, t = threadIdx.x;
    float ls = 0.f, cnt = 0.f;
    const long long ntiles = ((long long)N + 255) / 256;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long r0 = tile * 256;
        const int rows = (int)min((long long)256, (long long)N - r0);
        const float *zt = z + r0 * C;
        for (int e = t; e < rows * C; e += 256) ce_t[(e / C) * P + e % C] = zt[e];
        __syncthreads();
        if (t < rows) {
            float *zr = ce_t + t * P;
            const long long lb = label[r0 + t];
            float m = -INFINITY;
            for (int c = 0; c < C; c++) m = fmaxf(m, zr[c]);
            float se = 0.f;
            for (int c = 0; c < C; c++) se += expf(zr[c] - m);
            const float lse = m + logf(se);
            if (lb == ignore || lb < 0 || lb >= C) {
                for (int c = 0; c < C; c++) zr[c] = 0.f;
            } else {
                const float inv = 1.f / se, zl = zr[lb];
                for (int c = 0; c < C; c++) zr[c] = expf(zr[c] - m) * inv - (c == lb ? 1.f : 0.f);
                ls += lse - zl; cnt += 1.f;
            }
        }
        __syncthreads();
        float *gt = grad + r0 * C;
        for (int e = t; e < rows * C; e += 256) gt[e] = ce_t[(e / C) * P + e % C];
        __syncthreads();
    }
    s1[t] = ls; s2[t] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { s1[t] += s1[t + o]; s2[t] += s2[t + o]; }
        __syncthreads();
    }
    if (t == 0) { part[blockIdx.x * 2] = s1[0]; part[blockIdx.x * 2 + 1] = s2[0]; }
}
__global__ void ce_reduce_kernel(const float *part, int nblocks, float *out) {   // out[0] = mean loss, out[1] = count
    const int lane = threadIdx.x;
    double a = 0., b = 0.;
    for (int i = lane; i < nblocks; i += 64) { a += (double)part[i * 2]; b += (double)part[i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane == 0) { out[0] = (float)(b > 0. ? a / b : 0.); out[1] = (float)b; }
}

extern "C" size_t d3_cross_entropy_ws_bytes(void) { return (size_t)CE_GRID * 2 * sizeof(float); }

// nn.functional.cross_entropy(z, label, ignore_index) with mean reduction: out[0] = loss, out[1] = counted rows;
// grad (N,C) = d(sum of losses)/dz (the caller scales it by grad_out / count)
extern "C" int d3_cross_entropy(const float *z, const int64_t *label, float *grad, float *out, int N, int C,
                                int ignore_index, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (C < 1 || C > 64) return D3_ERR_ARG;
    if (ws_bytes < d3_cross_entropy_ws_bytes()) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int grid = (N + 255) / 256;
    if (grid > CE_GRID) grid = CE_GRID;
    if (grid < 1) grid = 1;
    if (C <= 32) ce_fwd_lds_kernel<<<grid, 256, (size_t)256 * (C | 1) * sizeof(float), s>>>(z, (const long long *)label, grad, (float *)ws, N, C, ignore_index);
    else ce_fwd_kernel<<<grid, 256, 0, s>>>(z, (const long long *)label, grad, (float *)ws, N, C, ignore_index);
    ce_reduce_kernel<<<1, 64, 0, s>>>((const float *)ws, grid, out);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ offset losses
// PointGroup.loss, offset part (reference: model/pointgroup.py:397-420): L1 offset loss and direction loss over the
// points that belong to an instance, ~15 elementwise library kernels forward and as many backward over N = 165k rows.
// One pass: per point the two loss terms and the two UNSCALED gradients w.r.t. pt_offsets,
//   g1 = sign(pt - gt) * valid,   g2 = -valid * d/dpt [ gt/(|gt|+1e-8) . pt/(|pt|+1e-8) ],
// and per-workgroup partial sums (sum dist*valid, sum dir*valid, sum valid) reduced in fixed order.
#define OL_GRID 1024
__global__ __launch_bounds__(256) void offset_loss_kernel(const float *__restrict__ pt, const float *__restrict__ coords,
                                                         const float *__restrict__ info, int ldi,
                                                         const long long *__restrict__ ids, long long ignore,
                                                         float *__restrict__ g1, float *__restrict__ g2,
                                                         float *__restrict__ part, int N) {
    __shared__ float s[3][256];
    float a = 0.f, b = 0.f, c = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const float valid = (ids[i] != ignore) ? 1.f : 0.f;
        float p[3], g[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { p[k] = pt[i * 3 + k]; g[k] = info[i * ldi + k] - coords[i * 3 + k]; }
        float dist = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float d = p[k] - g[k];
            dist += fabsf(d);
            g1[i * 3 + k] = valid * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        const float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), pn = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        const float ig = 1.f / (gn + 1e-8f), ip = 1.f / (pn + 1e-8f);
        const float dot = (g[0] * p[0] + g[1] * p[1] + g[2] * p[2]) * ig;      // gt_ . pt
        const float dir = -dot * ip;
        // d/dpt_k [ (gt_ . pt) / (|pt| + eps) ] = gt_k / (|pt|+eps) - (gt_ . pt) * pt_k / (|pt| (|pt|+eps)^2); |pt| = 0 -> first term only
        const float q = (pn > 0.f) ? dot * ip * ip / pn : 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) g2[i * 3 + k] = -valid * (g[k] * ig * ip - q * p[k]);
        a += dist * valid; b += dir * valid; c += valid;
    }
    s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s[0][threadIdx.x] += s[0][threadIdx.x + o]; s[1][threadIdx.x] += s[1][threadIdx.x + o]; s[2][threadIdx.x] += s[2][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = s[threadIdx.x][0];
}
__global__ void offset_loss_reduce_kernel(const float *part, int nblocks, float *out) {   // out: norm loss, dir loss, sum valid
    const int lane = threadIdx.x;
    double a = 0., b = 0., c = 0.;
    for (int i = lane; i < nblocks; i += 64) { a += (double)part[i * 3]; b += (double)part[i * 3 + 1]; c += (double)part[i * 3 + 2]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    if (lane == 0) { const double den = c + 1e-6; out[0] = (float)(a / den); out[1] = (float)(b / den); out[2] = (float)c; }
}
extern "C" size_t d3_offset_loss_ws_bytes(void) { return (size_t)OL_GRID * 3 * sizeof(float); }
// pt (N,3), coords (N,3), info (N, ldi) with the instance centre in columns 0..2, ids (N) int64.  out[0] = offset_norm_loss,
// out[1] = offset_dir_loss (both divided by sum(valid) + 1e-6), out[2] = sum(valid); g1 / g2 (N,3): unscaled gradients.
extern "C" int d3_offset_loss(const float *pt, const float *coords, const float *info, int ldi, const int64_t *ids,
                              long long ignore, float *g1, float *g2, float *out, int N, void *ws, size_t ws_bytes,
                              void *stream) {
    D3_CLEAR();
    if (ws_bytes < d3_offset_loss_ws_bytes()) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int grid = (N + 255) / 256;
    if (grid > OL_GRID) grid = OL_GRID;
    if (grid < 1) grid = 1;
    offset_loss_kernel<<<grid, 256, 0, s>>>(pt, coords, info, ldi, (const long long *)ids, ignore, g1, g2, (float *)ws, N);
    offset_loss_reduce_kernel<<<1, 64, 0, s>>>((const float *)ws, grid, out);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ row scatter-add
// Backward of the cluster feature gather `pt_feats[proposals_idx[:,1]]` (reference: model/pointgroup.py:130
// clusters_feats = feats[c_idxs]): out[idx[s], :] += g[s, :].  The library's index backward sorts the indices (a dozen
// launches).  A point is in at most one cluster of each of the two cluster sets, i.e. every output row receives at most
// two addends, and fp32 addition of two values onto zero is order independent ((0+a)+b == (0+b)+a): the atomics below
// are deterministic for this operator.
__global__ void scatter_add_rows_kernel(const float *__restrict__ g, const long long *__restrict__ idx, float *__restrict__ out,
                                        long long total, int C) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long long s = e / C;
    const int c = (int)(e - s * C);
    atomicAdd(&out[idx[s] * C + c], g[e]);
}
// out[r] = feats[idx[r]]: the row gathers of the point heads (`output.features[p2v_map]`, model/pointgroup.py:272; the cluster
// feature gather :139).  The library's index_select runs these 0.4-0.75 M-row x 16-float gathers at 0.6 TB/s (79 us each); one
// 16-byte load / store per thread, C % 4 == 0.
__global__ void gather_rows_kernel(const float *__restrict__ feats, const long long *__restrict__ idx, float *__restrict__ out,
                                   long long S, int C4) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S * C4) return;
    const long long r = e / C4;
    const int c = (int)(e - r * C4);
    ((float4 *)out)[e] = ((const float4 *)feats)[idx[r] * C4 + c];
}
extern "C" int d3_gather_rows(const float *feats, const int64_t *idx, float *out, long long S, int C, void *stream) {
    D3_CLEAR();
    if (S <= 0) return 0;
    if (C < 4 || (C & 3)) return D3_ERR_ARG;
    const long long n = S * (C / 4);
    gather_rows_kernel<<<(int)((n + 255) / 256), 256, 0, d3_stream(stream)>>>(feats, (const long long *)idx, out, S, C / 4);
    D3_LAUNCH_CHECK();
    return 0;
}

// out[r] = idx[r] in [0, rows) ? feats[idx[r]] : 0 -- a gather whose "no source" entries read a zero row (the library form
// concatenates a zero row to feats first: a full copy), any C; and its transpose for unique indices (out zero-filled by the caller)
__global__ void gather_rows_pad_kernel(const float *__restrict__ feats, long long rows, const long long *__restrict__ idx,
                                       float *__restrict__ out, long long total, int C, int scatter) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long long r = e / C;
    const int c = (int)(e - r * C);
    const long long i = idx[r];
    const bool ok = i >= 0 && i < rows;
    if (!scatter) out[e] = ok ? feats[i * C + c] : 0.f;
    else if (ok) out[i * C + c] = feats[e];       // (feats = the gradient rows, out = the (rows, C) gradient; indices unique)
}
extern "C" int d3_gather_rows_pad(const float *feats, long long rows, const int64_t *idx, float *out, long long S, int C, int scatter,
                                  void *stream) {
    D3_CLEAR();
    const long long total = S * C;
    if (total <= 0) return 0;
    gather_rows_pad_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(feats, rows, (const long long *)idx, out, total, C, scatter);
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_scatter_add_rows(const float *g, const int64_t *idx, float *out, long long S, int C, void *stream) {
    D3_CLEAR();
    const long long total = S * C;
    if (total <= 0) return 0;
    scatter_add_rows_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(g, (const long long *)idx, out, total, C);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ score loss
// PointGroup's proposal score loss (reference model/pointgroup.py:436-452) in one launch instead of ~30 elementwise
// launches over a few dozen proposals:  gt_iou = max_j ious[p, j];  gt_score = 1 above fg, 0 below bg, linear between;
// loss = mean_p BCEWithLogits(score_p, gt_score_p) with torch's stable form (1 - z) x + m + log(exp(-m) + exp(-x - m)),
// m = max(-x, 0).  out[0] = loss; dscore[p] = (sigmoid(x) - z) / P (scaled by the upstream gradient on the host side).
#define SL_T 256
// a wave per proposal, lanes along its IoU row, one proposal per wave over the whole grid (one workgroup walking all proposals
// paid a memory round trip per proposal and wave: 48-62 us at 430 x 160); term[p] = the proposal's loss
__global__ __launch_bounds__(SL_T) void score_rows_kernel(const float *__restrict__ scores, const float *__restrict__ ious, int P,
                                                         int nInst, float fg, float bg, float *__restrict__ gt_iou,
                                                         float *__restrict__ dscore, float *__restrict__ term) {
    const int p = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (p >= P) return;
    const float k = 1.f / (fg - bg), b = bg / (bg - fg);
    float m = -INFINITY;
    for (int j = lane; j < nInst; j += 64) m = fmaxf(m, ious[(long long)p * nInst + j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        gt_iou[p] = m;
        const float z = (m > fg) ? 1.f : (m < bg) ? 0.f : m * k + b;
        const float x = scores[p];
        const float mv = fmaxf(-x, 0.f);
        term[p] = (1.f - z) * x + mv + logf(expf(-mv) + expf(-x - mv));
        dscore[p] = (1.f / (1.f + expf(-x)) - z) / (float)P;
    }
}
// the terms added by one thread in proposal order: deterministic, and the order of the former single-workgroup kernel
__global__ void score_sum_kernel(const float *__restrict__ term, int P, float *__restrict__ out) {
    __shared__ float buf[1024];
    float part = 0.f;
    for (int p0 = 0; p0 < P; p0 += 1024) {
        const int n = min(1024, P - p0);
        if ((int)threadIdx.x < n) buf[threadIdx.x] = term[p0 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) for (int i = 0; i < n; i++) part += buf[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part / (float)P;
}
extern "C" int d3_score_loss(const float *scores, const float *ious, int P, int nInst, float fg, float bg, float *gt_iou,
                             float *dscore, float *out, void *stream) {
    D3_CLEAR();
    if (P <= 0 || nInst <= 0) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    float *term = out + 1;       // out: 1 + P floats (the per-proposal terms behind the loss)
    score_rows_kernel<<<(P + 3) / 4, SL_T, 0, s>>>(scores, ious, P, nInst, fg, bg, gt_iou, dscore, term);
    score_sum_kernel<<<1, 1024, 0, s>>>(term, P, out);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- caption cross-entropy
// `compute_cap_loss` (lib/captioning/loss_helper.py:177-224): XE over the words of the descriptions whose target box is good,
// target 0 = ignored, summed and divided by the number of counted words; word accuracy likewise.  One workgroup per
// (description, step) row of the (N, S, V) logits: row maximum / first arg-max, log-sum-exp, the row's loss term and -- in the
// same launch -- the gradient row (softmax - onehot) / count (the count of valid targets is recomputed by every workgroup from
// the N*S targets: 7 KB); a second one-workgroup launch adds the row terms in row order: deterministic.  Replaces ~26 library
// launches (where / compare / log-softmax / nll / argmax / reductions and their backward).
#define XE_T 256
__global__ __launch_bounds__(XE_T) void xe_rows_kernel(const float *__restrict__ pred, const long long *__restrict__ target,
                                                      long long ldt, const unsigned char *__restrict__ good, int N, int S, int V,
                                                      float *__restrict__ dpred, float *__restrict__ rowterm) {
    __shared__ float redf[XE_T];
    __shared__ int redi[XE_T];
    const int r = blockIdx.x, t = threadIdx.x;
    const int n = r / S, st = r - n * S;
    // number of counted words over the whole batch (every workgroup: N*S targets)
    int cnt = 0;
    for (int i = t; i < N * S; i += XE_T) {
        const int ni = i / S;
        cnt += (good[ni] && target[(long long)ni * ldt + (i - ni * S)] != 0) ? 1 : 0;
    }
    redi[t] = cnt;
    __syncthreads();
    for (int o = XE_T / 2; o > 0; o >>= 1) { if (t < o) redi[t] += redi[t + o]; __syncthreads(); }
    const int total = redi[0];
    __syncthreads();
    const float denom = (float)(total > 1 ? total : 1);
    const long long tg = good[n] ? target[(long long)n * ldt + st] : 0;
    const float *x = pred + (long long)r * V;
    float mx = -INFINITY;
    int am = 0x7FFFFFFF;
    for (int c = t; c < V; c += XE_T) { const float v = x[c]; if (v > mx) { mx = v; am = c; } }
    redf[t] = mx; redi[t] = am;
    __syncthreads();
    for (int o = XE_T / 2; o > 0; o >>= 1) {
        if (t < o) {
            const float v = redf[t + o]; const int a = redi[t + o];
            if (v > redf[t] || (v == redf[t] && a < redi[t])) { redf[t] = v; redi[t] = a; }
        }
        __syncthreads();
    }
    mx = redf[0]; am = redi[0];
    __syncthreads();
    float se = 0.f;
    for (int c = t; c < V; c += XE_T) se += expf(x[c] - mx);
    redf[t] = se;
    __syncthreads();
    for (int o = XE_T / 2; o > 0; o >>= 1) { if (t < o) redf[t] += redf[t + o]; __syncthreads(); }
    se = redf[0];
    const bool valid = tg != 0 && tg < V && tg > 0;
    const float scale = valid ? 1.f / denom : 0.f;
    float *d = dpred + (long long)r * V;
    const float inv = 1.f / se;
    for (int c = t; c < V; c += XE_T) d[c] = scale * (expf(x[c] - mx) * inv - (c == (int)tg ? 1.f : 0.f));
    if (t == 0) {
        rowterm[r * 2 + 0] = valid ? -((x[tg] - mx) - logf(se)) : 0.f;
        rowterm[r * 2 + 1] = (valid && am == (int)tg) ? 1.f : 0.f;
        if (r == 0) rowterm[(long long)N * S * 2] = denom;
    }
}
__global__ __launch_bounds__(XE_T) void xe_reduce_kernel(const float *__restrict__ rowterm, int R, float *__restrict__ out) {
    __shared__ float a[XE_T], b[XE_T];
    const int t = threadIdx.x;
    const int per = (R + XE_T - 1) / XE_T;
    float sa = 0.f, sb = 0.f;
    for (int i = t * per; i < min(R, (t + 1) * per); i++) { sa += rowterm[i * 2]; sb += rowterm[i * 2 + 1]; }
    a[t] = sa; b[t] = sb;
    __syncthreads();
    if (t == 0) {
        float x = 0.f, y = 0.f;
        for (int i = 0; i < XE_T; i++) { x += a[i]; y += b[i]; }
        const float denom = rowterm[(long long)R * 2];
        out[0] = x / denom; out[1] = y / denom;
    }
}
extern "C" size_t d3_masked_xe_ws_bytes(int N, int S) { return ((size_t)N * S * 2 + 1) * sizeof(float); }
extern "C" int d3_masked_xe(const float *pred, const long long *target, long long ld_target, const unsigned char *good, int N, int S,
                            int V, float *dpred, float *out2, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (N < 1 || S < 1 || V < 2) return D3_ERR_ARG;
    if (ws == nullptr || ws_bytes < d3_masked_xe_ws_bytes(N, S)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    xe_rows_kernel<<<N * S, XE_T, 0, s>>>(pred, target, ld_target, good, N, S, V, dpred, (float *)ws);
    xe_reduce_kernel<<<1, XE_T, 0, s>>>((const float *)ws, N * S, out2);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- orientation loss
// `compute_node_orientation_loss` (lib/captioning/loss_helper.py:244-307) in one launch: per graph edge the relative rotation
// of the GT objects assigned to its two ends (trace of R_s R_t^T -> angle -> bin), the rotation-mask / live-edge weight, the
// weighted cross-entropy over the num_bins orientation logits, the accuracy, and the gradient w.r.t. the logits -- ~45
// library launches (gathers of 3x3 matrices, a batched matmul, acos, bucketize, log-softmax, nll, argmax, their backward)
// on 10 k edges.  One edge per thread (a single workgroup walking ten edges per thread paid three dependent round trips per
// edge: 38 us), per-workgroup sums by a fixed-order tree, the workgroups' sums added in order by a second launch: deterministic.
// dpreds is the UNSCALED gradient mask * (softmax - onehot); the caller divides by out3[2].
#define OL_T 256
#define OL_MAXG 256
#define OL_MAXB 16
struct OlBounds { float v[OL_MAXB]; int n; };
__global__ __launch_bounds__(OL_T) void orient_loss_kernel(const float *__restrict__ preds, long long ldb, long long lde,
                                                          const float *__restrict__ eidx, const long long *__restrict__ nsrc,
                                                          const long long *__restrict__ ntar, const long long *__restrict__ assign,
                                                          const float *__restrict__ rot, const float *__restrict__ rmask, int B,
                                                          int E, int K, int G, int nb, OlBounds bd, float *__restrict__ dpreds,
                                                          float *__restrict__ part) {
    __shared__ float red[3][OL_T];
    const int t = threadIdx.x;
    const long long R = (long long)B * E;
    float s_ce = 0.f, s_m = 0.f, s_hit = 0.f;
    for (long long r = (long long)blockIdx.x * OL_T + t; r < R; r += (long long)gridDim.x * OL_T) {
        const int b = (int)(r / E), e = (int)(r - (long long)b * E);
        const float live = (long long)e < nsrc[b] * ntar[b] ? 1.f : 0.f;
        int sn = (int)(long long)eidx[((long long)b * 2 + 0) * E + e], tn = (int)(long long)eidx[((long long)b * 2 + 1) * E + e];
        sn = min(max(sn, 0), K - 1); tn = min(max(tn, 0), K - 1);
        long long as = assign[(long long)b * K + sn], at = assign[(long long)b * K + tn];
        as = as < 0 ? 0 : (as >= G ? G - 1 : as); at = at < 0 ? 0 : (at >= G ? G - 1 : at);
        const float *Rs = rot + ((long long)b * G + as) * 9, *Rt = rot + ((long long)b * G + at) * 9;
        float tr = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++) {       // diag_i of R_s R_t^T = sum_k Rs[i][k] Rt[i][k]
            float d = Rs[i * 3] * Rt[i * 3];
            d = fmaf(Rs[i * 3 + 1], Rt[i * 3 + 1], d);
            d = fmaf(Rs[i * 3 + 2], Rt[i * 3 + 2], d);
            tr = i == 0 ? d : tr + d;
        }
        const float c = fminf(fmaxf(0.5f * (tr - 1.f), -1.f), 1.f);
        const float rad = acosf(c);
        int label = 0;                       // torch.bucketize(right=False): number of boundaries strictly below the value
        for (int q = 0; q < bd.n; q++) label += bd.v[q] < rad ? 1 : 0;
        const float m = rmask[(long long)b * G + as] * rmask[(long long)b * G + at] * live;
        const float *x = preds + (long long)b * ldb + (long long)e * lde;
        float xv[OL_MAXB], mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int q = 0; q < OL_MAXB; q++) {
            xv[q] = q < nb ? x[q] : -INFINITY;
            if (xv[q] > mx) { mx = xv[q]; am = q; }
        }
        float se = 0.f;
#pragma unroll
        for (int q = 0; q < OL_MAXB; q++) se += q < nb ? expf(xv[q] - mx) : 0.f;
        const float lse = logf(se);
        if (label >= nb) label = nb - 1;      // (cannot happen: num_bins - 1 boundaries)
        float xl = 0.f;
#pragma unroll
        for (int q = 0; q < OL_MAXB; q++) xl = q == label ? xv[q] : xl;
        const float ce = -((xl - mx) - lse);
        s_ce += ce * m; s_m += m; s_hit += (am == label && m == 1.f) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < OL_MAXB; q++)
            if (q < nb) dpreds[r * nb + q] = m * (expf(xv[q] - mx) / se - (q == label ? 1.f : 0.f));
    }
    red[0][t] = s_ce; red[1][t] = s_m; red[2][t] = s_hit;
    __syncthreads();
    for (int o = OL_T / 2; o > 0; o >>= 1) {
        if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
        __syncthreads();
    }
    if (t == 0) { part[blockIdx.x * 3] = red[0][0]; part[blockIdx.x * 3 + 1] = red[1][0]; part[blockIdx.x * 3 + 2] = red[2][0]; }
}
// out3 = [loss, accuracy, sum of the edge weights + 1e-8]: the workgroups' partial sums added in workgroup order
__global__ void orient_sum_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    float a = 0.f, m = 0.f, h = 0.f;
    for (int i = 0; i < nblocks; i++) { a += part[i * 3]; m += part[i * 3 + 1]; h += part[i * 3 + 2]; }
    const float den = m + 1e-8f;
    out[0] = a / den; out[1] = h / den; out[2] = den;
}
extern "C" int d3_orientation_loss(const float *preds, long long ld_batch, long long ld_edge, const float *edge_index,
                                   const long long *num_src, const long long *num_tar, const long long *assign, const float *rotations,
                                   const float *rot_masks, int B, int E, int K, int G, int num_bins, const float *bounds_host,
                                   int nbounds, float *dpreds, float *out2, void *stream) {
    D3_CLEAR();
    if (B < 1 || E < 1 || K < 1 || G < 1 || num_bins < 1 || num_bins > OL_MAXB || nbounds < 0 || nbounds > OL_MAXB || (nbounds && !bounds_host))
        return D3_ERR_ARG;
    OlBounds bd;
    bd.n = nbounds;
    for (int i = 0; i < OL_MAXB; i++) bd.v[i] = i < nbounds ? bounds_host[i] : 0.f;
    const long long R = (long long)B * E;
    int grid = (int)((R + OL_T - 1) / OL_T);
    if (grid > OL_MAXG) grid = OL_MAXG;
    float *part = out2 + 3;      // out2: 3 + 3 * 256 floats (the workgroups' partial sums behind the results)
    orient_loss_kernel<<<grid, OL_T, 0, d3_stream(stream)>>>(preds, ld_batch, ld_edge, edge_index, num_src, num_tar, assign, rotations,
                                                           rot_masks, B, E, K, G, num_bins, bd, dpreds, part);
    orient_sum_kernel<<<1, 64, 0, d3_stream(stream)>>>(part, grid, out2);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- stack -> batch
// PointGroup.convert_stack_to_batch + get_object_assignments (reference model/pointgroup.py:216-263) in three launches
// instead of ~45 library launches over a few dozen proposals.  Proposal p of scene b with rank r among the kept
// proposals of b (stacking order) lands in slot b*K + inv_perm[b][r] when r < K (out[b][j] = buf[perm[b][j]] with
// buf[:n] = rows, as the reference's padded-then-shuffled copy); box corners are centre +- size/2 in fp64 like the
// numpy original (lib/utils/bbox.py:54-74, heading 0).
#define STB_MAXP 4096
#define STB_MAXBK 8192
__global__ __launch_bounds__(1024) void stb_slot_kernel(const int *__restrict__ bids, const long long *__restrict__ perm, int P,
                                                       int B, int K, long long *__restrict__ slot) {
    __shared__ int bS[STB_MAXP];
    __shared__ int invS[STB_MAXBK];
    const int t = threadIdx.x;
    for (int p = t; p < P; p += blockDim.x) bS[p] = bids[p];
    for (int i = t; i < B * K; i += blockDim.x) { const int b = i / K; invS[b * K + (int)perm[i]] = i - b * K; }
    __syncthreads();
    for (int p = t; p < P; p += blockDim.x) {
        const int b = bS[p];
        int rank = 0;
        for (int q = 0; q < p; q++) rank += (bS[q] == b) ? 1 : 0;
        slot[p] = (b >= 0 && b < B && rank < K) ? (long long)b * K + invS[b * K + rank] : -1;
    }
}
__global__ __launch_bounds__(64) void stb_scatter_kernel(const float *__restrict__ feats, const float *__restrict__ crop,
                                                        const float *__restrict__ scores, const long long *__restrict__ slot,
                                                        int m, float *__restrict__ feats_b, float *__restrict__ bbox_b,
                                                        float *__restrict__ center_b, float *__restrict__ sem_b,
                                                        float *__restrict__ scores_b, float *__restrict__ mask_b) {
    const int p = blockIdx.x, t = threadIdx.x;
    const long long s = slot[p];
    if (s < 0) return;
    for (int c = t; c < m; c += 64) feats_b[s * m + c] = feats[(long long)p * m + c];
    const float *cr = crop + (long long)p * 9;
    if (t < 24) {
        const int corner = t / 3, ax = t - corner * 3;
        // corner signs of get_3d_box_batch: x (+,+,-,-,+,+,-,-), y (+,-,-,+,+,-,-,+), z (+,+,+,+,-,-,-,-)
        const int sx = (corner & 2) ? -1 : 1, sy = ((corner + 1) & 2) ? -1 : 1, sz = (corner & 4) ? -1 : 1;
        const double sg = ax == 0 ? sx : ax == 1 ? sy : sz;
        bbox_b[s * 24 + t] = (float)((double)cr[3 + ax] / 2.0 * sg + (double)cr[ax]);
    }
    if (t >= 32 && t < 35) center_b[s * 3 + (t - 32)] = cr[t - 32];
    if (t == 40) { sem_b[s] = cr[7]; scores_b[s] = scores[p]; mask_b[s] = 1.f; }
}
// nearest GT centre in L1 for every slot (lib/utils/nn_distance.py:32-59 as used at :216-221): smallest index on ties
__global__ void stb_assign_kernel(const float *__restrict__ center_b, const float *__restrict__ gt, int B, int K, int G,
                                  long long *__restrict__ assign) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K;
    const float x = center_b[i * 3], y = center_b[i * 3 + 1], z = center_b[i * 3 + 2];
    float best = INFINITY; int bi = 0;
    for (int g = 0; g < G; g++) {
        const float *c = gt + ((long long)b * G + g) * 3;
        const float d = __fadd_rn(__fadd_rn(fabsf(__fsub_rn(x, c[0])), fabsf(__fsub_rn(y, c[1]))), fabsf(__fsub_rn(z, c[2])));
        if (d < best) { best = d; bi = g; }
    }
    assign[i] = bi;
}
extern "C" int d3_stack_to_batch(const float *feats, const float *crop, const float *scores, const int *bids,
                                 const long long *perm, const float *center_label, int G, int P, int m, int B, int K,
                                 float *feats_b, float *bbox_b, float *center_b, float *sem_b, float *scores_b, float *mask_b,
                                 long long *slot, long long *assign, void *stream) {
    D3_CLEAR();
    if (P < 0 || P > STB_MAXP || B < 1 || K < 1 || B * K > STB_MAXBK || m < 1) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    if (P > 0) {
        stb_slot_kernel<<<1, 1024, 0, s>>>(bids, perm, P, B, K, slot);
        stb_scatter_kernel<<<P, 64, 0, s>>>(feats, crop, scores, slot, m, feats_b, bbox_b, center_b, sem_b, scores_b, mask_b);
    }
    if (assign && center_label && G > 0) stb_assign_kernel<<<(B * K + 255) / 256, 256, 0, s>>>(center_b, center_label, B, K, G, assign);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ AdamW
// One launch for every parameter tensor of a group (torch.optim.AdamW semantics, decoupled weight decay, no amsgrad):
// a device table holds (param, grad, exp_avg, exp_avg_sq) pointers per tensor and a block map (tensor, chunk) per
// workgroup; HBM bound: 4 reads + 3 writes of 4 B per element.
#define ADAMW_CHUNK 4096
__global__ __launch_bounds__(256) void adamw_kernel(const long long *__restrict__ ptrs, const int *__restrict__ numel,
                                                   const int2 *__restrict__ blocks, float step_size, float beta2, float omb1,
                                                   float omb2, float eps, float lr_wd, float bc2_sqrt) {
    const int2 bm = blocks[blockIdx.x];
    float *p = (float *)ptrs[bm.x * 4 + 0];
    const float *g = (const float *)ptrs[bm.x * 4 + 1];
    float *m = (float *)ptrs[bm.x * 4 + 2], *v = (float *)ptrs[bm.x * 4 + 3];
    const int n = numel[bm.x], base = bm.y * ADAMW_CHUNK;
    float pv[16], gv[16], mv[16], vv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {   // all loads first
        const int i = base + j * 256 + threadIdx.x;
        const bool ok = i < n;
        pv[j] = ok ? p[i] : 0.f; gv[j] = ok ? g[i] : 0.f; mv[j] = ok ? m[i] : 0.f; vv[j] = ok ? v[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < n) {
            float pp = pv[j] - lr_wd * pv[j];
            const float mm = mv[j] + omb1 * (gv[j] - mv[j]);
            const float v2 = beta2 * vv[j] + omb2 * gv[j] * gv[j];
            const float denom = sqrtf(v2) / bc2_sqrt + eps;
            pp -= step_size * mm / denom;
            p[i] = pp; m[i] = mm; v[i] = v2;
        }
    }
}
extern "C" int d3_adamw(const long long *ptrs, const int *numel, const void *blocks, int nblocks, double lr, double beta1,
                        double beta2, double eps, double weight_decay, double bias_correction1, double bias_correction2_sqrt,
                        void *stream) {
    D3_CLEAR();
    if (nblocks <= 0) return 0;
    // scalars derived in double, as the library does before it narrows them
    adamw_kernel<<<nblocks, 256, 0, d3_stream(stream)>>>(ptrs, numel, (const int2 *)blocks, (float)(lr / bias_correction1),
                                                        (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                                                        (float)(lr * weight_decay), (float)bias_correction2_sqrt);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_adamw_chunk(void) { return ADAMW_CHUNK; }

