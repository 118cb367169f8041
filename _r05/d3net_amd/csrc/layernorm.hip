// layernorm.hip -- fused residual add + LayerNorm of the transformer match module (gfx950).
//
// Replaces `self.layer_norm(queries + out)` of MultiHeadAttention and the LayerNorm of `lang_fc` (reference:
// model/transformer/attention.py:134-176, model/match_module.py:170-173): as library calls an add, a LayerNorm forward and --
// in the backward -- a LayerNorm backward with two reduction launches plus an add, per attention layer and direction.
// Here y = LN(a + b) * gamma + beta is ONE pass (a wave per row of D <= 1024 channels: the row stays in registers between
// the statistics and the normalisation), and the backward is one pass for dx (= d(a) = d(b)) plus per-workgroup partial
// sums of dgamma / dbeta that a second launch adds in workgroup order: deterministic, no atomics.
// torch.nn.LayerNorm semantics: biased variance, eps inside the square root, elementwise affine.  HBM bound: 4*D bytes in
// (8*D with b), 4*D out per row.
#include "common.h"

#define LN_MAXE 16          // elements per lane: D <= 64 * LN_MAXE
#define LN_ROWS 16          // rows per workgroup (4 waves x 4 rows)

__device__ __forceinline__ float ln_wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd,
                                                    int R, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ne = (D + 63) >> 6;
    for (int rr = 0; rr < LN_ROWS / 4; rr++) {
        const int row = blockIdx.x * LN_ROWS + rr * 4 + wave;
        if (row >= R) return;        // (wave-uniform)
        float x[LN_MAXE];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++) {
            const int c = e * 64 + lane;
            x[e] = 0.f;
            if (e < ne && c < D) { x[e] = a[(size_t)row * D + c]; if (b) x[e] += b[(size_t)row * D + c]; }
            s += x[e];
        }
        const float mu = ln_wave_sum(s) / (float)D;
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++) {
            const int c = e * 64 + lane;
            if (e < ne && c < D) { const float d = x[e] - mu; v += d * d; }
        }
        const float rs = rsqrtf(ln_wave_sum(v) / (float)D + eps);
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++) {
            const int c = e * 64 + lane;
            if (e < ne && c < D) y[(size_t)row * D + c] = (x[e] - mu) * rs * gamma[c] + beta[c];
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); part[block][2][D] = (sum g*xhat, sum g) over the block's rows
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                    const float *__restrict__ gamma, const float *__restrict__ mean,
                                                    const float *__restrict__ rstd, const float *__restrict__ dy,
                                                    float *__restrict__ dx, float *__restrict__ part, int R, int D) {
    extern __shared__ float sh[];      // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ne = (D + 63) >> 6;
    float dg[LN_MAXE], db[LN_MAXE];
#pragma unroll
    for (int e = 0; e < LN_MAXE; e++) { dg[e] = 0.f; db[e] = 0.f; }
    for (int rr = 0; rr < LN_ROWS / 4; rr++) {
        const int row = blockIdx.x * LN_ROWS + rr * 4 + wave;
        if (row >= R) break;         // (wave-uniform)
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXE], gg[LN_MAXE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++) {
            const int c = e * 64 + lane;
            xh[e] = 0.f; gg[e] = 0.f;
            if (e < ne && c < D) {
                float x = a[(size_t)row * D + c]; if (b) x += b[(size_t)row * D + c];
                const float g = dy[(size_t)row * D + c];
                xh[e] = (x - mu) * rs;
                gg[e] = g * gamma[c];
                dg[e] += g * xh[e]; db[e] += g;
            }
            s1 += gg[e]; s2 += gg[e] * xh[e];
        }
        s1 = ln_wave_sum(s1) / (float)D; s2 = ln_wave_sum(s2) / (float)D;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++) {
            const int c = e * 64 + lane;
            if (e < ne && c < D) dx[(size_t)row * D + c] = rs * (gg[e] - s1 - xh[e] * s2);
        }
    }
#pragma unroll
    for (int e = 0; e < LN_MAXE; e++) {
        const int c = e * 64 + lane;
        if (e < ne && c < D) { sh[(wave * 2 + 0) * D + c] = dg[e]; sh[(wave * 2 + 1) * D + c] = db[e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, cc = c - which * D;
        part[((size_t)blockIdx.x * 2 + which) * D + cc] = sh[(0 * 2 + which) * D + cc] + sh[(1 * 2 + which) * D + cc] +
                                                         sh[(2 * 2 + which) * D + cc] + sh[(3 * 2 + which) * D + cc];
    }
}
__global__ void ln_bwd_final_kernel(const float *__restrict__ part, int nblocks, int D, float *dgamma, float *dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * D) return;
    const int which = c / D, cc = c - which * D;
    float s = 0.f;
    for (int b = 0; b < nblocks; b++) s += part[((size_t)b * 2 + which) * D + cc];
    (which ? dbeta : dgamma)[cc] = s;
}

extern "C" int d3_layernorm_fwd(const float *a, const float *b, const float *gamma, const float *beta, float *y, float *mean,
                                float *rstd, int R, int D, float eps, void *stream) {
    D3_CLEAR();
    if (R <= 0) return 0;
    if (D < 1 || D > 64 * LN_MAXE) return D3_ERR_ARG;
    ln_fwd_kernel<<<(R + LN_ROWS - 1) / LN_ROWS, 256, 0, d3_stream(stream)>>>(a, b, gamma, beta, y, mean, rstd, R, D, eps);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" size_t d3_layernorm_ws_bytes(int R, int D) { return (size_t)((R + LN_ROWS - 1) / LN_ROWS) * 2 * D * 4 + 256; }
extern "C" int d3_layernorm_bwd(const float *a, const float *b, const float *gamma, const float *mean, const float *rstd,
                                const float *dy, float *dx, float *dgamma, float *dbeta, int R, int D, void *ws, size_t ws_bytes,
                                void *stream) {
    D3_CLEAR();
    if (D < 1 || D > 64 * LN_MAXE) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    if (R <= 0) { D3_CHECK(hipMemsetAsync(dgamma, 0, (size_t)D * 4, s)); D3_CHECK(hipMemsetAsync(dbeta, 0, (size_t)D * 4, s)); return 0; }
    if (ws == nullptr || ws_bytes < d3_layernorm_ws_bytes(R, D)) return D3_ERR_WORKSPACE;
    const int nb = (R + LN_ROWS - 1) / LN_ROWS;
    ln_bwd_kernel<<<nb, 256, (size_t)4 * 2 * D * 4, s>>>(a, b, gamma, mean, rstd, dy, dx, (float *)ws, R, D);
    ln_bwd_final_kernel<<<(2 * D + 255) / 256, 256, 0, s>>>((const float *)ws, nb, D, dgamma, dbeta);
    D3_LAUNCH_CHECK();
    return 0;
}
