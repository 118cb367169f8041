// bn.hip -- MinkowskiBatchNorm (+ MinkowskiReLU) over the rows of an (M,C) feature matrix (gfx950).
//
// Reference: sp_norm = MinkowskiBatchNorm(eps=1e-4, momentum=0.1) followed by MinkowskiReLU
// (model/pointgroup.py:65,72-73; model/common.py:36-40,62-63,87-89,95-97), i.e. BatchNorm1d with
// per-rank batch statistics in training.  Pure streaming kernels: bytes = 4*M*C read for the
// statistics, 8*M*C for the normalise pass; the backward reads x, dy twice and writes dx.
#include "common.h"

#define BN_T 256

// per-channel sum / sum of squares -> per-workgroup fp64 partials ws[block][2C] (no memset, no atomics,
// deterministic); the finalize kernel adds the partials
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float *__restrict__ x, int M, int C, double *acc) {
    __shared__ float s1[BN_T], s2[BN_T];
    const int t = threadIdx.x;
    const int active = (BN_T / C) * C, rpp = active / C;
    float a = 0.f, b = 0.f;
    if (t < active) {
        const int c = t % C;
        for (long long r = (long long)blockIdx.x * rpp + t / C; r < M; r += (long long)gridDim.x * rpp) {
            const float v = x[r * C + c];
            a += v; b = fmaf(v, v, b);
        }
    }
    s1[t] = a; s2[t] = b;
    __syncthreads();
    if (t < C) {
        double da = 0., db = 0.;
        for (int k = t; k < active; k += C) { da += (double)s1[k]; db += (double)s2[k]; }
        acc[(size_t)blockIdx.x * 2 * C + t] = da;
        acc[(size_t)blockIdx.x * 2 * C + C + t] = db;
    }
}
// mean / biased variance + (optionally) the running-statistics update of nn.BatchNorm1d in training mode:
// running = (1-momentum)*running + momentum*stat, with the UNBIASED variance (M/(M-1))
__global__ void bn_finalize_kernel(const double *acc, int nblocks, int M, int C, float *mean, float *var,
                                   float *running_mean, float *running_var, float momentum) {
    // one wave per channel: lanes stride over the workgroup partials, fp64 wave reduction (fixed order: deterministic)
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double sa = 0., sb = 0.;
    for (int b = lane; b < nblocks; b += 64) { sa += acc[(size_t)b * 2 * C + c]; sb += acc[(size_t)b * 2 * C + C + c]; }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if (lane != 0) return;
    const double m = sa / (double)M;
    double v = sb / (double)M - m * m;
    if (v < 0.) v = 0.;
    mean[c] = (float)m; var[c] = (float)v;  // biased variance (what the normalisation uses)
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    }
}

__global__ void bn_relu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                   const float *__restrict__ var, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float *__restrict__ y, long long total4, int C,
                                   float eps, int relu) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread (C % 4 == 0)
    if (e >= total4) return;
    const int c = (int)((e * 4) % C);
    const float4 v = ((const float4 *)x)[e];
    float in[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float inv = rsqrtf(var[c + j] + eps);
        float r = fmaf((in[j] - mean[c + j]) * inv, gamma[c + j], beta[c + j]);
        o[j] = (relu && r < 0.f) ? 0.f : r;
    }
    ((float4 *)y)[e] = make_float4(o[0], o[1], o[2], o[3]);
}
// same, output stored as bf16 (round-to-nearest-even): the consumer is a convolution, whose MFMA operands are bf16
// anyway, so this is numerically identical to writing fp32 and converting in the conv -- at half the bytes
__device__ __forceinline__ unsigned int bn_pack2bf(float lo, float hi) {
    unsigned int a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7FFFu + ((a >> 16) & 1u); b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}
__global__ void bn_relu_fwd_bf16_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                        const float *__restrict__ var, const float *__restrict__ gamma,
                                        const float *__restrict__ beta, unsigned short *__restrict__ y, long long total4,
                                        int C, float eps, int relu) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // 4 channels per thread (C % 4 == 0)
    if (e >= total4) return;
    const int c = (int)((e * 4) % C);
    const float4 v = ((const float4 *)x)[e];
    float in[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float inv = rsqrtf(var[c + j] + eps);
        float r = fmaf((in[j] - mean[c + j]) * inv, gamma[c + j], beta[c + j]);
        o[j] = (relu && r < 0.f) ? 0.f : r;
    }
    ((uint2 *)y)[e] = make_uint2(bn_pack2bf(o[0], o[1]), bn_pack2bf(o[2], o[3]));
}
__global__ void bn_relu_fwd_scalar_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                          const float *__restrict__ var, const float *__restrict__ gamma,
                                          const float *__restrict__ beta, float *__restrict__ y, long long total,
                                          int C, float eps, int relu) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    float r = fmaf((x[e] - mean[c]) * rsqrtf(var[c] + eps), gamma[c], beta[c]);
    y[e] = (relu && r < 0.f) ? 0.f : r;
}

// backward reductions: sum g, sum g*xhat with g = dy * relu'(y)
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ dy,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ var,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int M, int C, float eps,
                                                            int relu, double *acc) {
    __shared__ float s1[BN_T], s2[BN_T];
    const int t = threadIdx.x;
    const int active = (BN_T / C) * C, rpp = active / C;
    float a = 0.f, b = 0.f;
    if (t < active) {
        const int c = t % C;
        const float mu = mean[c], inv = rsqrtf(var[c] + eps), ga = gamma[c], be = beta[c];
        for (long long r = (long long)blockIdx.x * rpp + t / C; r < M; r += (long long)gridDim.x * rpp) {
            const float xh = (x[r * C + c] - mu) * inv;
            float g = dy[r * C + c];
            if (relu && fmaf(xh, ga, be) <= 0.f) g = 0.f;
            a += g; b = fmaf(g, xh, b);
        }
    }
    s1[t] = a; s2[t] = b;
    __syncthreads();
    if (t < C) {
        double da = 0., db = 0.;
        for (int k = t; k < active; k += C) { da += (double)s1[k]; db += (double)s2[k]; }
        acc[(size_t)blockIdx.x * 2 * C + t] = da;
        acc[(size_t)blockIdx.x * 2 * C + C + t] = db;
    }
}
// adds the partials: sums[0..C) = sum g, sums[C..2C) = sum g*xhat (fp64), and writes dbeta / dgamma
__global__ void bn_bwd_params_kernel(const double *acc, int nblocks, int C, double *sums, float *dgamma,
                                     float *dbeta) {
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double sa = 0., sb = 0.;
    for (int b = lane; b < nblocks; b += 64) { sa += acc[(size_t)b * 2 * C + c]; sb += acc[(size_t)b * 2 * C + C + c]; }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if (lane != 0) return;
    sums[c] = sa; sums[C + c] = sb;
    dbeta[c] = (float)sa;
    dgamma[c] = (float)sb;
}
__global__ void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                    const float *__restrict__ mean, const float *__restrict__ var,
                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                    const double *__restrict__ acc, float *__restrict__ dx, long long total, int M,
                                    int C, float eps, int relu) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    const float inv = rsqrtf(var[c] + eps), ga = gamma[c];
    const float xh = (x[e] - mean[c]) * inv;
    float g = dy[e];
    if (relu && fmaf(xh, ga, beta[c]) <= 0.f) g = 0.f;
    const float mg = (float)(acc[c] / (double)M), mgx = (float)(acc[C + c] / (double)M);
    dx[e] = ga * inv * (g - mg - xh * mgx);
}
static int bn_grid(int M, int C) {
    const int rpp = (BN_T / C);
    long long blocks = ((long long)M + rpp - 1) / rpp;
    blocks = (blocks + 15) / 16;  // >= 16 row passes per block
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    return (int)blocks;
}

#define BN_MAXBLK 512
static int bn_blocks(int M, int C) { int g = bn_grid(M, C); return g > BN_MAXBLK ? BN_MAXBLK : g; }

extern "C" size_t d3_bn_ws_bytes(int C) { return (size_t)(BN_MAXBLK + 1) * 2 * C * sizeof(double); }

extern "C" int d3_bn_stats(const float *x, int M, int C, float *mean, float *var, float *running_mean,
                           float *running_var, float momentum, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (C < 1 || C > BN_T) return D3_ERR_ARG;
    if (ws == nullptr || ws_bytes < d3_bn_ws_bytes(C)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int nb = bn_blocks(M, C);
    bn_stats_kernel<<<nb, BN_T, 0, s>>>(x, M, C, (double *)ws);
    bn_finalize_kernel<<<(C + 3) / 4, 256, 0, s>>>((const double *)ws, nb, M, C, mean, var, running_mean, running_var,
                                                  momentum);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_bn_relu_fwd(const float *x, const float *mean, const float *var, const float *gamma,
                              const float *beta, float *y, int M, int C, float eps, int relu, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    hipStream_t s = d3_stream(stream);
    long long total = (long long)M * C;
    if ((C & 3) == 0) {
        long long t4 = total / 4;
        bn_relu_fwd_kernel<<<(int)((t4 + 255) / 256), 256, 0, s>>>(x, mean, var, gamma, beta, y, t4, C, eps, relu);
    } else {
        bn_relu_fwd_scalar_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(x, mean, var, gamma, beta, y, total, C,
                                                                           eps, relu);
    }
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_bn_relu_fwd_bf16(const float *x, const float *mean, const float *var, const float *gamma,
                                   const float *beta, void *y_bf16, int M, int C, float eps, int relu, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if ((C & 3) != 0) return D3_ERR_ARG;
    long long t4 = (long long)M * C / 4;
    bn_relu_fwd_bf16_kernel<<<(int)((t4 + 255) / 256), 256, 0, d3_stream(stream)>>>(x, mean, var, gamma, beta,
                                                                                  (unsigned short *)y_bf16, t4, C, eps, relu);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_bn_relu_bwd(const float *x, const float *dy, const float *mean, const float *var,
                              const float *gamma, const float *beta, float *dx, float *dgamma, float *dbeta, int M,
                              int C, float eps, int relu, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (C < 1 || C > BN_T) return D3_ERR_ARG;
    if (ws == nullptr || ws_bytes < d3_bn_ws_bytes(C)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int nb = bn_blocks(M, C);
    double *partials = (double *)ws, *sums = partials + (size_t)BN_MAXBLK * 2 * C;
    bn_bwd_reduce_kernel<<<nb, BN_T, 0, s>>>(x, dy, mean, var, gamma, beta, M, C, eps, relu, partials);
    bn_bwd_params_kernel<<<(C + 3) / 4, 256, 0, s>>>(partials, nb, C, sums, dgamma, dbeta);
    long long total = (long long)M * C;
    bn_bwd_apply_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(x, dy, mean, var, gamma, beta, sums, dx, total, M, C,
                                                                 eps, relu);
    D3_LAUNCH_CHECK();
    return 0;
}
