// unet.hip -- native executor for the sparse U-Nets of the PointGroup detector (gfx950).
//
// The reference runs its backbone (stem conv + 7-level UBlock + BN + ReLU, model/pointgroup.py:69-74) and its
// ScoreNet (UBlock([m,2m]) + BN + ReLU, :88-92) module by module through MinkowskiEngine: ~400 python-level calls
// and ~1,100 kernel launches per training step.  Here the python layer (d3net_amd/netexec.py) flattens the module
// tree ONCE into a layer program -- pre-activation convolutions (model/common.py:22-53 ResidualBlock, :56-70
// VGGBlock), strided / transposed convolutions, concatenations and residual adds of UBlock (:73-118) -- and this
// file executes the whole forward, and the whole backward, in one C-ABI call each:
//   * one activation arena per forward (layout planned from the level row counts; 288 GB of HBM: nothing is
//     recomputed or recycled inside a step), bf16 for BN->ReLU outputs (the MFMA operand precision), fp32 for the
//     residual stream;
//   * ME.cat never copies: both producers write their column half of the concatenated buffer (strided epilogue);
//     `x += identity` is the convolution epilogue; BatchNorm batch statistics come from the producing
//     convolution's epilogue partials (no separate statistics pass);
//   * backward: data gradients on the caller's stream, weight gradients on a side stream (they are off the
//     critical path: nothing downstream consumes them until the optimizer), residual gradients alias instead of
//     copy, gradient accumulation is fused into the kernels that produce the second contribution;
//   * all weights are re-packed to bf16 MFMA fragment order by ONE launch per forward;
//   * REFERENCE PRECISION: a program whose buffers are all fp32 (netexec.py builds it for minkowski.set_exact(True)) runs the
//     same schedule with fp32 activations / gradients, fp32 weight fragments and the D3_CONV_F32 kernels of spconv2.hip
//     (exact fp32 products on v_mfma_f32_16x16x4_f32) -- MinkowskiEngine's own precision (model/common.py:32-41).
// Kernels used: spconv2.hip (d3_spconv_fwd2 / d3_spconv_wgrad2) and the strided BatchNorm kernels below.
#include "common.h"
#include <vector>
#include <map>
#include <string.h>
#include <stdlib.h>

extern "C" size_t d3_spconv_pack_bytes(int K, int Cin, int Cout);
extern "C" size_t d3_spconv_pack_bytes_ex(int K, int Cin, int Cout, int flags);
extern "C" int d3_spconv_fwd2_nparts_ex(int Mout, int K, int Cin, int Cout, int flags);

// ------------------------------------------------------------------------------ small kernels
__device__ __forceinline__ unsigned int un_pack2bf(float lo, float hi) {
    unsigned int a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7FFFu + ((a >> 16) & 1u); b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}

// one element of a gradient row stored as fp32 or (bf != 0) bf16
__device__ __forceinline__ float un_ld1(const float *p, long long idx, int bf) {
    return bf ? __uint_as_float((unsigned int)((const unsigned short *)p)[idx] << 16) : p[idx];
}
// four consecutive elements of a BatchNorm input row: fp32, or bf16 (round 6: single-consumer convolution outputs).  The load is RAW
// (bf16: 8 bytes in .x / .y) and un_cvx4 widens it where the values are used -- a conversion right behind the load would make every
// load of a batch wait for its own data instead of all of them being in flight together
__device__ __forceinline__ float4 un_ldx4(const float *x, long long idx, int xbf) {
    if (xbf) { const uint2 v = *(const uint2 *)((const unsigned short *)x + idx); return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f); }
    return *(const float4 *)(x + idx);
}
__device__ __forceinline__ void un_cvx4(const float4 r, int xbf, float *o) {
    if (xbf) {
        const unsigned int a = __float_as_uint(r.x), b = __float_as_uint(r.y);
        o[0] = __uint_as_float(a << 16); o[1] = __uint_as_float(a & 0xFFFF0000u); o[2] = __uint_as_float(b << 16); o[3] = __uint_as_float(b & 0xFFFF0000u);
    } else { o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
}

struct PackJob { const float *W; size_t dst_off; int K, S, CinW, Cout, NT, flipk, transw, Cin, f32; long long start; };

// all convolution weights of a network -> bf16 MFMA fragment order (layout of spconv2.hip's spconv_pack_kernel)
__global__ void un_pack_batched_kernel(const PackJob *__restrict__ jobs, int njobs, long long total, char *arena) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].start <= e) lo = mid; else hi = mid - 1; }
    const PackJob j = jobs[lo];
    const long long l = e - j.start;
    const int col = (int)(l & 15);
    const int n = (int)((l >> 4) % j.NT);
    const int c8 = (int)((l / (16 * j.NT)) % j.S);
    const int k = (int)(l / ((long long)16 * j.NT * j.S));
    const int co = n * 16 + col, wk = j.flipk ? (j.K - 1 - k) : k;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int ci = c8 * 8 + q;
        v[q] = 0.f;
        // forward layout W (K, CinW, Cout); transposed use (data gradient): the call's Cin is the layer's Cout
        if (j.transw) { if (co < j.Cout && ci < j.CinW) v[q] = j.W[((long long)wk * j.Cout + co) * j.CinW + ci]; }
        else if (co < j.Cout && ci < j.CinW) v[q] = j.W[((long long)wk * j.CinW + ci) * j.Cout + co];
    }
    if (j.f32) {      // fp32 fragments (D3_CONV_F32): same element order, 32 bytes per element
        float4 *d = (float4 *)(arena + j.dst_off) + l * 2;
        d[0] = make_float4(v[0], v[1], v[2], v[3]); d[1] = make_float4(v[4], v[5], v[6], v[7]);
        return;
    }
    ((uint4 *)(arena + j.dst_off))[l] = make_uint4(un_pack2bf(v[0], v[1]), un_pack2bf(v[2], v[3]), un_pack2bf(v[4], v[5]), un_pack2bf(v[6], v[7]));
}
// ... and with an fp32 destination (reference-precision programs)
__global__ void un_padcast_f32_kernel(const float *__restrict__ x, float *__restrict__ y, long long M, int Cs, int Cd) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * Cd) return;
    const long long row = e / Cd;
    const int c = (int)(e - row * Cd);
    y[e] = (c < Cs) ? x[row * Cs + c] : 0.f;
}

// x (M, Cs) fp32 -> y (M, Cd) bf16, zero padded (Cd % 8 == 0): the stem convolution's operand
__global__ void un_padcast_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, long long M, int Cs, int Cd) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 2-channel pair per thread
    const int pairs = Cd >> 1;
    if (e >= M * pairs) return;
    const long long row = e / pairs;
    const int c = (int)(e - row * pairs) * 2;
    const float a = (c < Cs) ? x[row * Cs + c] : 0.f, b = (c + 1 < Cs) ? x[row * Cs + c + 1] : 0.f;
    ((unsigned int *)y)[e] = un_pack2bf(a, b);
}

#define UN_T 256
// per-channel sum / sum of squares of x (M, ld) -> partials [block][2][C] (fp32; summed in fp64 by the finalize)
#define UN_P2_ROWS 16      // rows of a second-level fp64 partial table (== C2_P2_ROWS of spconv2.hip: the producers add row = workgroup % 16)
__global__ __launch_bounds__(UN_T) void un_stats_parts_kernel(const float *__restrict__ x, int ld, int M, int C, float *part, double *part2) {
    __shared__ float s1[UN_T], s2[UN_T];
    const int t = threadIdx.x;
    const int active = (UN_T / C) * C, rpp = active / C;
    float a = 0.f, b = 0.f;
    if (t < active) {
        const int c = t % C;
        for (long long r = (long long)blockIdx.x * rpp + t / C; r < M; r += (long long)gridDim.x * rpp) {
            const float v = x[r * ld + c];
            a += v; b = fmaf(v, v, b);
        }
    }
    s1[t] = a; s2[t] = b;
    __syncthreads();
    if (t < C) {
        float da = 0.f, db = 0.f;
        for (int k = t; k < active; k += C) { da += s1[k]; db += s2[k]; }
        part[(size_t)blockIdx.x * 2 * C + t] = da;
        part[(size_t)blockIdx.x * 2 * C + C + t] = db;
        if (part2) {
            unsafeAtomicAdd(&part2[(size_t)(blockIdx.x % UN_P2_ROWS) * 2 * C + t], (double)da);
            unsafeAtomicAdd(&part2[(size_t)(blockIdx.x % UN_P2_ROWS) * 2 * C + C + t], (double)db);
        }
    }
}

// p2 (optional): the producer's second-level table [UN_P2_ROWS][2][width] of fp64 sums (spconv2.hip C2_P2_ROWS)
struct StatSrc { const float *part; int nparts, width, c0, cn; const double *p2; };

// mean / biased variance of C channels from up to two partial sets (a concatenated input has two producers), and
// the running-statistics update of nn.BatchNorm1d in training mode.  One wave per channel, fp64, fixed order.
__global__ void un_bn_finalize_kernel(StatSrc s0, StatSrc s1, int M, int C, float *mean, float *var,
                                      float *running_mean, float *running_var, float momentum) {
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const StatSrc s = (c >= s0.c0 && c < s0.c0 + s0.cn) ? s0 : s1;
    const int cc = c - s.c0;
    double sa = 0., sb = 0.;
    for (int b0 = lane; b0 < s.nparts; b0 += 256) {   // four partial rows per lane per round trip; same summation order
        float pa[4], pb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int b = b0 + q * 64;
            const size_t o = (size_t)(b < s.nparts ? b : 0) * 2 * s.width + cc;
            pa[q] = s.part[o]; pb[q] = s.part[o + s.width];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (b0 + q * 64 < s.nparts) { sa += (double)pa[q]; sb += (double)pb[q]; }
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if (lane != 0) return;
    const double m = sa / (double)M;
    double v = sb / (double)M - m * m;
    if (v < 0.) v = 0.;
    mean[c] = (float)m; var[c] = (float)v;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    }
}

// y = [relu]((x - mean) * rsqrt(var + eps) * gamma + beta); x (M, ldx) fp32; y (M, ldy) bf16 or fp32
template <bool BF16>
__global__ __launch_bounds__(256) void un_bn_apply_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ mean,
                                   const float *__restrict__ var, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, void *__restrict__ y, int ldy, long long M, int C,
                                   float eps, int relu, int xbf) {
    // (row-walking form: see un_bn_bwd_apply_kernel)
    const int c4 = C >> 2, rpb = 256 / c4, t = threadIdx.x;
    if (t >= rpb * c4) return;
    const int rl = t / c4, c = (t - rl * c4) * 4;
    float inv[4], ga[4], mu[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { inv[j] = rsqrtf(var[c + j] + eps); ga[j] = gamma[c + j]; mu[j] = mean[c + j]; be[j] = beta[c + j]; }
    const long long rows = (long long)((256 / c4) * 8);
    const long long r0 = (long long)blockIdx.x * rows, r1 = (r0 + rows < M) ? r0 + rows : M;
    for (long long rb = r0 + rl; rb < r1; rb += (long long)rpb * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long row = rb + (long long)u * rpb;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r1) v[u] = un_ldx4(x, row * ldx + c, xbf);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long row = rb + (long long)u * rpb;
            if (row >= r1) continue;
            float in[4];
            un_cvx4(v[u], xbf, in);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float r = fmaf((in[j] - mu[j]) * inv[j], ga[j], be[j]);
                o[j] = (relu && r < 0.f) ? 0.f : r;
            }
            if (BF16) *(uint2 *)((unsigned short *)y + row * ldy + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));
            else *(float4 *)((float *)y + row * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// backward reductions: sum g, sum g*xhat with g = dy * relu'(y) -> fp32 partials [block][2][C]
__global__ __launch_bounds__(UN_T) void un_bn_bwd_reduce_kernel(const float *__restrict__ x, int ldx,
                                                               const float *__restrict__ dy, int ldy,
                                                               const float *__restrict__ mean, const float *__restrict__ var,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               int M, int C, float eps, int relu, float *part, int dybf, int xbf) {
    __shared__ float s1[UN_T], s2[UN_T];
    const int t = threadIdx.x;
    const int active = (UN_T / C) * C, rpp = active / C;
    float a = 0.f, b = 0.f;
    if (t < active) {
        const int c = t % C;
        const float mu = mean[c], inv = rsqrtf(var[c] + eps), ga = gamma[c], be = beta[c];
        for (long long r = (long long)blockIdx.x * rpp + t / C; r < M; r += (long long)gridDim.x * rpp) {
            const float xh = (un_ld1(x, r * ldx + c, xbf) - mu) * inv;
            float g = un_ld1(dy, r * ldy + c, dybf);
            if (relu && fmaf(xh, ga, be) <= 0.f) g = 0.f;
            a += g; b = fmaf(g, xh, b);
        }
    }
    s1[t] = a; s2[t] = b;
    __syncthreads();
    if (t < C) {
        float da = 0.f, db = 0.f;
        for (int k = t; k < active; k += C) { da += s1[k]; db += s2[k]; }
        part[(size_t)blockIdx.x * 2 * C + t] = da;
        part[(size_t)blockIdx.x * 2 * C + C + t] = db;
    }
}
__global__ void un_bn_bwd_final_kernel(const float *part, int nparts, int C, float *sums, float *dgamma, float *dbeta,
                                       int accum) {
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double sa = 0., sb = 0.;
    for (int b0 = lane; b0 < nparts; b0 += 256) {   // four partial rows per lane per round trip; same summation order
        float pa[4], pb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int b = b0 + q * 64;
            const size_t o = (size_t)(b < nparts ? b : 0) * 2 * C + c;
            pa[q] = part[o]; pb[q] = part[o + C];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (b0 + q * 64 < nparts) { sa += (double)pa[q]; sb += (double)pb[q]; }
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if (lane != 0) return;
    sums[c] = (float)sa; sums[C + c] = (float)sb;
    if (dbeta) dbeta[c] = (accum ? dbeta[c] : 0.f) + (float)sa;
    if (dgamma) dgamma[c] = (accum ? dgamma[c] : 0.f) + (float)sb;
}
// dx = gamma*inv*(g - mean(g) - xhat*mean(g*xhat)) (+ dx when accum: the second contribution to a residual /
// concatenated gradient is added here instead of in a separate pass)
// Round 3: a thread keeps ONE channel quad and walks rows (UN_AP_U rows in flight): the six per-channel parameters are loaded
// once per thread instead of once per 16 bytes, the row index needs no 64-bit division, and a workgroup streams a contiguous
// row range.  Arithmetic (expressions and their order) unchanged.
#define UN_AP_U 4
__host__ __device__ __forceinline__ int un_ap_rows_per_block(int C) { return (256 / (C >> 2)) * UN_AP_U * 2; }
// workgroups of a row-walking apply kernel: (256 / (C / 4)) rows per pass, `passes` passes per workgroup
static inline int un_ap_grid(long long M, int C, int passes) {
    const long long rows = (long long)(256 / (C >> 2)) * passes;
    return (int)((M + rows - 1) / rows);
}
// raw 16 / 8 bytes of four consecutive gradient elements (fp32 / bf16: GBF), converted where they are used so that the loads of a
// batch of rows are issued back to back (a conversion right behind each load made every load wait for its own data)
template <bool GBF>
__device__ __forceinline__ float4 un_ldraw4(const float *p, long long idx) {
    if (GBF) { const uint2 v = *(const uint2 *)((const unsigned short *)p + idx); return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f); }
    return *(const float4 *)(p + idx);
}
template <bool GBF>
__device__ __forceinline__ void un_cvt4(const float4 r, float *g) {
    if (GBF) {
        const unsigned int a = __float_as_uint(r.x), b = __float_as_uint(r.y);
        g[0] = __uint_as_float(a << 16); g[1] = __uint_as_float(a & 0xFFFF0000u); g[2] = __uint_as_float(b << 16); g[3] = __uint_as_float(b & 0xFFFF0000u);
    } else { g[0] = r.x; g[1] = r.y; g[2] = r.z; g[3] = r.w; }
}
template <bool OBF, bool GBF>
__global__ __launch_bounds__(256) void un_bn_bwd_apply_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ dy, int ldy,
                                       const float *__restrict__ mean, const float *__restrict__ var,
                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                       const float *__restrict__ sums, float *__restrict__ dx, int ldo, long long M,
                                       int C, float eps, int relu, int accum, unsigned short *__restrict__ shadow, int xbf) {
    const int c4 = C >> 2, rpb = 256 / c4, t = threadIdx.x;
    if (t >= rpb * c4) return;
    const int rl = t / c4, c = (t - rl * c4) * 4;
    const float invM = 1.f / (float)M;
    float inv[4], ga[4], mu[4], be[4], mg[4], mgx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        inv[j] = rsqrtf(var[c + j] + eps); ga[j] = gamma[c + j]; mu[j] = mean[c + j]; be[j] = beta[c + j];
        mg[j] = sums[c + j] * invM; mgx[j] = sums[C + c + j] * invM;
    }
    const long long rows = (long long)un_ap_rows_per_block(C);
    const long long r0 = (long long)blockIdx.x * rows, r1 = (r0 + rows < M) ? r0 + rows : M;
    for (long long rb = r0 + rl; rb < r1; rb += (long long)rpb * UN_AP_U) {
        float4 xv[UN_AP_U], gv[UN_AP_U], ov[UN_AP_U];
#pragma unroll
        for (int u = 0; u < UN_AP_U; u++) {
            const long long row = rb + (long long)u * rpb;
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f); gv[u] = xv[u]; ov[u] = xv[u];
            if (row < r1) {
                xv[u] = un_ldx4(x, row * ldx + c, xbf);
                gv[u] = un_ldraw4<GBF>(dy, row * ldy + c);
                if (!OBF && accum) ov[u] = *(const float4 *)(dx + row * ldo + c);
            }
        }
#pragma unroll
        for (int u = 0; u < UN_AP_U; u++) {
            const long long row = rb + (long long)u * rpb;
            if (row >= r1) continue;
            float xi[4];
            un_cvx4(xv[u], xbf, xi);
            float gi[4];
            un_cvt4<GBF>(gv[u], gi);
            const float old[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float xh = (xi[j] - mu[j]) * inv[j];
                float g = gi[j];
                if (relu && fmaf(xh, ga[j], be[j]) <= 0.f) g = 0.f;
                o[j] = old[j] + ga[j] * inv[j] * (g - mg[j] - xh * mgx[j]);
            }
            if (OBF) *(uint2 *)((unsigned short *)dx + row * ldo + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));   // bf16 gradient buffer
            else *(float4 *)(dx + row * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (!OBF && shadow) *(uint2 *)(shadow + row * C + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));   // dense (M, C) bf16 copy
        }
    }
}
// ---- few-row levels (round 4): finalize + apply in ONE launch.
// Below ~16 k rows every kernel of the BatchNorm chain sits at the launch floor (~5 us for microseconds of work, + ~1.5 us per
// dependent boundary): conv -> finalize -> apply was three launches per convolution, 160 finalize launches per step.  Here a
// handful of workgroups each reduce the producer's partial rows themselves (the table is a few hundred KB at most and L2-resident;
// G <= 32 workgroups re-read it) and then normalise their slice of the rows.  The reduction is a fixed tree -- thread (slice j,
// channel c) adds rows j, j + S, ... in fp64, the slices are combined in slice order -- so every workgroup derives bit-identical
// statistics and the result does not depend on G.  Workgroup 0 stores mean / var (the backward reads them) and updates the
// running statistics.
#ifndef UN_FS_T
#define UN_FS_T 256
#endif
#define UN_FS_MAXC 256
#define UN_FS_MAX_PART_FLOATS 65536      // producer partial tables beyond this (2 * nparts * C floats) keep the separate finalize launch
// Per-channel fp64 sums of the producer's partial rows, identical in every workgroup: thread (slice j, channel quad q) adds rows
// j, j + S, ... (16-byte loads, eight rows in flight), the S slices are combined in slice order.  acc: [S][C][2] doubles (16 KB).
__device__ __forceinline__ void un_fs_reduce(const StatSrc &s0, const StatSrc &s1, int C, double *acc, double &sa, double &sb) {
    const int t = threadIdx.x;
    const int C4 = C >> 2, S = UN_FS_T / C4;
    const int q = t % C4, j = t / C4, c = q * 4;
    double a[4] = {0., 0., 0., 0.}, b[4] = {0., 0., 0., 0.};
    if (j < S) {
        const bool first = (c >= s0.c0 && c < s0.c0 + s0.cn);
        const float *part = first ? s0.part : s1.part;
        const int nparts = first ? s0.nparts : s1.nparts, width = first ? s0.width : s1.width, cc = c - (first ? s0.c0 : s1.c0);
        int r = j;
        for (; r + 7 * S < nparts; r += 8 * S) {
            float4 pa[8], pb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const float *o = part + (size_t)(r + u * S) * 2 * width + cc; pa[u] = *(const float4 *)o; pb[u] = *(const float4 *)(o + width); }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a[0] += (double)pa[u].x; a[1] += (double)pa[u].y; a[2] += (double)pa[u].z; a[3] += (double)pa[u].w;
                b[0] += (double)pb[u].x; b[1] += (double)pb[u].y; b[2] += (double)pb[u].z; b[3] += (double)pb[u].w;
            }
        }
        {   // tail: up to seven rows, requested together
            float4 pa[7], pb[7];
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int rr = r + u * S;
                const float *o = part + (size_t)(rr < nparts ? rr : 0) * 2 * width + cc;
                pa[u] = *(const float4 *)o; pb[u] = *(const float4 *)(o + width);
            }
#pragma unroll
            for (int u = 0; u < 7; u++) {
                if (r + u * S < nparts) {
                    a[0] += (double)pa[u].x; a[1] += (double)pa[u].y; a[2] += (double)pa[u].z; a[3] += (double)pa[u].w;
                    b[0] += (double)pb[u].x; b[1] += (double)pb[u].y; b[2] += (double)pb[u].z; b[3] += (double)pb[u].w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { acc[((size_t)j * C + c + k) * 2] = a[k]; acc[((size_t)j * C + c + k) * 2 + 1] = b[k]; }
    }
    __syncthreads();
    sa = 0.; sb = 0.;
    if (t < C) for (int k = 0; k < S; k++) { sa += acc[((size_t)k * C + t) * 2]; sb += acc[((size_t)k * C + t) * 2 + 1]; }
}
// The same sums from the producers' second-level tables (round 5): UN_P2_ROWS rows per source, ONE round trip -- thread `col` of
// the 2 C columns (sum | sum of squares) requests its 16 values together and adds them in row order.  acc: >= 2 C doubles.
__device__ __forceinline__ void un_fs_reduce2(const StatSrc &s0, const StatSrc &s1, int C, double *acc, double &sa, double &sb) {
    const int t = threadIdx.x;
    for (int col = t; col < 2 * C; col += UN_FS_T) {
        const int st = col >= C ? 1 : 0, c = col - st * C;
        const bool first = (c >= s0.c0 && c < s0.c0 + s0.cn);
        const StatSrc &src = first ? s0 : s1;
        const double *p = src.p2 + (size_t)st * src.width + (c - src.c0);
        double v[UN_P2_ROWS];
#pragma unroll
        for (int r = 0; r < UN_P2_ROWS; r++) v[r] = p[(size_t)r * 2 * src.width];
        double a = 0.;
#pragma unroll
        for (int r = 0; r < UN_P2_ROWS; r++) a += v[r];
        acc[col] = a;
    }
    __syncthreads();
    sa = 0.; sb = 0.;
    if (t < C) { sa = acc[t]; sb = acc[C + t]; }
}
template <bool BF16, bool XBF>      // XBF: the input rows are bf16 (a compile-time form: the run-time choice cost 2 us per launch)
__global__ __launch_bounds__(UN_FS_T) void un_bn_fused_small_kernel(StatSrc s0, StatSrc s1, const float *__restrict__ x, int ldx,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                      void *__restrict__ y, int ldy, int M, int C, float eps, int relu,
                                                                      float *mean_out, float *var_out, float *running_mean, float *running_var,
                                                                      float momentum, int rows_per_block) {
    constexpr int xbf = XBF ? 1 : 0;
    __shared__ double acc[UN_FS_T * 4 * 2];
    __shared__ float4 prm[UN_FS_MAXC];     // (mean, 1/std, gamma, beta)
    const int t = threadIdx.x;
    const int c4 = C >> 2, rpb = UN_FS_T / c4;
    const int rl = t / c4, c = (t - rl * c4) * 4;
    const bool worker = t < rpb * c4;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    // the first rows of this thread do not depend on the statistics: requested before the reduction, they arrive during it
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = r0 + rl + u * rpb;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (worker && row < r1) v[u] = un_ldx4(x, (long long)row * ldx + c, xbf);
    }
    double sa, sb;
    if (s0.p2) un_fs_reduce2(s0, s1, C, acc, sa, sb);
    else un_fs_reduce(s0, s1, C, acc, sa, sb);
    if (t < C) {
        const double m = sa / (double)M;
        double vv = sb / (double)M - m * m;
        if (vv < 0.) vv = 0.;
        prm[t] = make_float4((float)m, rsqrtf((float)vv + eps), gamma[t], beta[t]);
        if (blockIdx.x == 0) {
            mean_out[t] = (float)m; var_out[t] = (float)vv;
            if (running_mean) {
                running_mean[t] = (1.f - momentum) * running_mean[t] + momentum * (float)m;
                running_var[t] = (1.f - momentum) * running_var[t] + momentum * (float)(vv * ((double)M / (double)(M > 1 ? M - 1 : 1)));
            }
        }
    }
    __syncthreads();
    if (!worker) return;
    float inv[4], ga[4], mu[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const float4 p4 = prm[c + j]; mu[j] = p4.x; inv[j] = p4.y; ga[j] = p4.z; be[j] = p4.w; }
    for (int rb = r0 + rl; rb < r1; rb += rpb * 4) {
        if (rb != r0 + rl) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = rb + u * rpb;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < r1) v[u] = un_ldx4(x, (long long)row * ldx + c, xbf);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int row = rb + u * rpb;
            if (row >= r1) continue;
            float in[4];
            un_cvx4(v[u], xbf, in);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float r = fmaf((in[j] - mu[j]) * inv[j], ga[j], be[j]);
                o[j] = (relu && r < 0.f) ? 0.f : r;
            }
            if (BF16) *(uint2 *)((unsigned short *)y + (long long)row * ldy + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));
            else *(float4 *)((float *)y + (long long)row * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
// backward twin: sum g / sum g*xhat from the data gradient's epilogue partials (width C) -> sums, dgamma / dbeta (workgroup 0),
// then dx = gamma*inv*(g - mean(g) - xhat*mean(g*xhat)) for this workgroup's rows (the arithmetic of un_bn_bwd_apply_kernel)
template <bool OBF, bool GBF, bool XBF>
__global__ __launch_bounds__(UN_FS_T) void un_bn_bwd_fused_small_kernel(const float *__restrict__ part, int nparts, const float *__restrict__ x, int ldx,
                                                                          const float *__restrict__ dy, int ldy, const float *__restrict__ mean,
                                                                          const float *__restrict__ var, const float *__restrict__ gamma,
                                                                          const float *__restrict__ beta, float *sums, float *dgamma, float *dbeta,
                                                                          int paccum, float *__restrict__ dx, int ldo, int M, int C, float eps,
                                                                          int relu, int accum, unsigned short *__restrict__ shadow, int rows_per_block,
                                                                          const double *__restrict__ part2) {
    constexpr int xbf = XBF ? 1 : 0;
    __shared__ double acc[UN_FS_T * 4 * 2];
    __shared__ float2 sm[UN_FS_MAXC];
    const int t = threadIdx.x;
    const int c4 = C >> 2, rpb = UN_FS_T / c4;
    const int rl = t / c4, c = (t - rl * c4) * 4;
    const bool worker = dx != nullptr && t < rpb * c4;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 xv[UN_AP_U], gv[UN_AP_U], ov[UN_AP_U];
    auto load = [&](int rb) {
#pragma unroll
        for (int u = 0; u < UN_AP_U; u++) {
            const int row = rb + u * rpb;
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f); gv[u] = xv[u]; ov[u] = xv[u];
            if (worker && row < r1) {
                xv[u] = un_ldx4(x, (long long)row * ldx + c, xbf);
                gv[u] = un_ldraw4<GBF>(dy, (long long)row * ldy + c);
                if (!OBF && accum) ov[u] = *(const float4 *)(dx + (long long)row * ldo + c);
            }
        }
    };
    load(r0 + rl);          // (independent of the sums: in flight during the reduction)
    float pinv[4], pga[4], pmu[4], pbe[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { pinv[j] = 0.f; pga[j] = 0.f; pmu[j] = 0.f; pbe[j] = 0.f; }
    if (worker) {
#pragma unroll
        for (int j = 0; j < 4; j++) { pinv[j] = rsqrtf(var[c + j] + eps); pga[j] = gamma[c + j]; pmu[j] = mean[c + j]; pbe[j] = beta[c + j]; }
    }
    double sa, sb;
    const StatSrc s0{part, nparts, (C + 15) / 16 * 16, 0, C, part2};      // (the data gradient's partial rows are ceil(C / 16) * 16 wide)
    if (part2) un_fs_reduce2(s0, s0, C, acc, sa, sb);
    else un_fs_reduce(s0, s0, C, acc, sa, sb);
    if (t < C) {
        sm[t] = make_float2((float)sa, (float)sb);
        if (blockIdx.x == 0) {
            sums[t] = (float)sa; sums[C + t] = (float)sb;
            if (dbeta) dbeta[t] = (paccum ? dbeta[t] : 0.f) + (float)sa;
            if (dgamma) dgamma[t] = (paccum ? dgamma[t] : 0.f) + (float)sb;
        }
    }
    __syncthreads();
    if (!worker) return;
    const float invM = 1.f / (float)M;
    float mg[4], mgx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { mg[j] = sm[c + j].x * invM; mgx[j] = sm[c + j].y * invM; }
    for (int rb = r0 + rl; rb < r1; rb += rpb * UN_AP_U) {
        if (rb != r0 + rl) load(rb);
#pragma unroll
        for (int u = 0; u < UN_AP_U; u++) {
            const int row = rb + u * rpb;
            if (row >= r1) continue;
            float xi[4];
            un_cvx4(xv[u], xbf, xi);
            float gi[4];
            un_cvt4<GBF>(gv[u], gi);
            const float old[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float xh = (xi[j] - pmu[j]) * pinv[j];
                float g = gi[j];
                if (relu && fmaf(xh, pga[j], pbe[j]) <= 0.f) g = 0.f;
                o[j] = old[j] + pga[j] * pinv[j] * (g - mg[j] - xh * mgx[j]);
            }
            if (OBF) *(uint2 *)((unsigned short *)dx + (long long)row * ldo + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));
            else *(float4 *)(dx + (long long)row * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (!OBF && shadow) *(uint2 *)(shadow + (long long)row * C + c) = make_uint2(un_pack2bf(o[0], o[1]), un_pack2bf(o[2], o[3]));
        }
    }
}
// workgroups / rows per workgroup of the fused kernels: ~16 k elements per workgroup, at most `cap` workgroups (32 for the few-row
// levels; the big levels use up to 512 -- every workgroup re-reads the producer's partial table from L2, so their number is
// what the fusion costs)
static inline void un_fs_grid(int M, int C, int &G, int &rows_per_block, int cap = 32) {
    const long long per = cap > 512 ? (16384ll * 512) / cap : 16384ll;          // (experiments: caps beyond 512 also shrink the slice)
    long long g = ((long long)M * C + per - 1) / per;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    const int rpb = UN_FS_T / (C >> 2);                   // rows per pass
    int rows = (int)((M + g - 1) / g);
    rows = (rows + rpb - 1) / rpb * rpb;
    if (rows < rpb) rows = rpb;
    rows_per_block = rows;
    G = (M + rows - 1) / rows;
}

__global__ void un_add_kernel(float *__restrict__ dst, int ldd, const float *__restrict__ src, int lds, long long M, int C, int copy) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = C >> 2;
    if (e >= M * c4) return;
    const long long row = e / c4;
    const int c = (int)(e - row * c4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!copy) a = *(float4 *)(dst + row * ldd + c);
    const float4 b = *(const float4 *)(src + row * lds + c);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *(float4 *)(dst + row * ldd + c) = a;
}

// Row-split partials of ALL weight gradients of a backward -> dW, one launch (one 6 us reduction per layer otherwise, 69
// of them on the side stream).  Same arithmetic as spconv2.hip's wgrad2_reduce_kernel: 32 elements x 8 split groups per
// workgroup, group sums combined in group order.
struct RedJob { const float *part; float *dW; long long n; int R, accum; long long start; };   // start: first workgroup
__global__ __launch_bounds__(256) void un_wgrad_reduce_batched_kernel(const RedJob *__restrict__ jobs, int njobs) {
    // round 6: four consecutive elements per thread (16-byte loads, four rows in flight: the launch was bound by its ~32 dependent
    // 4-byte round trips per thread -- 0.35 ms per step at ~0.5 TB/s); 128 elements x 8 split groups per workgroup, same summation order
    __shared__ float4 sh[8][32];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].start <= (long long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const RedJob j = jobs[lo];
    const int el = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long e = (((long long)blockIdx.x - j.start) * 32 + el) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e + 3 < j.n && !(j.n & 3)) {
        int r = rg;
        for (; r + 24 < j.R; r += 32) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = *(const float4 *)(j.part + (long long)(r + 8 * u) * j.n + e);
#pragma unroll
            for (int u = 0; u < 4; u++) { v.x += p[u].x; v.y += p[u].y; v.z += p[u].z; v.w += p[u].w; }
        }
        for (; r < j.R; r += 8) { const float4 p = *(const float4 *)(j.part + (long long)r * j.n + e); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    } else if (e < j.n) {
        float *vv = &v.x;
        for (int q = 0; q < 4 && e + q < j.n; q++)
            for (int r = rg; r < j.R; r += 8) vv[q] += j.part[(long long)r * j.n + e + q];
    }
    sh[rg][el] = v;
    __syncthreads();
    if (rg == 0 && e < j.n) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 8; q++) { const float4 t = sh[q][el]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        const float sv[4] = {s.x, s.y, s.z, s.w};
        for (int q = 0; q < 4 && e + q < j.n; q++) j.dW[e + q] = (j.accum ? j.dW[e + q] : 0.f) + sv[q];
    }
}

// ------------------------------------------------------------------------------ the network object
enum { OP_CONV = 1, OP_BNACT = 2, OP_PADCAST = 3, OP_STATS = 4 };
enum { MAP_K1 = 0, MAP_K3 = 1, MAP_DOWN = 2, MAP_UP = 3 };

struct TensorD { int level, C, ld, coff, dtype, buf; };
struct BufD { int level, width, dtype; size_t off, goff; int need_grad; };
struct SrcRef { int op, c0, cn; };
struct OpD {
    int type, in, out, res, w, gamma, beta, rmean, rvar, map, mlevel, K, CinW, stats, relu;
    float eps, momentum;
    // derived
    int Cin, Cout;                    // padded channel counts of the call
    size_t wp_fwd, wp_bwd, part_off, state_off;   // arena offsets (bytes)
    int nparts, partw;
    std::vector<SrcRef> srcs;         // BNACT: producers of the input's batch statistics
    int in_grad_mode;                 // backward: 0 = input needs no gradient, 1 = write, 2 = accumulate
    int res_mode;                     // backward: 0 none, 1 alias (no work), 2 add, 3 copy (first contribution)
    int needs_dgrad_pack;
    int bn_of_in;                     // CONV: index of the BNACT op that produced `in` (-1: none) -> fused BN-backward dgrad
    int fused_by;                     // BNACT: index of the CONV whose dgrad epilogue already did this op's reductions
    size_t bpart_off; int bparts;     // BNACT: gradient-arena offset / count of those partials
    size_t part2_off, bpart2_off;     // second-level fp64 tables (UN_P2_ROWS rows; inside the per-call zeroed regions): producer ops / BNACT backward
    int wg_hazard;                    // CONV: its output gradient buffer is accumulated into in place later in the backward
    size_t wpart_off, wpart_bytes; int wsplits;   // CONV: weight-gradient partials in the gradient arena
    int use_shadow;                   // CONV: reads its output gradient from the bf16 shadow of that (fp32) gradient buffer
    int write_shadow;                 // BNACT: its backward apply also writes the bf16 shadow of the buffer it finalises
};
#define RED_RING 32
struct Net {
    std::vector<TensorD> T;
    std::vector<BufD> B;
    std::vector<OpD> ops;
    int nlevels = 0, nparams = 0, input_needs_grad = 0, out_tensor = -1;
    std::vector<int> rows;
    std::vector<int> galias;          // per buffer: tensor id whose gradient view this buffer's gradient aliases, or -1
    std::vector<int> gbf;             // per buffer: its gradient is stored as bf16 (single producer conv, single BatchNorm consumer)
    std::vector<int> gabf;            // per buffer: the gradient of an activation buffer (one BatchNorm writer, one convolution reader) is stored as bf16
    std::vector<int> gshadow;         // per buffer: width of the bf16 shadow of (a column window of) its fp32 gradient, 0 = none
    std::vector<size_t> gshadow_off;  //   its offset in the gradient arena
    size_t arena_bytes = 0, grad_bytes = 0, ws_bytes = 0, bnscr_off = 0, wgws_off = 0, bnscr_bytes = 0, wgws_bytes = 0;
    bool planned = false;
    bool f32 = false;                 // every buffer fp32: reference-precision program (D3_CONV_F32 kernels, no bf16 gradients)
    size_t cnt_off0 = 0, cnt_bytes = 0, bcnt_off0 = 0, bcnt_bytes = 0;   // the second-level partial tables' regions (zeroed once per call)
    // packing jobs (device copy refreshed when a parameter pointer or the arena moves)
    std::vector<PackJob> jobs;
    PackJob *jobs_dev = nullptr;
    size_t jobs_cap = 0;
    long long pack_total = 0;
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;      // second weight-gradient stream (D3_SIDE2)
    std::vector<hipEvent_t> ev;       // pool
    size_t ev_used = 0;
    // gradient chunks (data-parallel overlap): chunk k of the flat parameter-gradient buffer is complete once the backward has
    // processed op chunk_op[k] (ops run in reverse program order, parameters are registered in program order: a chunk is a
    // contiguous tail range); two events per chunk -- side stream (weight-gradient reductions) and caller's stream
    // (BatchNorm parameter gradients) -- let a collective stream start the chunk's all-reduce while the backward goes on
    int xp_op = -1;                    // index of the PADCAST op of the stem input (there is at most one)
    int xp_tensor = -1;                // the PADCAST op's output tensor (the stem's zero-padded bf16 input), -1: none
    const void *xp_ext = nullptr;      // its values prepared by the caller (d3_net_padcast) for the next forward / backward call, or NULL
    std::vector<const void *> k3_16;   // per level: 16-bit delta form of the k3 table for the next forward / backward call (or NULL)
    std::vector<const void *> ok16;    // per level: the lane table (spconv3.hip) for the next forward / backward call (or NULL)
    std::vector<int> chunk_op;
    std::vector<hipEvent_t> chunk_ev_side, chunk_ev_main;
    RedJob *red_host = nullptr, *red_dev = nullptr;   // pinned staging (ring of RED_RING slots) + device copies of the reduce jobs
    size_t red_cap = 0; int red_flip = 0;
    hipEvent_t next_event() {
        if (ev_used == ev.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; ev.push_back(e); }
        return ev[ev_used++];
    }
};

static int esize(int dtype) { return dtype == 1 ? 2 : 4; }

extern "C" int d3_spconv_fwd2_nparts(int Mout, int K, int Cin, int Cout);
extern "C" int d3_spconv_fwd2(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, const float *res,
                              int ldr, float *part, int Min, int Mout, int K, int Cin, int Cout, int flags, void *stream);
extern "C" int d3_spconv_fwd2_bnbwd(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, float *part,
                                    const float *bnx, int ldbx, const float *mean, const float *var, const float *gamma,
                                    const float *beta, float eps, int relu, int Min, int Mout, int K, int Cin, int Cout,
                                    int flags, void *stream);
extern "C" size_t d3_spconv_wgrad2_ws_bytes(int Min, int Mout, int K, int Cin, int Cout, int flags);
extern "C" int d3_spconv_wgrad2_splits(int Min, int Mout, int K, int Cin, int Cout, int flags);
extern "C" int d3_spconv_wgrad2(const void *x, int ldx, const int *tbl, const void *dy, int ldy, float *dW, int Min, int Mout,
                                int K, int Cin, int Cout, int CinW, int flags, void *ws, size_t ws_bytes, void *stream);

// prog: nops x 16 int64, tensors: nt x 6, bufs: nb x 3 (see d3net_amd/netexec.py, which builds them)
extern "C" void *d3_net_create(const int64_t *prog, int nops, const int64_t *tensors, int ntensors, const int64_t *bufs,
                               int nbufs, int nlevels, int nparams, int input_needs_grad, int out_tensor) {
    Net *n = new Net();
    n->nlevels = nlevels; n->nparams = nparams; n->input_needs_grad = input_needs_grad; n->out_tensor = out_tensor;
    for (int i = 0; i < ntensors; i++) {
        const int64_t *t = tensors + (size_t)i * 6;
        n->T.push_back(TensorD{(int)t[0], (int)t[1], (int)t[2], (int)t[3], (int)t[4], (int)t[5]});
    }
    for (int i = 0; i < nbufs; i++) {
        const int64_t *b = bufs + (size_t)i * 3;
        n->B.push_back(BufD{(int)b[0], (int)b[1], (int)b[2], 0, 0, 1});
    }
    n->f32 = true;
    for (auto &b : n->B) if (b.dtype == 1) n->f32 = false;
    for (int i = 0; i < nops; i++) {
        const int64_t *p = prog + (size_t)i * 16;
        OpD o;
        o.type = (int)p[0]; o.in = (int)p[1]; o.out = (int)p[2]; o.res = (int)p[3];
        o.w = o.gamma = o.beta = o.rmean = o.rvar = -1; o.map = 0; o.mlevel = 0; o.K = 1; o.CinW = 0; o.stats = 0; o.relu = 0;
        o.eps = 0.f; o.momentum = 0.f; o.Cin = o.Cout = 0; o.wp_fwd = o.wp_bwd = o.part_off = o.state_off = 0;
        o.nparts = 0; o.partw = 0; o.in_grad_mode = 0; o.res_mode = 0; o.needs_dgrad_pack = 0;
        o.bn_of_in = -1; o.fused_by = -1; o.bpart_off = 0; o.bparts = 0; o.part2_off = 0; o.bpart2_off = 0; o.wg_hazard = 0; o.wpart_off = 0; o.wpart_bytes = 0; o.wsplits = 1; o.use_shadow = 0; o.write_shadow = 0;
        if (o.type == OP_CONV) {
            o.w = (int)p[4]; o.map = (int)p[5]; o.mlevel = (int)p[6]; o.K = (int)p[7]; o.CinW = (int)p[8]; o.stats = (int)p[9];
            o.Cin = n->T[o.in].C; o.Cout = n->T[o.out].C;
        } else if (o.type == OP_BNACT) {
            o.gamma = (int)p[4]; o.beta = (int)p[5]; o.rmean = (int)p[6]; o.rvar = (int)p[7]; o.relu = (int)p[8];
            int32_t eb = (int32_t)p[9], mb = (int32_t)p[10];
            memcpy(&o.eps, &eb, 4); memcpy(&o.momentum, &mb, 4);
        } else if (o.type == OP_STATS) {
            o.stats = 1;
        }
        if (o.type == OP_PADCAST && n->xp_op < 0 && n->T[(size_t)o.in].buf < 0) { n->xp_op = (int)n->ops.size(); n->xp_tensor = o.out; }
        n->ops.push_back(o);
    }
    // statistics sources of every BNACT: producers whose columns lie inside the BN input's column range
    for (auto &o : n->ops) {
        if (o.type != OP_BNACT) continue;
        const TensorD &ti = n->T[o.in];
        for (size_t j = 0; j < n->ops.size(); j++) {
            const OpD &p = n->ops[j];
            if (!p.stats) continue;
            const TensorD &tp = n->T[p.type == OP_STATS ? p.in : p.out];
            if (tp.buf != ti.buf) continue;
            if (tp.coff >= ti.coff && tp.coff + tp.C <= ti.coff + ti.C) o.srcs.push_back(SrcRef{(int)j, tp.coff - ti.coff, tp.C});
        }
    }
    // backward plan: gradient write / accumulate modes and residual aliases (reverse walk)
    std::vector<int> ginit(n->B.size(), 0);
    int ginit_ext = 0;   // gradient of the external input
    n->galias.assign(n->B.size(), -1);
    if (out_tensor >= 0 && n->T[out_tensor].buf >= 0) ginit[n->T[out_tensor].buf] = 1;   // provided by the caller
    for (int i = (int)n->ops.size() - 1; i >= 0; i--) {
        OpD &o = n->ops[i];
        if (o.type == OP_CONV || o.type == OP_BNACT) {
            const int bi = n->T[o.in].buf;
            bool need = true;
            if (bi < 0) need = n->input_needs_grad != 0;
            else {
                // the producer of a buffer that nobody differentiates (the stem's padded input) needs no gradient
                bool produced_by_padcast = false;
                for (auto &q : n->ops) if (q.type == OP_PADCAST && n->T[q.out].buf == bi) produced_by_padcast = true;
                if (produced_by_padcast) need = false;
            }
            if (!need) o.in_grad_mode = 0;
            else if (bi < 0) { o.in_grad_mode = ginit_ext ? 2 : 1; ginit_ext = 1; }
            else { o.in_grad_mode = ginit[bi] ? 2 : 1; ginit[bi] = 1; }
            if (o.type == OP_CONV && o.in_grad_mode) o.needs_dgrad_pack = 1;
            if (o.type == OP_CONV && o.res >= 0) {
                const TensorD &tr = n->T[o.res];
                const int br = tr.buf;
                if (br >= 0 && !ginit[br] && tr.coff == 0 && tr.C == n->B[br].width) { n->galias[br] = o.out; ginit[br] = 1; o.res_mode = 1; }
                else if (br < 0) { o.res_mode = !n->input_needs_grad ? 0 : (ginit_ext ? 2 : 3); if (n->input_needs_grad) ginit_ext = 1; }
                else { o.res_mode = ginit[br] ? 2 : 3; ginit[br] = 1; }
            }
        }
    }
    for (auto &o : n->ops)
        if (o.type == OP_PADCAST) n->B[n->T[o.out].buf].need_grad = 0;
    // which convolutions' output gradients are written in place after their weight gradient was issued?
    auto groot = [&](int tensor) {
        for (int guard = 0; guard < 1000; guard++) {
            const TensorD &t = n->T[tensor];
            if (t.buf < 0) return -1;
            if (tensor == n->out_tensor) return -2;
            if (n->galias[t.buf] < 0) return t.buf;
            tensor = n->galias[t.buf];
        }
        return -3;
    };
    for (int i = 0; i < (int)n->ops.size(); i++) {
        OpD &o = n->ops[i];
        if (o.type != OP_CONV) continue;
        const int r = groot(o.out);
        if (r < 0) continue;
        for (int j = i - 1; j >= 0 && !o.wg_hazard; j--) {   // ops processed after op i in the backward
            const OpD &q = n->ops[j];
            if ((q.type == OP_CONV || q.type == OP_BNACT) && q.in_grad_mode == 2 && groot(q.in) == r) o.wg_hazard = 1;
            if (q.type == OP_CONV && q.res_mode == 2 && groot(q.res) == r) o.wg_hazard = 1;
        }
    }
    // BN -> ReLU -> conv units: the conv's data gradient does the BatchNorm-backward reductions in its epilogue
    for (size_t i = 0; i < n->ops.size(); i++) {
        OpD &o = n->ops[i];
        if (o.type != OP_CONV || o.in_grad_mode != 1) continue;
        for (size_t j = 0; j < n->ops.size(); j++) {
            OpD &b = n->ops[j];
            if (b.type == OP_BNACT && b.out == o.in && b.fused_by < 0 && (n->T[b.out].dtype == 1 || n->f32)) { o.bn_of_in = (int)j; b.fused_by = (int)i; break; }
        }
    }
    // Gradients with a single writer and a single reader pair are stored as bf16: the output gradient of a convolution whose
    // output feeds exactly one BatchNorm (the first conv of every residual block) is written once by that BatchNorm's backward
    // apply and read only by the convolution's own data- and weight-gradient kernels -- as MFMA operands, i.e. rounded to bf16
    // on load anyway.  Rounding at the store instead is numerically identical and halves the bytes of the data gradient's
    // 27-fold gather and of the weight gradient's dy operand.  D3_GRAD_BF16=0 keeps them in fp32 (A/B measurements).
    n->gbf.assign(n->B.size(), 0);
    {
        const bool on = d3_tune(D3T_GRAD_BF16) != 0 && !n->f32;
        for (size_t b = 0; on && b < n->B.size(); b++) {
            if (n->galias[b] >= 0) continue;
            if (n->out_tensor >= 0 && n->T[n->out_tensor].buf == (int)b) continue;
            bool ok = true;
            for (size_t x = 0; x < n->B.size(); x++)
                if (n->galias[x] >= 0 && n->T[n->galias[x]].buf == (int)b) ok = false;      // a residual gradient aliases into it
            int nprod = 0, ncons = 0;
            for (auto &o : n->ops) {
                if (o.type == OP_PADCAST && n->T[o.out].buf == (int)b) ok = false;
                if (o.type == OP_STATS) continue;
                if (o.type == OP_CONV && o.res >= 0 && n->T[o.res].buf == (int)b) ok = false;
                if ((o.type == OP_CONV || o.type == OP_BNACT) && n->T[o.out].buf == (int)b) {
                    const TensorD &t = n->T[o.out];
                    nprod++;
                    if (o.type != OP_CONV || o.res >= 0 || t.coff != 0 || t.C != n->B[b].width || (t.C & 7)) ok = false;
                }
                if ((o.type == OP_CONV || o.type == OP_BNACT) && n->T[o.in].buf == (int)b) {
                    const TensorD &t = n->T[o.in];
                    ncons++;
                    if (o.type != OP_BNACT || o.in_grad_mode != 1 || t.coff != 0 || t.C != n->B[b].width) ok = false;
                }
            }
            if (ok && nprod == 1 && ncons == 1) n->gbf[b] = 1;
        }
    }
    // The same for the gradient of an ACTIVATION buffer (BatchNorm -> ReLU output, bf16 in the forward): written once by the data
    // gradient of the one convolution that reads the activation, read once by that BatchNorm's backward.  The convolution's epilogue
    // takes the BatchNorm-backward partial sums from the unrounded values and stores bf16 (D3_CONV_OUTBF16); the apply pass reads 2
    // bytes per element instead of 4.  Round 4 measured it neutral (the BatchNorm backward was bound by its reduction of the partial
    // table then) and left it off; with the second-level tables and the lane-table data gradient it is worth 3 - 7 % on both kernels of
    // levels 0 - 1 (rocprofv3, gpurun_out/r06_j34: 20.4 -> 19.4 us and 25.7 -> 24.0 us) and 0.7 GB less traffic per backward: on since round 6.
    n->gabf.assign(n->B.size(), 0);
    {
        const bool on = !n->f32;
        for (size_t b = 0; on && b < n->B.size(); b++) {
            if (n->galias[b] >= 0 || n->gbf[b]) continue;
            if (n->out_tensor >= 0 && n->T[n->out_tensor].buf == (int)b) continue;
            bool ok = n->B[b].dtype == 1;
            for (size_t x = 0; x < n->B.size(); x++)
                if (n->galias[x] >= 0 && n->T[n->galias[x]].buf == (int)b) ok = false;
            int nprod = 0, ncons = 0;
            for (auto &o : n->ops) {
                if (o.type == OP_PADCAST && n->T[o.out].buf == (int)b) ok = false;
                if (o.type == OP_STATS) continue;
                if (o.type == OP_CONV && o.res >= 0 && n->T[o.res].buf == (int)b) ok = false;
                if ((o.type == OP_CONV || o.type == OP_BNACT) && n->T[o.out].buf == (int)b) {
                    const TensorD &t = n->T[o.out];
                    nprod++;
                    if (o.type != OP_BNACT || t.coff != 0 || t.C != n->B[b].width || (t.C & 7)) ok = false;
                }
                if ((o.type == OP_CONV || o.type == OP_BNACT) && n->T[o.in].buf == (int)b) {
                    const TensorD &t = n->T[o.in];
                    ncons++;
                    if (o.type != OP_CONV || o.in_grad_mode != 1 || t.coff != 0 || t.C != n->B[b].width || o.CinW != t.C) ok = false;
                }
            }
            if (ok && nprod == 1 && ncons == 1) n->gabf[b] = 1;
        }
    }
    // Residual-stream gradients stay fp32 (they are accumulated in place and feed fp32 consumers), but the convolution that
    // reads one as ITS output gradient uses it as a bf16 MFMA operand, gathered 27 times per row.  Where the last kernel to
    // touch the buffer before that read is a BatchNorm backward apply over the whole buffer (the first BatchNorm of the next
    // residual block adding its contribution), that kernel also writes the finished values as bf16 into a shadow buffer and
    // the convolution's data- and weight-gradient kernels read the shadow: identical operands, half the gathered bytes.
    n->gshadow.assign(n->B.size(), 0);
    n->gshadow_off.assign(n->B.size(), 0);
    {
        const bool on = d3_tune(D3T_GRAD_BF16) != 0 && !n->f32;
        auto root_of = [&](int tensor, int &coff, int &C) {          // as gptr(): follow residual aliases
            const TensorD *t = &n->T[tensor];
            coff = t->coff; C = t->C;
            for (int guard = 0; guard < 1000; guard++) {
                if (t->buf < 0) return -1;
                if (tensor == n->out_tensor) return -2;
                if (n->galias[t->buf] < 0) return t->buf;
                tensor = n->galias[t->buf];
                t = &n->T[tensor];
                coff = t->coff;
            }
            return -3;
        };
        // root buffer -> (BNACT op whose apply was the last writer of the buffer, column window it wrote) or op -1.  A window is
        // the whole buffer (residual streams) or one column half of a concatenation buffer's gradient (the skip connection:
        // its last writer is the BatchNorm in front of the stride-2 convolution, accumulating into the left half only).
        struct LastBn { int op, coff, C; };
        std::map<int, LastBn> last_bn;
        for (int i = (int)n->ops.size() - 1; on && i >= 0; i--) {
            OpD &o = n->ops[i];
            int coff, C;
            if (o.type == OP_CONV) {
                const int ro = root_of(o.out, coff, C);
                if (ro >= 0 && !n->gbf[ro] && !(C & 7)) {
                    auto it = last_bn.find(ro);
                    if (it != last_bn.end() && it->second.op >= 0 && it->second.coff == coff && it->second.C == C &&
                        (!n->gshadow[ro] || n->gshadow[ro] == C)) {       // (one shadow window per buffer)
                        o.use_shadow = 1; n->ops[it->second.op].write_shadow = 1; n->gshadow[ro] = C;
                    }
                }
                if (o.in_grad_mode) { const int ri = root_of(o.in, coff, C); if (ri >= 0) last_bn[ri] = LastBn{-1, 0, 0}; }
                if (o.res >= 0 && o.res_mode >= 2) { const int rr = root_of(o.res, coff, C); if (rr >= 0) last_bn[rr] = LastBn{-1, 0, 0}; }
            } else if (o.type == OP_BNACT && o.in_grad_mode) {
                const int ri = root_of(o.in, coff, C);
                if (ri >= 0) last_bn[ri] = n->gbf[ri] ? LastBn{-1, 0, 0} : LastBn{i, coff, C};
            }
        }
    }
    {   // weight gradients run on their own (plain-priority) streams.  A lowest-priority stream won 0.3 ms in round 2 and was starved once
        // the process owned more streams in round 3 (18.3 -> 21.7 ms per step): plain it is
        hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&n->side2, hipStreamNonBlocking);
    }
    return n;
}

extern "C" void d3_net_destroy(void *h) {
    Net *n = (Net *)h;
    if (!n) return;
    if (n->jobs_dev) hipFree(n->jobs_dev);
    if (n->red_host) hipHostFree(n->red_host);
    if (n->red_dev) hipFree(n->red_dev);
    for (auto e : n->ev) hipEventDestroy(e);
    for (auto e : n->chunk_ev_side) hipEventDestroy(e);
    for (auto e : n->chunk_ev_main) hipEventDestroy(e);
    if (n->side) hipStreamDestroy(n->side);
    if (n->side2) hipStreamDestroy(n->side2);
    delete n;
}

// op_idx[k] (descending): the backward has finished chunk k's parameters when it has processed op op_idx[k]
extern "C" int d3_net_set_chunks(void *h, const int *op_idx, int nchunks) {
    Net *n = (Net *)h;
    // every chunk boundary may flush the pending reductions through one slot of the pinned staging ring (+ the tail flush and the
    // final one): a slot must not come round again while its host-to-device copy can still be queued, i.e. within two backward
    // calls -- 2 * (nchunks + 2) <= RED_RING (64 chunks wrapped the ring inside ONE backward)
    if (!n || nchunks < 0 || 2 * (nchunks + 2) > RED_RING) return D3_ERR_ARG;
    for (int k = 0; k < nchunks; k++) {
        if (op_idx[k] < 0 || op_idx[k] >= (int)n->ops.size() || (k > 0 && op_idx[k] >= op_idx[k - 1])) return D3_ERR_ARG;
    }
    n->chunk_op.assign(op_idx, op_idx + nchunks);
    while ((int)n->chunk_ev_side.size() < nchunks) {
        hipEvent_t a, b;
        D3_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        D3_CHECK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        n->chunk_ev_side.push_back(a); n->chunk_ev_main.push_back(b);
    }
    return 0;
}
extern "C" int d3_net_set_k3_16(void *h, const void *const *k3_16, const int *const *ok16) {
    Net *n = (Net *)h;
    if (!n) return D3_ERR_ARG;
    n->k3_16.assign((size_t)n->nlevels, nullptr);
    n->ok16.assign((size_t)n->nlevels, nullptr);
    if (d3_tune(D3T_KMAP16) != 0 && !n->f32)
        for (int l = 0; l < n->nlevels; l++) {
            if (k3_16 && k3_16[l]) n->k3_16[(size_t)l] = k3_16[l];
            if (ok16 && ok16[l]) n->ok16[(size_t)l] = (const void *)ok16[l];      // (second array: the lane tables, round 6)
        }
    return 0;
}
// Round 5 (input prefetch): the stem's zero-padded bf16 input prepared OUTSIDE the forward -- d3_net_padcast writes it (M rows x
// d3_net_padded_channels() bf16) from the fp32 voxel features, d3_net_set_padded_input hands it to the NEXT d3_net_forward /
// d3_net_backward call (like the 16-bit tables: one call, then dropped), which skips its own PADCAST launch and reads the stem's
// operand from there.  bf16 executors with a stem only (padded channels 0 otherwise).
extern "C" int d3_net_padded_channels(void *h) {
    Net *n = (Net *)h;
    if (!n || n->f32 || n->xp_tensor < 0 || n->T[(size_t)n->xp_tensor].dtype != 1) return 0;
    return n->T[(size_t)n->xp_tensor].C;
}
extern "C" int d3_net_padcast(void *h, const void *input, void *out, long long M, void *stream) {
    D3_CLEAR();
    Net *n = (Net *)h;
    const int Cd = d3_net_padded_channels(h);
    if (!n || Cd <= 0 || !input || !out || M < 0) return D3_ERR_ARG;
    const int Cs = n->T[(size_t)n->ops[(size_t)n->xp_op].in].C;
    const long long total = M * (Cd / 2);
    if (total > 0) un_padcast_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>((const float *)input, (unsigned short *)out, M, Cs, Cd);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_net_set_padded_input(void *h, const void *xp) {
    Net *n = (Net *)h;
    if (!n) return D3_ERR_ARG;
    n->xp_ext = d3_net_padded_channels(h) > 0 ? xp : nullptr;
    return 0;
}
// make `stream` wait until chunk k of the LAST d3_net_backward call is complete (its events were recorded by that call)
extern "C" int d3_net_chunk_wait(void *h, int k, void *stream) {
    Net *n = (Net *)h;
    if (!n || k < 0 || k >= (int)n->chunk_op.size()) return D3_ERR_ARG;
    D3_CHECK(hipStreamWaitEvent(d3_stream(stream), n->chunk_ev_side[k], 0));
    D3_CHECK(hipStreamWaitEvent(d3_stream(stream), n->chunk_ev_main[k], 0));
    return 0;
}

static int bn_blocks2(int M, int C) {
    const int rpp = UN_T / C;
    long long blocks = ((long long)M + rpp - 1) / rpp;
    blocks = (blocks + 15) / 16;
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    return (int)blocks;
}

static void conv_dims(const Net *n, const OpD &o, int &Min, int &Mout) {
    Min = n->rows[n->T[o.in].level]; Mout = n->rows[n->T[o.out].level];
}

// rows[nlevels] -> arena / gradient-arena / workspace sizes; offsets are fixed until the next plan
extern "C" int d3_net_plan(void *h, const int *rows, size_t *arena_bytes, size_t *grad_bytes) {
    Net *n = (Net *)h;
    n->rows.assign(rows, rows + n->nlevels);
    size_t off = 0, goff = 0;
    for (auto &b : n->B) {
        b.off = off; off += d3_align((size_t)n->rows[b.level] * b.width * esize(b.dtype));
        const size_t bi = &b - &n->B[0];
        b.goff = goff; if (b.need_grad) goff += d3_align((size_t)n->rows[b.level] * b.width * ((n->gbf[bi] || n->gabf[bi]) ? 2 : 4));
        if (n->gshadow[bi]) { n->gshadow_off[bi] = goff; goff += d3_align((size_t)n->rows[b.level] * n->gshadow[bi] * 2); }
    }
    size_t bnscr = 0, wgws = 16;
    for (auto &o : n->ops) {
        if (o.type == OP_CONV) {
            int Min, Mout; conv_dims(n, o, Min, Mout);
            const int pf = n->f32 ? D3_CONV_F32 : 0;
            o.wp_fwd = off; off += d3_align(d3_spconv_pack_bytes_ex(o.K, o.Cin, o.Cout, pf));
            o.wp_bwd = off; if (o.needs_dgrad_pack) off += d3_align(d3_spconv_pack_bytes_ex(o.K, o.Cout, o.Cin, pf));
            if (o.stats) {
                o.nparts = d3_spconv_fwd2_nparts_ex(Mout, o.K, o.Cin, o.Cout, pf);
                o.partw = (o.Cout + 15) / 16 * 16;
                o.part_off = off; off += d3_align((size_t)o.nparts * 2 * o.partw * 4);
            }
            const int xstat = ((o.Cin > o.Cout) ? D3_CONV_XSTAT : 0) | (n->T[o.in].dtype == 1 ? D3_CONV_XBF16 : 0) |
                              ((n->T[o.out].buf >= 0 && n->gbf[n->T[o.out].buf]) || o.use_shadow ? D3_CONV_DYBF16 : 0) |
                              (n->f32 ? D3_CONV_F32 : 0);   // as in d3_net_backward
            o.wsplits = d3_spconv_wgrad2_splits(Min, Mout, o.K, o.Cin, o.Cout, xstat);
            o.wpart_bytes = d3_align((size_t)o.wsplits * o.K * o.Cin * o.Cout * 4 + 256);
        } else if (o.type == OP_STATS) {
            const TensorD &t = n->T[o.in];
            o.nparts = bn_blocks2(n->rows[t.level], t.C); o.partw = t.C;
            o.part_off = off; off += d3_align((size_t)o.nparts * 2 * t.C * 4);
        } else if (o.type == OP_BNACT) {
            const TensorD &t = n->T[o.in];
            o.state_off = off; off += d3_align((size_t)4 * t.C * 4);   // mean, var, bwd sums (2C)
            const size_t s = (size_t)bn_blocks2(n->rows[t.level], t.C) * 2 * t.C * 4;
            if (s > bnscr) bnscr = s;
        }
    }
    for (auto &o : n->ops) {
        if (o.type != OP_BNACT || o.fused_by < 0) continue;
        const OpD &cv = n->ops[o.fused_by];
        int Min, Mout; conv_dims(n, cv, Min, Mout);
        o.bparts = d3_spconv_fwd2_nparts_ex(Min, cv.K, cv.Cout, cv.CinW, n->f32 ? D3_CONV_F32 : 0);
        o.bpart_off = goff; goff += d3_align((size_t)o.bparts * 2 * ((cv.CinW + 15) / 16 * 16) * 4);
    }
    for (auto &o : n->ops) if (o.type == OP_CONV) { o.wpart_off = goff; goff += o.wpart_bytes; }
    n->cnt_off0 = off;
    off = d3_align(off);
    // second-level partial tables (round 5) live in the same zeroed-once-per-call region as the ticket counters: no extra fill launch
    for (auto &o : n->ops)
        if ((o.type == OP_CONV || o.type == OP_STATS) && o.nparts > 0) { o.part2_off = off; off += d3_align((size_t)UN_P2_ROWS * 2 * o.partw * 8); }
    n->cnt_bytes = off - n->cnt_off0;
    n->bcnt_off0 = goff;
    goff = d3_align(goff);
    for (auto &o : n->ops) {
        if (o.type != OP_BNACT || o.fused_by < 0) continue;
        const OpD &cv = n->ops[o.fused_by];
        o.bpart2_off = goff; goff += d3_align((size_t)UN_P2_ROWS * 2 * ((cv.CinW + 15) / 16 * 16) * 8);
    }
    n->bcnt_bytes = goff - n->bcnt_off0;
    n->arena_bytes = off;
    n->bnscr_off = goff; goff += d3_align(bnscr); n->bnscr_bytes = bnscr;
    n->wgws_off = goff; goff += d3_align(wgws); n->wgws_bytes = wgws;
    n->grad_bytes = goff;
    n->planned = true;
    *arena_bytes = n->arena_bytes; *grad_bytes = n->grad_bytes;
    return 0;
}

extern "C" long long d3_net_tensor_offset(void *h, int tensor) {
    Net *n = (Net *)h;
    const TensorD &t = n->T[tensor];
    if (t.buf < 0) return -1;
    return (long long)(n->B[t.buf].off + (size_t)t.coff * esize(t.dtype));
}

struct Maps { const int *const *k3; const int *const *child; const int *const *up; };

static void conv_tables(const OpD &o, const Maps &m, const int *&tf, const int *&tb, int &bwd_flip) {
    tf = tb = nullptr; bwd_flip = 0;
    if (o.map == MAP_K3) { tf = tb = m.k3[o.mlevel]; bwd_flip = 1; }
    else if (o.map == MAP_DOWN) { tf = m.child[o.mlevel]; tb = m.up[o.mlevel]; }
    else if (o.map == MAP_UP) { tf = m.up[o.mlevel]; tb = m.child[o.mlevel]; }
}

static inline char *tptr(const Net *n, char *arena, const void *input, int tensor) {
    const TensorD &t = n->T[tensor];
    if (t.buf < 0) return (char *)input;
    if (tensor == n->xp_tensor && n->xp_ext) return (char *)n->xp_ext;      // (prepared outside the call: d3_net_set_padded_input)
    return arena + n->B[t.buf].off + (size_t)t.coff * esize(t.dtype);
}

// The 16-bit tables handed over by d3_net_set_k3_16 serve exactly ONE forward or backward call (a caller that freed
// its coordinate manager and ran again without re-setting them read freed tables): every return path of the two entry points
// drops them, and the thread-local hint of d3_spconv_next_tbl16 with them (an error return taken between setting the hint and
// the convolution that consumes it must not hand the table to an unrelated K = 27 convolution on this thread).
static void net_drop_k3_16(Net *n) {
    if (n) { n->k3_16.assign(n->k3_16.size(), nullptr); n->ok16.assign(n->ok16.size(), nullptr); n->xp_ext = nullptr; }
    d3_spconv_next_tbl16(nullptr, nullptr);
    d3_spconv_next_part2(nullptr);
}
static int net_forward_impl(void *h, const void *const *params, const int *const *k3, const int *const *child,
                            const int *const *up, const void *input, void *arena_, int training, void *stream);
extern "C" int d3_net_forward(void *h, const void *const *params, const int *const *k3, const int *const *child,
                              const int *const *up, const void *input, void *arena_, int training, void *stream) {
    if (!h) return D3_ERR_ARG;
    const int rc = net_forward_impl(h, params, k3, child, up, input, arena_, training, stream);
    net_drop_k3_16((Net *)h);
    return rc;
}
static int net_forward_impl(void *h, const void *const *params, const int *const *k3, const int *const *child,
                            const int *const *up, const void *input, void *arena_, int training, void *stream) {
    D3_CLEAR();
    Net *n = (Net *)h;
    if (!n->planned) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    char *arena = (char *)arena_;
    Maps maps{k3, child, up};
    if (training && n->cnt_bytes) D3_CHECK(hipMemsetAsync(arena + n->cnt_off0, 0, n->cnt_bytes, s));
    const bool use_p2 = training && d3_tune(D3T_BN_PART2) != 0;      // second-level fp64 partial tables (zeroed with the counters above)
    // ---- all weights -> bf16 fragment order, one launch
    {
        std::vector<PackJob> jobs;
        long long start = 0;
        for (auto &o : n->ops) {
            if (o.type != OP_CONV) continue;
            for (int dir = 0; dir < 2; dir++) {
                if (dir == 1 && !o.needs_dgrad_pack) continue;
                PackJob j;
                memset(&j, 0, sizeof(j));   // padding bytes too: the job table is compared with memcmp
                j.W = (const float *)params[o.w];
                j.dst_off = (dir == 0 ? o.wp_fwd : o.wp_bwd);
                j.K = o.K;
                if (dir == 0) { j.Cin = o.Cin; j.S = o.Cin / 8; j.CinW = o.CinW; j.Cout = o.Cout; j.flipk = 0; j.transw = 0; }
                else {   // data gradient: reduction over the layer's Cout, output = the layer's (real) Cin columns
                    j.Cin = o.Cout; j.S = o.Cout / 8; j.CinW = o.Cout; j.Cout = o.CinW; j.flipk = (o.map == MAP_K3) ? 1 : 0; j.transw = 1;
                }
                j.NT = (j.Cout + 15) / 16;
                j.f32 = n->f32 ? 1 : 0;
                j.start = start;
                start += (long long)j.K * j.S * j.NT * 16;
                jobs.push_back(j);
            }
        }
        bool same = jobs.size() == n->jobs.size() && n->jobs_dev != nullptr &&
                    memcmp(jobs.data(), n->jobs.data(), jobs.size() * sizeof(PackJob)) == 0;
        if (!same) {
            if (jobs.size() > n->jobs_cap) {
                if (n->jobs_dev) D3_CHECK(hipFree(n->jobs_dev));
                D3_CHECK(hipMalloc((void **)&n->jobs_dev, jobs.size() * sizeof(PackJob)));
                n->jobs_cap = jobs.size();
            }
            D3_CHECK(hipMemcpyAsync(n->jobs_dev, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, s));
            D3_CHECK(hipStreamSynchronize(s));   // the host vector is about to be replaced
            n->jobs = jobs;
        }
        n->pack_total = start;
        if (start > 0) {
            un_pack_batched_kernel<<<(int)((start + 255) / 256), 256, 0, s>>>(n->jobs_dev, (int)n->jobs.size(), start, arena);
            D3_LAUNCH_CHECK();
        }
    }
    for (auto &o : n->ops) {
        if (o.type == OP_PADCAST) {
            if (n->xp_ext && n->xp_tensor >= 0 && &o == &n->ops[(size_t)n->xp_op]) continue;      // (the caller prepared it: input prefetch)
            const TensorD &to = n->T[o.out];
            const long long M = n->rows[to.level];
            const int Cs = n->T[o.in].C;
            const long long total = M * (to.C / 2);
            if (to.dtype != 1) {
                const long long tot32 = M * to.C;
                if (tot32 > 0) un_padcast_f32_kernel<<<(int)((tot32 + 255) / 256), 256, 0, s>>>((const float *)input, (float *)tptr(n, arena, input, o.out), M, Cs, to.C);
            } else
            if (total > 0) un_padcast_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>((const float *)input, (unsigned short *)tptr(n, arena, input, o.out), M, Cs, to.C);
        } else if (o.type == OP_STATS) {
            if (!training) continue;
            const TensorD &t = n->T[o.in];
            const int M = n->rows[t.level];
            if (M > 0) un_stats_parts_kernel<<<o.nparts, UN_T, 0, s>>>((const float *)tptr(n, arena, input, o.in), t.ld, M, t.C, (float *)(arena + o.part_off),
                                                                       use_p2 ? (double *)(arena + o.part2_off) : nullptr);
        } else if (o.type == OP_CONV) {
            const TensorD &ti = n->T[o.in], &to = n->T[o.out];
            int Min, Mout; conv_dims(n, o, Min, Mout);
            const int *tf, *tb; int flip; conv_tables(o, maps, tf, tb, flip);
            const float *res = nullptr; int ldr = 0;
            if (o.res >= 0) { res = (const float *)tptr(n, arena, input, o.res); ldr = n->T[o.res].ld; }
            float *part = (o.stats && training) ? (float *)(arena + o.part_off) : nullptr;
            if (o.map == MAP_K3 && (size_t)o.mlevel < n->k3_16.size() && (n->k3_16[(size_t)o.mlevel] || n->ok16[(size_t)o.mlevel]))
                d3_spconv_next_tbl16(n->k3_16[(size_t)o.mlevel], use_p2 || !part ? n->ok16[(size_t)o.mlevel] : nullptr);
            if (part && use_p2) d3_spconv_next_part2((double *)(arena + o.part2_off));
            int rc;
            rc = d3_spconv_fwd2(tptr(n, arena, input, o.in), ti.ld, tf, arena + o.wp_fwd, (float *)tptr(n, arena, input, o.out), to.ld,
                                res, ldr, part, Min, Mout, o.K, o.Cin, o.Cout, (ti.dtype == 1 ? D3_CONV_XBF16 : 0) | (n->f32 ? D3_CONV_F32 : 0) | (to.dtype == 1 ? D3_CONV_OUTBF16 : 0), stream);
            if (rc) return rc;
            if (part) o.nparts = d3_spconv_last_nparts();      // (rows actually written: the kernel depends on the tables at hand)
        } else if (o.type == OP_BNACT) {
            const TensorD &ti = n->T[o.in], &to = n->T[o.out];
            const int M = n->rows[ti.level], C = ti.C;
            float *mean = (float *)(arena + o.state_off), *var = mean + C;
            const float *gamma = (const float *)params[o.gamma], *beta = (const float *)params[o.beta];
            const float *use_mean = mean, *use_var = var;
            if (training) {
                if (o.srcs.empty() || o.srcs.size() > 2) return D3_ERR_ARG;
                StatSrc ss[2];
                for (size_t q = 0; q < 2; q++) {
                    const SrcRef &r = o.srcs[q < o.srcs.size() ? q : 0];
                    const OpD &p = n->ops[r.op];
                    ss[q] = StatSrc{(const float *)(arena + p.part_off), p.nparts, p.partw, r.c0, r.cn, use_p2 ? (const double *)(arena + p.part2_off) : nullptr};
                }
                const int fs_rows = d3_tune(D3T_BN_FUSED_ROWS);
                // (with the second-level tables the reduction is 16 rows whatever the producer's grid was: no size limit)
                const long long part_floats = use_p2 ? 0ll : 2ll * ss[0].nparts * ss[0].cn + (o.srcs.size() > 1 ? 2ll * ss[1].nparts * ss[1].cn : 0ll);
                const int fs_big = d3_tune(D3T_BN_FUSED_BIG);
                if (M > 0 && (M <= fs_rows || (fs_big && fs_rows > 0)) && C <= UN_FS_MAXC && part_floats <= UN_FS_MAX_PART_FLOATS) {
                    // statistics + normalisation in one launch (un_bn_fused_small_kernel); big levels: up to 512 workgroups
                    int G, rows_pb; un_fs_grid(M, C, G, rows_pb, M <= fs_rows ? 32 : (fs_big > 1 ? fs_big : 512));
                    float *rm = o.rmean >= 0 ? (float *)params[o.rmean] : nullptr, *rv = o.rvar >= 0 ? (float *)params[o.rvar] : nullptr;
#define UN_FSF(OBFV, XBFV)                                                                                                             \
                    un_bn_fused_small_kernel<OBFV, XBFV><<<G, UN_FS_T, 0, s>>>(ss[0], ss[1], (const float *)tptr(n, arena, input, o.in), ti.ld, gamma, beta, \
                                                                              tptr(n, arena, input, o.out), to.ld, M, C, o.eps, o.relu, mean, var, rm, rv, o.momentum, rows_pb)
                    if (to.dtype == 1) { if (ti.dtype == 1) UN_FSF(true, true); else UN_FSF(true, false); }
                    else { if (ti.dtype == 1) UN_FSF(false, true); else UN_FSF(false, false); }
#undef UN_FSF
                    continue;
                }
                if (M > 0)
                    un_bn_finalize_kernel<<<(C + 3) / 4, 256, 0, s>>>(ss[0], ss[1], M, C, mean, var,
                                                                   o.rmean >= 0 ? (float *)params[o.rmean] : nullptr,
                                                                   o.rvar >= 0 ? (float *)params[o.rvar] : nullptr, o.momentum);
            } else {
                if (o.rmean < 0) return D3_ERR_ARG;
                use_mean = (const float *)params[o.rmean]; use_var = (const float *)params[o.rvar];
            }
            const long long total = (long long)M * (C / 4);
            if (total > 0) {
                if (to.dtype == 1)
                    un_bn_apply_kernel<true><<<un_ap_grid(M, C, 8), 256, 0, s>>>((const float *)tptr(n, arena, input, o.in), ti.ld, use_mean, use_var, gamma, beta,
                                                                                     tptr(n, arena, input, o.out), to.ld, M, C, o.eps, o.relu, ti.dtype == 1 ? 1 : 0);
                else
                    un_bn_apply_kernel<false><<<un_ap_grid(M, C, 8), 256, 0, s>>>((const float *)tptr(n, arena, input, o.in), ti.ld, use_mean, use_var, gamma, beta,
                                                                                      tptr(n, arena, input, o.out), to.ld, M, C, o.eps, o.relu, ti.dtype == 1 ? 1 : 0);
            }
        }
    }
    D3_LAUNCH_CHECK();
    return 0;
}

// gradient view of a tensor: (pointer, row stride); residual aliases are followed to their root
static float *gptr(const Net *n, char *garena, const float *gout, float *gin, int tensor, int &ld, int &root, int *bf16 = nullptr) {
    const TensorD *t = &n->T[tensor];
    if (bf16) *bf16 = 0;
    int coff = t->coff;
    for (int guard = 0; guard < 1000; guard++) {
        if (t->buf < 0) { ld = t->ld; root = -1; return gin; }
        if (tensor == n->out_tensor) { ld = t->ld; root = -2; return (float *)gout; }
        if (n->galias[t->buf] < 0) break;
        tensor = n->galias[t->buf];
        t = &n->T[tensor];
        coff = t->coff;     // the aliased buffer is a whole-buffer view of the target tensor
    }
    ld = n->B[t->buf].width;
    root = t->buf;
    if (n->gbf[t->buf] || n->gabf[t->buf]) {          // (whole-buffer views only: coff == 0)
        if (bf16) *bf16 = 1;
        return (float *)(garena + n->B[t->buf].goff);
    }
    return (float *)(garena + n->B[t->buf].goff) + coff;
}

// gout: gradient of the output tensor (rows x C fp32, dense); pgrads[param]: where to write / accumulate the
// parameter gradient (NULL = frozen); paccum[param] != 0 -> accumulate.  gin: gradient of the external input
// (only when the network was created with input_needs_grad).
static int net_backward_impl(void *h, const void *const *params, const int *const *k3, const int *const *child,
                             const int *const *up, const void *input, void *arena_, void *garena_, const float *gout,
                             float *const *pgrads, const int *paccum, float *gin, void *stream);
extern "C" int d3_net_backward(void *h, const void *const *params, const int *const *k3, const int *const *child,
                               const int *const *up, const void *input, void *arena_, void *garena_, const float *gout,
                               float *const *pgrads, const int *paccum, float *gin, void *stream) {
    if (!h) return D3_ERR_ARG;
    const int rc = net_backward_impl(h, params, k3, child, up, input, arena_, garena_, gout, pgrads, paccum, gin, stream);
    net_drop_k3_16((Net *)h);
    return rc;
}
static int net_backward_impl(void *h, const void *const *params, const int *const *k3, const int *const *child,
                             const int *const *up, const void *input, void *arena_, void *garena_, const float *gout,
                             float *const *pgrads, const int *paccum, float *gin, void *stream) {
    D3_CLEAR();
    Net *n = (Net *)h;
    if (!n->planned) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    char *arena = (char *)arena_, *garena = (char *)garena_;
    Maps maps{k3, child, up};
    n->ev_used = 0;
    // Small networks (ScoreNet: a few thousand rows, every kernel ~5 us) are bound by host API calls: their weight
    // gradients stay on the caller's stream (no events).  Big ones use the side stream; the "a side-stream kernel still
    // reads this gradient buffer" hazard gets an event only for the convolutions whose output gradient is later
    // accumulated into in place (residual aliases: known from the program, OpD::wg_hazard).
    const int side_min_rows = d3_tune(D3T_SIDE_MIN_ROWS);   // (experiments)
    const bool use_side = n->rows[0] >= side_min_rows;
    hipStream_t ws_stream = use_side ? n->side : s;
    std::map<int, hipEvent_t> pending;   // gradient buffer root -> event after its last side-stream reader
    bool side_used = false;
    bool main_wgrads = false;                               // ... and the batched reduction (side stream) is ordered behind them by one event
    auto wait_pending = [&](int root) {
        if (!use_side) return;
        auto it = pending.find(root);
        if (it != pending.end()) { hipStreamWaitEvent(s, it->second, 0); pending.erase(it); }
    };
    if (n->bcnt_bytes) D3_CHECK(hipMemsetAsync(garena + n->bcnt_off0, 0, n->bcnt_bytes, s));
    const bool use_p2 = d3_tune(D3T_BN_PART2) != 0;
    float *bnscr = (float *)(garena + n->bnscr_off);
    std::vector<RedJob> red;
    long long red_blocks = 0;
    // row-split partials -> dW in batched launches on the weight-gradient stream: one when only the last few convolutions of
    // the backward (the first of the network: level 0, small weights) are left, one at the very end -- a single launch at the
    // end left its 137 us exposed behind the stem's weight gradient, after the caller's stream had nothing left to do
    const int side2_mode = (use_side && n->side2) ? d3_tune(D3T_SIDE2) : 0;
    bool side2_dirty = false;                               // side2 holds weight gradients the batched reduction / the join has not been ordered behind yet
    int side2_flip = 0;
    auto join_side2 = [&]() -> int {                        // ws_stream waits for everything enqueued on side2
        if (!side2_dirty) return 0;
        hipEvent_t e = n->next_event();
        if (!e) return D3_ERR_OVERFLOW;
        D3_CHECK(hipEventRecord(e, n->side2));
        D3_CHECK(hipStreamWaitEvent(ws_stream, e, 0));
        side2_dirty = false;
        return 0;
    };
    auto flush_red = [&]() -> int {
        if (red.empty()) return 0;
        { int jrc = join_side2(); if (jrc) return jrc; }
        if (main_wgrads && use_side) {
            hipEvent_t e = n->next_event();
            if (!e) return D3_ERR_OVERFLOW;
            D3_CHECK(hipEventRecord(e, s));
            D3_CHECK(hipStreamWaitEvent(ws_stream, e, 0));
            main_wgrads = false;
        }
        if (red.size() > n->red_cap) {
            D3_CHECK(hipStreamSynchronize(ws_stream));   // (growing: nothing may still read the old tables -- BEFORE they are freed)
            if (n->red_host) hipHostFree(n->red_host);
            if (n->red_dev) hipFree(n->red_dev);
            n->red_cap = red.size() + 16;
            D3_CHECK(hipHostMalloc((void **)&n->red_host, RED_RING * n->red_cap * sizeof(RedJob)));
            D3_CHECK(hipMalloc((void **)&n->red_dev, RED_RING * n->red_cap * sizeof(RedJob)));
        }
        n->red_flip = (n->red_flip + 1) % RED_RING;   // <= nchunks + 2 flushes per backward (d3_net_set_chunks bounds nchunks): a slot comes round again >= 2 backward calls later
        RedJob *hj = n->red_host + (size_t)n->red_flip * n->red_cap, *dj = n->red_dev + (size_t)n->red_flip * n->red_cap;
        memcpy(hj, red.data(), red.size() * sizeof(RedJob));
        D3_CHECK(hipMemcpyAsync(dj, hj, red.size() * sizeof(RedJob), hipMemcpyHostToDevice, ws_stream));
        un_wgrad_reduce_batched_kernel<<<(int)red_blocks, 256, 0, ws_stream>>>(dj, (int)red.size());
        red.clear();
        red_blocks = 0;
        return 0;
    };
    const int flush_tail = d3_tune(D3T_RED_TAIL);
    int tail_idx = -1;
    for (int i = 0, c = 0; i < (int)n->ops.size() && flush_tail > 0; i++)
        if (n->ops[i].type == OP_CONV && pgrads[n->ops[i].w] != nullptr && ++c == flush_tail) { tail_idx = i; break; }
    size_t next_chunk = 0;
    auto chunk_done = [&](int i) -> int {      // op i has been processed: close every chunk that ends here
        while (next_chunk < n->chunk_op.size() && n->chunk_op[next_chunk] >= i) {
            int frc = flush_red(); if (frc) return frc;
            frc = join_side2(); if (frc) return frc;       // (weight gradients written straight to dW on the second side stream)
            D3_CHECK(hipEventRecord(n->chunk_ev_side[next_chunk], ws_stream));
            D3_CHECK(hipEventRecord(n->chunk_ev_main[next_chunk], s));
            next_chunk++;
        }
        return 0;
    };
    for (int i = (int)n->ops.size() - 1; i >= 0; i--) {
        OpD &o = n->ops[i];
        if (i < (int)n->ops.size() - 1) { int crc = chunk_done(i + 1); if (crc) return crc; }
        if (i == tail_idx) { int frc = flush_red(); if (frc) return frc; }
        if (o.type == OP_CONV) {
            const TensorD &ti = n->T[o.in];
            int Min, Mout; conv_dims(n, o, Min, Mout);
            const int *tf, *tb; int flip; conv_tables(o, maps, tf, tb, flip);
            int ldgo, root_o, gobf; float *go = gptr(n, garena, gout, gin, o.out, ldgo, root_o, &gobf);
            float *go32 = go;                      // (the residual add below reads the fp32 buffer)
            const int ldgo32 = ldgo;
            if (o.use_shadow && root_o >= 0 && n->gshadow[root_o]) { go = (float *)(garena + n->gshadow_off[root_o]); gobf = 1; ldgo = n->gshadow[root_o]; }   // dense (M, C) bf16
            const bool op_side = use_side;
            hipEvent_t e1 = nullptr;
            if (pgrads[o.w] != nullptr && op_side) {      // dy is complete here
                e1 = n->next_event();
                if (!e1) return D3_ERR_OVERFLOW;
                D3_CHECK(hipEventRecord(e1, s));
            }
            // data gradient
            if (o.in_grad_mode) {
                int ldgi, root_i, gibf; float *gi = gptr(n, garena, gout, gin, o.in, ldgi, root_i, &gibf);
                const int obf = gibf ? D3_CONV_OUTBF16 : 0;
                if (root_i >= 0) wait_pending(root_i);
                const bool t16 = o.map == MAP_K3 && (size_t)o.mlevel < n->k3_16.size() && (n->k3_16[(size_t)o.mlevel] || n->ok16[(size_t)o.mlevel]);
                if (t16) d3_spconv_next_tbl16(n->k3_16[(size_t)o.mlevel], (use_p2 || o.bn_of_in < 0) ? n->ok16[(size_t)o.mlevel] : nullptr);
                int rc;
                if (o.bn_of_in >= 0) {
                    const OpD &b = n->ops[o.bn_of_in];
                    const TensorD &tx = n->T[b.in];
                    float *mean = (float *)(arena + b.state_off), *var = mean + tx.C;
                    if (use_p2) d3_spconv_next_part2((double *)(garena + b.bpart2_off));
                    rc = d3_spconv_fwd2_bnbwd(go, ldgo, tb, arena + o.wp_bwd, gi, ldgi, (float *)(garena + b.bpart_off),
                                              (const float *)tptr(n, arena, input, b.in), tx.ld, mean, var, (const float *)params[b.gamma],
                                              (const float *)params[b.beta], b.eps, b.relu, Mout, Min, o.K, o.Cout, o.CinW, (gobf ? D3_CONV_XBF16 : 0) | (n->f32 ? D3_CONV_F32 : 0) | obf | (tx.dtype == 1 ? D3_CONV_BNXBF16 : 0), stream);
                    if (!rc) n->ops[o.bn_of_in].bparts = d3_spconv_last_nparts();
                } else {
                    rc = d3_spconv_fwd2(go, ldgo, tb, arena + o.wp_bwd, gi, ldgi, nullptr, 0, nullptr, Mout, Min, o.K, o.Cout, o.CinW,
                                        (o.in_grad_mode == 2 ? D3_CONV_ACCUM : 0) | (gobf ? D3_CONV_XBF16 : 0) | (n->f32 ? D3_CONV_F32 : 0) | obf, stream);
                }
                if (rc) return rc;
            }
            if (o.res_mode >= 2) {
                int ldr, root_r; float *gr = gptr(n, garena, gout, gin, o.res, ldr, root_r);
                if (root_r >= 0) wait_pending(root_r);
                const long long total = (long long)Mout * (o.Cout / 4);
                if (total > 0) un_add_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(gr, ldr, go32, ldgo32, Mout, o.Cout, o.res_mode == 3 ? 1 : 0);
            }
            // weight gradient on the side stream -- enqueued AFTER the data gradient: at the deep levels the caller's stream runs ~10 us
            // kernels and the host is the bottleneck (event pair + plan + launches of the weight gradient took ~20 us of host time in
            // front of every data gradient: profiles/r03_b step gaps); the event is recorded where dy is complete, i.e. before the
            // data gradient, so the side stream loses nothing
            if (pgrads[o.w] != nullptr) {
                if (!op_side) main_wgrads = true;
                hipStream_t wst = ws_stream;                 // the stream of THIS weight gradient
                if (op_side) {
                    if (side2_mode == 2 ? (side2_flip++ & 1) : (side2_mode == 1 && root_o >= 0 && o.wg_hazard)) { wst = n->side2; side2_dirty = true; }
                    D3_CHECK(hipStreamWaitEvent(wst, e1, 0));
                    side_used = true;
                }
                const bool xstat = o.Cin > o.Cout;
                int flags = (ti.dtype == 1 ? D3_CONV_XBF16 : 0) | (paccum[o.w] ? D3_CONV_ACCUM : 0) | (gobf ? D3_CONV_DYBF16 : 0) | (n->f32 ? D3_CONV_F32 : 0);
                const int *tw = tf;
                if (xstat) { flags |= D3_CONV_XSTAT | (flip ? D3_CONV_FLIPK : 0); tw = tb; }
                float *dW = pgrads[o.w];
                // (the stem's x carries zero-padded channels: dW has CinW rows per offset)
                char *wpart = garena + o.wpart_off;
                if (o.map == MAP_K3 && (size_t)o.mlevel < n->k3_16.size() && n->k3_16[(size_t)o.mlevel])
                    d3_spconv_next_tbl16(n->k3_16[(size_t)o.mlevel], nullptr);
                int rc = d3_spconv_wgrad2(tptr(n, arena, input, o.in), ti.ld, tw, go, ldgo, dW, Min, Mout, o.K, o.Cin, o.Cout, o.CinW,
                                          flags | D3_CONV_NOREDUCE, wpart, o.wpart_bytes, (void *)(op_side ? wst : s));
                if (o.wsplits > 1 || paccum[o.w] || n->f32) {   // (a single bf16-path split without accumulation was written to dW directly)
                    RedJob j;
                    memset(&j, 0, sizeof(j));
                    j.part = (const float *)wpart; j.dW = dW; j.n = (long long)o.K * o.CinW * o.Cout; j.R = o.wsplits;
                    j.accum = paccum[o.w] ? 1 : 0; j.start = red_blocks;
                    red_blocks += (j.n + 127) / 128;
                    red.push_back(j);
                }
                if (rc) return rc;
                if (op_side && root_o >= 0 && o.wg_hazard) {
                    //  one event per gradient-buffer root: if an earlier reader of this root on the OTHER side stream is still
                    // pending, this stream waits for it first, so that the recorded event covers both readers
                    auto prev = pending.find(root_o);
                    if (prev != pending.end()) D3_CHECK(hipStreamWaitEvent(wst, prev->second, 0));
                    hipEvent_t e2 = n->next_event();
                    if (!e2) return D3_ERR_OVERFLOW;
                    D3_CHECK(hipEventRecord(e2, wst));
                    pending[root_o] = e2;
                }
            }
        } else if (o.type == OP_BNACT) {
            if (!o.in_grad_mode && pgrads[o.gamma] == nullptr) continue;
            const TensorD &ti = n->T[o.in];
            const int M = n->rows[ti.level], C = ti.C;
            if (M <= 0) continue;
            float *mean = (float *)(arena + o.state_off), *var = mean + C, *sums = var + C;
            const float *gamma = (const float *)params[o.gamma], *beta = (const float *)params[o.beta];
            int ldgo, root_o, gobf; float *go = gptr(n, garena, gout, gin, o.out, ldgo, root_o, &gobf);
            const float *x = (const float *)tptr(n, arena, input, o.in);
            int relu = o.relu;
            const int fs_rows_b = d3_tune(D3T_BN_FUSED_ROWS);
            if (o.fused_by >= 0 && (M <= fs_rows_b || (d3_tune(D3T_BN_FUSED_BIG) && fs_rows_b > 0)) && C <= UN_FS_MAXC &&
                (use_p2 || 2ll * o.bparts * C <= UN_FS_MAX_PART_FLOATS)) {
                // the epilogue partials -> sums / dgamma / dbeta and the input gradient in one launch
                int G, rows_pb; un_fs_grid(M, C, G, rows_pb, M <= fs_rows_b ? 32 : (d3_tune(D3T_BN_FUSED_BIG) > 1 ? d3_tune(D3T_BN_FUSED_BIG) : 512));
                float *gi = nullptr; int ldgi = 0, root_i = -1, gibf = 0;
                if (o.in_grad_mode) {
                    gi = gptr(n, garena, gout, gin, o.in, ldgi, root_i, &gibf);
                    if (root_i >= 0) wait_pending(root_i);
                }
                unsigned short *sh = (o.in_grad_mode && o.write_shadow && root_i >= 0 && n->gshadow[root_i] == C) ? (unsigned short *)(garena + n->gshadow_off[root_i]) : nullptr;
                if (!o.in_grad_mode) G = 1;
#define UN_FSB2(OBFV, GBFV, XBFV, RELU_, ACC_, SH_)                                                                                   \
                un_bn_bwd_fused_small_kernel<OBFV, GBFV, XBFV><<<G, UN_FS_T, 0, s>>>((const float *)(garena + o.bpart_off), o.bparts, x, ti.ld, go, ldgo, mean, \
                                                                              var, gamma, beta, sums, pgrads[o.gamma], pgrads[o.beta], paccum[o.gamma], gi, \
                                                                              ldgi, M, C, o.eps, RELU_, ACC_, SH_, rows_pb, \
                                                                              use_p2 ? (const double *)(garena + o.bpart2_off) : nullptr)
#define UN_FSB(OBFV, GBFV, RELU_, ACC_, SH_) do { if (ti.dtype == 1) UN_FSB2(OBFV, GBFV, true, RELU_, ACC_, SH_); else UN_FSB2(OBFV, GBFV, false, RELU_, ACC_, SH_); } while (0)
                if (gibf) { if (gobf) UN_FSB(true, true, 0, 0, nullptr); else UN_FSB(true, false, 0, 0, nullptr); }
                else if (gobf) UN_FSB(false, true, 0, o.in_grad_mode == 2 ? 1 : 0, sh);
                else UN_FSB(false, false, 0, o.in_grad_mode == 2 ? 1 : 0, sh);
#undef UN_FSB2
#undef UN_FSB
                continue;
            }
            if (o.fused_by >= 0) {   // reductions (and the ReLU mask) done by the consumer conv's data gradient
                un_bn_bwd_final_kernel<<<(C + 3) / 4, 256, 0, s>>>((const float *)(garena + o.bpart_off), o.bparts, C, sums,
                                                               pgrads[o.gamma], pgrads[o.beta], paccum[o.gamma]);
                relu = 0;
            } else {
                const int nb = bn_blocks2(M, C);
                un_bn_bwd_reduce_kernel<<<nb, UN_T, 0, s>>>(x, ti.ld, go, ldgo, mean, var, gamma, beta, M, C, o.eps, o.relu, bnscr, gobf, ti.dtype == 1 ? 1 : 0);
                un_bn_bwd_final_kernel<<<(C + 3) / 4, 256, 0, s>>>(bnscr, nb, C, sums, pgrads[o.gamma], pgrads[o.beta], paccum[o.gamma]);
            }
            if (o.in_grad_mode) {
                int ldgi, root_i, gibf; float *gi = gptr(n, garena, gout, gin, o.in, ldgi, root_i, &gibf);
                if (root_i >= 0) wait_pending(root_i);
                const long long total = (long long)M * (C / 4);
                unsigned short *sh = (o.write_shadow && root_i >= 0 && n->gshadow[root_i] == C) ? (unsigned short *)(garena + n->gshadow_off[root_i]) : nullptr;
#define UN_APB(OBFV, GBFV, ACC_, SH_)                                                                                                     \
                un_bn_bwd_apply_kernel<OBFV, GBFV><<<un_ap_grid(M, C, UN_AP_U * 2), 256, 0, s>>>(x, ti.ld, go, ldgo, mean, var, gamma, beta, sums, gi, ldgi, M, C, \
                                                                                              o.eps, relu, ACC_, SH_, ti.dtype == 1 ? 1 : 0)
                if (gibf) { if (gobf) UN_APB(true, true, 0, nullptr); else UN_APB(true, false, 0, nullptr); }
                else if (gobf) UN_APB(false, true, o.in_grad_mode == 2 ? 1 : 0, sh);
                else UN_APB(false, false, o.in_grad_mode == 2 ? 1 : 0, sh);
#undef UN_APB
            }
        }
    }
    { int frc = flush_red(); if (frc) return frc; }
    { int crc = chunk_done(0); if (crc) return crc; }
    { int jrc = join_side2(); if (jrc) return jrc; }
    // join: the caller's stream waits for the last weight gradient
    if (side_used) {
        hipEvent_t e = n->next_event();
        if (!e) return D3_ERR_OVERFLOW;
        D3_CHECK(hipEventRecord(e, n->side));
        D3_CHECK(hipStreamWaitEvent(s, e, 0));
    }
    D3_LAUNCH_CHECK();
    return 0;
}
