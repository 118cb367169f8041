// attention.hip -- proposal-level multi-head attention core for the listener's match module (gfx950).
//
// Replaces the dense chain of ScaledDotProductAttention.forward between the projections
// (reference: model/transformer/attention.py:61-75):
//     att = q k^T / sqrt(d_k) (+ attention_weights) ; masked_fill(mask == 0, -inf) ; softmax ; att v
// which the reference runs as 5 kernels on materialised (B*C, h, 128, 128) tensors, after replicating the
// pairwise-distance weights and the key masks per description chunk with `.repeat` (model/match_module.py:191-197,
// 324-326).  Here one workgroup owns one (batch item, head): K and V (<= 128 x 32 fp32) live in LDS, each wave
// walks query rows, keeps the 128 scores of a row in registers (2 per lane), does the softmax with wave
// reductions and multiplies by V straight from LDS.  The additive weights are read from the UN-replicated
// (B, h, nq, nk) tensor (index b / bias_div) and the mask from (B, nk).
// Sizes are tiny (0.13 GFLOP per layer at B*C = 32): exact fp32 FMA, no MFMA -- the kernel is launch/latency bound and
// its job is to remove launches and HBM round trips.  Bytes: q,k,v,out once + P written once (kept for backward).
#include "common.h"

#define AT_MAXN 128   // max queries / keys
#define AT_MAXD 32    // max head dim
#define AT_T 256

__device__ __forceinline__ float at_wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float at_wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// q (B,nq,h*dk)  k (B,nk,h*dk)  v (B,nk,h*dv)  bias (B/bias_div,h,nq,nk)|null  mask (B,nk)|null (0 = masked)
// out (B,nq,h*dv)  P (B,h,nq,nk)
__global__ __launch_bounds__(AT_T) void attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, const float *__restrict__ bias,
                                                        const float *__restrict__ mask, float *__restrict__ out,
                                                        float *__restrict__ P, int h, int nq, int nk, int dk, int dv,
                                                        int bias_div, float scale) {
    __shared__ float Ks[AT_MAXN][AT_MAXD + 1];
    __shared__ float Vs[AT_MAXN][AT_MAXD];
    __shared__ float qs[AT_T / 64][AT_MAXD];
    __shared__ float ps[AT_T / 64][AT_MAXN];
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = AT_T / 64;
    for (int e = t; e < nk * dk; e += AT_T) { const int j = e / dk, d = e % dk; Ks[j][d] = k[((size_t)(b * nk + j) * h + head) * dk + d]; }
    for (int e = t; e < nk * dv; e += AT_T) { const int j = e / dv, d = e % dv; Vs[j][d] = v[((size_t)(b * nk + j) * h + head) * dv + d]; }
    __syncthreads();
    const float *brow_base = bias ? bias + ((size_t)(b / bias_div) * h + head) * nq * nk : nullptr;
    const float *mrow = mask ? mask + (size_t)b * nk : nullptr;
    const int j0 = lane, j1 = lane + 64;
    for (int i = wave; i < nq; i += nw) {
        if (lane < dk) qs[wave][lane] = q[((size_t)(b * nq + i) * h + head) * dk + lane];
        __builtin_amdgcn_wave_barrier();
        float s0 = -INFINITY, s1 = -INFINITY;
        if (j0 < nk) {
            float a = 0.f;
            for (int d = 0; d < dk; d++) a = fmaf(qs[wave][d], Ks[j0][d], a);
            a *= scale;
            if (brow_base) a += brow_base[(size_t)i * nk + j0];
            s0 = (mrow && mrow[j0] == 0.f) ? -INFINITY : a;
        }
        if (j1 < nk) {
            float a = 0.f;
            for (int d = 0; d < dk; d++) a = fmaf(qs[wave][d], Ks[j1][d], a);
            a *= scale;
            if (brow_base) a += brow_base[(size_t)i * nk + j1];
            s1 = (mrow && mrow[j1] == 0.f) ? -INFINITY : a;
        }
        const float m = at_wave_max(fmaxf(s0, s1));
        const float e0 = (j0 < nk) ? expf(s0 - m) : 0.f, e1 = (j1 < nk) ? expf(s1 - m) : 0.f;
        const float inv = 1.f / at_wave_sum(e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        float *Prow = P + (((size_t)b * h + head) * nq + i) * nk;
        if (j0 < nk) { Prow[j0] = p0; ps[wave][j0] = p0; }
        if (j1 < nk) { Prow[j1] = p1; ps[wave][j1] = p1; }
        __builtin_amdgcn_wave_barrier();
        // out[i][d] = sum_j p_j V[j][d]: lane = (half, d), halves take alternate keys
        const int d = lane & 31, half = lane >> 5;
        float o = 0.f;
        if (d < dv) for (int j = half; j < nk; j += 2) o = fmaf(ps[wave][j], Vs[j][d], o);
        o += __shfl_xor(o, 32);
        if (half == 0 && d < dv) out[((size_t)(b * nq + i) * h + head) * dv + d] = o;
        __builtin_amdgcn_wave_barrier();
    }
}

// backward, row pass: dS = P * (dP - sum_j P dP) * scale, dq = dS K ; dS is written over P's storage (dSP)
__global__ __launch_bounds__(AT_T) void attn_bwd_rows_kernel(const float *__restrict__ k, const float *__restrict__ v,
                                                             const float *__restrict__ dout, float *__restrict__ P,
                                                             float *__restrict__ dS, float *__restrict__ dq, int h,
                                                             int nq, int nk, int dk, int dv, float scale) {
    __shared__ float Ks[AT_MAXN][AT_MAXD + 1];
    __shared__ float Vs[AT_MAXN][AT_MAXD + 1];
    __shared__ float gs[AT_T / 64][AT_MAXD];
    __shared__ float ds[AT_T / 64][AT_MAXN];
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = AT_T / 64;
    for (int e = t; e < nk * dk; e += AT_T) { const int j = e / dk, d = e % dk; Ks[j][d] = k[((size_t)(b * nk + j) * h + head) * dk + d]; }
    for (int e = t; e < nk * dv; e += AT_T) { const int j = e / dv, d = e % dv; Vs[j][d] = v[((size_t)(b * nk + j) * h + head) * dv + d]; }
    __syncthreads();
    const int j0 = lane, j1 = lane + 64;
    for (int i = wave; i < nq; i += nw) {
        if (lane < dv) gs[wave][lane] = dout[((size_t)(b * nq + i) * h + head) * dv + lane];
        __builtin_amdgcn_wave_barrier();
        const size_t row = (((size_t)b * h + head) * nq + i) * nk;
        float p0 = 0.f, p1 = 0.f, g0 = 0.f, g1 = 0.f;
        if (j0 < nk) { p0 = P[row + j0]; for (int d = 0; d < dv; d++) g0 = fmaf(gs[wave][d], Vs[j0][d], g0); }
        if (j1 < nk) { p1 = P[row + j1]; for (int d = 0; d < dv; d++) g1 = fmaf(gs[wave][d], Vs[j1][d], g1); }
        const float D = at_wave_sum(p0 * g0 + p1 * g1);
        const float d0 = p0 * (g0 - D), d1 = p1 * (g1 - D);   // gradient w.r.t. the pre-softmax score
        if (j0 < nk) { dS[row + j0] = d0; ds[wave][j0] = d0 * scale; }
        if (j1 < nk) { dS[row + j1] = d1; ds[wave][j1] = d1 * scale; }
        __builtin_amdgcn_wave_barrier();
        const int d = lane & 31, half = lane >> 5;
        float o = 0.f;
        if (d < dk) for (int j = half; j < nk; j += 2) o = fmaf(ds[wave][j], Ks[j][d], o);
        o += __shfl_xor(o, 32);
        if (half == 0 && d < dk) dq[((size_t)(b * nq + i) * h + head) * dk + d] = o;
        __builtin_amdgcn_wave_barrier();
    }
}

// backward, column pass: dk[j] = scale * sum_i dS[i][j] q[i], dv[j] = sum_i P[i][j] dout[i]
__global__ __launch_bounds__(AT_T) void attn_bwd_cols_kernel(const float *__restrict__ q, const float *__restrict__ dout,
                                                             const float *__restrict__ P, const float *__restrict__ dS,
                                                             float *__restrict__ dk_, float *__restrict__ dv_, int h,
                                                             int nq, int nk, int dk, int dv, float scale) {
    __shared__ float Qs[AT_MAXN][AT_MAXD];
    __shared__ float Gs[AT_MAXN][AT_MAXD];
    __shared__ float Pt[16][AT_MAXN + 1];    // P[:, jb:jb+16] transposed
    __shared__ float St[16][AT_MAXN + 1];    // dS[:, jb:jb+16] transposed
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = AT_T / 64;
    for (int e = t; e < nq * dk; e += AT_T) { const int i = e / dk, d = e % dk; Qs[i][d] = q[((size_t)(b * nq + i) * h + head) * dk + d]; }
    for (int e = t; e < nq * dv; e += AT_T) { const int i = e / dv, d = e % dv; Gs[i][d] = dout[((size_t)(b * nq + i) * h + head) * dv + d]; }
    const size_t base = ((size_t)b * h + head) * nq * nk;
    for (int jb = 0; jb < nk; jb += 16) {
        __syncthreads();
        for (int e = t; e < nq * 16; e += AT_T) {
            const int i = e >> 4, jj = e & 15;
            const bool ok = jb + jj < nk;
            Pt[jj][i] = ok ? P[base + (size_t)i * nk + jb + jj] : 0.f;
            St[jj][i] = ok ? dS[base + (size_t)i * nk + jb + jj] : 0.f;
        }
        __syncthreads();
        for (int jj = wave; jj < 16 && jb + jj < nk; jj += nw) {
            const int d = lane & 31, half = lane >> 5;
            float ak = 0.f, av = 0.f;
            for (int i = half; i < nq; i += 2) {
                if (d < dk) ak = fmaf(St[jj][i], Qs[i][d], ak);
                if (d < dv) av = fmaf(Pt[jj][i], Gs[i][d], av);
            }
            ak += __shfl_xor(ak, 32); av += __shfl_xor(av, 32);
            const int j = jb + jj;
            if (half == 0 && d < dk) dk_[((size_t)(b * nk + j) * h + head) * dk + d] = ak * scale;
            if (half == 0 && d < dv) dv_[((size_t)(b * nk + j) * h + head) * dv + d] = av;
        }
    }
}

extern "C" int d3_attn_fwd(const float *q, const float *k, const float *v, const float *bias, const float *mask,
                           float *out, float *P, int B, int h, int nq, int nk, int dk, int dv, int bias_div,
                           void *stream) {
    D3_CLEAR();
    if (B <= 0) return 0;
    if (nq > AT_MAXN || nk > AT_MAXN || dk > AT_MAXD || dv > AT_MAXD || nq < 1 || nk < 1 || bias_div < 1) return D3_ERR_ARG;
    const float scale = (float)(1.0 / sqrt((double)dk));
    attn_fwd_kernel<<<B * h, AT_T, 0, d3_stream(stream)>>>(q, k, v, bias, mask, out, P, h, nq, nk, dk, dv, bias_div, scale);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_attn_bwd(const float *q, const float *k, const float *v, const float *P, const float *dout,
                           float *dS, float *dq, float *dk_, float *dv_, int B, int h, int nq, int nk, int dk, int dv,
                           void *stream) {
    D3_CLEAR();
    if (B <= 0) return 0;
    if (nq > AT_MAXN || nk > AT_MAXN || dk > AT_MAXD || dv > AT_MAXD) return D3_ERR_ARG;
    const float scale = (float)(1.0 / sqrt((double)dk));
    hipStream_t s = d3_stream(stream);
    attn_bwd_rows_kernel<<<B * h, AT_T, 0, s>>>(k, v, dout, (float *)P, dS, dq, h, nq, nk, dk, dv, scale);
    attn_bwd_cols_kernel<<<B * h, AT_T, 0, s>>>(q, dout, P, dS, dk_, dv_, h, nq, nk, dk, dv, scale);
    D3_LAUNCH_CHECK();
    return 0;
}
