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
// Two forms: the scalar one below (exact fp32 FMA; round 1, kept behind D3_ATTN_SCALAR=1 as the cross-check) and the MFMA
// form further down, which d3_attn_fwd / d3_attn_bwd launch.  Bytes: q,k,v,out once + P written once (kept for backward).
#include "common.h"
#include <stdlib.h>

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

// ------------------------------------------------------------------------------ MFMA form (round 2)
// The scalar kernels above spend their time on LDS reads (one per FMA): 160 / 131 / 237 us per call at B*C = 32, h = 4,
// 128 x 128 -- 2.7 ms of the listener step (profiles/r02_v).  Same arithmetic on v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulation; only the summation order differs from the scalar form): one workgroup per (batch item, head), one wave
// per 16-query tile (forward / row pass) or 16-key tile (column pass).
//   forward : S = Q K^T as 8 key tiles x 8 MFMAs (A = query rows from memory, B = K rows from LDS), scale + distance bias +
//             key mask on the accumulators, softmax across the 8 tiles and the 16 lanes of a row, P to memory (kept for the
//             backward) and to a wave-private LDS tile, O = P V with V transposed in LDS;
//   row pass: dP = dO V^T the same way, dS = P (dP - sum_j P dP), dq = scale dS K with K transposed in LDS;
//   col pass: dk = scale dS^T Q, dv = P^T dO -- A operands are column tiles of dS / P read straight from memory (16 lanes
//             = 16 consecutive keys of one query row), B operands Q^T / dO^T from LDS.
// Operand layout of v_mfma_f32_16x16x4_f32 as used in hgemm.hip: lane (i = lane & 15, g = lane >> 4) supplies A[row i][k] and
// B[k][col i] for the four k = kb*16 + g*4 + c of a 16-wide k block (one MFMA per c), and receives D[row g*4 + c][col i].
typedef float am_f32x4 __attribute__((ext_vector_type(4)));
#define AM_NW 8
#define AM_RS 36       // row stride (floats) of the row-major K / V images: 128 x 36
#define AM_PS 132      // row stride of the transposed images (32 x 132) and of the wave-private P tiles (16 x 132)
#define AM_LDS_FWD ((AT_MAXN * AM_RS + AT_MAXD * AM_PS + AM_NW * 16 * AM_PS) * 4)
#define AM_LDS_COLS ((2 * AT_MAXD * AM_PS) * 4)

__device__ __forceinline__ float am_grp_max(float v) { for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float am_grp_sum(float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ void am_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// four consecutive channels of row `row` of a (rows, h*dd) tensor at head `head`, zero outside
__device__ __forceinline__ am_f32x4 am_row4(const float *__restrict__ x, int b, int n, int row, int h, int head, int dd, int d0) {
    am_f32x4 r = (am_f32x4){0.f, 0.f, 0.f, 0.f};
    if (row < n) {
        const float *p = x + ((size_t)(b * n + row) * h + head) * dd;
#pragma unroll
        for (int c = 0; c < 4; c++) if (d0 + c < dd) r[c] = p[d0 + c];
    }
    return r;
}
// row-major image R[j][d] (stride AM_RS) and transposed image T[d][j] (stride AM_PS) of a (n, h*dd) operand's head slice
__device__ __forceinline__ void am_stage(const float *__restrict__ x, int b, int n, int h, int head, int dd, float *R, float *T, int t, int nt) {
    for (int e = t; e < AT_MAXN * AT_MAXD; e += nt) {
        const int j = e >> 5, d = e & 31;
        const float val = (j < n && d < dd) ? x[((size_t)(b * n + j) * h + head) * dd + d] : 0.f;
        if (R) R[j * AM_RS + d] = val;
        if (T) T[d * AM_PS + j] = val;
    }
}

__global__ __launch_bounds__(AM_NW * 64) void attn_fwd_mfma_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                   const float *__restrict__ v, const float *__restrict__ bias,
                                                                   const float *__restrict__ mask, float *__restrict__ out,
                                                                   float *__restrict__ P, int h, int nq, int nk, int dk, int dv,
                                                                   int bias_div, float scale) {
    extern __shared__ __attribute__((aligned(16))) float am_sm[];
    float *Ks = am_sm, *Vt = Ks + AT_MAXN * AM_RS, *Pw = Vt + AT_MAXD * AM_PS;
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    am_stage(k, b, nk, h, head, dk, Ks, nullptr, t, AM_NW * 64);
    am_stage(v, b, nk, h, head, dv, nullptr, Vt, t, AM_NW * 64);
    __syncthreads();
    float *Ps = Pw + wave * 16 * AM_PS;
    const int nqt = (nq + 15) >> 4, nkt = (nk + 15) >> 4;
    const float *bbase = bias ? bias + ((size_t)(b / bias_div) * h + head) * nq * nk : nullptr;
    const float *mrow = mask ? mask + (size_t)b * nk : nullptr;
    for (int qt = wave; qt < nqt; qt += AM_NW) {
        am_f32x4 s[8];
#pragma unroll
        for (int kt = 0; kt < 8; kt++) s[kt] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            if (kb * 16 >= dk) break;
            const am_f32x4 a = am_row4(q, b, nq, qt * 16 + i, h, head, dk, kb * 16 + g * 4);
#pragma unroll
            for (int kt = 0; kt < 8; kt++) {
                if (kt >= nkt) break;
                const am_f32x4 bb = *(const am_f32x4 *)&Ks[(kt * 16 + i) * AM_RS + kb * 16 + g * 4];
#pragma unroll
                for (int c = 0; c < 4; c++) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bb[c], s[kt], 0, 0, 0);
            }
        }
        // element (c, kt) of this lane: query qt*16 + g*4 + c, key kt*16 + i
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < 8; kt++) {
            const int key = kt * 16 + i;
            const bool kin = kt < nkt && key < nk && !(mrow && mrow[key < nk ? key : 0] == 0.f);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c;
                float val = s[kt][c] * scale;
                if (bbase && qr < nq && key < nk) val += bbase[(size_t)qr * nk + key];
                val = kin ? val : -INFINITY;
                s[kt][c] = val;
                mx[c] = fmaxf(mx[c], val);
            }
        }
        float sum[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { mx[c] = am_grp_max(mx[c]); sum[c] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < 8; kt++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float e = (kt < nkt && kt * 16 + i < nk) ? expf(s[kt][c] - mx[c]) : 0.f;
                s[kt][c] = e; sum[c] += e;
            }
#pragma unroll
        for (int c = 0; c < 4; c++) sum[c] = 1.f / am_grp_sum(sum[c]);
#pragma unroll
        for (int kt = 0; kt < 8; kt++) {
            const int key = kt * 16 + i;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c;
                const float pv = s[kt][c] * sum[c];
                Ps[(g * 4 + c) * AM_PS + key] = (kt < nkt && key < nk) ? pv : 0.f;
                if (kt < nkt && key < nk && qr < nq) P[(((size_t)b * h + head) * nq + qr) * nk + key] = pv;
            }
        }
        am_wave_sync();
        am_f32x4 o[2] = {(am_f32x4){0.f, 0.f, 0.f, 0.f}, (am_f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            if (kb >= nkt) break;
            const am_f32x4 a = *(const am_f32x4 *)&Ps[i * AM_PS + kb * 16 + g * 4];
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                const am_f32x4 bb = *(const am_f32x4 *)&Vt[(dt * 16 + i) * AM_PS + kb * 16 + g * 4];
#pragma unroll
                for (int c = 0; c < 4; c++) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bb[c], o[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c, d = dt * 16 + i;
                if (qr < nq && d < dv) out[((size_t)(b * nq + qr) * h + head) * dv + d] = o[dt][c];
            }
        am_wave_sync();     // Ps is rewritten by this wave's next tile
    }
}

__global__ __launch_bounds__(AM_NW * 64) void attn_bwd_rows_mfma_kernel(const float *__restrict__ k, const float *__restrict__ v,
                                                                        const float *__restrict__ dout, const float *__restrict__ P,
                                                                        float *__restrict__ dS, float *__restrict__ dq, int h,
                                                                        int nq, int nk, int dk, int dv, float scale) {
    extern __shared__ __attribute__((aligned(16))) float am_sm[];
    float *Vs = am_sm, *Kt = Vs + AT_MAXN * AM_RS, *Pw = Kt + AT_MAXD * AM_PS;
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    am_stage(v, b, nk, h, head, dv, Vs, nullptr, t, AM_NW * 64);
    am_stage(k, b, nk, h, head, dk, nullptr, Kt, t, AM_NW * 64);
    __syncthreads();
    float *Ps = Pw + wave * 16 * AM_PS;
    const int nqt = (nq + 15) >> 4, nkt = (nk + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += AM_NW) {
        am_f32x4 gp[8];
#pragma unroll
        for (int kt = 0; kt < 8; kt++) gp[kt] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            if (kb * 16 >= dv) break;
            const am_f32x4 a = am_row4(dout, b, nq, qt * 16 + i, h, head, dv, kb * 16 + g * 4);
#pragma unroll
            for (int kt = 0; kt < 8; kt++) {
                if (kt >= nkt) break;
                const am_f32x4 bb = *(const am_f32x4 *)&Vs[(kt * 16 + i) * AM_RS + kb * 16 + g * 4];
#pragma unroll
                for (int c = 0; c < 4; c++) gp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bb[c], gp[kt], 0, 0, 0);
            }
        }
        am_f32x4 pp[8];
        float D[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 8; kt++) {
            const int key = kt * 16 + i;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c;
                const float pv = (kt < nkt && key < nk && qr < nq) ? P[(((size_t)b * h + head) * nq + qr) * nk + key] : 0.f;
                pp[kt][c] = pv;
                D[c] += pv * gp[kt][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) D[c] = am_grp_sum(D[c]);
#pragma unroll
        for (int kt = 0; kt < 8; kt++) {
            const int key = kt * 16 + i;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c;
                const float d = pp[kt][c] * (gp[kt][c] - D[c]);        // gradient w.r.t. the pre-softmax score
                Ps[(g * 4 + c) * AM_PS + key] = d * scale;
                if (kt < nkt && key < nk && qr < nq) dS[(((size_t)b * h + head) * nq + qr) * nk + key] = d;
            }
        }
        am_wave_sync();
        am_f32x4 o[2] = {(am_f32x4){0.f, 0.f, 0.f, 0.f}, (am_f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            if (kb >= nkt) break;
            const am_f32x4 a = *(const am_f32x4 *)&Ps[i * AM_PS + kb * 16 + g * 4];
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                const am_f32x4 bb = *(const am_f32x4 *)&Kt[(dt * 16 + i) * AM_PS + kb * 16 + g * 4];
#pragma unroll
                for (int c = 0; c < 4; c++) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bb[c], o[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qr = qt * 16 + g * 4 + c, d = dt * 16 + i;
                if (qr < nq && d < dk) dq[((size_t)(b * nq + qr) * h + head) * dk + d] = o[dt][c];
            }
        am_wave_sync();
    }
}

__global__ __launch_bounds__(AM_NW * 64) void attn_bwd_cols_mfma_kernel(const float *__restrict__ q, const float *__restrict__ dout,
                                                                        const float *__restrict__ P, const float *__restrict__ dS,
                                                                        float *__restrict__ dk_, float *__restrict__ dv_, int h,
                                                                        int nq, int nk, int dk, int dv, float scale) {
    extern __shared__ __attribute__((aligned(16))) float am_sm[];
    float *Qt = am_sm, *Gt = Qt + AT_MAXD * AM_PS;
    const int b = blockIdx.x / h, head = blockIdx.x % h;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    am_stage(q, b, nq, h, head, dk, nullptr, Qt, t, AM_NW * 64);
    am_stage(dout, b, nq, h, head, dv, nullptr, Gt, t, AM_NW * 64);
    __syncthreads();
    const int nqt = (nq + 15) >> 4, nkt = (nk + 15) >> 4;
    const size_t base = ((size_t)b * h + head) * nq * nk;
    for (int jt = wave; jt < nkt; jt += AM_NW) {
        am_f32x4 aK[2] = {(am_f32x4){0.f, 0.f, 0.f, 0.f}, (am_f32x4){0.f, 0.f, 0.f, 0.f}};
        am_f32x4 aV[2] = {(am_f32x4){0.f, 0.f, 0.f, 0.f}, (am_f32x4){0.f, 0.f, 0.f, 0.f}};
        const int key = jt * 16 + i;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            if (kb >= nqt) break;
            float as[4], ap[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {       // A[row = key][k = query]: a column tile of dS / P, 16 consecutive keys per query row
                const int qr = kb * 16 + g * 4 + c;
                const bool ok = qr < nq && key < nk;
                as[c] = ok ? dS[base + (size_t)qr * nk + key] : 0.f;
                ap[c] = ok ? P[base + (size_t)qr * nk + key] : 0.f;
            }
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                const am_f32x4 bq = *(const am_f32x4 *)&Qt[(dt * 16 + i) * AM_PS + kb * 16 + g * 4];
                const am_f32x4 bg = *(const am_f32x4 *)&Gt[(dt * 16 + i) * AM_PS + kb * 16 + g * 4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    aK[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[c], bq[c], aK[dt], 0, 0, 0);
                    aV[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[c], bg[c], aV[dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int kr = jt * 16 + g * 4 + c, d = dt * 16 + i;
                if (kr < nk && d < dk) dk_[((size_t)(b * nk + kr) * h + head) * dk + d] = aK[dt][c] * scale;
                if (kr < nk && d < dv) dv_[((size_t)(b * nk + kr) * h + head) * dv + d] = aV[dt][c];
            }
    }
}

static bool am_scalar() { return d3_tune(D3T_ATTN_SCALAR) == 1; }   // (A/B measurements, tests)
static int am_attrs() {
    static bool done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!done[dev]) {
        D3_CHECK(hipFuncSetAttribute((const void *)attn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AM_LDS_FWD));
        D3_CHECK(hipFuncSetAttribute((const void *)attn_bwd_rows_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AM_LDS_FWD));
        done[dev] = true;
    }
    return 0;
}

extern "C" int d3_attn_fwd(const float *q, const float *k, const float *v, const float *bias, const float *mask,
                           float *out, float *P, int B, int h, int nq, int nk, int dk, int dv, int bias_div,
                           void *stream) {
    D3_CLEAR();
    if (B <= 0) return 0;
    if (nq > AT_MAXN || nk > AT_MAXN || dk > AT_MAXD || dv > AT_MAXD || nq < 1 || nk < 1 || bias_div < 1) return D3_ERR_ARG;
    const float scale = (float)(1.0 / sqrt((double)dk));
    if (am_scalar()) attn_fwd_kernel<<<B * h, AT_T, 0, d3_stream(stream)>>>(q, k, v, bias, mask, out, P, h, nq, nk, dk, dv, bias_div, scale);
    else {
        int rc = am_attrs(); if (rc) return rc;
        attn_fwd_mfma_kernel<<<B * h, AM_NW * 64, AM_LDS_FWD, d3_stream(stream)>>>(q, k, v, bias, mask, out, P, h, nq, nk, dk, dv, bias_div, scale);
    }
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
    if (am_scalar()) {
        attn_bwd_rows_kernel<<<B * h, AT_T, 0, s>>>(k, v, dout, (float *)P, dS, dq, h, nq, nk, dk, dv, scale);
        attn_bwd_cols_kernel<<<B * h, AT_T, 0, s>>>(q, dout, P, dS, dk_, dv_, h, nq, nk, dk, dv, scale);
    } else {
        int rc = am_attrs(); if (rc) return rc;
        attn_bwd_rows_mfma_kernel<<<B * h, AM_NW * 64, AM_LDS_FWD, s>>>(k, v, dout, P, dS, dq, h, nq, nk, dk, dv, scale);
        attn_bwd_cols_mfma_kernel<<<B * h, AM_NW * 64, AM_LDS_COLS, s>>>(q, dout, P, dS, dk_, dv_, h, nq, nk, dk, dv, scale);
    }
    D3_LAUNCH_CHECK();
    return 0;
}
