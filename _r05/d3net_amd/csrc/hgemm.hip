// hgemm.hip -- small-batch fp32 GEMMs on the matrix cores for the proposal-level heads (gfx950).
//
// The speaker / listener heads of D3Net (model/caption_module.py:72-133 the top-down captioner step,
// model/graph_module.py:101-108 the EdgeConv message MLP, model/lang_module.py:51-55 the GRU language encoder) are chains
// of nn.Linear / nn.GRUCell calls on a few dozen rows (batch 32 = 4 scenes x 8 descriptions; 128 proposals per scene).
// The reference issues them through cuBLAS one by one -- ~25 launches per decode step, ~4,000 per training step; the BLAS
// library's kernels for M = 32 take 5-25 us each.  Here ONE kernel family covers every one of them:
//
//     C (M,N) [+]= act( sum_seg A_seg (M,K_seg) . B_seg (N,K_seg)^T + bias[N] + add (M,N) )
//
//   * fp32 in, fp32 accumulate on the matrix cores: v_mfma_f32_16x16x4_f32 is EXACT fp32 (an fmaf chain) at the fp32
//     vector rate (155 TFLOP/s measured, MI355X_MICROARCH.md) -- the heads keep the reference's precision;
//   * up to three K-segments per problem with their own operand pointers: torch.cat([a, b, c], -1) @ W^T never
//     materialises the concatenation (map_topdown: [embedding, hidden_2, target], map_lang: [attended, hidden_1]); a
//     segment's A rows may be gathered through an index vector (the embedding lookup: one-hot x table in the reference,
//     caption_module.py:95-98);
//   * either operand may be "k-major" (element (r,k) at base[k*ld + r]): the same kernel does y = x W^T (forward),
//     dx = dy W (data gradient, B k-major) and dW = dy^T x (weight gradient, both k-major, K = rows) without transposes;
//   * skinny problems (M <= 64: a decode step) give one 16-column tile to a workgroup whose 4 waves split the K loop and
//     are summed through LDS (a 32 x 512 x 512 step has only 32 column tiles: without the split 32 waves would each walk
//     a 128-deep dependent chain); tall problems give every wave its own tile;
//   * up to four independent problems per launch (blockIdx.z): the two GEMMs of a GRU backward share one launch.
// Every lane loads 16 bytes of a row per 16-wide k block (64-byte segments per row across the 4 lane groups) and four
// k blocks are requested before the first MFMA of the batch (the loads are L2 / Infinity-Cache hits: weights of a step
// total ~20 MB and are re-read every step).
// Roofline: these GEMMs are latency / launch bound at M = 32 (0.3 GFLOP per decode step); the batched ones (classifier
// over all time steps: 992 x 512 x 3004) are bound by the fp32 matrix rate.
#include <mutex>
#include "common.h"
#include "prof.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HG_MAXP 4
struct HgBatch { d3_gemm_prob p[HG_MAXP]; };

// 4 consecutive-k values of one operand row for this lane: k = k0 .. k0+3
__device__ __forceinline__ f32x4 hg_load4(const float *base, long long ld, int kmajor, int vec, int k0, int K, bool valid) {
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!valid || k0 >= K) return v;
    if (!kmajor) {
        if (vec && k0 + 3 < K) return *(const f32x4 *)(base + k0);
#pragma unroll
        for (int s = 0; s < 4; s++) if (k0 + s < K) v[s] = base[k0 + s];
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) if (k0 + s < K) v[s] = base[(long long)(k0 + s) * ld];
    }
    return v;
}

// NW: waves per workgroup.  KSPLIT: the NW waves split the k blocks of ONE 16-column tile (skinny problems: the deeper the
// reduction, the more waves -- each wave should need a single batch of loads); otherwise (NW = 4) every wave owns a tile.
template <int RT, bool KSPLIT, int NW>
__global__ __launch_bounds__(NW * 64) void hg_gemm_kernel(const HgBatch batch) {
    __shared__ float red[KSPLIT ? NW * RT * 256 : 1];
    const d3_gemm_prob &p = batch.p[blockIdx.z];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int ctiles = (p.N + 15) >> 4, rgroups = (p.M + RT * 16 - 1) / (RT * 16);
    const int ct = KSPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int rg = blockIdx.y;
    if (rg >= rgroups || (KSPLIT ? ct >= ctiles : (int)blockIdx.x * 4 >= ctiles)) return;
    const bool tile_ok = ct < ctiles;
    f32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int col = ct * 16 + i;
    const bool cvalid = tile_ok && col < p.N;
    int gkb = 0;   // k-block counter over the concatenated segments (K-split assignment)
    for (int s = 0; s < p.nseg; s++) {
        const d3_gemm_seg &sg = p.seg[s];
        const int K = sg.K, nkb = (K + 15) >> 4;
        const int avec = (!sg.a_kmajor && (sg.lda & 3) == 0 && (((size_t)sg.A) & 15) == 0) ? 1 : 0;
        const int bvec = (!sg.b_kmajor && (sg.ldb & 3) == 0 && (((size_t)sg.B) & 15) == 0) ? 1 : 0;
        const float *ab[RT];
        bool av[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int row = (rg * RT + r) * 16 + i;
            av[r] = tile_ok && row < p.M;
            long long ar = av[r] ? (sg.ia ? (long long)sg.ia[row] : (long long)row) : 0;
            ab[r] = sg.a_kmajor ? sg.A + ar : sg.A + ar * sg.lda;
        }
        const float *bb = sg.b_kmajor ? sg.B + (cvalid ? col : 0) : sg.B + (long long)(cvalid ? col : 0) * sg.ldb;
        // this wave's k blocks of the segment: kb = first, first + step, ...
        const int step = KSPLIT ? NW : 1;
        int first = KSPLIT ? ((wave - gkb) & (NW - 1)) : 0;
        gkb += nkb;
#ifndef HG_U16
#define HG_U16 2
#endif
#ifndef HG_U16_RT1
#define HG_U16_RT1 3
#endif
        // (1024-thread workgroups: 128 VGPRs per lane; with ONE row tile a wave's whole share of a 1536-deep reduction -- 6 k blocks --
        // is a two batches of loads (HG_U16_RT1 = 3; 6 spills))
        constexpr int U = RT > 2 ? 2 : (NW == 16 ? (RT == 1 ? HG_U16_RT1 : HG_U16) : 4);
        for (int kb0 = first; kb0 < nkb; kb0 += step * U) {
            f32x4 a[U][RT], b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k0 = (kb0 + u * step) * 16 + g * 4;
                const bool in = kb0 + u * step < nkb;
                b[u] = hg_load4(bb, sg.ldb, sg.b_kmajor, bvec, k0, K, in && cvalid);
#pragma unroll
                for (int r = 0; r < RT; r++) a[u][r] = hg_load4(ab[r], sg.lda, sg.a_kmajor, avec, k0, K, in && av[r]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (kb0 + u * step < nkb) {   // wave-uniform
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int r = 0; r < RT; r++)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][r][q], b[u][q], acc[r], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: D layout col = lane & 15, row = (lane >> 4) * 4 + q
    auto finish = [&](int r, int q, int ln, float v) {
        const int row = (rg * RT + r) * 16 + (ln >> 4) * 4 + q, c = ct * 16 + (ln & 15);
        if (row >= p.M || c >= p.N) return;
        if (p.bias) v += p.bias[c];
        if (p.add) v += p.add[(long long)row * p.ldadd + c];
        if (p.relu && v < 0.f) v = 0.f;
        const long long orow = p.perm_nb > 0 ? (long long)(row % p.perm_nb) * p.perm_s + row / p.perm_nb : (long long)row;
        float *o = p.C + orow * p.ldc + c;
        if (p.gru) {      // GRUCell gate backward on the finished element (d3hip.h; same expressions as topdown.hip's td_gru_bwd_gates_kernel)
            const float tot = p.accum ? *o + v : v;
            const int H = p.gru_H;
            const long long e = (long long)row * H + c;
            float dh = 0.f;
            if (p.g_d0) dh += p.g_d0[(long long)row * p.g_ld0 + c];
            if (p.g_d1) dh += p.g_d1[(long long)row * p.g_ld1 + c];
            dh += tot;
            const float rr = p.g_r[e], zz = p.g_z[e], nv = p.g_n[e];
            const float dn = dh * (1.f - zz), dz = dh * (p.g_hp[(long long)row * p.g_ldh + c] - nv);
            const float dnp = dn * (1.f - nv * nv);
            const float drp = dnp * p.g_ghn[e] * rr * (1.f - rr);
            const float dzp = dz * zz * (1.f - zz);
            const long long og = (long long)row * 3 * H + c, oi = (long long)row * p.g_lddgi + c;
            p.g_dgi[oi] = drp; p.g_dgi[oi + H] = dzp; p.g_dgi[oi + 2 * H] = dnp;
            p.g_dgh[og] = drp; p.g_dgh[og + H] = dzp; p.g_dgh[og + 2 * H] = dnp * rr;
            p.g_dhp[e] = dh * zz;
            return;
        }
        *o = p.accum ? *o + v : v;
    };
    if (KSPLIT) {
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[((wave * RT + r) * 4 + q) * 64 + lane] = acc[r][q];
        __syncthreads();
        for (int e = t; e < RT * 256; e += NW * 64) {
            const int r = e >> 8, q = (e >> 6) & 3, ln = e & 63;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) v += red[((w * RT + r) * 4 + q) * 64 + ln];
            finish(r, q, ln, v);
        }
    } else if (tile_ok) {
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) finish(r, q, lane, acc[r][q]);
    }
}

// Tall problems (the batched ones: classifier over all time steps 992 x 512 x 3004 and its two gradients, feature projections
// over B*K rows, the EdgeConv message MLPs): the wave-per-tile kernel above re-reads A once per 16-column tile and B once per
// 64-row group straight from L2 (480 MB for the classifier forward: L2 bound at ~17 TFLOP/s).  Here a workgroup owns a 64 x 64
// tile of C and stages 32-deep slabs of both operands through LDS (row-major, one 16-byte read per lane feeds four MFMAs),
// the next slab's global loads in flight behind the current slab's MFMAs; each wave
// computes a 32 x 32 quarter (2 x 2 MFMA tiles).  Same operand forms (row gather, k-major, segments) and epilogues.
#define HT_BM 64
#define HT_BN 64
#define HT_BK 32     // two 16-deep k quads per thread and slab: every slab costs one memory round trip, 3 workgroups per CU at 992 x 3004
#define HT_KQ (HT_BK / 16)
#define HT_RP (HT_BK + 4)   // LDS row pitch (floats) of the row-major operand slabs: 16-byte aligned rows, 16 lanes of a k quad on distinct banks
__global__ __launch_bounds__(256) void hg_gemm_tiled_kernel(const HgBatch batch) {
    __shared__ __attribute__((aligned(16))) float As[2][HT_BM][HT_RP], Bs[2][HT_BN][HT_RP];
    const d3_gemm_prob &p = batch.p[blockIdx.z];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * HT_BM, n0 = blockIdx.x * HT_BN;
    if (m0 >= p.M || n0 >= p.N) return;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;        // this wave's quarter
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging assignment: thread -> (row r of the tile, 4 consecutive k) of A and of B: 64 rows x 4 k-quads = 256 threads
    const int sr = t >> 2, sk = (t & 3) * 4;
    f32x4 ra[HT_KQ], rb[HT_KQ];
    auto fetch = [&](const d3_gemm_seg &sg, int kb) {
        const int row = m0 + sr, col = n0 + sr;
        const bool av = row < p.M, bv = col < p.N;
        const long long ar = av ? (sg.ia ? (long long)sg.ia[row] : (long long)row) : 0;
        const float *ab = sg.a_kmajor ? sg.A + ar : sg.A + ar * sg.lda;
        const float *bb = sg.b_kmajor ? sg.B + (bv ? col : 0) : sg.B + (long long)(bv ? col : 0) * sg.ldb;
        const int avec = (!sg.a_kmajor && (sg.lda & 3) == 0 && (((size_t)sg.A) & 15) == 0) ? 1 : 0;
        const int bvec = (!sg.b_kmajor && (sg.ldb & 3) == 0 && (((size_t)sg.B) & 15) == 0) ? 1 : 0;
#pragma unroll
        for (int u = 0; u < HT_KQ; u++) {
            const int k0 = kb * HT_BK + u * 16 + sk;
            ra[u] = hg_load4(ab, sg.lda, sg.a_kmajor, avec, k0, sg.K, av);
            rb[u] = hg_load4(bb, sg.ldb, sg.b_kmajor, bvec, k0, sg.K, bv);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < HT_KQ; u++) { *(f32x4 *)&As[buf][sr][u * 16 + sk] = ra[u]; *(f32x4 *)&Bs[buf][sr][u * 16 + sk] = rb[u]; }
    };
    // flat list of (segment, k slab)
    int nslab = 0;
    for (int s = 0; s < p.nseg; s++) nslab += (p.seg[s].K + HT_BK - 1) / HT_BK;
    auto locate = [&](int slab, int &sidx, int &kb) {
        sidx = 0; kb = slab;
        while (sidx < p.nseg - 1 && kb >= (p.seg[sidx].K + HT_BK - 1) / HT_BK) { kb -= (p.seg[sidx].K + HT_BK - 1) / HT_BK; sidx++; }
    };
    int sidx, kb;
    locate(0, sidx, kb);
    fetch(p.seg[sidx], kb);
    stash(0);
    __syncthreads();
    for (int slab = 0; slab < nslab; slab++) {
        const int buf = slab & 1;
        if (slab + 1 < nslab) { locate(slab + 1, sidx, kb); fetch(p.seg[sidx], kb); }      // in flight behind the MFMAs below
#pragma unroll
        for (int u = 0; u < HT_KQ; u++) {
            // lane (i, g): four consecutive k (u*16 + g*4 ..) of row / column i -- one 16-byte LDS read feeds four MFMAs
            f32x4 a[2], b[2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) a[mt] = *(const f32x4 *)&As[buf][wm + mt * 16 + i][u * 16 + g * 4];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) b[nt] = *(const f32x4 *)&Bs[buf][wn + nt * 16 + i][u * 16 + g * 4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][q], b[nt][q], acc[mt][nt], 0, 0, 0);
        }
        if (slab + 1 < nslab) stash(buf ^ 1);
        __syncthreads();
    }
    // epilogue: D layout col = lane & 15, row = (lane >> 4) * 4 + q
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = m0 + wm + mt * 16 + g * 4 + q, c = n0 + wn + nt * 16 + i;
                if (row >= p.M || c >= p.N) continue;
                float v = acc[mt][nt][q];
                if (p.bias) v += p.bias[c];
                if (p.add) v += p.add[(long long)row * p.ldadd + c];
                if (p.relu && v < 0.f) v = 0.f;
                const long long orow = p.perm_nb > 0 ? (long long)(row % p.perm_nb) * p.perm_s + row / p.perm_nb : (long long)row;
                float *o = p.C + orow * p.ldc + c;
                *o = p.accum ? *o + v : v;
            }
}

// bf16 x 3 form of the tiled kernel (round 4).  The fp32 matrix rate is 1/16 of the bf16 rate, and the tall problems (the
// classifier over all time steps, the listener's projections over 96 x 128 tokens, their gradients) spend their time in
// v_mfma_f32_16x16x4_f32.  Here every fp32 operand is split once, on its way into LDS, into hi = bf16(x) and lo = bf16(x - hi)
// and a product is three v_mfma_f32_16x16x32_bf16 -- lo*hi + hi*lo + hi*hi, fp32 accumulate: the dropped lo*lo term and the
// rounding of lo are ~2^-17 relative per product (fp32 itself: 2^-24), i.e. ~1e-5 relative on a dot product -- two orders below
// the 1e-3 the heads are held to -- at 12 bf16 MFMAs per 64 x 64 x 32 slab and wave instead of 32 fp32 ones (192 vs 1024
// matrix-core cycles).  Measured (profiles/r04): 152 -> 140 us per launch in the joint step, 26 -> 23 us in the speaker step --
// the 64 x 64 tile moves 16 KB per 262 kFLOP slab and is bound by L2 traffic, not by the matrix rate.  NOT adopted: the switch
// D3_HG_BF16X3 is off by default (the heads stay exact fp32); minkowski.set_exact forces it off.
typedef __bf16 hg_bf16x8 __attribute__((ext_vector_type(8)));
#define HT3_RP (HT_BK + 8)     // LDS row pitch in bf16: 80-byte rows, 16-byte aligned k groups
__device__ __forceinline__ unsigned short hg_bf16_rne(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void hg_split4(const f32x4 v, uint2 &hi, uint2 &lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        h[q] = hg_bf16_rne(v[q]);
        l[q] = hg_bf16_rne(v[q] - __uint_as_float((unsigned int)h[q] << 16));
    }
    hi = make_uint2((unsigned int)h[0] | ((unsigned int)h[1] << 16), (unsigned int)h[2] | ((unsigned int)h[3] << 16));
    lo = make_uint2((unsigned int)l[0] | ((unsigned int)l[1] << 16), (unsigned int)l[2] | ((unsigned int)l[3] << 16));
}
__global__ __launch_bounds__(256) void hg_gemm_tiled3_kernel(const HgBatch batch) {
    __shared__ __attribute__((aligned(16))) unsigned short Ah[2][HT_BM][HT3_RP], Al[2][HT_BM][HT3_RP], Bh[2][HT_BN][HT3_RP], Bl[2][HT_BN][HT3_RP];
    const d3_gemm_prob &p = batch.p[blockIdx.z];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * HT_BM, n0 = blockIdx.x * HT_BN;
    if (m0 >= p.M || n0 >= p.N) return;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;        // this wave's quarter
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int sr = t >> 2, sk = (t & 3) * 4;
    f32x4 ra[HT_KQ], rb[HT_KQ];
    auto fetch = [&](const d3_gemm_seg &sg, int kb) {
        const int row = m0 + sr, col = n0 + sr;
        const bool av = row < p.M, bv = col < p.N;
        const long long ar = av ? (sg.ia ? (long long)sg.ia[row] : (long long)row) : 0;
        const float *ab = sg.a_kmajor ? sg.A + ar : sg.A + ar * sg.lda;
        const float *bb = sg.b_kmajor ? sg.B + (bv ? col : 0) : sg.B + (long long)(bv ? col : 0) * sg.ldb;
        const int avec = (!sg.a_kmajor && (sg.lda & 3) == 0 && (((size_t)sg.A) & 15) == 0) ? 1 : 0;
        const int bvec = (!sg.b_kmajor && (sg.ldb & 3) == 0 && (((size_t)sg.B) & 15) == 0) ? 1 : 0;
#pragma unroll
        for (int u = 0; u < HT_KQ; u++) {
            const int k0 = kb * HT_BK + u * 16 + sk;
            ra[u] = hg_load4(ab, sg.lda, sg.a_kmajor, avec, k0, sg.K, av);
            rb[u] = hg_load4(bb, sg.ldb, sg.b_kmajor, bvec, k0, sg.K, bv);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < HT_KQ; u++) {
            uint2 h, l;
            hg_split4(ra[u], h, l);
            *(uint2 *)&Ah[buf][sr][u * 16 + sk] = h; *(uint2 *)&Al[buf][sr][u * 16 + sk] = l;
            hg_split4(rb[u], h, l);
            *(uint2 *)&Bh[buf][sr][u * 16 + sk] = h; *(uint2 *)&Bl[buf][sr][u * 16 + sk] = l;
        }
    };
    int nslab = 0;
    for (int s = 0; s < p.nseg; s++) nslab += (p.seg[s].K + HT_BK - 1) / HT_BK;
    auto locate = [&](int slab, int &sidx, int &kb) {
        sidx = 0; kb = slab;
        while (sidx < p.nseg - 1 && kb >= (p.seg[sidx].K + HT_BK - 1) / HT_BK) { kb -= (p.seg[sidx].K + HT_BK - 1) / HT_BK; sidx++; }
    };
    int sidx, kb;
    locate(0, sidx, kb);
    fetch(p.seg[sidx], kb);
    stash(0);
    __syncthreads();
    for (int slab = 0; slab < nslab; slab++) {
        const int buf = slab & 1;
        if (slab + 1 < nslab) { locate(slab + 1, sidx, kb); fetch(p.seg[sidx], kb); }      // in flight behind the MFMAs below
        // lane (i, g): eight consecutive k (g * 8 ..) of row / column i -- the whole 32-deep slab is ONE bf16 MFMA per tile
        hg_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            ah[mt] = *(const hg_bf16x8 *)&Ah[buf][wm + mt * 16 + i][g * 8];
            al[mt] = *(const hg_bf16x8 *)&Al[buf][wm + mt * 16 + i][g * 8];
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            bh[nt] = *(const hg_bf16x8 *)&Bh[buf][wn + nt * 16 + i][g * 8];
            bl[nt] = *(const hg_bf16x8 *)&Bl[buf][wn + nt * 16 + i][g * 8];
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        if (slab + 1 < nslab) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = m0 + wm + mt * 16 + g * 4 + q, c = n0 + wn + nt * 16 + i;
                if (row >= p.M || c >= p.N) continue;
                float v = acc[mt][nt][q];
                if (p.bias) v += p.bias[c];
                if (p.add) v += p.add[(long long)row * p.ldadd + c];
                if (p.relu && v < 0.f) v = 0.f;
                const long long orow = p.perm_nb > 0 ? (long long)(row % p.perm_nb) * p.perm_s + row / p.perm_nb : (long long)row;
                float *o = p.C + orow * p.ldc + c;
                *o = p.accum ? *o + v : v;
            }
}

static int hg_check(const d3_gemm_prob &p) {
    if (p.nseg < 1 || p.nseg > 3 || p.M < 0 || p.N < 1 || !p.C) return D3_ERR_ARG;
    if (p.gru && (p.N != p.gru_H || p.perm_nb > 0 || p.relu || !p.g_r || !p.g_z || !p.g_n || !p.g_ghn || !p.g_hp || !p.g_dgi ||
                  !p.g_dgh || !p.g_dhp)) return D3_ERR_ARG;      // (the gate epilogue lives in the decode-step kernels only)
    for (int s = 0; s < p.nseg; s++)
        if (!p.seg[s].A || !p.seg[s].B || p.seg[s].K < 1) return D3_ERR_ARG;
    return 0;
}

// host-side launcher shared with topdown.hip / edgeconv.hip (C++ linkage; the C entry point is d3_hgemm)
// kernel class of ONE problem: 0 = decode step (M <= 32), 1 = few output tiles (K split over the waves of a workgroup),
// 2 = tall (64 x 64 LDS-tiled kernel)
static int hg_class(const d3_gemm_prob &p) {
    if (p.M <= 32) return 0;
    const long long tiles16 = (long long)((p.N + 15) / 16) * ((p.M + 15) / 16);
    return tiles16 < 2048 ? 1 : 2;
}

// ---- deep reductions over few output tiles (round 4): K split over WORKGROUPS.
// dW = dy^T x of the listener's projections is 128 x 128 (or 768 x 300) over K = 4,096 ... 12,288 rows (the split serves K >= 8,192): 32 - 456 workgroups whose
// 16 waves each walk 256 - 768 k.  The reduction is cut into HG_KS slices launched as ONE batch (one problem per slice, raw
// partial outputs in a library-owned scratch buffer), and a second small launch adds the slices in slice order and applies the
// epilogue (bias / add / ReLU / accumulate / row permutation): deterministic, 73.8 -> ~25 us at 128 x 128 x 12,288.
#define HG_KS 4
#define HG_DECLINED 1000000
#define HG_KS_MINK 8192          // measured in-process: K = 12,288 (joint) -0.55 ms / step, K = 4,096 (listener) +0.2 ms (the second launch costs more than the slices save)
// grow-only, one per (device, stream) that ever ran a split: torch's default stream is handle 0 on EVERY device, so the stream
// alone does not identify the buffer.  The buffer lives in the stream's own order (hipMallocAsync / hipFreeAsync on
// the launch stream): growing needs no host synchronisation, and the kernels already enqueued on the stream finish with the old
// buffer before it is released.  The table's mutex is held until this call's launches are enqueued, so a concurrent grow on
// another thread cannot slip between reading the pointer and using it.
struct HgScratch { int dev; hipStream_t s; float *p; size_t floats; };
static HgScratch g_hg_scr[32];
static int g_hg_nscr = 0;
static std::mutex g_hg_scr_mu;
__global__ void hg_splitk_reduce_kernel(const float *__restrict__ part, d3_gemm_prob p, long long slice) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)p.M * p.N) return;
    const int row = (int)(e / p.N), c = (int)(e - (long long)row * p.N);
    float v = 0.f;
#pragma unroll
    for (int z = 0; z < HG_KS; z++) v += part[z * slice + e];
    if (p.bias) v += p.bias[c];
    if (p.add) v += p.add[(long long)row * p.ldadd + c];
    if (p.relu && v < 0.f) v = 0.f;
    const long long orow = p.perm_nb > 0 ? (long long)(row % p.perm_nb) * p.perm_s + row / p.perm_nb : (long long)row;
    float *o = p.C + orow * p.ldc + c;
    *o = p.accum ? *o + v : v;
}
static int hg_launch_batch(const d3_gemm_prob *probs, int nprobs, hipStream_t s);
static int hg_splitk(const d3_gemm_prob &p, hipStream_t s) {
    const long long slice = (long long)p.M * p.N;
    float *g_hg_scratch = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HG_DECLINED;
    std::lock_guard<std::mutex> lk(g_hg_scr_mu);          // (held until the launches below are enqueued)
    {
        HgScratch *e = nullptr;
        for (int i = 0; i < g_hg_nscr; i++) if (g_hg_scr[i].s == s && g_hg_scr[i].dev == dev) e = &g_hg_scr[i];
        if (!e) {
            if (g_hg_nscr == 32) return HG_DECLINED;          // (more streams than slots: the caller keeps the one-workgroup reduction)
            e = &g_hg_scr[g_hg_nscr++];
            e->dev = dev; e->s = s; e->p = nullptr; e->floats = 0;
        }
        if ((size_t)(slice * HG_KS) > e->floats) {
            if (e->p) D3_CHECK(hipFreeAsync(e->p, s));          // (stream-ordered: earlier launches on s still read it safely)
            e->floats = (size_t)(slice * HG_KS) * 2;
            e->p = nullptr;
            D3_CHECK(hipMallocAsync((void **)&e->p, e->floats * sizeof(float), s));
        }
        g_hg_scratch = e->p;
    }
    d3_gemm_prob sub[HG_KS];
    const d3_gemm_seg &sg = p.seg[0];
    const int kper = ((sg.K + HG_KS - 1) / HG_KS + 15) & ~15;          // slices start at multiples of 16 (the kernels' k block)
    for (int z = 0; z < HG_KS; z++) {
        const int k0 = z * kper, k1 = k0 + kper < sg.K ? k0 + kper : sg.K;
        d3_gemm_prob q = p;
        q.nseg = 1; q.bias = nullptr; q.add = nullptr; q.ldadd = 0; q.relu = 0; q.accum = 0; q.perm_nb = 0; q.perm_s = 0;
        q.C = g_hg_scratch + z * slice; q.ldc = p.N;
        q.seg[0].K = k1 > k0 ? k1 - k0 : 0;
        q.seg[0].A = sg.a_kmajor ? sg.A + (long long)k0 * sg.lda : sg.A + k0;
        q.seg[0].B = sg.b_kmajor ? sg.B + (long long)k0 * sg.ldb : sg.B + k0;
        if (q.seg[0].K == 0) { q.seg[0].K = 1; q.M = 0; }          // (empty tail slice: a zero-row problem writes nothing ...)
        sub[z] = q;
    }
    // (... so its scratch slice is cleared instead)
    for (int z = 0; z < HG_KS; z++)
        if (sub[z].M == 0) D3_CHECK(hipMemsetAsync(g_hg_scratch + z * slice, 0, (size_t)slice * sizeof(float), s));
    int rc = hg_launch_batch(sub, HG_KS, s);
    if (rc) return rc;
    hg_splitk_reduce_kernel<<<(int)((slice + 255) / 256), 256, 0, s>>>(g_hg_scratch, p, slice);
    D3_LAUNCH_CHECK();
    return 0;
}
static bool hg_wants_splitk(const d3_gemm_prob &p) {
    const int cap = d3_tune(D3T_HG_SPLITK);          // the switch's value is the largest number of 16 x 16 output tiles
    if (cap <= 0 || p.nseg != 1 || p.seg[0].ia || p.M <= 32) return false;
    const long long tiles16 = (long long)((p.N + 15) / 16) * ((p.M + 15) / 16);
    return p.seg[0].K >= HG_KS_MINK && tiles16 <= cap;
}

int hg_launch(const d3_gemm_prob *probs, int nprobs, hipStream_t s) {
    if (nprobs < 1 || nprobs > HG_MAXP) return D3_ERR_ARG;
    // deep few-tile problems leave the batch and run K-split over workgroups
    bool any = false;
    for (int i = 0; i < nprobs; i++) any |= (probs[i].M > 0 && hg_check(probs[i]) == 0 && hg_wants_splitk(probs[i]));
    if (!any) return hg_launch_batch(probs, nprobs, s);
    d3_gemm_prob rest[HG_MAXP];
    int n = 0;
    for (int i = 0; i < nprobs; i++) {
        bool done = false;
        if (probs[i].M > 0 && hg_check(probs[i]) == 0 && hg_wants_splitk(probs[i])) {
            const int rc = hg_splitk(probs[i], s);
            if (rc != 0 && rc != HG_DECLINED) return rc;
            done = rc == 0;
        }
        if (!done) rest[n++] = probs[i];
    }
    return n > 0 ? hg_launch_batch(rest, n, s) : 0;
}
static int hg_launch_batch(const d3_gemm_prob *probs, int nprobs, hipStream_t s) {
    if (nprobs < 1 || nprobs > HG_MAXP) return D3_ERR_ARG;
    // Round 4: a batch is launched with the kernel its LARGEST problem asks for, which is right for the captioner's homogeneous
    // batches and very wrong for the listener's backward pairs: (dx = dy W: 4096 x 128 x 128, tall) batched with
    // (dW = dy^T x: 128 x 128 x 4096) ran the weight gradient on FOUR workgroups of the tiled kernel walking 128 (joint step: 384)
    // k slabs one after the other -- 212 us (643 us) per launch, 14 % of the listener step.  A mixed batch is split by class:
    // the deep, few-tile problems go to the K-split kernel (128 x 128 x 4096: ~17 us).
    if (nprobs > 1 && d3_tune(D3T_HG_CLASS_SPLIT) != 0) {
        int cls[HG_MAXP], first = -1;
        bool mixed = false;
        for (int i = 0; i < nprobs; i++) {
            int rc = hg_check(probs[i]);
            if (rc) return rc;
            cls[i] = probs[i].M > 0 ? hg_class(probs[i]) : -1;
            if (cls[i] < 0) continue;
            if (first < 0) first = cls[i]; else if (cls[i] != first) mixed = true;
        }
        if (mixed) {
            for (int c = 0; c < 3; c++) {
                d3_gemm_prob sub[HG_MAXP];
                int n = 0;
                for (int i = 0; i < nprobs; i++) if (cls[i] == c) sub[n++] = probs[i];
                if (n > 0) { int rc = hg_launch(sub, n, s); if (rc) return rc; }
            }
            return 0;
        }
    }
    HgBatch b;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < nprobs; i++) {
        int rc = hg_check(probs[i]);
        if (rc) return rc;
        b.p[i] = probs[i];
        if (probs[i].M > maxM) maxM = probs[i].M;
        if (probs[i].N > maxN) maxN = probs[i].N;
    }
    for (int i = nprobs; i < HG_MAXP; i++) b.p[i] = probs[0];
    if (maxM == 0) return 0;
    for (int i = 0; i < nprobs; i++)      // (the 64 x 64 tiled kernels do not carry the gate epilogue: only hg_gemm_kernel's variants do)
        if (probs[i].gru && maxM > 32 && (long long)((maxN + 15) / 16) * ((maxM + 15) / 16) >= 2048) return D3_ERR_ARG;
    const int ctiles = (maxN + 15) / 16;
    int kblocks = 0;             // deepest reduction of the batch, in 16-wide k blocks
    for (int i = 0; i < nprobs; i++) {
        int kb = 0;
        for (int q = 0; q < probs[i].nseg; q++) kb += (probs[i].seg[q].K + 15) / 16;
        if (kb > kblocks) kblocks = kb;
    }
    // launch timing (bench.py): exact-fp32 MFMA GEMM -- flops 2 M N K, bytes = operands + outputs once (SURVEY 8(d) "Heads")
    double pflops = 0.0, pbytes = 0.0;
    for (int i = 0; i < nprobs; i++) {
        long long ksum = 0;
        for (int q = 0; q < probs[i].nseg; q++) ksum += probs[i].seg[q].K;
        pflops += 2.0 * (double)probs[i].M * probs[i].N * (double)ksum;
        pbytes += 4.0 * ((double)probs[i].M * (double)ksum + (double)ksum * probs[i].N + (double)probs[i].M * probs[i].N);
    }
    void *pr = d3_prof_begin(3, pbytes, pflops, s);
    int variant[3] = {0, 0, 0};          // {kernel: 0 hg_gemm_tiled_kernel, 1 hg_gemm_kernel, RT, waves}
#define HG_SPLIT(RTV, GY)                                                                                  \
    do {                                                                                                   \
        variant[0] = 1; variant[1] = RTV; variant[2] = kblocks >= 40 ? 16 : (kblocks >= 20 ? 8 : 4);        \
        if (kblocks >= 40) hg_gemm_kernel<RTV, true, 16><<<dim3(ctiles, GY, nprobs), 1024, 0, s>>>(b);      \
        else if (kblocks >= 20) hg_gemm_kernel<RTV, true, 8><<<dim3(ctiles, GY, nprobs), 512, 0, s>>>(b);   \
        else hg_gemm_kernel<RTV, true, 4><<<dim3(ctiles, GY, nprobs), 256, 0, s>>>(b);                      \
    } while (0)
    if (maxM <= 32) {            // a decode step: K split over the waves of a workgroup
        if (maxM <= 16) HG_SPLIT(1, 1);
        else if (d3_tune(D3T_HG_RT1) != 0) HG_SPLIT(1, 2);     // (one row tile per workgroup: twice the workgroups; 0: two tiles)
        else HG_SPLIT(2, 1);
    } else {
        const long long tiles16 = (long long)ctiles * ((maxM + 15) / 16);
        if (tiles16 < 2048) {    // few tiles: still split K so that the chip is covered
            HG_SPLIT(2, (maxM + 31) / 32);
        } else {
            const bool tiled = d3_tune(D3T_HG_TILED) != 0;   // (A/B)
            if (tiled && d3_tune(D3T_HG_BF16X3) != 0) { variant[0] = 2; hg_gemm_tiled3_kernel<<<dim3((ctiles * 16 + HT_BN - 1) / HT_BN, (maxM + HT_BM - 1) / HT_BM, nprobs), 256, 0, s>>>(b); }
            else if (tiled) hg_gemm_tiled_kernel<<<dim3((ctiles * 16 + HT_BN - 1) / HT_BN, (maxM + HT_BM - 1) / HT_BM, nprobs), 256, 0, s>>>(b);
            else { variant[0] = 1; variant[1] = 4; variant[2] = 0; hg_gemm_kernel<4, false, 4><<<dim3((ctiles + 3) / 4, (maxM + 63) / 64, nprobs), 256, 0, s>>>(b); }
        }
    }
#undef HG_SPLIT
    if (pr) {
        const int tags[7] = {maxM, maxN, kblocks * 16, nprobs, variant[0], variant[1], variant[2]};
        for (int i = 0; i < 7; i++) d3_prof_tag(pr, i, tags[i]);
        d3_prof_end(pr, s);
    }
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_hgemm(const d3_gemm_prob *probs, int nprobs, void *stream) {
    D3_CLEAR();
    return hg_launch(probs, nprobs, d3_stream(stream));
}

// out[c] (+)= sum_r x[r, c]  (bias gradients), up to HG_MAXCS matrices per call: stage 1 sums HG_RS row slices per
// 64-column block (coalesced 256-byte rows, 4 waves on interleaved rows), stage 2 adds the slices in fixed order.
#define HG_MAXCS 12
#define HG_RS 16
struct HgColsumJobs { const float *x[HG_MAXCS]; float *out[HG_MAXCS]; long long ld[HG_MAXCS]; int R[HG_MAXCS], C[HG_MAXCS], accum[HG_MAXCS]; int cmax; };

__global__ __launch_bounds__(256) void hg_colsum1_kernel(const HgColsumJobs j, float *__restrict__ part) {
    __shared__ float sh[4][64];
    const int job = blockIdx.z, rs = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, R = j.R[job], C = j.C[job];
    if (blockIdx.x * 64 >= C) return;
    const int per = (R + HG_RS - 1) / HG_RS, r0 = rs * per, r1 = min(R, r0 + per);
    const float *x = j.x[job];
    const long long ld = j.ld[job];
    float s = 0.f;
    if (c < C) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {   // four rows in flight per lane
            const float a0 = x[(long long)r * ld + c], a1 = x[(long long)(r + 4) * ld + c], a2 = x[(long long)(r + 8) * ld + c],
                        a3 = x[(long long)(r + 12) * ld + c];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; r < r1; r += 4) s += x[(long long)r * ld + c];
    }
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) part[((long long)job * HG_RS + rs) * j.cmax + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
__global__ void hg_colsum2_kernel(const HgColsumJobs j, const float *__restrict__ part) {
    const int job = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= j.C[job]) return;
    float s = 0.f;
#pragma unroll
    for (int rs = 0; rs < HG_RS; rs++) s += part[((long long)job * HG_RS + rs) * j.cmax + c];
    float *o = j.out[job];
    o[c] = j.accum[job] ? o[c] + s : s;
}

size_t hg_colsum_ws_bytes(int njobs, int cmax) { return (size_t)njobs * HG_RS * cmax * 4; }

// n <= HG_MAXCS column sums in two launches; ws >= hg_colsum_ws_bytes(n, max C)
int hg_colsum_multi(const float *const *x, const long long *ld, const int *R, const int *C, float *const *out, const int *accum, int n,
                    void *ws, size_t ws_bytes, hipStream_t s) {
    if (n < 1 || n > HG_MAXCS) return D3_ERR_ARG;
    HgColsumJobs j;
    j.cmax = 0;
    for (int i = 0; i < HG_MAXCS; i++) {
        const int q = i < n ? i : 0;
        j.x[i] = x[q]; j.out[i] = out[q]; j.ld[i] = ld[q]; j.R[i] = R[q]; j.C[i] = C[q]; j.accum[i] = accum ? accum[q] : 0;
        if (i < n && C[i] > j.cmax) j.cmax = C[i];
    }
    if (j.cmax <= 0) return 0;
    if (ws_bytes < hg_colsum_ws_bytes(n, j.cmax)) return D3_ERR_WORKSPACE;
    hg_colsum1_kernel<<<dim3((j.cmax + 63) / 64, HG_RS, n), 256, 0, s>>>(j, (float *)ws);
    hg_colsum2_kernel<<<dim3((j.cmax + 255) / 256, n), 256, 0, s>>>(j, (const float *)ws);
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t d3_colsum_ws_bytes(int C) { return hg_colsum_ws_bytes(1, C); }
extern "C" int d3_colsum(const float *x, long long ld, int R, int C, float *out, int accum, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    return hg_colsum_multi(&x, &ld, &R, &C, &out, &accum, 1, ws, ws_bytes, d3_stream(stream));
}
