// spconv3.hip -- third-generation forward / data-gradient kernel of the K = 27 sparse convolutions (round 6).
//
// Same contraction as spconv2.hip (MinkowskiConvolution forward and data gradient, reference call sites
// model/common.py:32-41,88-118; model/pointgroup.py:70):
//
//   out[u,:] = sum_k x[nbr[u,k],:] @ W[k]   (+ res[u,:])        nbr: (M, 27) kernel map of the level (coordmap.hip)
//
// What bound spconv_fwd2_kernel (profiles/r05_k_pmc_sq_summary.txt): 388 instructions per 14-MFMA tile loop --
// table rows staged through LDS, one LDS read + a compare + a 24-bit multiply + a select in front of every gather, an epilogue
// with every fusion as a run-time flag -- i.e. issue- and latency-bound at 0.14 of HBM.  This kernel removes the instructions
// instead of hiding them:
//   * LANE TABLE.  The level's kernel map is stored a second time in the order the lanes consume it (cm_k3_kernel /
//     d3_kmap_k3_packq): per 16-row tile 64 lanes x 8 uint16; lane (r = lane & 15, g = lane >> 4) holds entry q = 0..6 = the
//     neighbour of row r at offset k = 4q + g, as e = nbr - row0 + 32768 (row0 = first row of the tile; 0xFFFF = absent; offset 27
//     and slot 7 are pads).  ONE 16-byte load per lane and tile brings every index the lane will ever need into four registers:
//     no LDS staging, no wave barrier, no LDS read per gather (64 B per row against the dense table's 108).
//   * REDUCTION ORDER.  MFMA step (q, c8) contracts over (offset 4q + g, channels c8*8 .. +8) for the four lane groups g: a lane
//     gathers ALL channel groups of its one neighbour row per q (ST loads of 16 B off one address register, immediate offsets),
//     so a table entry is decoded once per ST gathers -- one and_or_shift + one 24-bit multiply.
//   * ABSENT NEIGHBOURS COST NOTHING.  The tile's buffer descriptor is rebased to row0 - 32768 and its extent clamped to
//     65535 rows: entry 0xFFFF lands beyond the extent whatever the row size, and the hardware returns zeros without a memory
//     request -- no compare, no select, no branch.
//   * weights: the spconv2 fragment order (d3_spconv_pack), resident in LDS with a zero 28th offset; lane (col, g) reads the
//     fragment of (offset 4q + g, c8, column tile n) at base + immediate.
//   * the epilogue is a template parameter (plain | + residual | BatchNorm-backward), as are the output type (fp32 | bf16) and the
//     type of the BatchNorm input re-read by the backward epilogue.
// Per 16-row tile of a 16 -> 16 layer: 1 table load, 7 decodes (2 VALU), 14 gathers, 14 LDS reads, 14 MFMAs + the epilogue --
// ~95 instructions against 388.
// Roofline: HBM: 2 (Nin Cin + Nout Cout) + 2 K Cin Cout + 8 P bytes per launch.
#include "common.h"
#include "prof.h"
#include <atomic>

typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float c3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int c3_u32x4 __attribute__((ext_vector_type(4)));

#define C3_RSRC_FLAGS 0x00020000          // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
#define C3_P2_ROWS 16                     // rows of the second-level fp64 BatchNorm partial table (= C2_P2_ROWS / UN_P2_ROWS)

__device__ __forceinline__ unsigned int c3_pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    const c3_bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, p);
}
__device__ __forceinline__ float c3_bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float c3_bf_hi(unsigned int w) { return __uint_as_float(w & 0xFFFF0000u); }

// ------------------------------------------------------------------------------ lane table
// Per 16-row tile: the LIVE offsets (offsets some row of the tile has a neighbour at) in ascending order k_0 < k_1 < ..., padded
// with 27 to 28 slots; lq = ceil(live / 4) reduction groups.  Lane (r, g) of the tile holds, in slot q < 7, the neighbour of row r at
// offset k_{4q+g}:
//   tq[(tile * 64 + g * 16 + r) * 8 + q] = nbr[tile * 16 + r][k_{4q+g}] - tile * 16 + 32768   (uint16; 0xFFFF: absent / pad)
// and the tile's record (8 words behind the lane tables: rec[tile * 8 + w]) lists the offsets: byte g of word q < 7 = k_{4q+g},
// word 7 = lq.  Raster-ordered rows of a 2 cm indoor level have ~16 of 27 offsets live per tile (a planar patch: 9), so the
// kernel issues 4 - 5 gather groups per tile instead of 7 -- the texture addresser's cycles (16 per 64-lane 16-byte load whether
// the lanes are in range or not: TA busy 73 % of the 16 -> 16 launch with all 7 groups, profiles/r06_*) are what bounds it.
// *okq is cleared when an entry does not fit [1, 65534].  One wave per tile.
__global__ __launch_bounds__(256) void c3_packq_kernel(const int *__restrict__ nbr, int M, int ntiles, uint4 *__restrict__ tq,
                                                       unsigned int *__restrict__ rec, int *okq) {
    __shared__ int idsS[4][28];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + w;
    if (tile >= ntiles) return;                      // (wave-uniform; no workgroup barrier below)
    const int u = tile * 16 + r;
    // ---- live mask: lane (r, part g) looks at offsets 7g .. 7g + 6
    unsigned int m = 0u;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const int k = g * 7 + j;
        if (k < 27 && u < M && nbr[(long long)u * 27 + k] >= 0) m |= 1u << k;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m |= __shfl_xor(m, o);
    const int L = __popc(m);
    if (lane < 28) idsS[w][lane] = 27;
    __builtin_amdgcn_wave_barrier();
    if (lane < 27 && ((m >> lane) & 1u)) idsS[w][__popc(m & ((1u << lane) - 1u))] = lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- this lane's seven entries
    unsigned int e[8];
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int k = idsS[w][4 * q + g];
        unsigned int v = 0xFFFFu;
        if (k < 27 && u < M) {
            const int nb = nbr[(long long)u * 27 + k];
            if (nb >= 0) {
                const int d = nb - tile * 16 + 32768;
                if (d < 1 || d > 65534) bad = true; else v = (unsigned int)d;
            }
        }
        e[q] = v;
    }
    e[7] = 0xFFFFu;
    tq[(size_t)tile * 64 + lane] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    if (lane < 7) {
        const int *id = &idsS[w][4 * lane];
        rec[(size_t)tile * 8 + lane] = (unsigned int)id[0] | ((unsigned int)id[1] << 8) | ((unsigned int)id[2] << 16) | ((unsigned int)id[3] << 24);
    } else if (lane == 7) rec[(size_t)tile * 8 + 7] = (unsigned int)((L + 3) / 4);
    if (__any(bad) && lane == 0) *okq = 0;
}
__global__ void c3_set1_kernel(int *p) { *p = 1; }

// lane tables (1024 B per tile) followed by the tile records (32 B per tile)
extern "C" size_t d3_kmap_k3_q16_bytes(int M) { return (size_t)((M + 15) / 16) * (1024 + 32); }

extern "C" int d3_kmap_k3_packq(const int *nbr, int M, void *tq, int *okq, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (!nbr || !tq || !okq) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const int ntiles = (M + 15) / 16;
    c3_set1_kernel<<<1, 1, 0, s>>>(okq);
    c3_packq_kernel<<<(ntiles + 3) / 4, 256, 0, s>>>(nbr, M, ntiles, (uint4 *)tq, (unsigned int *)((char *)tq + (size_t)ntiles * 1024), okq);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ forward / data gradient
struct Conv3Args {
    const void *x;              // (Min, ldx) bf16
    const uint4 *tq;            // lane table: [ntiles][64] x 16 bytes
    const unsigned int *rec;    // tile records: [ntiles][8]
    const uint4 *Wp;            // packed bf16 fragments (d3_spconv_pack order), 27 offsets
    void *out;                  // (Mout, ldo) fp32 or bf16
    const float *res;           // EPI 1: residual (Mout, ldr) fp32
    float *part;                // optional BatchNorm partials [grid][2][NT*16]
    double *part2;              // optional second-level table [C3_P2_ROWS][2][NT*16] (zeroed by the caller)
    int ldx, ldo, ldr, Mout, ntiles;
    unsigned long long xbytes;  // extent of x in bytes: ((Min - 1) * ldx + Cin) * 2
    // EPI 2 (BatchNorm-backward epilogue): the stored value is g = acc * relu'(bn(bnx)), partials (sum g, sum g * xhat)
    const void *bnx; const float *bn_mean, *bn_var, *bn_gamma, *bn_beta;
    int ldbx, bn_relu; float bn_eps;
};

enum { C3_EPI_PLAIN = 0, C3_EPI_RES = 1, C3_EPI_BNBWD = 2 };

// gather groups (q) whose loads are in flight together: everything of the tile up to 32 input channels, 2 / 1 groups beyond
// (two buffers: the next chunk is requested before the products of the current one)
// QC: 7 = every gather of the tile requested before the first product (one memory round trip per tile; ST * 28 registers);
// smaller: chunks of QC groups in two buffers

// The reduction of one tile with LQ live gather groups: straight-line code per LQ (the waits count exactly the loads behind them).
template <int LQ, int ST, int NT, int QC>
__device__ __forceinline__ void c3_reduce(c3_f32x4 (&acc)[NT], const __amdgpu_buffer_rsrc_t rx, const unsigned int (&words)[4],
                                          const unsigned int (&ids)[7], const unsigned int rowb, const uint4 *wS, const int wlane,
                                          const int g8, uint4 &tw_next, unsigned int (&ids_next)[8], const uint4 *tq_next, const unsigned int *rec_next,
                                          const bool has_next) {
    constexpr int NCH = (LQ + QC - 1) / QC;
    c3_u32x4 buf[NCH > 1 ? 2 : 1][QC * ST];
    int wq[LQ > 0 ? LQ : 1];                // fragment base of (offset k_{4q+g}, c8 0, n 0) for this lane
#pragma unroll
    for (int q = 0; q < LQ; q++) wq[q] = (int)__umul24((ids[q] >> g8) & 0xFFu, (unsigned int)(ST * NT * 16)) + wlane;
    auto issue = [&](const int c, c3_u32x4 (&dst)[QC * ST]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QC; j++) {
            const int q = c * QC + j;
            if (q >= LQ) continue;
            const unsigned int e = (q & 1) ? (words[q >> 1] >> 16) : (words[q >> 1] & 0xFFFFu);
            const unsigned int off = __umul24(e, rowb);
#pragma unroll
            for (int c8 = 0; c8 < ST; c8++) dst[j * ST + c8] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned int)c8 * 16u, 0, 0);
        }
    };
    auto products = [&](const int c, const c3_u32x4 (&src)[QC * ST]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QC; j++) {
            const int q = c * QC + j;
            if (q >= LQ) continue;
#pragma unroll
            for (int c8 = 0; c8 < ST; c8++) {
                const c3_bf16x8 B = __builtin_bit_cast(c3_bf16x8, src[j * ST + c8]);
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const uint4 w = wS[wq[q] + (c8 * NT + n) * 16];
                    // transposed product: D[m = channel][n = row] += W^T[channel][k] * X^T[k][row]
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c3_bf16x8, w), B, acc[n], 0, 0, 0);
                }
            }
        }
    };
    auto prefetch = [&]() __attribute__((always_inline)) {      // the next tile's table row + record ride behind the gathers
        tw_next = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
        for (int i = 0; i < 8; i++) ids_next[i] = 0u;
        if (has_next) {
            tw_next = *tq_next;
#pragma unroll
            for (int i = 0; i < 8; i++) ids_next[i] = rec_next[i];
        }
    };
    if (LQ > 0) issue(0, buf[0]);
    if (NCH <= 1) prefetch();
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH) issue(c + 1, buf[(c + 1) & 1]);
        if (NCH > 1 && c + 1 == NCH - 1) prefetch();
        __builtin_amdgcn_sched_barrier(0);
        products(c, buf[c & 1]);
    }
}

// ST = Cin / 8, NT = Cout / 16, EPI as above, OBF: bf16 output, BXBF: bnx is bf16, NW waves per workgroup
template <int ST, int NT, int EPI, bool OBF, bool BXBF, int NW, int QC>
__global__ __launch_bounds__(NW * 64) void spconv_fwd3_kernel(const Conv3Args a, const unsigned int *__restrict__ rec_, const uint4 *__restrict__ tq_) {
    // (the table and the records as restrict-qualified kernel arguments: loads the stores of this kernel provably do not clobber --
    // the records then travel through the scalar cache, s_load_dwordx8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WREAL = 27 * ST * NT * 16, WELEMS = 28 * ST * NT * 16;     // 16-byte fragments (offset 27: zeros)
    uint4 *wS = (uint4 *)smem;
    float *redS = (float *)(smem + (size_t)WELEMS * 16);                     // [NW][2][NT*16]
    float4 *bnS = (float4 *)(redS + NW * 2 * NT * 16);                       // [NT*16] (mean, 1/std, gamma, beta)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, g = lane >> 4;
    // ---- stage the weights (and the BatchNorm parameters)
    for (int i = t; i < WELEMS; i += NW * 64) {
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (i < WREAL) w = a.Wp[i];
        wS[i] = w;
    }
    if (EPI == C3_EPI_BNBWD && t < NT * 16) {
        float4 bp;
        bp.x = a.bn_mean[t]; bp.y = rsqrtf(a.bn_var[t] + a.bn_eps);
        bp.z = a.bn_relu ? a.bn_gamma[t] : 0.f; bp.w = a.bn_relu ? a.bn_beta[t] : 1.f;      // (no ReLU: gamma 0, beta 1 -> the mask test is always "keep")
        bnS[t] = bp;
    }
    __syncthreads();
    // ---- tile range of this workgroup: the nb / 8 workgroups of an XCD (block b runs on XCD b % 8) take the XCD's eighth of the
    // tile groups in turn, so that an XCD works on one moving window of rows
    const int nb = gridDim.x, b = blockIdx.x;
    const int ntg = (a.ntiles + NW - 1) / NW;
    int tg0, tg1, tstride;
    if ((nb & 7) == 0) {
        const int per = (ntg + nb - 1) / nb, nx = nb >> 3;
        tg0 = (b & 7) * nx * per + (b >> 3); tstride = nx;
        tg1 = min(ntg, ((b & 7) + 1) * nx * per);
    } else { tg0 = b; tstride = nb; tg1 = ntg; }
    c3_f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int n = 0; n < NT; n++) { ssum[n] = (c3_f32x4){0.f, 0.f, 0.f, 0.f}; ssq[n] = (c3_f32x4){0.f, 0.f, 0.f, 0.f}; }
    const unsigned int rowb = (unsigned int)a.ldx * 2u;
    const int g8 = g * 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint4 tw = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    unsigned int ids[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    {
        const int tile = tg0 * NW + wave_u;
        if (tg0 < tg1 && tile < a.ntiles) {
            tw = tq_[(size_t)tile * 64 + lane];
#pragma unroll
            for (int i = 0; i < 8; i++) ids[i] = rec_[(size_t)tile * 8 + i];
        }
    }
    for (int tg = tg0; tg < tg1; tg += tstride) {
        const int tile = __builtin_amdgcn_readfirstlane(tg * NW + wave_u);
        if (tile >= a.ntiles) break;
        const int row0 = tile * 16, urow = row0 + r;
        // the tile's window of x: rows [row0 - 32768, row0 + 32767)
        const long long base_b = ((long long)row0 - 32768) * (long long)rowb;
        const long long avail = (long long)a.xbytes - base_b;
        const long long win = 65535ll * (long long)rowb;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.x + base_b), 0, (int)(avail < win ? avail : win), C3_RSRC_FLAGS);
        const unsigned int words[4] = {tw.x, tw.y, tw.z, tw.w};
        unsigned int idq[7];
#pragma unroll
        for (int i = 0; i < 7; i++) idq[i] = (unsigned int)__builtin_amdgcn_readfirstlane((int)ids[i]);
        const int lq = __builtin_amdgcn_readfirstlane((int)ids[7]);
        // ---- epilogue operands that do not depend on the products: requested ahead of the gathers
        c3_f32x4 e_res[NT];
        uint2 e_bx16[NT];
        const bool rowok = urow < a.Mout;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            e_res[n] = (c3_f32x4){0.f, 0.f, 0.f, 0.f}; e_bx16[n] = make_uint2(0u, 0u);
            if (EPI == C3_EPI_RES && rowok) e_res[n] = *(const c3_f32x4 *)(a.res + (long long)urow * a.ldr + n * 16 + g * 4);
            if (EPI == C3_EPI_BNBWD && rowok) {
                if (BXBF) e_bx16[n] = *(const uint2 *)((const unsigned short *)a.bnx + (long long)urow * a.ldbx + n * 16 + g * 4);
                else e_res[n] = *(const c3_f32x4 *)((const float *)a.bnx + (long long)urow * a.ldbx + n * 16 + g * 4);
            }
        }
        c3_f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = (c3_f32x4){0.f, 0.f, 0.f, 0.f};
        const int ntile = __builtin_amdgcn_readfirstlane(tile + tstride * NW);
        const bool has_next = tg + tstride < tg1 && ntile < a.ntiles;
        const int ptile = __builtin_amdgcn_readfirstlane(has_next ? ntile : tile);       // (provably uniform: the record is a scalar load)
        const uint4 *tq_next = tq_ + ((size_t)ptile * 64 + lane);
        const unsigned int *rec_next = rec_ + (size_t)ptile * 8;
        uint4 tw_n; unsigned int ids_n[8];
#define C3_LQ(V) case V: c3_reduce<V, ST, NT, QC>(acc, rx, words, idq, rowb, wS, r, g8, tw_n, ids_n, tq_next, rec_next, has_next); break;
        switch (lq) { C3_LQ(0) C3_LQ(1) C3_LQ(2) C3_LQ(3) C3_LQ(4) C3_LQ(5) C3_LQ(6) default: c3_reduce<7, ST, NT, QC>(acc, rx, words, idq, rowb, wS, r, g8, tw_n, ids_n, tq_next, rec_next, has_next); break; }
#undef C3_LQ
        tw = tw_n;
#pragma unroll
        for (int i = 0; i < 8; i++) ids[i] = ids_n[i];
        // ---- epilogue.  D layout: column (= output row) lane & 15, rows (= channels) g * 4 + j
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int col = n * 16 + g * 4;
            c3_f32x4 vv = acc[n];
            if (EPI == C3_EPI_RES) vv += e_res[n];
            if (EPI == C3_EPI_BNBWD) {
                c3_f32x4 bx, xh;
                if (BXBF) { bx[0] = c3_bf_lo(e_bx16[n].x); bx[1] = c3_bf_hi(e_bx16[n].x); bx[2] = c3_bf_lo(e_bx16[n].y); bx[3] = c3_bf_hi(e_bx16[n].y); }
                else bx = e_res[n];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 bp = bnS[col + j];
                    xh[j] = (bx[j] - bp.x) * bp.y;
                    if (fmaf(xh[j], bp.z, bp.w) <= 0.f) vv[j] = 0.f;
                }
                ssum[n] += vv; ssq[n] += vv * xh;      // (rows beyond Mout: vv = 0)
            } else { ssum[n] += vv; ssq[n] += vv * vv; }
            if (rowok) {
                if (OBF) *(uint2 *)((unsigned short *)a.out + (long long)urow * a.ldo + col) = make_uint2(c3_pack2(vv[0], vv[1]), c3_pack2(vv[2], vv[3]));
                else *(c3_f32x4 *)((float *)a.out + (long long)urow * a.ldo + col) = vv;
            }
        }
    }
    if (a.part) {   // per-workgroup BatchNorm partials (fixed order: deterministic)
#pragma unroll
        for (int n = 0; n < NT; n++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float s1 = ssum[n][j], s2 = ssq[n][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (r == 0) { redS[wave * 2 * NT * 16 + n * 16 + g * 4 + j] = s1; redS[wave * 2 * NT * 16 + NT * 16 + n * 16 + g * 4 + j] = s2; }
            }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += redS[w * 2 * NT * 16 + t];
            a.part[(long long)b * 2 * NT * 16 + t] = s;
            if (a.part2) unsafeAtomicAdd(&a.part2[(b % C3_P2_ROWS) * 2 * NT * 16 + t], (double)s);
        }
    }
}

// ------------------------------------------------------------------------------ host side
static int c3_ncu() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t pr; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    return n;
}
// one row per instantiated shape: ST = Cin / 8, NT = Cout / 16, waves per workgroup, gather groups per chunk
struct Conv3Shape { int st, nt, nw, qc; };
#define C3_SHAPES(X) X(2, 1, 4, 7) X(2, 2, 8, 7) X(4, 1, 8, 2) X(4, 2, 8, 2) X(4, 4, 16, 2) X(6, 3, 16, 1) X(8, 2, 16, 1)
#define C3_ROW(STV, NTV, NWV, QCV) {STV, NTV, NWV, QCV},
static const Conv3Shape c3_shapes[] = {C3_SHAPES(C3_ROW)};

struct Conv3Plan { int ok, nw, qc, maxgrid; size_t lds; };
static size_t c3_lds_bytes(int ST, int NT, int nw) { return (size_t)28 * ST * NT * 256 + (size_t)nw * 2 * NT * 16 * 4 + (size_t)NT * 16 * 16; }

static Conv3Plan conv3_plan(int Mout, int Cin, int Cout) {
    Conv3Plan p{0, 4, 7, 1, 0};
    if (Mout <= 0 || (Cin & 7) || (Cout & 15)) return p;
    const int ST = Cin / 8, NT = Cout / 16;
    const Conv3Shape *pick = nullptr;
    for (const Conv3Shape &c : c3_shapes)
        if (c.st == ST && c.nt == NT && !pick) pick = &c;
    if (!pick) return p;
    p.nw = pick->nw; p.qc = pick->qc; p.lds = c3_lds_bytes(ST, NT, p.nw);
    if (p.lds > 160 * 1024) return p;
    p.maxgrid = c3_ncu() * (16 / p.nw);          // upper bound of the grid (the partial table's rows): 16 waves per CU
    p.ok = 1;
    return p;
}

// upper bound of the partial rows a launch writes (0: no instance for the shape); the launch itself may use fewer workgroups
// (d3_spconv_fwd3_last_nparts)
extern "C" int d3_spconv_fwd3_nparts(int Mout, int Cin, int Cout) {
    const Conv3Plan p = conv3_plan(Mout, Cin, Cout);
    return p.ok ? p.maxgrid : 0;
}

static std::atomic<long long> g_c3_launches{0};
extern "C" long long d3_spconv_fwd3_launches(void) { return g_c3_launches.load(); }

template <int ST, int NT, int EPI, bool OBF, bool BXBF, int NW, int QC>
static int c3_launch_inst(const Conv3Args &a, const Conv3Plan &p, int *grid_out, hipStream_t s) {
    static int occ_dev[64] = {0};                // resident workgroups per CU of this instance (0: not asked yet)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!occ_dev[dev]) {
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW, QC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nblk = 0;
        D3_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void *)spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW, QC>, NW * 64, p.lds));
        occ_dev[dev] = nblk < 1 ? 1 : (nblk > 16 / NW ? 16 / NW : nblk);      // (measured: 16 -> 16 at five waves per SIMD runs no faster than at four -- 23.7 vs 23.5 us)
    }
    // persistent workgroups: every resident slot of the chip once, fewer when there are fewer tile groups
    const int ntg = (a.ntiles + NW - 1) / NW;
    const int cap = c3_ncu() * occ_dev[dev];
    const int per = (ntg + cap - 1) / cap;
    int grid = (ntg + per - 1) / per;
    if (grid >= 8) grid = (grid + 7) / 8 * 8;      // (XCD-interleaved ranges want a multiple of 8; a surplus workgroup finds no tile and writes a zero partial row)
    if (grid > p.maxgrid) grid = p.maxgrid;
    spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW, QC><<<grid, NW * 64, p.lds, s>>>(a, a.rec, a.tq);
    D3_LAUNCH_CHECK();
    g_c3_launches++;
    *grid_out = grid;
    return 0;
}
template <int ST, int NT, int NW, int QC>
static int c3_launch_shape(const Conv3Args &a, const Conv3Plan &p, int epi, bool obf, bool bxbf, int *grid_out, hipStream_t s) {
    if (epi == C3_EPI_PLAIN) return obf ? c3_launch_inst<ST, NT, C3_EPI_PLAIN, true, false, NW, QC>(a, p, grid_out, s) : c3_launch_inst<ST, NT, C3_EPI_PLAIN, false, false, NW, QC>(a, p, grid_out, s);
    if (epi == C3_EPI_RES) return obf ? D3_ERR_ARG : c3_launch_inst<ST, NT, C3_EPI_RES, false, false, NW, QC>(a, p, grid_out, s);
    if (obf) return bxbf ? c3_launch_inst<ST, NT, C3_EPI_BNBWD, true, true, NW, QC>(a, p, grid_out, s) : c3_launch_inst<ST, NT, C3_EPI_BNBWD, true, false, NW, QC>(a, p, grid_out, s);
    return bxbf ? c3_launch_inst<ST, NT, C3_EPI_BNBWD, false, true, NW, QC>(a, p, grid_out, s) : c3_launch_inst<ST, NT, C3_EPI_BNBWD, false, false, NW, QC>(a, p, grid_out, s);
}

struct Conv3Bn { const void *x; const float *mean, *var, *gamma, *beta; int ldx, relu, xbf16; float eps; };

// internal entry (spconv2.hip's dispatcher and the C ABI below).  Returns D3_ERR_ARG for shapes without an instance.
int d3_conv3_run(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, const float *res, int ldr, float *part,
                 double *part2, int Min, int Mout, int Cin, int Cout, int obf16, const Conv3Bn *bn, int *nparts_out, hipStream_t s) {
    const Conv3Plan p = conv3_plan(Mout, Cin, Cout);
    if (!p.ok || !x || !tq || !Wp || !out) return D3_ERR_ARG;
    if ((ldx & 7) || ldx < Cin || ldo < Cout || (ldo & 3) || (res && (ldr & 3)) || (res && bn) || (res && obf16)) return D3_ERR_ARG;
    if (bn && (bn->ldx & 3)) return D3_ERR_ARG;
    const unsigned long long rowb = (unsigned long long)ldx * 2ull;
    const unsigned long long xb = Min > 0 ? ((unsigned long long)(Min - 1) * ldx + Cin) * 2ull : 0ull;
    if (Min <= 0 || Min >= (1 << 24) || rowb > 16384ull || xb > 0x7FFFFFFFull) return D3_ERR_RANGE;
    Conv3Args a;
    a.x = x; a.tq = (const uint4 *)tq; a.rec = (const unsigned int *)((const char *)tq + (size_t)((Mout + 15) / 16) * 1024); a.Wp = (const uint4 *)Wp;
    a.out = out; a.res = res; a.part = part; a.part2 = part ? part2 : nullptr;
    a.ldx = ldx; a.ldo = ldo; a.ldr = ldr; a.Mout = Mout; a.ntiles = (Mout + 15) / 16; a.xbytes = xb;
    a.bnx = nullptr; a.bn_mean = a.bn_var = a.bn_gamma = a.bn_beta = nullptr; a.ldbx = 0; a.bn_relu = 0; a.bn_eps = 0.f;
    if (bn) { a.bnx = bn->x; a.bn_mean = bn->mean; a.bn_var = bn->var; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta; a.ldbx = bn->ldx; a.bn_relu = bn->relu; a.bn_eps = bn->eps; }
    const int epi = bn ? C3_EPI_BNBWD : (res ? C3_EPI_RES : C3_EPI_PLAIN);
    const bool obf = obf16 != 0, bxbf = bn && bn->xbf16;
    const int ST = Cin / 8, NT = Cout / 16;
    const double bytes = 2.0 * (double)Min * Cin + (obf ? 2.0 : 4.0) * (double)Mout * Cout + 2.0 * 27.0 * Cin * Cout + 66.0 * (double)Mout + (res ? 4.0 * (double)Mout * Cout : 0.0);
    void *pr = d3_prof_begin(0, bytes, 0.0, s);
    int rc = D3_ERR_ARG, grid = 0;
#define C3_CASE(STV, NTV, NWV, QCV) if (rc == D3_ERR_ARG && ST == STV && NT == NTV && p.nw == NWV && p.qc == QCV) rc = c3_launch_shape<STV, NTV, NWV, QCV>(a, p, epi, obf, bxbf, &grid, s);
    C3_SHAPES(C3_CASE)
#undef C3_CASE
    if (nparts_out) *nparts_out = grid;
    if (pr) {
        const int dims[12] = {Min, Mout, 27, Cin, Cout, NT, epi, (obf ? 1 : 0) | (bxbf ? 2 : 0), p.nw, p.qc, 27, ST + 8000};      // (+ 8000: spconv_fwd3_kernel; EPI / OBF | BXBF << 1 / QC in the
                                                                                                                               //  WLDS / XBF / F32M slots: bench.py's kernel naming)
        for (int i = 0; i < 12; i++) d3_prof_tag(pr, i, dims[i]);
    }
    d3_prof_end(pr, s);
    return rc;
}

// C ABI (tests, tools): one K = 27 forward / data-gradient launch on the lane table.
//   flags: D3_CONV_OUTBF16 (out is bf16; not with a residual); x is bf16 always.
//   part (optional): [d3_spconv_fwd3_nparts()][2][Cout] fp32 partial sums / sums of squares; part2 (optional, zeroed by the caller):
//   [16][2][Cout] fp64 second-level table.
extern "C" int d3_spconv_fwd3(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, const float *res, int ldr,
                              float *part, double *part2, int Min, int Mout, int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (Mout <= 0) return 0;
    int np = 0;
    const int rc = d3_conv3_run(x, ldx, tq, Wp, out, ldo, res, ldr, part, part2, Min, Mout, Cin, Cout, (flags & D3_CONV_OUTBF16) ? 1 : 0, nullptr, &np, d3_stream(stream));
    d3_spconv_set_last_nparts(np);
    return rc;
}
// data gradient of a BatchNorm -> ReLU -> convolution unit (as d3_spconv_fwd2_bnbwd); bnx: the BatchNorm input, fp32, or bf16 with
// D3_CONV_BNXBF16 in flags
extern "C" int d3_spconv_fwd3_bnbwd(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, float *part, double *part2,
                                    const void *bnx, int ldbx, const float *mean, const float *var, const float *gamma, const float *beta,
                                    float eps, int relu, int Min, int Mout, int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (Mout <= 0) return 0;
    Conv3Bn bn{bnx, mean, var, gamma, beta, ldbx, relu, (flags & D3_CONV_BNXBF16) ? 1 : 0, eps};
    int np = 0;
    const int rc = d3_conv3_run(x, ldx, tq, Wp, out, ldo, nullptr, 0, part, part2, Min, Mout, Cin, Cout, (flags & D3_CONV_OUTBF16) ? 1 : 0, &bn, &np, d3_stream(stream));
    d3_spconv_set_last_nparts(np);
    return rc;
}
