// spconv2.hip -- second-generation sparse convolution kernels for gfx950: wave-autonomous gather -> MFMA.
//
// Same contraction as spconv.hip (MinkowskiConvolution / MinkowskiConvolutionTranspose forward and data gradient,
// reference call sites model/common.py:32,38,41,66,90,98; model/pointgroup.py:70):
//
//   out[u,:] = sum_k x[tbl[u,k],:] @ W[k]   (+ res[u,:])        tbl: dense (Mout,K) kernel map (coordmap.hip)
//
// Design (what changed against spconv.hip, and why -- profiles/r01_h: the 64-row LDS-staged tile spends one barrier
// and one LDS round trip per (offset, 32 channels) to feed a single 16x16x16 MFMA at C=16):
//   * a WAVE owns a 16-row output tile.  The MFMA A operand (16 rows x 32 reduction elements, 8 bf16 per lane) is
//     gathered straight from HBM/L2 into registers: lane (r = lane&15, g = lane>>4) loads 16 bytes (8 channels) of
//     input row tbl[row0+r][k]; no LDS staging of activations, no workgroup barrier in the main loop.
//   * the reduction index is the flattened list of (active offset, 8-channel group) "slots"; one
//     v_mfma_f32_16x16x32_bf16 consumes four slots (one per lane group), so Cin=16 packs two offsets into one MFMA
//     and offsets unused by the 16-row tile cost nothing (4x finer skipping than a 64-row tile).
//   * weights are pre-packed once per step into bf16 MFMA-B fragment order (d3_spconv_pack); a fragment is one
//     contiguous 16-byte read per lane, from LDS when the layer's weights fit (big levels, persistent workgroups)
//     or from L2 (deep levels).
//   * few-row levels: the waves of a workgroup split the slots of ONE tile and reduce through LDS -- no atomics,
//     no zero fill, deterministic, and the complete tile is available to the epilogue.
//   * epilogue fusions: residual add, accumulate-into, strided output (writes straight into a concatenated
//     buffer) and per-channel sum / sum-of-squares partials for the following BatchNorm.
// Roofline: HBM; algorithmic bytes per launch as in spconv.hip.
#include "common.h"
#include "prof.h"
#include <cstdlib>
#include <cstring>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2bf2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    const bf16x2_t p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, p);
}

#define C2_MAXK 27
#ifndef C2_UBIG
#define C2_UBIG 8
#endif
#ifndef C2_OCC_SMALL
#define C2_OCC_SMALL 4
#endif
#ifndef C2_PREFETCH
#define C2_PREFETCH 1   // prefetch the next tile's kernel-map rows into registers (7 VGPRs)
#endif
#define C2_U(NTV) ((NTV) <= 2 ? C2_UBIG : 4)   // MFMA steps whose gathers are issued together

// ------------------------------------------------------------------------------ weight packing
// Wp[((k*S + c8)*NT + n)*16 + col][8] = bf16(Weff[k][c8*8 + j][n*16 + col]),  Weff = W[flipk ? K-1-k : k] (or its
// transpose when W is laid out (K, Cout, Cin)); columns >= Cout are zero.
__global__ void spconv_pack_kernel(const float *__restrict__ W, uint4 *__restrict__ Wp, int K, int Cin, int Cout,
                                   int NT, int flipk, int transw) {
    const int S = Cin >> 3;
    const long long total = (long long)K * S * NT * 16;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int col = (int)(e & 15);
    const int n = (int)((e >> 4) % NT);
    const int c8 = (int)((e / (16 * NT)) % S);
    const int k = (int)(e / ((long long)16 * NT * S));
    const int co = n * 16 + col, wk = flipk ? (K - 1 - k) : k;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ci = c8 * 8 + j;
        v[j] = 0.f;
        if (co < Cout) v[j] = transw ? W[((long long)wk * Cout + co) * Cin + ci] : W[((long long)wk * Cin + ci) * Cout + co];
    }
    Wp[e] = make_uint4(pack2bf2(v[0], v[1]), pack2bf2(v[2], v[3]), pack2bf2(v[4], v[5]), pack2bf2(v[6], v[7]));
}

// fp32 fragments (D3_CONV_F32: the reference's precision on v_mfma_f32_16x16x4_f32): same element order, 8 floats per element
__global__ void spconv_pack_f32_kernel(const float *__restrict__ W, float4 *__restrict__ Wp, int K, int Cin, int Cout,
                                       int NT, int flipk, int transw) {
    const int S = Cin >> 3;
    const long long total = (long long)K * S * NT * 16;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int col = (int)(e & 15);
    const int n = (int)((e >> 4) % NT);
    const int c8 = (int)((e / (16 * NT)) % S);
    const int k = (int)(e / ((long long)16 * NT * S));
    const int co = n * 16 + col, wk = flipk ? (K - 1 - k) : k;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ci = c8 * 8 + j;
        v[j] = 0.f;
        if (co < Cout) v[j] = transw ? W[((long long)wk * Cout + co) * Cin + ci] : W[((long long)wk * Cin + ci) * Cout + co];
    }
    Wp[e * 2] = make_float4(v[0], v[1], v[2], v[3]);
    Wp[e * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
}

extern "C" size_t d3_spconv_pack_bytes(int K, int Cin, int Cout) {
    return (size_t)K * (Cin / 8) * ((Cout + 15) / 16) * 256;
}
extern "C" size_t d3_spconv_pack_bytes_ex(int K, int Cin, int Cout, int flags) {
    return d3_spconv_pack_bytes(K, Cin, Cout) * ((flags & D3_CONV_F32) ? 2 : 1);
}

extern "C" int d3_spconv_pack(const float *W, void *Wp, int K, int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (K < 1 || K > C2_MAXK || Cin < 8 || (Cin & 7) || Cout < 1) return D3_ERR_ARG;
    const int NT = (Cout + 15) / 16;
    const long long total = (long long)K * (Cin / 8) * NT * 16;
    if (flags & D3_CONV_F32)
        spconv_pack_f32_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(
            W, (float4 *)Wp, K, Cin, Cout, NT, (flags & D3_CONV_FLIPK) ? 1 : 0, (flags & D3_CONV_TRANSW) ? 1 : 0);
    else
    spconv_pack_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(
        W, (uint4 *)Wp, K, Cin, Cout, NT, (flags & D3_CONV_FLIPK) ? 1 : 0, (flags & D3_CONV_TRANSW) ? 1 : 0);
    D3_LAUNCH_CHECK();
    return 0;
}

// Second-level BatchNorm partials (round 5).  Every BatchNorm launch used to reduce its producer's whole partial table (one row per
// convolution workgroup: ~1000 rows at the big levels) IN EVERY ONE of its <= 512 workgroups before it could touch a row: 8 - 19 us
// of dependent L2 round trips per launch, ~155 launches per step -- more than the normalisation passes themselves.  The producer
// now also adds its row into a 16-row fp64 table (hardware fp64 atomics, row = workgroup % 16: <= 64 adds per address); the
// consumer's reduction is ONE round trip over 16 rows.  Every addend is an fp32 value, so the fp64 sums are exact -- independent of
// the order the atomics land in -- unless the addends of one channel span more than 2^29 in magnitude (then: 2^-53 relative).
#define C2_P2_ROWS 16
// ------------------------------------------------------------------------------ forward / data gradient
struct Conv2Args {
    const void *x;              // (Min, ldx) fp32 or bf16
    const int *tbl;             // (Mout, K) or NULL (identity, K = 1)
    const unsigned short *Wp;   // packed bf16 fragments
    float *out;                 // (Mout, ldo)
    const float *res;           // optional residual (Mout, ldr), added before the store
    float *part;                // optional BatchNorm partials: [nparts][2][NT*16] (sum, sum of squares per column)
    double *part2;              // optional (round 5): second-level table [C2_P2_ROWS][2][NT*16] of fp64 accumulators; workgroup b adds its
                                // partial row to row b % C2_P2_ROWS (zeroed by the caller), so a consumer reads 16 rows instead of ~1000
    int ldx, ldo, ldr;
    int Mout, K, Cout, S;       // S = Cin / 8 slots per offset
    unsigned int inv;           // ceil(65536 / S): i = (s * inv) >> 16 == s / S for s < 4096
    int xbf16, accum, ntiles, NT;   // NT = ceil(Cout / 16)
    int obf16;                      // D3_CONV_OUTBF16: out is (Mout, ldo) bf16
    int f32;                        // D3_CONV_F32: fp32 weight fragments, v_mfma_f32_16x16x4_f32 (host-side dispatch only)
    unsigned int xbytes;            // extent of x in bytes for the raw buffer gathers (0: beyond 2 GiB / 2^24 rows, refused for the wave-per-tile kernel)
    unsigned int invK;          // ceil(65536 / K): e / K for e < 16*27
    int interleave;             // wave-per-tile kernel: the workgroups of an XCD take consecutive tile groups in turn (one moving window per L2)
    const unsigned int *tbl16;  // optional 16-bit delta form of tbl (coordmap.hip cm_pack16_kernel; validated by the caller), read as 32-bit
                                // words by the T16 instances of the wave-per-tile kernel
    // BatchNorm-backward epilogue (data gradient of a BN -> ReLU -> conv unit): the stored value is g = dy * relu'(bn(x))
    // and the partials are (sum g, sum g * xhat) -- the two reductions of the BatchNorm backward, fused here
    const float *bnx; const float *bn_mean, *bn_var, *bn_gamma, *bn_beta;
    int ldbx, bn_relu; float bn_eps;
    int bnx_bf16;               // D3_CONV_BNXBF16: bnx is stored as bf16 (a single-consumer convolution output, round 6)
};

__device__ __forceinline__ bf16x8_t c2_zero() {
    uint4 z = make_uint4(0u, 0u, 0u, 0u);
    return __builtin_bit_cast(bf16x8_t, z);
}
__device__ __forceinline__ bf16x8_t c2_load_a(const void *x, int xbf16, long long off) {
    if (xbf16) {
        uint4 v = *(const uint4 *)((const unsigned short *)x + off);
        return __builtin_bit_cast(bf16x8_t, v);
    }
    const float4 f0 = *(const float4 *)((const float *)x + off);
    const float4 f1 = *(const float4 *)((const float *)x + off + 4);
    uint4 v = make_uint4(pack2bf2(f0.x, f0.y), pack2bf2(f0.z, f0.w), pack2bf2(f1.x, f1.y), pack2bf2(f1.z, f1.w));
    return __builtin_bit_cast(bf16x8_t, v);
}

// Gathers are issued RAW (no conversion next to the load): a use right behind a load makes the compiler wait for it
// before the next load is issued, i.e. one memory round trip per gathered row instead of one per batch.
template <bool XBF>
__device__ __forceinline__ void c2_load_raw(const void *x, long long off, uint4 &lo, uint4 &hi) {
    if (XBF) lo = *(const uint4 *)((const unsigned short *)x + off);
    else { lo = *(const uint4 *)((const float *)x + off); hi = *(const uint4 *)((const float *)x + off + 4); }
}
typedef unsigned int c2_u32x4 __attribute__((ext_vector_type(4)));
#define C2_RSRC_FLAGS 0x00020000          // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
__device__ __forceinline__ uint4 c2_from_u32x4(const c2_u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }
template <bool XBF>
__device__ __forceinline__ bf16x8_t c2_cvt_raw(const uint4 lo, const uint4 hi) {
    if (XBF) return __builtin_bit_cast(bf16x8_t, lo);
    const uint4 v = make_uint4(pack2bf2(__uint_as_float(lo.x), __uint_as_float(lo.y)), pack2bf2(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                               pack2bf2(__uint_as_float(hi.x), __uint_as_float(hi.y)), pack2bf2(__uint_as_float(hi.z), __uint_as_float(hi.w)));
    return __builtin_bit_cast(bf16x8_t, v);
}

// One reduction step of a 16x16 tile (transposed product: first operand = weight fragment, second = gathered rows).
// bf16: one v_mfma_f32_16x16x32_bf16 over the lane's 8 channels; F32M (D3_CONV_F32, fp32 gathers only): eight
// v_mfma_f32_16x16x4_f32 -- step j pairs float j of the weight element with float j of the gathered 8 channels, i.e. the
// reduction index (lane group, j) is the same channel on both sides: exact fp32 products, fp32 accumulation.
template <bool XBF, bool F32M>
__device__ __forceinline__ f32x4 c2_mma(f32x4 acc, const uint4 wlo, const uint4 whi, const uint4 rlo, const uint4 rhi) {
    if (F32M) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wlo.x), __uint_as_float(rlo.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wlo.y), __uint_as_float(rlo.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wlo.z), __uint_as_float(rlo.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wlo.w), __uint_as_float(rlo.w), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(whi.x), __uint_as_float(rhi.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(whi.y), __uint_as_float(rhi.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(whi.z), __uint_as_float(rhi.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(whi.w), __uint_as_float(rhi.w), acc, 0, 0, 0);
        return acc;
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wlo), c2_cvt_raw<XBF>(rlo, rhi), acc, 0, 0, 0);
}
// weight element `e` (16-byte bf16 fragment, or 32-byte fp32 fragment) of a packed buffer
template <bool F32M>
__device__ __forceinline__ void c2_wload(const unsigned short *Wb, int e, uint4 &lo, uint4 &hi) {
    if (F32M) { const uint4 *p = (const uint4 *)Wb + (size_t)e * 2; lo = p[0]; hi = p[1]; }
    else { lo = *((const uint4 *)Wb + e); hi = lo; }
}

// LDS use of the wave-per-tile kernel besides the weights
#define C2_TBL_SENT (16 * C2_MAXK)        // one more slot per wave that always holds -1 (steps beyond the reduction read it)
#define C2_TBL_INTS (16 * C2_MAXK + 16)
#define C2_WAVE_LDS_BASE(NTV, NWV) ((NWV) * C2_TBL_INTS * 4 + (NWV) * 32 * 4 + (NWV) * 2 * (NTV) * 16 * 4)
#define C2_WAVE_LDS_BYTES(NTV, NWV) (C2_WAVE_LDS_BASE(NTV, NWV) + (NTV) * 16 * 16)   // + BatchNorm parameters, float4 per channel

// Wave-per-tile kernel (big levels): NW independent waves per workgroup (4, or 16 when the layer's packed weights are
// large: ONE LDS copy then serves 16 waves -- a 117 KB stem / 124 KB 48->48 weight set fits the CU's 160 KB once, and a
// 55 KB 32->32 set is staged by 256 workgroups instead of 1024), persistent over a contiguous range of
// NW-tile groups (XCD-contiguous: block b runs on XCD b % 8, so XCD x gets the x-th eighth of the rows and the
// neighbour rows its tiles gather stay in that XCD's L2).
// Every dependent global access of a wave is a full L2/HBM round trip and the MFMA work between them is tiny, so the
// kernel is organised around round trips, not FLOPs: (1) the kernel-map rows of the NEXT tile are prefetched into
// registers while the current tile computes; (2) all gathers of a batch are issued before the first MFMA; (3) no
// per-tile offset mask / compaction (its shuffle + LDS chain cost more than the skipped MFMAs: a 16-row tile uses
// nearly all offsets) -- absent neighbours simply gather nothing; (4) the MFMA is issued TRANSPOSED (A = weights,
// B = gathered rows), so a lane ends up with 4 consecutive output channels of one row: the tile is stored (and the
// residual read) as one contiguous float4 per lane instead of four 64-byte row fragments.
// (fp32 gathers hold twice the registers of bf16 ones until they are converted: one occupancy step less)
#ifndef C2_F32_U
#define C2_F32_U 8          // gathers per batch of the fp32-input variants with <= 2 column tiles (experiments: 4 with C2_F32_OCCDROP 0)
#endif
#ifndef C2_F32_OCCDROP
#define C2_F32_OCCDROP 1
#endif
#define C2_OCC(NTV, XB) ((NTV) <= 4 ? ((XB) ? C2_OCC_SMALL : C2_OCC_SMALL - C2_F32_OCCDROP) : (NTV) <= 9 ? ((XB) ? 3 : 2) : 2)
// KT / ST > 0: kernel size and slots per offset (Cin / 8) known at compile time (round 3: the shapes that carry the step -- K = 27
// with 16 / 32 / 64 input channels): the reduction loop is fully unrolled and every (offset, channel group) of a step is a
// constant per lane group -- the ~10 index instructions in front of each gather fold away.
// CMP (round 5, the statically shaped instances): the offsets NO row of the tile has are dropped before the reduction loop.  The
// kernel is bound by the dependent round trips of a wave (table -> gathers -> products, one per batch), not by a throughput: with
// the rows in raster order a 16-row tile of the 2 cm level uses 15.9 of the 27 offsets on average (a planar patch: 9), so the
// batches of a tile shrink from 14 to ~8 (stem) / from 2 to mostly 1 (16 -> 16).  One 27-lane pass over the tile's table in LDS
// gives the mask; the live offsets are then taken from it with scalar instructions (no list in memory).
template <int NT, bool WLDS, bool XBF, int NW, bool F32M, int KT, int ST, bool T16, bool CMP = false>
__device__ __forceinline__ void spconv_fwd2_body(const Conv2Args &a) {
    static_assert(!CMP || (KT == 27 && (ST == 2 || ST >= 4)), "offset compaction: the statically shaped instances");
    static_assert(!T16 || KT == 27, "the 16-bit kernel map is read by the statically shaped K = 27 instances");
    constexpr int NWT = NW;             // tiles of a workgroup's turn
    static_assert(!(F32M && XBF), "fp32 MFMA needs fp32 gathers");
    static_assert((KT > 0) == (ST > 0), "static shapes fix both the kernel size and the channel groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (16-wave workgroups run at 128 VGPRs: the fp32-input variants keep 4 gathers of 32 B in flight there instead of 8)
    // statically shaped, >= 32 input channels: a batch is KB whole offsets of Q = ceil(ST / 4) steps each
    constexpr int Q = ST >= 4 ? (ST + 3) / 4 : 1;
    constexpr int KB = ST >= 4 ? (Q >= 5 ? 2 : (8 / Q > 0 ? 8 / Q : 1)) : 1;
    constexpr int U = ST >= 4 ? KB * Q : (!XBF && NT <= 2) ? (NW == 16 ? 4 : C2_F32_U) : C2_U(NT);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, g = lane >> 4;
    const int slot = wave;
    const int K = KT ? KT : a.K, S = ST ? ST : a.S;
    const size_t wbytes = WLDS ? (size_t)K * S * NT * (F32M ? 512 : 256) : 0;
    int *tblS = (int *)(smem + wbytes) + wave * C2_TBL_INTS;
    float *redS = (float *)(smem + wbytes + NW * C2_TBL_INTS * 4);
    const unsigned short *Wb = WLDS ? (const unsigned short *)smem : a.Wp;
    const int nb = gridDim.x, b = blockIdx.x;
    const int lb = ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int ntg = (a.ntiles + NWT - 1) / NWT;
    const int per = (ntg + nb - 1) / nb;
    // Round 4: with `interleave` the nb / 8 workgroups of an XCD take the XCD's tile groups IN TURN (iteration i of workgroup j:
    // group xcd_base + i * (nb / 8) + j) instead of one contiguous range each: at any moment the XCD works on ONE window of
    // (nb / 8) * NW * 16 consecutive rows plus its neighbourhood, which fits its 4 MB L2, where 32 separate windows do not (the
    // stem gathers 272-byte rows from +-1 x-slab: measured 1032 MB of HBM-side traffic per launch against 245 MB algorithmic).
    const bool il = a.interleave && (nb & 7) == 0;
    const int tstride = il ? (nb >> 3) : 1;
    const int tg0 = il ? (b & 7) * (nb >> 3) * per + (b >> 3) : lb * per;
    const int tg1 = il ? min(ntg, ((b & 7) + 1) * (nb >> 3) * per) : min(ntg, tg0 + per);
    int v[7];
    // 16-bit table (round 4): entries (2d, 2d + 1) of the tile travel as ONE 32-bit word d = lane + it * 64 < 8 K; v[0..3] hold the
    // raw words, the LDS store decodes them (row + delta; 0x8000 = absent).  K is odd: the word that straddles the end of the
    // table's last row reads one of the two pad entries behind it.
    constexpr bool t16 = T16;                                  // (the host validated the table: d3_spconv_next_tbl16)
#define C2_LOAD_TBL(TILE)                                                                                     \
    {                                                                                                         \
        const int tile_ = (TILE);                                                                             \
        const long long base_ = (long long)tile_ * 16 * K, lim_ = (long long)a.Mout * K;                      \
        if constexpr (t16) {                                                                                  \
            _Pragma("unroll") for (int it = 0; it < 4; it++) {                                                \
                const int d = lane + it * 64;                                                                 \
                v[it] = (int)0x80008000u;                                                                     \
                if (tile_ < a.ntiles && d < 8 * K && base_ + 2 * d < lim_) v[it] = (int)a.tbl16[(base_ >> 1) + d]; \
            }                                                                                                 \
        } else {                                                                                              \
        _Pragma("unroll") for (int it = 0; it < 7; it++) {                                                    \
            const int e = lane + it * 64;                                                                     \
            v[it] = -1;                                                                                       \
            if (tile_ < a.ntiles && e < 16 * K && base_ + e < lim_) v[it] = a.tbl ? a.tbl[base_ + e] : (int)(base_ + e); \
        }                                                                                                     \
        }                                                                                                     \
    }
    if (C2_PREFETCH) C2_LOAD_TBL(tg0 * NWT + slot)
    float4 *bnS = (float4 *)(smem + wbytes + C2_WAVE_LDS_BASE(NT, NW));   // (mean, 1/std, gamma, beta) per channel
    if (a.bnx && t < NT * 16) {
        float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < a.Cout) {
            bp.x = a.bn_mean[t]; bp.y = rsqrtf(a.bn_var[t] + a.bn_eps);
            if (a.bn_relu) { bp.z = a.bn_gamma[t]; bp.w = a.bn_beta[t]; }
        }
        bnS[t] = bp;
    }
    if (WLDS) {
        const uint4 *src = (const uint4 *)a.Wp;
        uint4 *dst = (uint4 *)smem;
        const int n16 = (int)(wbytes >> 4);
        for (int i0 = 0; i0 < n16; i0 += 4 * 64 * NW) {
            uint4 w4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + q * 64 * NW + t; w4[q] = make_uint4(0u, 0u, 0u, 0u); if (i < n16) w4[q] = src[i]; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + q * 64 * NW + t; if (i < n16) dst[i] = w4[q]; }
        }
    }
    if (WLDS || a.bnx) __syncthreads();
    f32x4 ssum[NT], ssq[NT];   // per lane: its row's values, channels n*16 + g*4 + q
#pragma unroll
    for (int n = 0; n < NT; n++) { ssum[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; ssq[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int nsteps = (K * S + 3) >> 2;
    constexpr int NSTEPS_T = (KT * ST + 3) / 4;   // static shapes below 32 channels: steps of a tile
    const unsigned int xrowb = (unsigned int)a.ldx * (XBF ? 2u : 4u);
    const int rK = r * K;
    if (lane == 0) tblS[C2_TBL_SENT] = -1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, C2_RSRC_FLAGS);

    for (int tg = tg0; tg < tg1; tg += tstride) {
        const int tile = tg * NWT + slot;
        if (tile >= a.ntiles) continue;   // wave-uniform; there is no workgroup barrier inside this loop
        const int row0 = tile * 16;
        if (!C2_PREFETCH) C2_LOAD_TBL(tile)
        if constexpr (t16) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int d = lane + it * 64;
                if (d < 8 * K) {
                    const int e0 = 2 * d, e1 = e0 + 1;
                    const int lo = (int)(short)(v[it] & 0xFFFF), hi = v[it] >> 16;          // (arithmetic shift: sign-extended)
                    const int u0 = e0 / (KT ? KT : 1), u1 = e1 / (KT ? KT : 1);
                    // (entries behind the table's end are absent already: words not loaded are 0x80008000 and the one word
                    // that straddles the end carries a pad entry, which cm_pack16_kernel wrote as absent)
                    tblS[e0] = lo == -32768 ? -1 : row0 + u0 + lo;
                    tblS[e1] = hi == -32768 ? -1 : row0 + u1 + hi;
                }
            }
        } else {
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const int e = lane + it * 64;
            if (e < 16 * K) tblS[e] = v[it];
        }
        }
        if (C2_PREFETCH && tg + tstride < tg1) C2_LOAD_TBL(tile + NWT * tstride)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned int kmask = 0u;
        if constexpr (CMP) {      // bit k: some row of the tile has offset k (absent entries are -1: the AND of a column keeps the sign bit)
            int av = -1;
            if (lane < KT) {
#pragma unroll
                for (int rr = 0; rr < 16; rr++) av &= tblS[rr * KT + lane];
            }
            kmask = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)__ballot(lane < KT && av >= 0));
        }

        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Epilogue operands (residual, accumulate target, BatchNorm input) do not depend on the gathers: for narrow
        // outputs they are requested BEFORE the gathers and arrive with them; wider outputs request them in chunks of
        // NB column tiles at the epilogue (one round trip per chunk, not one per operand).
        constexpr int NB = NT <= 4 ? NT : 4;
        constexpr bool HOIST = NT == 1 || (NT == 2 && !XBF);   // (register budget of the bf16 variants: 128)
        f32x4 e_res[NB], e_out[NB], e_bnx[NB];
        const int urow = row0 + r;
#define C2_EPI_LOAD(N0)                                                                                       \
        _Pragma("unroll") for (int j = 0; j < NB; j++) {                                                      \
            const int col = ((N0) + j) * 16 + g * 4;                                                          \
            e_res[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; e_out[j] = e_res[j]; e_bnx[j] = e_res[j];                   \
            if ((N0) + j < NT && urow < a.Mout && col < a.Cout) {                                             \
                if (a.res) e_res[j] = *(const f32x4 *)(a.res + (long long)urow * a.ldr + col);                \
                if (a.accum) e_out[j] = *(const f32x4 *)(a.out + (long long)urow * a.ldo + col);              \
                if (a.bnx) {                                                                                  \
                    if (a.bnx_bf16) { const uint2 b2 = *(const uint2 *)((const unsigned short *)a.bnx + (long long)urow * a.ldbx + col); \
                        e_bnx[j] = (f32x4){__uint_as_float(b2.x << 16), __uint_as_float(b2.x & 0xFFFF0000u), __uint_as_float(b2.y << 16), __uint_as_float(b2.y & 0xFFFF0000u)}; } \
                    else e_bnx[j] = *(const f32x4 *)(a.bnx + (long long)urow * a.ldbx + col);                 \
                }                                                                                             \
            }                                                                                                 \
        }
        if (HOIST) { C2_EPI_LOAD(0) }
        // (CMP: ko[] = the live offsets of this batch, taken from the mask; KT = none)
        constexpr int NKO = !CMP ? 1 : (ST >= 4 ? KB : 2 * U);
        auto batch = [&](const int m0, const int (&ko)[NKO]) __attribute__((always_inline)) {
            uint4 rlo[U], rhi[U];
            int boff[U], idxv[U], c8v[U];
            // Round 3 (ISA review): the kernel-map entries of the whole batch are read from LDS back to back and unconditionally
            // (a clamped slot: a conditional read put an exec-masked branch and a full LDS wait in front of EVERY gather -- eight
            // serialized LDS round trips per batch), the row offset is one unsigned 32 x 32 -> 64 multiply-add (the signed
            // long long form took three), and the weight element of slot s is simply s * NT * 16 + r (k * S + c8 == s).
            int lidx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                int s = 4 * (m0 + u) + g, k;
                bool ok;
                if constexpr (CMP && ST < 4) {
                    // 16 input channels: a step is two live offsets (lane groups 0-1 the first, 2-3 the second)
                    k = (g >> 1) ? ko[(2 * u + 1) % NKO] : ko[(2 * u) % NKO];
                    c8v[u] = g & 1;
                    ok = k < KT;
                    s = k * ST + c8v[u];
                } else if (ST >= 4) {
                    // static shapes with >= 32 input channels: a step stays inside ONE offset (here m0 is the first OFFSET of the
                    // batch; the channel group is a constant per lane group; 136 channels run 5 steps per offset, the last one
                    // with a single live lane group)
                    k = CMP ? ko[(u / Q) % NKO] : m0 + u / Q;
                    c8v[u] = 4 * (u % Q) + g;
                    ok = (k < KT) && (c8v[u] < ST);
                    s = k * ST + c8v[u];
                } else {
                    k = ST ? s / (ST ? ST : 1) : (int)(__umul24((unsigned int)s, a.inv) >> 16);      // (24-bit multiplies: full rate)
                    c8v[u] = s - (int)__umul24((unsigned int)k, (unsigned int)S);
                    ok = (m0 + u < nsteps) && (k < K);
                }
                lidx[u] = ok ? rK + k : C2_TBL_SENT;                              // (the sentinel slot holds -1)
                boff[u] = ok ? s * (NT * 16) + r : r;                 // weight ELEMENT index (16 B bf16 / 32 B fp32 each)
            }
#pragma unroll
            for (int u = 0; u < U; u++) idxv[u] = tblS[lidx[u]];
            // raw buffer gathers: an absent neighbour (-1) lands beyond the buffer's extent and the hardware returns zeros
            // without a memory request -- no exec-masked branch, no zero fill and no 64-bit address per gather
            // (inputs beyond 2 GiB are refused by the host: 32-bit offsets, absent rows at offset 2^31)
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (CMP ? (ko[(ST >= 4 ? u / Q : 2 * u) % NKO] >= KT)
                        : ((ST > 0 && ST < 4 && m0 + u >= NSTEPS_T) || (ST >= 4 && m0 + u / Q >= KT))) {   // (folded where m0 is a constant; wave-uniform otherwise)
                    rlo[u] = make_uint4(0u, 0u, 0u, 0u); rhi[u] = rlo[u]; continue;
                }
                const unsigned int off = idxv[u] >= 0 ? __umul24((unsigned int)idxv[u], xrowb) + (unsigned int)c8v[u] * (XBF ? 16u : 32u) : 0x80000000u;
                rlo[u] = c2_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                if (!XBF) rhi[u] = c2_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rx, off + 16u, 0, 0));
                else rhi[u] = rlo[u];
            }
            if (WLDS || F32M) {
                __builtin_amdgcn_sched_barrier(0);   // every gather is in flight before the first conversion
                // (steps beyond nsteps gathered nothing: their products add zero, and without a branch per step the weight
                // fragments of the batch are read ahead of the products instead of one LDS round trip in front of each)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (CMP ? (ko[(ST >= 4 ? u / Q : 2 * u) % NKO] >= KT)
                            : ((ST > 0 && ST < 4 && m0 + u >= NSTEPS_T) || (ST >= 4 && m0 + u / Q >= KT))) continue;
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        uint4 wl, wh;
                        c2_wload<F32M>(Wb, boff[u] + n * 16, wl, wh);
                        // transposed product: D[m = channel][n = row] += W^T[channel][k] * X^T[k][row]
                        acc[n] = c2_mma<XBF, F32M>(acc[n], wl, wh, rlo[u], rhi[u]);
                    }
                }
            } else {
                // weights from global memory (L2)
                if (NT <= 4) {   // the fragments of step u + 1 are requested before the MFMAs of step u
                    uint4 wnext[NT];
#pragma unroll
                    for (int n = 0; n < NT; n++) wnext[n] = *((const uint4 *)Wb + boff[0] + n * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (m0 + u < nsteps) {   // wave-uniform
                            uint4 wcur[NT];
#pragma unroll
                            for (int n = 0; n < NT; n++) wcur[n] = wnext[n];
                            if (u + 1 < U && m0 + u + 1 < nsteps) {
#pragma unroll
                                for (int n = 0; n < NT; n++) wnext[n] = *((const uint4 *)Wb + boff[u + 1] + n * 16);
                            }
                            const bf16x8_t A = c2_cvt_raw<XBF>(rlo[u], rhi[u]);
#pragma unroll
                            for (int n = 0; n < NT; n++)
                                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wcur[n]), A, acc[n], 0, 0, 0);
                        }
                    }
                } else {         // (registers) the fragments of a step in batches of 4: one round trip per batch
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (m0 + u < nsteps) {   // wave-uniform
                            const bf16x8_t A = c2_cvt_raw<XBF>(rlo[u], rhi[u]);
#pragma unroll
                            for (int nb = 0; nb < NT; nb += 4) {
                                uint4 wv[4];
#pragma unroll
                                for (int j = 0; j < 4; j++) if (nb + j < NT) wv[j] = *((const uint4 *)Wb + boff[u] + (nb + j) * 16);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                    if (nb + j < NT) acc[nb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[j]), A, acc[nb + j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        };
        const int ko0[NKO] = {0};
        if constexpr (CMP) {
            unsigned int m = kmask;
#pragma unroll 1
            while (m) {      // (wave-uniform: the mask lives in scalar registers)
                int ko[NKO];
#pragma unroll
                for (int j = 0; j < NKO; j++) { ko[j] = m ? (int)__builtin_ctz(m) : KT; m &= m - 1u; }
                batch(0, ko);
            }
        } else if constexpr (ST >= 4 && (KT * Q > 56 || NT >= 3)) {   // (the stem: 135 steps; >= 48 output channels -- unrolled completely they spill)
#pragma unroll 1
            for (int k0 = 0; k0 < KT; k0 += KB) batch(k0, ko0);
        } else if constexpr (ST >= 4) {
#pragma unroll
            for (int k0 = 0; k0 < KT; k0 += KB) batch(k0, ko0);
        } else if constexpr (ST > 0) {
#pragma unroll
            for (int m0 = 0; m0 < NSTEPS_T; m0 += U) batch(m0, ko0);
        } else {
            for (int m0 = 0; m0 < nsteps; m0 += U) batch(m0, ko0);
        }
        // D layout: column (= output row) lane & 15, rows (= channels) (lane >> 4) * 4 + q
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += NB) {
            if (!HOIST) { C2_EPI_LOAD(n0) __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int n = n0 + j;
                if (n >= NT) continue;
                const int col = n * 16 + g * 4;
                if (urow < a.Mout && col < a.Cout) {   // Cout % 4 == 0 (checked on the host)
                    f32x4 vv = acc[n];
                    if (a.res) vv += e_res[j];
                    f32x4 *o = (f32x4 *)(a.out + (long long)urow * a.ldo + col);
                    if (a.accum) vv += e_out[j];
                    if (a.bnx) {
                        f32x4 xh;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float4 bp = bnS[col + q];
                            xh[q] = (e_bnx[j][q] - bp.x) * bp.y;
                            if (a.bn_relu && fmaf(xh[q], bp.z, bp.w) <= 0.f) vv[q] = 0.f;
                        }
                        if (a.obf16) *(uint2 *)((unsigned short *)a.out + (long long)urow * a.ldo + col) = make_uint2(pack2bf2(vv[0], vv[1]), pack2bf2(vv[2], vv[3]));
                        else *o = vv;
                        ssum[n] += vv; ssq[n] += vv * xh;
                    } else {
                        if (a.obf16) *(uint2 *)((unsigned short *)a.out + (long long)urow * a.ldo + col) = make_uint2(pack2bf2(vv[0], vv[1]), pack2bf2(vv[2], vv[3]));
                        else *o = vv;
                        ssum[n] += vv; ssq[n] += vv * vv;
                    }
                }
            }
        }
#undef C2_EPI_LOAD
        __builtin_amdgcn_wave_barrier();   // tblS is rewritten by the next tile
    }
#undef C2_LOAD_TBL
    if (a.part) {   // per-workgroup BatchNorm partials (fixed order: deterministic)
#pragma unroll
        for (int n = 0; n < NT; n++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float s1 = ssum[n][q], s2 = ssq[n][q];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (r == 0) { redS[wave * 2 * NT * 16 + n * 16 + g * 4 + q] = s1; redS[wave * 2 * NT * 16 + NT * 16 + n * 16 + g * 4 + q] = s2; }
            }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += redS[w * 2 * NT * 16 + t];
            a.part[(long long)b * 2 * NT * 16 + t] = s;
            if (a.part2) unsafeAtomicAdd(&a.part2[(b % C2_P2_ROWS) * 2 * NT * 16 + t], (double)s);
        }
    }
}
template <int NT, bool WLDS, bool XBF, int NW = 4, bool F32M = false, int KT = 0, int ST = 0, bool T16 = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 4 ? C2_OCC(NT, XBF) : 4, NW == 4 ? 8 : 4))) void spconv_fwd2_kernel(const Conv2Args a) {
    spconv_fwd2_body<NT, WLDS, XBF, NW, F32M, KT, ST, T16>(a);
}
// the statically shaped instances (K = 27, bf16 rows, weights in LDS) with the tile's dead offsets dropped (CMP above)
template <int NT, int NW, int ST, bool T16>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 4 ? C2_OCC(NT, true) : 4, NW == 4 ? 8 : 4))) void spconv_fwd2_c_kernel(const Conv2Args a) {
    spconv_fwd2_body<NT, true, true, NW, false, 27, ST, T16, true>(a);
}
// Workgroup-per-tile kernel (few-row levels): grid = (16-row tiles, column groups of NTW 16-wide tiles).  The
// W = blockDim.x/64 waves split the MFMA steps of the tile, their accumulators are summed through LDS in wave order,
// and the workgroup owns complete output columns: no atomics, no cross-workgroup reduction, deterministic.
template <int NTW, bool XBF, bool F32M = false>
__global__ __launch_bounds__(1024) void spconv_fwd2_split_kernel(const Conv2Args a) {
    static_assert(!(F32M && XBF), "fp32 MFMA needs fp32 gathers");
    constexpr int U = 4;
    constexpr int CW = NTW * 16;                     // output columns of this workgroup
    constexpr bool SMALL = !F32M && NTW <= (XBF ? 3 : 2);     // register budget: 128 VGPRs at 1024 threads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, g = lane >> 4;
    const int W = blockDim.x >> 6, K = a.K, S = a.S;
    int *tblS = (int *)smem;                         // 16*27 ints
    int *actS = tblS + C2_TBL_INTS;                  // 32 ints
    unsigned int *kmaskS = (unsigned int *)(actS + 32);   // 1 (+3 pad)
    float *redS = (float *)(kmaskS + 4);             // W * NTW*256 floats
    float *finS = redS + (size_t)W * NTW * 256;      // 16 x NTW*16: stored values
    float *fin2S = finS + 16 * NTW * 16;             // 16 x NTW*16: second statistic (v*v, or g*xhat)
    const int row0 = blockIdx.x * 16, n0 = blockIdx.y * NTW;
    // The kernel is a short chain of dependent memory round trips (kernel-map rows -> gathers -> epilogue operands), so
    // everything whose address is known up front is requested up front: the kernel-map rows and, for narrow column
    // groups, the epilogue operands of this thread's output elements e = t + i * blockDim.x (blockDim.x >= 256: at most
    // NTW of them).  Wide groups request the epilogue operands in one batch at the epilogue instead (registers).
    float e_res[NTW], e_out[NTW], e_bnx[NTW], e_mean[NTW], e_var[NTW], e_gam[NTW], e_bet[NTW];
#define C2S_EPI_LOAD                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NTW; i++) {                                                         \
        const int e = t + i * (int)blockDim.x;                                                                \
        const int row = e / CW, cl = e - row * CW;                                                            \
        const int u = row0 + row, col = n0 * 16 + cl;                                                         \
        e_res[i] = 0.f; e_out[i] = 0.f; e_bnx[i] = 0.f; e_mean[i] = 0.f; e_var[i] = 1.f; e_gam[i] = 0.f; e_bet[i] = 0.f; \
        if (e < 16 * CW && u < a.Mout && col < a.Cout) {                                                      \
            if (a.res) e_res[i] = a.res[(long long)u * a.ldr + col];                                          \
            if (a.accum) e_out[i] = a.out[(long long)u * a.ldo + col];                                        \
            if (a.bnx) {                                                                                      \
                {   /* fp32, or bf16 (round 6): ONE 32-bit load either way -- the word that holds the element */              \
                    const long long bi_ = (long long)u * a.ldbx + col;                                                \
                    const unsigned int bw_ = ((const unsigned int *)a.bnx)[a.bnx_bf16 ? (bi_ >> 1) : bi_];            \
                    e_bnx[i] = __uint_as_float(a.bnx_bf16 ? ((bi_ & 1) ? (bw_ & 0xFFFF0000u) : (bw_ << 16)) : bw_);   \
                }                                                                                                     \
                e_mean[i] = a.bn_mean[col]; e_var[i] = a.bn_var[col]; \
                if (a.bn_relu) { e_gam[i] = a.bn_gamma[col]; e_bet[i] = a.bn_beta[col]; }                     \
            }                                                                                                 \
        }                                                                                                     \
    }
    int v[2];
    const long long base = (long long)row0 * K, lim = (long long)a.Mout * K;
#pragma unroll
    for (int it = 0; it < 2; it++) {   // blockDim.x >= 256 and 16*K <= 432
        const int e = t + it * blockDim.x;
        v[it] = -1;
        if (e < 16 * K && base + e < lim) v[it] = a.tbl ? a.tbl[base + e] : (int)(base + e);
    }
    if (SMALL) { C2S_EPI_LOAD }
    if (t == 0) *kmaskS = 0u;
    __syncthreads();
    {
        unsigned int bits = 0u;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int e = t + it * blockDim.x;
            if (e < 16 * K) {
                tblS[e] = v[it];
                if (v[it] >= 0) bits |= 1u << (e - (int)(((unsigned int)e * a.invK) >> 16) * K);
            }
        }
        if (bits) atomicOr(kmaskS, bits);
    }
    __syncthreads();
    const unsigned int kmask = *kmaskS;
    const int na = __popc(kmask);
    if (t < K && ((kmask >> t) & 1u)) actS[__popc(kmask & ((1u << t) - 1u))] = t;
    __syncthreads();
    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; n++) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nsteps = (na * S + 3) >> 2;
    for (int m0 = wave; m0 < nsteps; m0 += W * U) {
        uint4 rlo[U], rhi[U];
        int boff[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int m = m0 + u * W;
            const int s = 4 * m + g;
            const int i = (int)(((unsigned int)s * a.inv) >> 16);
            const int c8 = s - i * S;
            const bool ok = (m < nsteps) && (i < na);
            const int k = actS[ok ? i : 0];
            const int idx = ok ? tblS[r * K + k] : -1;
            boff[u] = ((k * S + (ok ? c8 : 0)) * a.NT + n0) * 16 + r;      // weight ELEMENT index
            rlo[u] = make_uint4(0u, 0u, 0u, 0u); rhi[u] = rlo[u];
            if (idx >= 0) c2_load_raw<XBF>(a.x, (long long)idx * a.ldx + c8 * 8, rlo[u], rhi[u]);
        }
        if (SMALL) {
            // gathers and all weight fragments of the batch in flight together: one round trip per batch
            uint4 wv[U][NTW];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int n = 0; n < NTW; n++) {
                    wv[u][n] = make_uint4(0u, 0u, 0u, 0u);
                    if (m0 + u * W < nsteps && n0 + n < a.NT) wv[u][n] = *((const uint4 *)a.Wp + boff[u] + n * 16);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (m0 + u * W < nsteps) {
                    const bf16x8_t A = c2_cvt_raw<XBF>(rlo[u], rhi[u]);
#pragma unroll
                    for (int n = 0; n < NTW; n++)
                        if (n0 + n < a.NT) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8_t, wv[u][n]), acc[n], 0, 0, 0);
                }
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (m0 + u * W < nsteps) {
                    uint4 wv[NTW], wh[NTW];
#pragma unroll
                    for (int n = 0; n < NTW; n++) {
                        wv[n] = make_uint4(0u, 0u, 0u, 0u); wh[n] = wv[n];
                        if (n0 + n < a.NT) c2_wload<F32M>(a.Wp, boff[u] + n * 16, wv[n], wh[n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (F32M) {      // (not transposed here: first operand = gathered rows, second = weights)
#pragma unroll
                        for (int n = 0; n < NTW; n++)
                            if (n0 + n < a.NT) acc[n] = c2_mma<false, true>(acc[n], rlo[u], rhi[u], wv[n], wh[n]);
                    } else {
                    const bf16x8_t A = c2_cvt_raw<XBF>(rlo[u], rhi[u]);
#pragma unroll
                    for (int n = 0; n < NTW; n++)
                        if (n0 + n < a.NT) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8_t, wv[n]), acc[n], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NTW; n++)
#pragma unroll
        for (int q = 0; q < 4; q++) redS[(wave * NTW * 4 + n * 4 + q) * 64 + lane] = acc[n][q];
    if (!SMALL) { C2S_EPI_LOAD }
    __syncthreads();
    // final tile: element e = row * CW + col, summed in wave order
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        const int e = t + i * (int)blockDim.x;
        if (e >= 16 * CW) continue;
        const int row = e / CW, cl = e - row * CW;
        const int n = cl >> 4, ln = (row >> 2) * 16 + (cl & 15), q = row & 3;
        float v = 0.f, w2 = 0.f;
        for (int w = 0; w < W; w++) v += redS[(w * NTW * 4 + n * 4 + q) * 64 + ln];
        const int u = row0 + row, col = n0 * 16 + cl;
        if (u < a.Mout && col < a.Cout) {
            if (a.res) v += e_res[i];
            if (a.accum) v += e_out[i];
            if (a.bnx) {
                const float xh = (e_bnx[i] - e_mean[i]) * rsqrtf(e_var[i] + a.bn_eps);
                if (a.bn_relu && fmaf(xh, e_gam[i], e_bet[i]) <= 0.f) v = 0.f;
                w2 = v * xh;
            } else w2 = v * v;
            if (a.obf16) ((unsigned short *)a.out)[(long long)u * a.ldo + col] = (unsigned short)(pack2bf2(v, 0.f) & 0xFFFFu);
            else a.out[(long long)u * a.ldo + col] = v;
        } else { v = 0.f; w2 = 0.f; }
        finS[e] = v; fin2S[e] = w2;
    }
#undef C2S_EPI_LOAD
    if (a.part) {
        __syncthreads();
        if (t < 2 * CW) {
            const int cl = (t < CW) ? t : t - CW;
            float s = 0.f;
            for (int row = 0; row < 16; row++) s += (t < CW) ? finS[row * CW + cl] : fin2S[row * CW + cl];
            if (n0 * 16 + cl < a.NT * 16) {
                float *pp = &a.part[(long long)blockIdx.x * 2 * a.NT * 16 + (t < CW ? 0 : a.NT * 16) + n0 * 16 + cl];
                *pp = s;
                if (a.part2) unsafeAtomicAdd(&a.part2[(long long)(blockIdx.x % C2_P2_ROWS) * 2 * a.NT * 16 + (t < CW ? 0 : a.NT * 16) + n0 * 16 + cl], (double)s);
            }
        }
    }
}

#define C2_NW16_MAXNT 4      // 16-wave variants are instantiated for <= 4 column tiles
#define C2_GRIDCAP 1024      // persistent workgroups per convolution launch; levels of fewer 16-row tiles run the workgroup-per-tile kernel
static int c2_ncu() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t pr; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    return n;
}
struct Conv2Plan { int split, W, grid, wlds, ntw, gy, nw; size_t lds; };

static Conv2Plan conv2_plan(int Mout, int K, int Cin, int Cout, bool f32 = false) {
    Conv2Plan p;
    const int NT = (Cout + 15) / 16, ntiles = (Mout + 15) / 16;
    const size_t wbytes = (size_t)K * (Cin / 8) * NT * (f32 ? 512 : 256);
    p.ntw = NT; p.gy = 1; p.nw = 4;
    if (ntiles >= 1024) {
        p.split = 0; p.W = 1;
        // 16 waves around ONE LDS copy of a large weight set (one workgroup per CU), 4 waves per workgroup otherwise
        const size_t big_from = (size_t)d3_tune(D3T_C2_NW16_KB) * 1024, lds_max = (size_t)160 * 1024;
        if (NT <= C2_NW16_MAXNT && wbytes >= big_from && wbytes + C2_WAVE_LDS_BYTES(NT, 16) <= lds_max && wbytes + C2_WAVE_LDS_BYTES(NT, 16) <= 160 * 1024)
            p.nw = 16;
        const int ntg = (ntiles + p.nw - 1) / p.nw;
        int cap = C2_GRIDCAP;
        if (p.nw == 16) cap = c2_ncu();      // LDS admits one such workgroup per CU
        const int per = (ntg + cap - 1) / cap;
        p.grid = (ntg + per - 1) / per;
        p.wlds = (p.nw == 16 || wbytes + C2_WAVE_LDS_BYTES(NT, 4) <= 72 * 1024) ? 1 : 0;
        p.lds = (p.wlds ? wbytes : 0) + C2_WAVE_LDS_BYTES(NT, p.nw);
    } else {
        p.split = 1; p.grid = ntiles; p.wlds = 0;
        // few tiles: one column tile per workgroup (the gather is repeated per column group, from L2)
        if (ntiles < 256) p.ntw = 1; else if (NT > 4) p.ntw = (NT + 1) / 2;
        if (p.ntw > 7) p.ntw = 7;
        // (round 6: the 4-tile instance of the workgroup-per-tile kernel spills at its 128-register budget since the last-block finalize left
        // its tail -- 78 registers before, 128 + 64 B of scratch after, 13 -> 20 us per launch; the 5-tile instance, 86 registers, serves 64
        // columns with its fifth tile idle)
        if (p.ntw == 4) p.ntw = 5;
        p.gy = (NT + p.ntw - 1) / p.ntw;
        const int steps = K * (Cin / 8) / 4 + 1;     // upper bound of MFMA steps per tile
        int W = ntiles * p.gy >= 512 ? 4 : 8;
        if (ntiles * p.gy < 128) W = 16;
        while (W > 4 && W * 2 > steps) W >>= 1;
        p.W = W;
        p.lds = (size_t)(C2_TBL_INTS + 32 + 4) * 4 + (size_t)W * p.ntw * 1024 + (size_t)p.ntw * 2048;
    }
    return p;
}

extern "C" int d3_spconv_fwd3_nparts(int Mout, int Cin, int Cout);
// rows of the BatchNorm partial table a forward / data-gradient call may write: the larger of the two kernels that can serve the
// shape (which one runs depends on the tables the call is handed); d3_spconv_last_nparts() says how many the call did write
extern "C" int d3_spconv_fwd2_nparts(int Mout, int K, int Cin, int Cout) {
    const int g2 = conv2_plan(Mout, K, Cin, Cout).grid, g3 = K == 27 ? d3_spconv_fwd3_nparts(Mout, Cin, Cout) : 0;
    return g2 > g3 ? g2 : g3;
}
// flags: D3_CONV_F32 changes the weight footprint and with it the workgroup shape
extern "C" int d3_spconv_fwd2_nparts_ex(int Mout, int K, int Cin, int Cout, int flags) {
    const int g2 = conv2_plan(Mout, K, Cin, Cout, (flags & D3_CONV_F32) != 0).grid;
    const int g3 = (K == 27 && !(flags & D3_CONV_F32)) ? d3_spconv_fwd3_nparts(Mout, Cin, Cout) : 0;
    return g2 > g3 ? g2 : g3;
}

// which kernel d3_spconv_fwd2* runs for this shape: out[6] = {split (1: spconv_fwd2_split_kernel), waves per workgroup,
// grid.x, weights resident in LDS, column tiles per workgroup, grid.y}
extern "C" int d3_spconv_fwd2_plan(int Mout, int K, int Cin, int Cout, int *out) {
    if (!out || K < 1 || K > C2_MAXK || Cin < 8 || (Cin & 7) || Cout < 1 || Cout > 224) return D3_ERR_ARG;
    const Conv2Plan p = conv2_plan(Mout, K, Cin, Cout);
    out[0] = p.split; out[1] = p.split ? p.W : p.nw; out[2] = p.grid; out[3] = p.wlds; out[4] = p.ntw; out[5] = p.gy;
    return 0;
}

// (the dynamic-LDS attribute is per device: one flag per device ordinal)
static bool c2_attr_needed(bool *done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

#include <atomic>
static std::atomic<long long> g_t16_launches{0};       // launches that read a 16-bit kernel map (tests: the path really ran)
extern "C" long long d3_spconv_t16_launches(void) { return g_t16_launches.load(); }
// template arguments of the instance the last launch_fwd2* call of this thread ran: {NT, WLDS, XBF, NW, F32M, KT, ST} for
// spconv_fwd2_kernel, {NTW, XBF, F32M} for spconv_fwd2_split_kernel -- the profiling record names the kernel as rocprofv3 prints it
static thread_local int g_c2_inst[7];
static inline void c2_inst(int a0, int a1, int a2, int a3, int a4, int a5, int a6) {
    g_c2_inst[0] = a0; g_c2_inst[1] = a1; g_c2_inst[2] = a2; g_c2_inst[3] = a3; g_c2_inst[4] = a4; g_c2_inst[5] = a5; g_c2_inst[6] = a6;
}
template <int NT>
static int launch_fwd2_f32(const Conv2Args &a, const Conv2Plan &p, hipStream_t s) {
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev)) {
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, false, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        if constexpr (NT <= C2_NW16_MAXNT)
            D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, false, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if constexpr (NT <= C2_NW16_MAXNT) {
        if (p.nw == 16) { c2_inst(NT, 1, 0, 16, 1, 0, 0); spconv_fwd2_kernel<NT, true, false, 16, true><<<p.grid, 1024, p.lds, s>>>(a); D3_LAUNCH_CHECK(); return 0; }
    }
    c2_inst(NT, p.wlds ? 1 : 0, 0, 4, 1, 0, 0);
    if (p.wlds) spconv_fwd2_kernel<NT, true, false, 4, true><<<p.grid, 256, p.lds, s>>>(a);
    else spconv_fwd2_kernel<NT, false, false, 4, true><<<p.grid, 256, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}
// the statically shaped instances (K = 27, bf16 rows, weights in LDS)
template <int NT, int NW, int ST>
static int launch_fwd2_static(const Conv2Args &a, const Conv2Plan &p, hipStream_t s) {
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev))
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, true, NW, false, 27, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // the stem (136 -> 16: five products per offset) drops the offsets no row of a 16-row tile has before its reduction loop
    // (spconv_fwd2_c_kernel: 266 -> 231 us at 649 k rows; for the 16 / 32-channel instances the mask pass cost more than the dropped
    // steps saved -- measured in round 5 -- and they run spconv_fwd3_kernel on the lane table since round 6 anyway)
    if constexpr (ST == 17) {
        static bool attrc_done_dev[64] = {false};
        if (c2_attr_needed(attrc_done_dev)) {
            D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_c_kernel<NT, NW, ST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_c_kernel<NT, NW, ST, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        c2_inst(NT, 1, 1, NW, 0, 27, ST + (a.tbl16 ? 1000 : 0) + 4000);     // (+ 4000: spconv_fwd2_c_kernel, see bench.py's kernel naming)
        if (a.tbl16) { spconv_fwd2_c_kernel<NT, NW, ST, true><<<p.grid, 64 * NW, p.lds, s>>>(a); g_t16_launches++; }
        else spconv_fwd2_c_kernel<NT, NW, ST, false><<<p.grid, 64 * NW, p.lds, s>>>(a);
        D3_LAUNCH_CHECK();
        return 0;
    }
    c2_inst(NT, 1, 1, NW, 0, 27, ST);
    spconv_fwd2_kernel<NT, true, true, NW, false, 27, ST><<<p.grid, 64 * NW, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}
template <int NT>
static int launch_fwd2(const Conv2Args &a, const Conv2Plan &p, hipStream_t s) {
    if (a.f32) return launch_fwd2_f32<NT>(a, p, s);
    if (a.xbf16 && a.K == 27 && p.wlds && d3_tune(D3T_C2_STATIC) != 0) {
        if constexpr (NT == 1) {
            if (a.S == 2 && p.nw == 4) return launch_fwd2_static<1, 4, 2>(a, p, s);      // 16 -> 16
            if (a.S == 4 && p.nw == 16) return launch_fwd2_static<1, 16, 4>(a, p, s);    // 32 -> 16
            if (a.S == 17 && p.nw == 16) return launch_fwd2_static<1, 16, 17>(a, p, s);  // the stem: 134 (+2) -> 16
        }
        if constexpr (NT == 2) {
            if (a.S == 2 && p.nw == 16) return launch_fwd2_static<2, 16, 2>(a, p, s);    // 16 -> 32
            if (a.S == 4 && p.nw == 16) return launch_fwd2_static<2, 16, 4>(a, p, s);    // 32 -> 32
            if (a.S == 8 && p.nw == 16) return launch_fwd2_static<2, 16, 8>(a, p, s);    // 64 -> 32
        }
        // (48 -> 48 with the offset loop rolled: measured no faster than the generic instance -- 36 k rows are one tile per wave)
        // (32 -> 64 spills at 128 registers even with the rolled loop: generic instance)
    }
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev)) {   // allow more than 64 KB of dynamic LDS
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    if constexpr (NT <= C2_NW16_MAXNT) {
        if (p.nw == 16) {
            static bool attr16_done_dev[64] = {false};
            if (c2_attr_needed(attr16_done_dev)) {
                D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_kernel<NT, true, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            c2_inst(NT, 1, a.xbf16 ? 1 : 0, 16, 0, 0, 0);
            if (a.xbf16) spconv_fwd2_kernel<NT, true, true, 16><<<p.grid, 1024, p.lds, s>>>(a);
            else spconv_fwd2_kernel<NT, true, false, 16><<<p.grid, 1024, p.lds, s>>>(a);
            D3_LAUNCH_CHECK();
            return 0;
        }
    }
    c2_inst(NT, p.wlds ? 1 : 0, a.xbf16 ? 1 : 0, 4, 0, 0, 0);
    if (a.xbf16) {
        if (p.wlds) spconv_fwd2_kernel<NT, true, true><<<p.grid, 256, p.lds, s>>>(a);
        else spconv_fwd2_kernel<NT, false, true><<<p.grid, 256, p.lds, s>>>(a);
    } else {
        if (p.wlds) spconv_fwd2_kernel<NT, true, false><<<p.grid, 256, p.lds, s>>>(a);
        else spconv_fwd2_kernel<NT, false, false><<<p.grid, 256, p.lds, s>>>(a);
    }
    D3_LAUNCH_CHECK();
    return 0;
}
template <int NTW>
static int launch_fwd2_split(const Conv2Args &a, const Conv2Plan &p, hipStream_t s) {
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev)) {
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_split_kernel<NTW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_split_kernel<NTW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    }
    c2_inst(NTW, a.f32 ? 0 : (a.xbf16 ? 1 : 0), a.f32 ? 1 : 0, -1, 0, 0, 0);
    if (a.f32) {
        static bool attr32_done_dev[64] = {false};
        if (c2_attr_needed(attr32_done_dev))
            D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd2_split_kernel<NTW, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        spconv_fwd2_split_kernel<NTW, false, true><<<dim3(p.grid, p.gy), p.W * 64, p.lds, s>>>(a);
    } else
    if (a.xbf16) spconv_fwd2_split_kernel<NTW, true><<<dim3(p.grid, p.gy), p.W * 64, p.lds, s>>>(a);
    else spconv_fwd2_split_kernel<NTW, false><<<dim3(p.grid, p.gy), p.W * 64, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}

static thread_local const void *g_next_tbl16 = nullptr;
static thread_local const void *g_next_tblq = nullptr;
void d3_spconv_next_tbl16(const void *tbl16, const void *tblq) { g_next_tbl16 = tbl16; g_next_tblq = tblq; }
static thread_local int g_last_nparts = 0;
extern "C" int d3_spconv_last_nparts(void) { return g_last_nparts; }
void d3_spconv_set_last_nparts(int n) { g_last_nparts = n; }
// spconv3.hip
struct Conv3Bn { const void *x; const float *mean, *var, *gamma, *beta; int ldx, relu, xbf16; float eps; };
int d3_conv3_run(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, const float *res, int ldr, float *part,
                 double *part2, int Min, int Mout, int Cin, int Cout, int obf16, const Conv3Bn *bn, int *nparts_out, hipStream_t s);
extern "C" int d3_spconv_fwd3_nparts(int Mout, int Cin, int Cout);
// second-level partial table of the NEXT forward / data-gradient call of this thread (same hand-over as the 16-bit map hint):
// [C2_P2_ROWS][2][ceil(Cout / 16) * 16] doubles, zeroed by the caller; ignored when the call takes no partials
static thread_local double *g_next_part2 = nullptr;
void d3_spconv_next_part2(double *part2) { g_next_part2 = part2; }

struct Conv2Bn { const float *x, *mean, *var, *gamma, *beta; int ldx, relu; float eps; int xbf16; };

static int conv2_run(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, const float *res, int ldr,
                     float *part, int Min, int Mout, int K, int Cin, int Cout, int flags, const Conv2Bn *bn, void *stream) {
    D3_CLEAR();
    const void *tbl16 = g_next_tbl16, *tblq = g_next_tblq;      // the hints belong to THIS call, whatever it does with them
    g_next_tbl16 = nullptr; g_next_tblq = nullptr;
    double *part2 = g_next_part2;
    g_next_part2 = nullptr;
    if (Mout <= 0) return 0;
    if (K < 1 || K > C2_MAXK || Cin < 8 || (Cin & 7) || Cout < 1 || Cout > 224) return D3_ERR_ARG;
    if (tbl == nullptr && K != 1) return D3_ERR_ARG;
    const int xbf16 = (flags & D3_CONV_XBF16) ? 1 : 0;
    const int f32 = (flags & D3_CONV_F32) ? 1 : 0;
    if (f32 && xbf16) return D3_ERR_ARG;      // the reference-precision path gathers fp32 rows
    if ((xbf16 && (ldx & 7)) || (!xbf16 && (ldx & 3)) || ldx < Cin || ldo < Cout) return D3_ERR_ARG;
    if ((Cout & 3) || (ldo & 3) || (res && (ldr & 3))) return D3_ERR_ARG;   // float4 epilogue
    hipStream_t s = d3_stream(stream);
    // round 6: the big levels' K = 27 layers on the lane table (spconv3.hip) -- bf16 rows, no accumulate-into
    if (tblq && tbl && K == 27 && xbf16 && !f32 && !(flags & D3_CONV_ACCUM) && d3_tune(D3T_C3) != 0 &&
        (d3_tune(D3T_C3) == 1 || (d3_tune(D3T_C3) == 2 && !bn) || (d3_tune(D3T_C3) == 3 && bn) || (d3_tune(D3T_C3) == 4 && !bn && !res) || (d3_tune(D3T_C3) == 5 && res)) &&      // (2 .. 5: debugging -- forward only / data gradients only / plain forward / residual forward)
        Mout >= C2_GRIDCAP * 16 && d3_spconv_fwd3_nparts(Mout, Cin, Cout) > 0 && !(bn && res)) {
        Conv3Bn b3;
        if (bn) b3 = Conv3Bn{bn->x, bn->mean, bn->var, bn->gamma, bn->beta, bn->ldx, bn->relu, bn->xbf16, bn->eps};
        int np = 0;
        const int rc3 = d3_conv3_run(x, ldx, tblq, Wp, out, ldo, res, ldr, part, part ? part2 : nullptr, Min, Mout, Cin, Cout,
                                     (flags & D3_CONV_OUTBF16) ? 1 : 0, bn ? &b3 : nullptr, &np, s);
        if (rc3 != D3_ERR_ARG) { g_last_nparts = np; return rc3; }
    }
    Conv2Args a;
    a.x = x; a.tbl = tbl; a.Wp = (const unsigned short *)Wp; a.out = out; a.res = res; a.part = part;
    a.part2 = part ? part2 : nullptr;
    a.ldx = ldx; a.ldo = ldo; a.ldr = ldr; a.Mout = Mout; a.K = K; a.Cout = Cout; a.S = Cin / 8;
    a.inv = (65536u + a.S - 1) / a.S;
    a.invK = (65536u + K - 1) / K;
    a.interleave = d3_tune(D3T_C2_INTERLEAVE) != 0 ? 1 : 0;
    a.tbl16 = (tbl && tbl16 && K == 27 && !f32 && xbf16) ? (const unsigned int *)tbl16 : nullptr;   // (only the static instances launch with it)
    a.xbf16 = xbf16; a.f32 = f32; a.accum = (flags & D3_CONV_ACCUM) ? 1 : 0; a.ntiles = (Mout + 15) / 16;
    a.obf16 = (flags & D3_CONV_OUTBF16) ? 1 : 0;
    if (a.obf16 && (a.accum || res || f32)) return D3_ERR_ARG;
    {   // the last row of a column view ends after Cin elements; an absent neighbour's offset (2^32 - row bytes + ...) must stay outside
        const unsigned long long elt = xbf16 ? 2ull : 4ull, rowb = (unsigned long long)ldx * elt;
        const unsigned long long xb = Min > 0 ? ((unsigned long long)(Min - 1) * ldx + Cin) * elt : 0ull;
        a.xbytes = (Min > 0 && Min < (1 << 24) && rowb < (1ull << 24) && xb <= 0x7FFFFFFFull) ? (unsigned int)xb : 0u;   // (24-bit row x row-bytes multiply; absent rows address 2 GiB)
    }
    a.bnx = nullptr; a.bn_mean = a.bn_var = a.bn_gamma = a.bn_beta = nullptr; a.ldbx = 0; a.bn_relu = 0; a.bn_eps = 0.f; a.bnx_bf16 = 0;
    if (bn) {
        if (bn->ldx & 3) return D3_ERR_ARG;
        a.bnx = bn->x; a.bn_mean = bn->mean; a.bn_var = bn->var; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta;
        a.ldbx = bn->ldx; a.bn_relu = bn->relu; a.bn_eps = bn->eps; a.bnx_bf16 = bn->xbf16;
    }
    const Conv2Plan p = conv2_plan(Mout, K, Cin, Cout, f32 != 0);
    g_last_nparts = p.grid;
    if (!p.split && a.xbytes == 0u) return D3_ERR_RANGE;   // the wave-per-tile kernel addresses x through a raw buffer: <= 2 GiB, < 2^24 rows
    const double bytes = (xbf16 ? 2.0 : 4.0) * (double)Min * Cin + (a.obf16 ? 2.0 : 4.0) * (double)Mout * Cout + (f32 ? 4.0 : 2.0) * (double)K * Cin * Cout +
                         (tbl ? 4.0 * (double)Mout * K : 0.0) + (res ? 4.0 * (double)Mout * Cout : 0.0);
    void *pr = d3_prof_begin(p.split ? 2 : 0, bytes, 0.0, s);
    auto tag_rec = [&]() {
        if (!pr) return;
        const int dims[5] = {Min, Mout, K, Cin, Cout};
        for (int i = 0; i < 5; i++) d3_prof_tag(pr, i, dims[i]);
        for (int i = 0; i < 7; i++) d3_prof_tag(pr, 5 + i, g_c2_inst[i]);
    };
    int rc;
    a.NT = (Cout + 15) / 16;
    if (p.split) {
        switch (p.ntw) {
            case 1: rc = launch_fwd2_split<1>(a, p, s); break;
            case 2: rc = launch_fwd2_split<2>(a, p, s); break;
            case 3: rc = launch_fwd2_split<3>(a, p, s); break;
            case 4: rc = launch_fwd2_split<4>(a, p, s); break;
            case 5: rc = launch_fwd2_split<5>(a, p, s); break;
            case 6: rc = launch_fwd2_split<6>(a, p, s); break;
            default: rc = launch_fwd2_split<7>(a, p, s); break;
        }
        tag_rec();
        d3_prof_end(pr, s);
        return rc;
    }
#define C2_CASE(NTV) case NTV: rc = launch_fwd2<NTV>(a, p, s); break;
    switch ((Cout + 15) / 16) {
        C2_CASE(1) C2_CASE(2) C2_CASE(3) C2_CASE(4) C2_CASE(5) C2_CASE(6) C2_CASE(7) C2_CASE(8) C2_CASE(9)
        C2_CASE(10) C2_CASE(11) C2_CASE(12) C2_CASE(13) C2_CASE(14)
        default: rc = D3_ERR_ARG;
    }
#undef C2_CASE
    tag_rec();
    d3_prof_end(pr, s);
    return rc;
}

extern "C" int d3_spconv_fwd2(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo,
                              const float *res, int ldr, float *part, int Min, int Mout, int K, int Cin, int Cout,
                              int flags, void *stream) {
    return conv2_run(x, ldx, tbl, Wp, out, ldo, res, ldr, part, Min, Mout, K, Cin, Cout, flags, nullptr, stream);
}

// Data gradient of a BatchNorm -> ReLU -> convolution unit with the BatchNorm backward reductions fused in: the stored
// value is g = (sum_k dy[tbl[u,k]] @ Wk) * relu'(bn(bnx[u])) and part receives (sum g, sum g * xhat) per channel, where
// xhat = (bnx - mean) * rsqrt(var + eps): exactly what d3_bn_relu_bwd's reduction pass computes from a second read of
// x and dy.  bnx (Mout, ldbx) fp32 is the BatchNorm INPUT.
extern "C" int d3_spconv_fwd2_bnbwd(const void *x, int ldx, const int *tbl, const void *Wp, float *out, int ldo, float *part,
                                    const float *bnx, int ldbx, const float *mean, const float *var, const float *gamma,
                                    const float *beta, float eps, int relu, int Min, int Mout, int K, int Cin, int Cout,
                                    int flags, void *stream) {
    Conv2Bn bn{bnx, mean, var, gamma, beta, ldbx, relu, eps, (flags & D3_CONV_BNXBF16) ? 1 : 0};
    return conv2_run(x, ldx, tbl, Wp, out, ldo, nullptr, 0, part, Min, Mout, K, Cin, Cout, flags, &bn, stream);
}

// ------------------------------------------------------------------------------ weight gradient
// dW[k] = sum_u x[tbl[u,k],:]^T dy[u,:].  One operand is read contiguously ("stationary": rows u of the table),
// the other is gathered through the table; the host gathers the narrower one (the gather is re-done per offset).
//   P[k] (Cg x Cs) = sum_rows G[tbl[row,k],:]^T S[row,:]       MFMA: M = gathered channel, N = stationary channel,
//                                                               reduction = 32 rows per v_mfma_f32_16x16x32_bf16
// Both MFMA operands need 8 consecutive ROWS per lane, i.e. columns of the row-major matrices: each wave stages its
// 32-row chunk transposed in a private LDS region (St once per chunk, shared by all offsets of the pass -- the
// spconv.hip kernel re-read dy once per offset: profiles/r01_h, 85 MB of traffic against 8 MB algorithmic).
// grid = (row splits R, offset groups, tile passes).  A wave keeps 16 accumulator tiles: OPW = 16/TPO offsets x
// TPO tiles per offset; the 4 waves of a workgroup take alternate chunks and are summed through LDS in wave order;
// with R > 1 the workgroup writes a partial dW that wgrad2_reduce_kernel sums in split order (deterministic; no
// atomics).
#define WG2_LDT 40      // shorts per transposed LDS row: 32 rows + 8 pad (80 B)
#ifndef WG2_TARGET_WGS
#define WG2_TARGET_WGS 512   // workgroups per launch the row split aims at
#endif
#ifndef WG2_PART_MB
#define WG2_PART_MB 8         // cap of the partial-dW buffer
#endif
#ifndef WG2_T
#define WG2_T 16        // accumulator tiles per wave (64 AGPRs): occupancy matters more than reuse here
#endif

struct Wg2Args {
    const void *G; const void *Sm; const int *tbl; float *dst;
    int ldg, lds, gbf16, sbf16;
    int Ms, K, mt, nt;          // mt / nt: 16-channel tiles of the gathered / stationary operand
    int Cg8, Cs8;               // 8-channel units per row
    unsigned int invg, invs;    // ceil(65536 / Cg8), ceil(65536 / Cs8)
    int cpw;                    // chunks per workgroup
    int gx, flipk, Cin, Cout;   // gx: the gathered operand is x (P = dW[k]); else it is dy (P = dW[k]^T)
    int rsg, dg, rss, dss;      // row-major LDS images (TR kernels): row stride and 8-row shift in bytes, per operand
    int imgg, imgs;             // image sizes in bytes
};

// Row-major LDS image of a 32-row chunk read back through gfx950's transposing LDS read.  ds_read_b64_tr_b16: the 16 lanes of
// a group address a 4 x 16 bf16 block (lane i: row i/4, columns 4(i%4)..+3) and lane i receives column i, rows 0..3 -- two
// reads give the 8 consecutive rows of one channel that both MFMA operands need, without the 8 x ds_write_b16 transposed
// staging (~80 instructions per MFMA in the first version of this kernel).  A 32-lane half of the wave holds the blocks of
// rows 8g.. and 8(g+1)..: the row stride RSB (a multiple of 32 B, odd multiple where C*2 is a multiple of 128) and a shift
// D per 8 rows keep the eight 32-byte row segments of a half on distinct banks.
typedef short v4s16_t __attribute__((ext_vector_type(4)));
typedef short v8s16_t __attribute__((ext_vector_type(8)));
static void wg2_img(int C8, int *rsb, int *d, int *bytes) {
    int r = (C8 * 16 + 31) / 32 * 32;
    if ((r & 127) == 0) r += 32;
    const int dd = (r & 63) == 0 ? 32 : 128;
    *rsb = r; *d = dd; *bytes = 32 * r + 3 * dd;
}
__device__ __forceinline__ void wg2_put_r(unsigned char *img, int rsb, int d, int c8, int row, uint4 v) {
    *(uint4 *)(img + row * rsb + (row >> 3) * d + c8 * 16) = v;
}
// lane base of the fragment reads: rows 8g + (r>>2) (+4 for the second read), 8 bytes per lane inside the 32-byte tile row
__device__ __forceinline__ int wg2_lane_base(int rsb, int d, int r, int g) { return (8 * g + (r >> 2)) * rsb + g * d + (r & 3) * 8; }
__device__ __forceinline__ bf16x8_t wg2_frag_tr(const unsigned char *img, int lane_base, int rsb, int tile) {
    typedef v4s16_t __attribute__((address_space(3))) *lds_p;
    const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + lane_base + tile * 32));
    const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + lane_base + tile * 32 + 4 * rsb));
    const v8s16_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ uint4 wg2_load8(const void *p, int bf16, long long off) {
    if (bf16) return *(const uint4 *)((const unsigned short *)p + off);
    const float4 f0 = *(const float4 *)((const float *)p + off);
    const float4 f1 = *(const float4 *)((const float *)p + off + 4);
    return make_uint4(pack2bf2(f0.x, f0.y), pack2bf2(f0.z, f0.w), pack2bf2(f1.x, f1.y), pack2bf2(f1.z, f1.w));
}
__device__ __forceinline__ void wg2_store_t(unsigned short *T, int c8, int row, uint4 v) {
    unsigned short *d = T + (c8 * 8) * WG2_LDT + row;
    d[0 * WG2_LDT] = (unsigned short)(v.x & 0xFFFFu); d[1 * WG2_LDT] = (unsigned short)(v.x >> 16);
    d[2 * WG2_LDT] = (unsigned short)(v.y & 0xFFFFu); d[3 * WG2_LDT] = (unsigned short)(v.y >> 16);
    d[4 * WG2_LDT] = (unsigned short)(v.z & 0xFFFFu); d[5 * WG2_LDT] = (unsigned short)(v.z >> 16);
    d[6 * WG2_LDT] = (unsigned short)(v.w & 0xFFFFu); d[7 * WG2_LDT] = (unsigned short)(v.w >> 16);
}

template <int TPO, int NU, bool TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WG2_T > 16 ? (NU <= 7 ? 2 : 1) : (NU <= 2 ? 3 : NU <= 7 ? 2 : 1), 8))) void spconv_wgrad2_kernel(const Wg2Args a) {
    constexpr int OPW = WG2_T / TPO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, g = lane >> 4;
    const int K = a.K;
    // per-wave LDS: the two operand images (TR: row-major, else transposed mt*16 / nt*16 x LDT bf16), table chunk 32*K ints
    const size_t gt_bytes = TR ? (size_t)a.imgg : (size_t)a.mt * 16 * WG2_LDT * 2, st_bytes = TR ? (size_t)a.imgs : (size_t)a.nt * 16 * WG2_LDT * 2;
    const size_t wave_bytes = gt_bytes + st_bytes + (size_t)32 * C2_MAXK * 4;
    unsigned short *Gt = (unsigned short *)(smem + (size_t)wave * wave_bytes);
    unsigned short *St = (unsigned short *)((unsigned char *)Gt + gt_bytes);
    int *tblW = (int *)((unsigned char *)St + st_bytes);
    const int lbg = wg2_lane_base(a.rsg, a.dg, r, g), lbs = wg2_lane_base(a.rss, a.dss, r, g);
    const int k0 = blockIdx.y * OPW;
    const int tile0 = blockIdx.z * TPO, ntl = a.mt * a.nt;
    const int nchunks = (a.Ms + 31) >> 5;
    const int c_begin = blockIdx.x * a.cpw, c_end = min(nchunks, c_begin + a.cpw);
    // stationary column window of this pass (8-channel units)
    int sc8lo = 0, sc8n = a.Cs8;
    {
        const int tlast = min(ntl, tile0 + TPO) - 1;
        if (tlast >= tile0 && tile0 / a.nt == tlast / a.nt) {
            sc8lo = (tile0 % a.nt) * 2;
            sc8n = min(a.Cs8, (tlast % a.nt + 1) * 2) - sc8lo;
        }
    }

    f32x4 acc[OPW][TPO];
#pragma unroll
    for (int j = 0; j < OPW; j++)
#pragma unroll
        for (int i = 0; i < TPO; i++) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c = c_begin + wave; c < c_end; c += 4) {
        const int u0 = c * 32;
        // table chunk (32 rows x K, contiguous) -> LDS
        if (a.tbl) {   // loads first, LDS stores after (a rolled loop would serialise one round trip per pass)
            const long long base = (long long)u0 * K, lim = (long long)a.Ms * K;
            int v[14];
#pragma unroll
            for (int it = 0; it < 14; it++) {
                const int e = lane + it * 64;
                v[it] = -1;
                if (e < 32 * K && base + e < lim) v[it] = a.tbl[base + e];
            }
#pragma unroll
            for (int it = 0; it < 14; it++) {
                const int e = lane + it * 64;
                if (e < 32 * K) tblW[e] = v[it];
            }
        }
        // stationary rows, transposed (batches of 4 units per lane in flight); only the columns this pass's tiles
        // use (a wide stationary operand with a narrow gathered one is split into column passes by the host)
        for (int ub = 0; ub < 32 * sc8n; ub += 256) {
            uint4 sv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int unit = ub + q * 64 + lane;
                sv[q] = make_uint4(0u, 0u, 0u, 0u);
                if (unit < 32 * sc8n) {
                    const int row = unit / sc8n, c8 = sc8lo + unit - row * sc8n;
                    if (u0 + row < a.Ms) sv[q] = wg2_load8(a.Sm, a.sbf16, (long long)(u0 + row) * a.lds + c8 * 8);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int unit = ub + q * 64 + lane;
                if (unit < 32 * sc8n) {
                    const int row = unit / sc8n, c8 = sc8lo + unit - row * sc8n;
                    if constexpr (TR) wg2_put_r((unsigned char *)St, a.rss, a.dss, c8, row, sv[q]);
                    else wg2_store_t(St, c8, row, sv[q]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // offsets in groups of PF: the gathers of a group are issued together (memory-level parallelism -- every
        // offset is one dependent LDS -> L2/HBM -> LDS -> MFMA chain, and the MFMA work per offset is tiny)
#ifndef WG2_PF1
#define WG2_PF1 4
#endif
#ifndef WG2_PF2
#define WG2_PF2 2
#endif
        constexpr int PF = NU == 1 ? WG2_PF1 : NU == 2 ? WG2_PF2 : 1;
        uint4 pre[PF][NU];
        bool pre_any[PF];
#pragma unroll
        for (int j0 = 0; j0 < OPW; j0 += PF) {
#pragma unroll
            for (int pf = 0; pf < PF; pf++) {
                const int k = k0 + j0 + pf;
                bool any = false;
#pragma unroll
                for (int q = 0; q < NU; q++) {
                    const int unit = lane + q * 64;
                    pre[pf][q] = make_uint4(0u, 0u, 0u, 0u);
                    if (j0 + pf < OPW && k < K && unit < 32 * a.Cg8) {
                        const int row = (int)(((unsigned int)unit * a.invg) >> 16), c8 = unit - row * a.Cg8;
                        int idx = -1;
                        if (u0 + row < a.Ms) idx = a.tbl ? tblW[row * K + k] : (u0 + row);
                        if (idx >= 0) { pre[pf][q] = wg2_load8(a.G, a.gbf16, (long long)idx * a.ldg + c8 * 8); any = true; }
                    }
                }
                pre_any[pf] = __any(any) != 0;
            }
#pragma unroll
            for (int pf = 0; pf < PF; pf++) {
                const int j = j0 + pf;
                if (j < OPW && k0 + j < K && pre_any[pf]) {   // uniform
#pragma unroll
                    for (int q = 0; q < NU; q++) {
                        const int unit = lane + q * 64;
                        if (unit < 32 * a.Cg8) {
                            const int row = (int)(((unsigned int)unit * a.invg) >> 16), c8 = unit - row * a.Cg8;
                            if constexpr (TR) wg2_put_r((unsigned char *)Gt, a.rsg, a.dg, c8, row, pre[pf][q]);
                            else wg2_store_t(Gt, c8, row, pre[pf][q]);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int i = 0; i < TPO; i++) {
                        const int tile = tile0 + i;
                        if (tile < ntl) {   // uniform
                            const int mi = tile / a.nt, ni = tile - mi * a.nt;
                            bf16x8_t av, bv;
                            if constexpr (TR) {
                                av = wg2_frag_tr((const unsigned char *)Gt, lbg, a.rsg, mi);
                                bv = wg2_frag_tr((const unsigned char *)St, lbs, a.rss, ni);
                            } else {
                                av = __builtin_bit_cast(bf16x8_t, *(const uint4 *)&Gt[(mi * 16 + r) * WG2_LDT + g * 8]);
                                bv = __builtin_bit_cast(bf16x8_t, *(const uint4 *)&St[(ni * 16 + r) * WG2_LDT + g * 8]);
                            }
                            acc[j < OPW ? j : 0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[j < OPW ? j : 0][i], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // St / tblW are rewritten by the next chunk
    }
    // sum the four waves through LDS (8 tiles per round) and store
    __syncthreads();
    float *red = (float *)smem;   // 4 waves x 8 tiles x 256 floats = 32 KB
    const long long wsz = (long long)K * a.Cin * a.Cout;
    float *dst = a.dst + (long long)blockIdx.x * wsz;
#pragma unroll
    for (int rd = 0; rd < WG2_T / 8; rd++) {
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) {
            const int f = rd * 8 + q8, j = f / TPO, i = f % TPO;
#pragma unroll
            for (int q = 0; q < 4; q++) red[((wave * 8 + q8) * 4 + q) * 64 + lane] = acc[j][i][q];
        }
        __syncthreads();
        // wave w finishes tiles 2w, 2w+1 of the round
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q8 = wave * 2 + h, f = rd * 8 + q8, j = f / TPO, i = f % TPO;
            const int k = k0 + j, tile = tile0 + i;
            if (k < K && tile < ntl) {
                const int mi = tile / a.nt, ni = tile - mi * a.nt;
                const int wk = a.flipk ? (K - 1 - k) : k;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; w++) v += red[((w * 8 + q8) * 4 + q) * 64 + lane];
                    const int cg = mi * 16 + g * 4 + q, cs = ni * 16 + r;
                    const int ci = a.gx ? cg : cs, co = a.gx ? cs : cg;
                    if (ci < a.Cin && co < a.Cout) dst[((long long)wk * a.Cin + ci) * a.Cout + co] = v;
                }
            }
        }
        __syncthreads();
    }
}

// Wide-stationary weight gradient (the stem: x 136 channels stationary, dy 16 channels gathered, K = 27).  The generic kernel
// above gives a workgroup 16 accumulator tiles, i.e. 4 offsets x 4 of the 9 column tiles: 21 (offset group, column pass)
// combinations, each of which re-reads its slice of x and RE-GATHERS dy -- 1.96 GB of fabric traffic per launch against 288 MB
// algorithmic (profiles/r02_g: 1.14 ms, alone on the GPU at the end of the backward, on the critical path).  Here ONE
// 16-wave workgroup holds all K x nt = 243 tiles: x's 32-row chunk is staged (transposed) once and shared by all waves; wave
// w owns offsets {w, w + 16} with all nt column tiles (18 accumulator tiles = 72 VGPRs), so every dy row is gathered exactly
// once per offset; the next chunk's kernel-map rows and x units are requested before the current chunk's MFMAs.  Each tile
// has a single owner: no cross-wave reduction; row splits write partial dW summed by the fixed-order reduction.
#define WGW_WAVES 16
#define WGW_MAXNT 9
template <int NTV, bool TR>
__global__ __launch_bounds__(WGW_WAVES * 64) void spconv_wgrad2_wide_kernel(const Wg2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, g = lane >> 4;
    const int K = a.K;
    const size_t st_bytes = TR ? (size_t)a.imgs : (size_t)NTV * 16 * WG2_LDT * 2;
    const size_t gslot = TR ? (size_t)a.imgg : (size_t)16 * WG2_LDT * 2;          // one gathered 32 x 16 image
    unsigned short *St = (unsigned short *)smem;                                   // stationary chunk, shared by the waves
    int *tblS = (int *)(smem + st_bytes);                                          // 32 x K
    unsigned short *Gt = (unsigned short *)((unsigned char *)(tblS + 32 * C2_MAXK) + (size_t)wave * 2 * gslot);   // 2 slots per wave
    const int lbg = wg2_lane_base(a.rsg, a.dg, r, g), lbs = wg2_lane_base(a.rss, a.dss, r, g);
    const int nchunks = (a.Ms + 31) >> 5;
    const int c_begin = blockIdx.x * a.cpw, c_end = min(nchunks, c_begin + a.cpw);
    const int k0 = wave, k1 = wave + WGW_WAVES;                                    // this wave's offsets
    f32x4 acc[2][NTV];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < NTV; i++) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // prefetch registers: one kernel-map entry and one 8-channel unit of the stationary operand per thread
    const int sunits = 32 * a.Cs8;                      // <= 1024 (host check)
    int tv = -1;
    uint4 sv = make_uint4(0u, 0u, 0u, 0u);
    auto prefetch = [&](int c) {
        const int u0 = c * 32;
        tv = -1;
        if (a.tbl && t < 32 * K) { const long long e = (long long)u0 * K + t; if (e < (long long)a.Ms * K) tv = a.tbl[e]; }
        sv = make_uint4(0u, 0u, 0u, 0u);
        if (t < sunits) {
            const int row = t / a.Cs8, c8 = t - row * a.Cs8;
            if (u0 + row < a.Ms) sv = wg2_load8(a.Sm, a.sbf16, (long long)(u0 + row) * a.lds + c8 * 8);
        }
    };
    if (c_begin < c_end) prefetch(c_begin);
    for (int c = c_begin; c < c_end; c++) {
        const int u0 = c * 32;
        if (t < 32 * K) tblS[t] = a.tbl ? tv : (u0 + t < a.Ms ? u0 + t : -1);
        if (t < sunits) {
            const int row = t / a.Cs8, c8 = t - row * a.Cs8;
            if constexpr (TR) wg2_put_r((unsigned char *)St, a.rss, a.dss, c8, row, sv);
            else wg2_store_t(St, c8, row, sv);
        }
        __syncthreads();
        if (c + 1 < c_end) prefetch(c + 1);
        // gathers of this wave's two offsets (32 rows x 16 channels = 64 units: one per lane and offset)
        uint4 gv[2];
        bool any[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = j == 0 ? k0 : k1;
            gv[j] = make_uint4(0u, 0u, 0u, 0u);
            bool got = false;
            if (k < K) {
                const int row = lane >> 1, c8 = lane & 1;
                const int idx = (u0 + row < a.Ms) ? tblS[row * K + k] : -1;
                if (idx >= 0) { gv[j] = wg2_load8(a.G, a.gbf16, (long long)idx * a.ldg + c8 * 8); got = true; }
            }
            any[j] = __any(got) != 0;
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (any[j]) {
                if constexpr (TR) wg2_put_r((unsigned char *)Gt + j * gslot, a.rsg, a.dg, lane & 1, lane >> 1, gv[j]);
                else wg2_store_t(Gt + (size_t)j * 16 * WG2_LDT, lane & 1, lane >> 1, gv[j]);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (!any[j]) continue;      // wave-uniform
            bf16x8_t av;
            if constexpr (TR) av = wg2_frag_tr((const unsigned char *)Gt + j * gslot, lbg, a.rsg, 0);
            else av = __builtin_bit_cast(bf16x8_t, *(const uint4 *)&Gt[(size_t)j * 16 * WG2_LDT + r * WG2_LDT + g * 8]);
#pragma unroll
            for (int i = 0; i < NTV; i++) {
                bf16x8_t bv;
                if constexpr (TR) bv = wg2_frag_tr((const unsigned char *)St, lbs, a.rss, i);
                else bv = __builtin_bit_cast(bf16x8_t, *(const uint4 *)&St[(i * 16 + r) * WG2_LDT + g * 8]);
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[j][i], 0, 0, 0);
            }
        }
        __syncthreads();   // St / tblS are rewritten by the next chunk
    }
    // every tile has one owner: store (gathered operand = dy: P = dW[k]^T, rows = Cout channel, columns = Cin channel)
    const long long wsz = (long long)K * a.Cin * a.Cout;
    float *dst = a.dst + (long long)blockIdx.x * wsz;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k = j == 0 ? k0 : k1;
        if (k >= K) continue;
        const int wk = a.flipk ? (K - 1 - k) : k;
#pragma unroll
        for (int i = 0; i < NTV; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cg = g * 4 + q, cs = i * 16 + r;
                const int ci = a.gx ? cg : cs, co = a.gx ? cs : cg;
                if (ci < a.Cin && co < a.Cout) dst[((long long)wk * a.Cin + ci) * a.Cout + co] = acc[j][i][q];
            }
    }
}

// dW[e] = sum_r part[r][e]: 32 elements x 8 split groups per workgroup; group sums are combined in group order
__global__ __launch_bounds__(256) void wgrad2_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, long long n, int R, int accum) {
    __shared__ float sh[8][32];
    const int el = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long e = (long long)blockIdx.x * 32 + el;
    float v = 0.f;
    if (e < n)
        for (int r = rg; r < R; r += 8) v += part[(long long)r * n + e];
    sh[rg][el] = v;
    __syncthreads();
    if (rg == 0 && e < n) {
        float s = accum ? dW[e] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) s += sh[q][el];
        dW[e] = s;
    }
}

// ------------------------------------------------------------------------------ weight gradient, third generation
// What bounds spconv_wgrad2_kernel at the big levels is instruction issue, not memory: ~110 VALU/SALU instructions per
// (32-row chunk, offset) and lane for one MFMA -- run-time operand types (both conversion paths compiled in), bounds checks
// and exec-mask juggling around every gather, 64-bit address arithmetic -- and every offset group re-reads the stationary
// chunk and the kernel-map rows (profiles/r02_i: 75 MB of HBM traffic per launch against 32 MB algorithmic).  This kernel
// generalises the wide-stationary kernel above to every shape of levels 0-2:
//   * one workgroup of NW waves shares an iteration's rows (S sub-chunks of 32): kernel-map rows and the stationary operand are
//     staged ONCE (row-major, converted to bf16 on the way), double-buffered in LDS, the next iteration's requested from
//     memory before this iteration's gathers (one barrier per iteration);
//   * wave w owns the offsets kbase + w + j*NW (j < OW) with all MT x NT tiles: no cross-wave reduction, every gathered row
//     is fetched once per offset;
//   * gathers are raw buffer loads (an absent neighbour, index -1, is an out-of-range offset: the hardware returns zeros;
//     rows past the end likewise), all OW*S*MT of a wave's iteration in flight together; compile-time shapes, 32-bit offsets:
//     ~10 instructions per gathered unit;
//   * transposing LDS reads (ds_read_b64_tr_b16) deliver both MFMA operands from the row-major images.
// Row splits write partial dW (single owner per tile and split: deterministic) summed by the fixed-order reduction.
struct Wg3Args {
    const void *G; const void *Sm; const int *tbl; float *dst;
    unsigned int gbytes, sbytes, tbytes;   // buffer extents in bytes
    int growb, srowb;                      // row pitch in bytes
    int Ms, Cs8, cpw, flipk, Cin, Cout, K;
    unsigned int invs;                     // ceil(65536 / Cs8)
    int rss, dss, imgs;                    // stationary image (wg2_img)
    const void *tbl16; const int *ok16; unsigned int t16bytes;   // optional 16-bit delta form of tbl (KV = 27; see spconv_fwd2_kernel)
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define WG3_RSRC_FLAGS 0x00020000          // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)

__device__ __forceinline__ uint4 wg3_cvt8(const u32x4_t lo, const u32x4_t hi) {
    return make_uint4(pack2bf2(__uint_as_float(lo.x), __uint_as_float(lo.y)), pack2bf2(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                      pack2bf2(__uint_as_float(hi.x), __uint_as_float(hi.y)), pack2bf2(__uint_as_float(hi.z), __uint_as_float(hi.w)));
}

// GX: the gathered operand is x (bf16), the stationary one dy; else dy is gathered and x (bf16) stationary.  DYBF: dy is stored
// as bf16 (the executor's single-consumer gradient buffers), else fp32 and converted on the way into LDS.
// NW * OW * KG >= KV; with equality (27 = 9 waves x 3 offsets, 8 = 4 x 2 = 8 x 1) no wave carries an idle offset slot.
typedef v4s16_t __attribute__((address_space(3))) *wg3_lds_p;
template <int MT, int NT, int KV, int NW, int OW, int KG, int S, bool GX, bool DYBF, bool T16>
__global__ __launch_bounds__(NW * 64) void spconv_wgrad3_kernel(const Wg3Args a) {
    static_assert(!T16 || KV == 27, "the 16-bit table exists for the 27-offset maps only");
    constexpr int NTH = NW * 64;
    constexpr int TE = S * 32 * KV;                                    // kernel-map entries per iteration
    constexpr int TL = (TE + NTH - 1) / NTH;                           //   ... per thread
    constexpr int SU = (S * 32 * NT * 2 + NTH - 1) / NTH;              // stationary 8-channel units per thread and iteration
    constexpr int RSBG = (MT * 32) % 128 == 0 ? MT * 32 + 32 : MT * 32, DG = RSBG % 64 == 0 ? 32 : 128;
    constexpr int IMGG = (32 * RSBG + 3 * DG + 15) & ~15;
    constexpr int GE = (GX || DYBF) ? 1 : 2, SE = (GX && !DYBF) ? 2 : 1;   // 16-byte loads per gathered / stationary 8-channel unit
    constexpr int CG8 = 2 * MT;
    constexpr bool FULL = NW * OW * KG == KV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // LDS (byte offsets): 2 x stationary images | 2 x kernel-map rows | one gather image per wave
    const int st_bytes = S * a.imgs;
    const int tb_base = 2 * st_bytes;
    const int gs_off = tb_base + 2 * TE * 4 + wave * IMGG;
    const int lbg = gs_off + wg2_lane_base(RSBG, DG, r, g), lbs = wg2_lane_base(a.rss, a.dss, r, g);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)a.G, 0, a.gbytes, WG3_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.Sm, 0, a.sbytes, WG3_RSRC_FLAGS);
    constexpr bool t16 = T16;                             // (a compile-time form: a run-time choice put both table loops into the iteration
                                                          //  and a full vmcnt(0) between the table loads and the gathers)
    const __amdgpu_buffer_rsrc_t rt = t16 ? __builtin_amdgcn_make_buffer_rsrc((void *)a.tbl16, 0, a.t16bytes, WG3_RSRC_FLAGS)
                                          : __builtin_amdgcn_make_buffer_rsrc((void *)a.tbl, 0, a.tbytes, WG3_RSRC_FLAGS);
    const int nit = (a.Ms + 32 * S - 1) / (32 * S);
    const int it_begin = blockIdx.x * a.cpw, it_end = min(nit, it_begin + a.cpw);
    const int k0 = blockIdx.y * (NW * OW) + wave;                      // this wave's offsets: k0 + j * NW

    f32x4 acc[OW][MT][NT];
#pragma unroll
    for (int j = 0; j < OW; j++)
#pragma unroll
        for (int mi = 0; mi < MT; mi++)
#pragma unroll
            for (int ni = 0; ni < NT; ni++) acc[j][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the stationary units this thread stages (the same image slots every iteration)
    int s_img[SU];
    unsigned int s_off[SU];
#pragma unroll
    for (int i = 0; i < SU; i++) {
        const int u = t + i * NTH;
        s_img[i] = -1; s_off[i] = 0xFFFFFFE0u;                          // out of range: the loads return zeros
        if (u < S * 32 * a.Cs8) {
            const int srow = (int)(((unsigned int)u * a.invs) >> 16), c8 = u - srow * a.Cs8, r32 = srow & 31;
            s_img[i] = (srow >> 5) * a.imgs + r32 * a.rss + (r32 >> 3) * a.dss + c8 * 16;
            s_off[i] = (unsigned int)(srow * a.srowb + c8 * (SE == 2 ? 32 : 16));
        }
    }
    // gather lanes: row / unit of this lane's q-th gathered unit; byte offset of its kernel-map entry (offset k0, sub-chunk 0)
    int g_row[MT], g_c8[MT];
#pragma unroll
    for (int q = 0; q < MT; q++) { const int unit = lane + q * 64; g_row[q] = unit / CG8; g_c8[q] = unit - g_row[q] * CG8; }

    int tv[TL];
    int erow[TL];                     // row (inside the iteration's 32 * S rows) of this thread's i-th kernel-map entry
#pragma unroll
    for (int i = 0; i < TL; i++) erow[i] = (t + i * NTH) / KV;
    u32x4_t sv[SU][SE];
    auto prefetch = [&](int it) {
        const unsigned int row0 = (unsigned int)it * (32 * S);
        if constexpr (t16) {
#pragma unroll
            for (int i = 0; i < TL; i++) {
                const int e = t + i * NTH;
                tv[i] = 0;
                if (TL * NTH == TE || e < TE) tv[i] = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(rt, (row0 * KV + e) * 2u, 0, 0);   // (beyond the table: 0; decoded where it is stored)
            }
        } else {
#pragma unroll
        for (int i = 0; i < TL; i++) {
            const int e = t + i * NTH;
            tv[i] = 0;
            if (TL * NTH == TE || e < TE) tv[i] = __builtin_amdgcn_raw_buffer_load_b32(rt, (row0 * KV + e) * 4u, 0, 0);
        }
        }
#pragma unroll
        for (int i = 0; i < SU; i++) {
            const unsigned int off = s_img[i] >= 0 ? row0 * (unsigned int)a.srowb + s_off[i] : 0xFFFFFFE0u;
#pragma unroll
            for (int h = 0; h < SE; h++) sv[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * h, 0, 0);
        }
    };
    if (it_begin < it_end) prefetch(it_begin);
    for (int it = it_begin; it < it_end; it++) {
        const int buf = (it - it_begin) & 1;
        const int st_off = buf * st_bytes, tb_off = tb_base + buf * (TE * 4);
#pragma unroll
        for (int i = 0; i < TL; i++) {
            const int e = t + i * NTH;
            // (the 16-bit delta is decoded HERE, an iteration after its load was issued: decoding in prefetch() made every wave wait for the
            // table's round trip before it could issue its gathers)
            if (TL * NTH == TE || e < TE) *(int *)(smem + tb_off + e * 4) = !t16 ? tv[i] : (tv[i] == -32768 ? -1 : it * (32 * S) + erow[i] + tv[i]);
        }
#pragma unroll
        for (int i = 0; i < SU; i++)
            if (s_img[i] >= 0) {
                uint4 v;
                if constexpr (SE == 2) v = wg3_cvt8(sv[i][0], sv[i][SE - 1]);
                else v = make_uint4(sv[i][0].x, sv[i][0].y, sv[i][0].z, sv[i][0].w);
                *(uint4 *)(smem + st_off + s_img[i]) = v;
            }
        __syncthreads();
        if (it + 1 < it_end) prefetch(it + 1);
        // this wave's gathers: OW offsets x S sub-chunks x MT units per lane, all in flight together
        u32x4_t gv[OW][S][MT][GE];
#pragma unroll
        for (int j = 0; j < OW; j++) {
            const int k = k0 + j * NW;
            if (FULL || k < KV) {   // wave-uniform (scalar)
#pragma unroll
                for (int s = 0; s < S; s++)
#pragma unroll
                    for (int q = 0; q < MT; q++) {
                        const int idx = *(const int *)(smem + tb_off + ((s * 32 + g_row[q]) * KV + k) * 4);
                        const unsigned int off = (unsigned int)idx * (unsigned int)a.growb + g_c8[q] * (GE == 2 ? 32 : 16);
#pragma unroll
                        for (int h = 0; h < GE; h++) gv[j][s][q][h] = __builtin_amdgcn_raw_buffer_load_b128(rg, off + 16u * h, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            bf16x8_t af[OW][MT];
#pragma unroll
            for (int j = 0; j < OW; j++) {
                const int k = k0 + j * NW;
                if (FULL || k < KV) {
#pragma unroll
                    for (int q = 0; q < MT; q++) {
                        uint4 v;
                        if constexpr (GE == 1) v = make_uint4(gv[j][s][q][0].x, gv[j][s][q][0].y, gv[j][s][q][0].z, gv[j][s][q][0].w);
                        else v = wg3_cvt8(gv[j][s][q][0], gv[j][s][q][GE - 1]);
                        *(uint4 *)(smem + gs_off + g_row[q] * RSBG + (g_row[q] >> 3) * DG + g_c8[q] * 16) = v;
                    }
#pragma unroll
                    for (int mi = 0; mi < MT; mi++) {
                        const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_p)(smem + lbg + mi * 32));
                        const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_p)(smem + lbg + mi * 32 + 4 * RSBG));
                        af[j][mi] = __builtin_bit_cast(bf16x8_t, (v8s16_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < NT; ni++) {
                const int bo = st_off + s * a.imgs + lbs + ni * 32;
                const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_p)(smem + bo));
                const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_p)(smem + bo + 4 * a.rss));
                const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, (v8s16_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int j = 0; j < OW; j++) {
                    const int k = k0 + j * NW;
                    if (FULL || k < KV) {
#pragma unroll
                        for (int mi = 0; mi < MT; mi++)
                            acc[j][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][mi], bv, acc[j][mi][ni], 0, 0, 0);
                    }
                }
            }
        }
    }
    const long long wsz = (long long)KV * a.Cin * a.Cout;
    float *dst = a.dst + (long long)blockIdx.x * wsz;
#pragma unroll
    for (int j = 0; j < OW; j++) {
        const int k = k0 + j * NW;
        if (!FULL && k >= KV) continue;
        const int wk = a.flipk ? (KV - 1 - k) : k;
#pragma unroll
        for (int mi = 0; mi < MT; mi++)
#pragma unroll
            for (int ni = 0; ni < NT; ni++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cg = mi * 16 + g * 4 + q, cs = ni * 16 + r;
                    const int ci = GX ? cg : cs, co = GX ? cs : cg;
                    if (ci < a.Cin && co < a.Cout) dst[((long long)wk * a.Cin + ci) * a.Cout + co] = acc[j][mi][ni][q];
                }
    }
}

// shapes the third-generation kernel is instantiated for: (MT, NT, K, gx) -> (NW, OW, S)
struct Wg3Cfg { int mt, nt, k, gx, nw, ow, kg, s; };
#define WG3_CONFIGS(X)                                                        \
    X(1, 1, 27, 1, 9, 3, 1, 8)   /* 16 -> 16, level 0 */                       \
    X(1, 2, 27, 0, 9, 3, 1, 4)   /* 32 -> 16 (first conv behind a concatenation) */ \
    X(1, 2, 8, 1, 4, 2, 1, 4)    /* down 16 -> 32 */                           \
    X(1, 2, 8, 0, 4, 2, 1, 4)    /* up 32 -> 16 */                             \
    X(2, 2, 27, 1, 9, 3, 1, 4)   /* 32 -> 32, level 1 */                       \
    X(2, 2, 27, 1, 9, 1, 3, 4)   /*   ... three offset groups (more workgroups per partial dW) */ \
    X(2, 4, 27, 0, 9, 3, 1, 1)   /* 64 -> 32 */                                \
    X(2, 4, 27, 0, 9, 1, 3, 4)                                                 \
    X(2, 3, 8, 1, 8, 1, 1, 4)    /* down 32 -> 48 */                           \
    X(2, 3, 8, 0, 8, 1, 1, 4)    /* up 48 -> 32 */                             \
    X(3, 3, 27, 1, 9, 3, 1, 1)   /* 48 -> 48, level 2 */                       \
    X(3, 3, 27, 1, 9, 1, 3, 2)                                                 \
    X(3, 6, 27, 0, 9, 1, 3, 2)   /* 96 -> 48 */
/* (S: sub-chunks of 32 rows per iteration, i.e. gathers in flight per wave -- swept per shape in round 3.)  Measured and left to
 * the other kernels (tools/wgrad_bench.py, profiles/r02_k): the stem 136 -> 16 (the 16-wave wide-stationary kernel: 208 us against
 * 273 us here at 649 k rows) and the stride-2 pairs of level 2 and deeper (within noise).  Round 6: two 7-wave workgroups per compute unit for
 * 16 -> 16 (4 offsets per wave, S = 3: 128 VGPRs) ran 68 us against 45 us for the one 9-wave workgroup (gpurun_out/r06_j30): not kept */
#define WG3_ROW(MT, NT, KV, GXV, NW, OW, KG, SV) {MT, NT, KV, GXV, NW, OW, KG, SV},
static const Wg3Cfg wg3_cfgs[] = {WG3_CONFIGS(WG3_ROW)};
#undef WG3_ROW
static bool wg3_enabled() { return d3_tune(D3T_WG3) != 0; }   // D3_WG3=0: A/B measurements
// Row splits of a configuration.  One workgroup per compute unit: measured on MI355X (tools/wgrad_bench.py, 649 k rows, 16 -> 16)
// 256 / 384 / 512 / 1024 workgroups = 57 / 75 / 68 / 90 us -- a multiple of the CU count keeps the CUs evenly loaded, every extra
// split is another partial dW written and read back.  The partials stay below max(16 MB, 25 % of the algorithmic bytes).
static int wg3_ncu() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t p; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    return n;
}
static int wg3_splits(const Wg3Cfg &c, int Ms, int Mg, int K, int Cg, int Cs, int Cin, int Cout, bool gbf, bool sbf, int *cpw, bool *capped) {
    const int nit = (Ms + 32 * c.s - 1) / (32 * c.s);
    const long long wsz = (long long)K * Cin * Cout * 4;
    const double alg = (double)Ms * K * 4 + (double)Mg * Cg * (gbf ? 2 : 4) + (double)Ms * Cs * (sbf ? 2 : 4);
    double cap = 0.25 * alg; if (cap < 16.0 * 1048576) cap = 16.0 * 1048576;
    int target = wg3_ncu();
    int R = target / c.kg; if (R < 1) R = 1;
    const int capR = (int)(cap / (double)wsz);
    *capped = R > capR;
    if (R > capR) R = capR;
    if (R > (nit + 1) / 2) R = (nit + 1) / 2;
    if (R < 2) R = 2;            // (always row-split: the partials go through the reduction; Ms >= 2048 gives nit >= 8)
    *cpw = (nit + R - 1) / R;
    return (nit + *cpw - 1) / *cpw;
}
static const Wg3Cfg *wg3_pick(int Ms, int Mg, int K, int Cg, int Cs, int Cin, int Cout, bool gx, bool gbf, bool sbf) {
    if (!wg3_enabled() || Ms < 2048 || (Cg & 15) || (Cs & 7)) return nullptr;
    if (gx ? !gbf : !sbf) return nullptr;                            // x bf16 only (dy fp32 or bf16)
    // 32-bit buffer offsets: operand extents with up to 2x row pitch (views of concatenated buffers)
    if ((long long)Mg * Cg * 2 * (gbf ? 2 : 4) >= (1ll << 31) || (long long)Ms * Cs * 2 * (sbf ? 2 : 4) >= (1ll << 31) || (long long)Ms * K * 4 >= (1ll << 31)) return nullptr;
    const int mt = Cg / 16, nt = (Cs + 15) / 16;
    const Wg3Cfg *best = nullptr;
    int best_wgs = 0;
    for (const Wg3Cfg &c : wg3_cfgs)
        if (c.mt == mt && c.nt == nt && c.k == K && c.gx == (gx ? 1 : 0)) {
            int cpw; bool capped;
            const int wgs = wg3_splits(c, Ms, Mg, K, Cg, Cs, Cin, Cout, gbf, sbf, &cpw, &capped) * c.kg;
            if (!capped) return &c;              // the first (fewest offset groups) whose splits fill the chip within the budget
            if (wgs > best_wgs) { best = &c; best_wgs = wgs; }
        }
    return best;
}


// ------------------------------------------------------------------------------ weight gradient, reference precision
// D3_CONV_F32: dW[k] = sum_u G[tbl[u,k]]^T (x) Sm[u] with exact fp32 products on v_mfma_f32_16x16x4_f32.  No LDS staging:
// the MFMA operands are read straight from memory -- A[i = lane & 15][kk = lane >> 4] = (x side)[row kk][ci0 + i],
// B[kk][j = lane & 15] = (dy side)[row kk][co0 + j], four rows per step, 64 contiguous bytes per row and tile.
// A workgroup owns (a row range, a group of OW offsets, a BG x BS block of tiles): the stationary operand's rows are read
// ONCE per step and serve all OW offsets (one workgroup per offset re-read them 27 times: 2.3 GB per level-0 launch); its
// 4 waves take interleaved 4-row groups, keep OW x BG x BS accumulator tiles (<= 18), and are summed through LDS in
// wave order; row ranges write partial dW that wgrad2_reduce_kernel adds in range order: deterministic, no atomics.
struct WgfArgs {
    const float *G, *Sm;        // gathered operand (rows tbl[u][k]) and stationary operand (row u)
    const int *tbl;             // (Ms, K) or NULL (identity, K == 1)
    float *dst;                 // partials [R][K][CinW][Cout]
    int ldg, lds, Ms, K, gx;    // gx: the gathered operand is x (else dy: D3_CONV_XSTAT)
    int Cg, Cs;                 // channels of the gathered / stationary operand
    int CinW, Cout, flipk, rows_per;   // rows per range (multiple of 16)
    int ngb, nsb;               // tile blocks of the gathered / stationary side
};
template <int BG, int BS, int OW>
__global__ __launch_bounds__(256) void spconv_wgrad_f32_kernel(const WgfArgs a) {
    __shared__ float redS[4][4][64];     // wave, q, lane: one tile at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kk = lane >> 4;
    const int r = blockIdx.x;
    const int k0 = blockIdx.y * OW;
    const int gb = (int)blockIdx.z / a.nsb, sb = (int)blockIdx.z - gb * a.nsb;
    const int u0 = r * a.rows_per, u1 = min(a.Ms, u0 + a.rows_per);
    f32x4 acc[OW][BG][BS];
#pragma unroll
    for (int o = 0; o < OW; o++)
#pragma unroll
        for (int p = 0; p < BG; p++)
#pragma unroll
            for (int q = 0; q < BS; q++) acc[o][p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cg[BG], cs[BS];
#pragma unroll
    for (int p = 0; p < BG; p++) cg[p] = (gb * BG + p) * 16 + i16;
#pragma unroll
    for (int q = 0; q < BS; q++) cs[q] = (sb * BS + q) * 16 + i16;
    for (int u = u0 + wave * 4; u < u1; u += 16) {
        const int row = u + kk;
        const bool live = row < u1;
        const long long sr = live ? row : 0;
        float sv[BS];
#pragma unroll
        for (int q = 0; q < BS; q++) { sv[q] = a.Sm[sr * a.lds + (cs[q] < a.Cs ? cs[q] : 0)]; if (!live || cs[q] >= a.Cs) sv[q] = 0.f; }
        // every load of the step is issued before the first MFMA (a use right behind a load serialises the round trips):
        // the OW kernel-map entries, then the OW x BG gathered values -- absent neighbours / offsets read row 0 and are zeroed
        int g[OW];
#pragma unroll
        for (int o = 0; o < OW; o++) g[o] = (live && k0 + o < a.K) ? (a.tbl ? a.tbl[(long long)row * a.K + k0 + o] : row) : -1;
        float gv[OW][BG];
#pragma unroll
        for (int o = 0; o < OW; o++) {
            const long long gr = g[o] >= 0 ? g[o] : 0;
#pragma unroll
            for (int p = 0; p < BG; p++) gv[o][p] = a.G[gr * a.ldg + (cg[p] < a.Cg ? cg[p] : 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < OW; o++) {
#pragma unroll
            for (int p = 0; p < BG; p++) {
                const float v = (g[o] < 0 || cg[p] >= a.Cg) ? 0.f : gv[o][p];
#pragma unroll
                for (int q = 0; q < BS; q++)     // D[x-side channel][dy-side channel]: the x operand goes first
                    acc[o][p][q] = a.gx ? __builtin_amdgcn_mfma_f32_16x16x4f32(v, sv[q], acc[o][p][q], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x4f32(sv[q], v, acc[o][p][q], 0, 0, 0);
            }
        }
    }
    // D layout: row (= ci) (lane >> 4) * 4 + e, column (= co) lane & 15
#pragma unroll
    for (int o = 0; o < OW; o++) {
        if (k0 + o >= a.K) break;      // (uniform)
        const int kd = a.flipk ? a.K - 1 - (k0 + o) : k0 + o;
        float *out = a.dst + ((size_t)r * a.K + kd) * a.CinW * a.Cout;
#pragma unroll
        for (int p = 0; p < BG; p++)
#pragma unroll
            for (int q = 0; q < BS; q++) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; e++) redS[wave][e][lane] = acc[o][p][q][e];
                __syncthreads();
                const int t = threadIdx.x, ln = t & 63, e = t >> 6;
                const float v = redS[0][e][ln] + redS[1][e][ln] + redS[2][e][ln] + redS[3][e][ln];
                const int tg = (gb * BG + p) * 16, ts = (sb * BS + q) * 16;
                const int ci = (a.gx ? tg : ts) + (ln >> 4) * 4 + e, co = (a.gx ? ts : tg) + (ln & 15);
                if (ci < a.CinW && co < a.Cout) out[(size_t)ci * a.Cout + co] = v;
            }
    }
}
struct WgfCfg { int bg, bs, ow; };
// block of tiles per workgroup and offsets per group: OW * BG * BS <= 18 accumulator tiles
static WgfCfg wgf_cfg(int Cg, int Cs, int K) {
    const int tg = (Cg + 15) / 16, ts = (Cs + 15) / 16;
    WgfCfg c;
    c.bg = tg >= 3 ? 3 : tg; c.bs = ts >= 3 ? 3 : ts;
    // (register budget: accumulators + the step's gathered values must leave room for >= 2 waves per SIMD -- the kernel is a
    // chain of memory round trips, one wave per SIMD left it at 400 us for a level-0 16 -> 16 layer)
    static const int ow_of[4][4] = {{0, 0, 0, 0}, {0, 14, 9, 5}, {0, 9, 4, 3}, {0, 5, 3, 2}};
    c.ow = ow_of[c.bg][c.bs];
    if (K <= 8 && c.ow > 8) c.ow = 8;
    if (K == 1) c.ow = 1;
    return c;
}
static int launch_wgf(const WgfArgs &a, const WgfCfg &c, int R, hipStream_t s) {
    const dim3 grid(R, (a.K + c.ow - 1) / c.ow, a.ngb * a.nsb);
#define WGF_CASE(BGV, BSV, OWV) if (c.bg == BGV && c.bs == BSV && c.ow == OWV) { spconv_wgrad_f32_kernel<BGV, BSV, OWV><<<grid, 256, 0, s>>>(a); D3_LAUNCH_CHECK(); return 0; }
    WGF_CASE(1, 1, 14) WGF_CASE(1, 1, 8) WGF_CASE(1, 2, 9) WGF_CASE(1, 2, 8) WGF_CASE(2, 1, 9) WGF_CASE(2, 1, 8) WGF_CASE(1, 3, 5)
    WGF_CASE(3, 1, 5) WGF_CASE(2, 2, 4) WGF_CASE(2, 3, 3) WGF_CASE(3, 2, 3) WGF_CASE(3, 3, 2)
    WGF_CASE(1, 1, 1) WGF_CASE(1, 2, 1) WGF_CASE(2, 1, 1) WGF_CASE(1, 3, 1) WGF_CASE(3, 1, 1) WGF_CASE(2, 2, 1) WGF_CASE(2, 3, 1) WGF_CASE(3, 2, 1) WGF_CASE(3, 3, 1)
#undef WGF_CASE
    return D3_ERR_ARG;
}

struct Wg2Plan { int tpo, nu, opw, kg, passes, R, cpw, wide, tr; int rsg, dg, imgg, rss, dss, imgs; size_t lds, ws_bytes; const Wg3Cfg *w3; };

// D3_WG2_TR=0 selects the first staging scheme (transposed ds_write_b16 images) for A/B measurements
static bool wg2_use_tr() { return d3_tune(D3T_WG2_TR) != 0; }

static Wg2Plan wg2_plan(int Ms, int Mg, int K, int Cg, int Cs, int Cin, int Cout, bool gx, bool gbf, bool sbf) {
    Wg2Plan p;
    const int mt = (Cg + 15) / 16, nt = (Cs + 15) / 16, ntl = mt * nt;
    p.wide = 0;
    p.tr = wg2_use_tr() ? 1 : 0;
    wg2_img(Cg / 8, &p.rsg, &p.dg, &p.imgg);
    wg2_img(Cs / 8, &p.rss, &p.dss, &p.imgs);
    p.imgg = (p.imgg + 15) & ~15; p.imgs = (p.imgs + 15) & ~15;
    p.w3 = wg3_pick(Ms, Mg, K, Cg, Cs, Cin, Cout, gx, gbf, sbf);
    if (p.w3) {
        const Wg3Cfg &c = *p.w3;
        p.kg = c.kg;
        bool capped;
        p.R = wg3_splits(c, Ms, Mg, K, Cg, Cs, Cin, Cout, gbf, sbf, &p.cpw, &capped);
        const long long wsz = (long long)K * Cin * Cout * 4;
        p.lds = (size_t)2 * c.s * p.imgs + (size_t)2 * c.s * 32 * K * 4 + (size_t)c.nw * (((32 * ((c.mt * 32) % 128 == 0 ? c.mt * 32 + 32 : c.mt * 32) + 3 * 128) + 15) & ~15);
        p.ws_bytes = (size_t)p.R * wsz;
        p.tpo = 0; p.nu = 0; p.opw = c.ow; p.passes = 1;
        return p;
    }
    if (mt == 1 && nt > 4 && nt <= WGW_MAXNT && K <= 2 * WGW_WAVES && 32 * (Cs / 8) <= WGW_WAVES * 64 && Ms >= 4096) {
        // one 16-wave workgroup per row split holds all K x nt tiles (spconv_wgrad2_wide_kernel)
        p.wide = 1; p.tpo = nt; p.nu = 1; p.opw = 2; p.kg = 1; p.passes = 1;
        const int nchunks = (Ms + 31) / 32;
        int R = 256; if (R > (nchunks + 3) / 4) R = (nchunks + 3) / 4; if (R < 1) R = 1;
        p.cpw = (nchunks + R - 1) / R;
        p.R = (nchunks + p.cpw - 1) / p.cpw;
        p.lds = p.tr ? (size_t)p.imgs + (size_t)32 * C2_MAXK * 4 + (size_t)WGW_WAVES * 2 * p.imgg
                     : (size_t)nt * 16 * WG2_LDT * 2 + (size_t)32 * C2_MAXK * 4 + (size_t)WGW_WAVES * 2 * 16 * WG2_LDT * 2;
        p.ws_bytes = (size_t)p.R * K * Cin * Cout * 4;
        return p;
    }
    p.tpo = ntl <= 1 ? 1 : ntl <= 2 ? 2 : ntl <= 4 ? 4 : ntl <= 8 ? 8 : 16;
    if (mt == 1 && nt > 4) p.tpo = 4;   // column passes of 4 tiles, 4 offsets per wave (the stem: 16 x 136 channels)
    p.nu = (Cg / 8 * 32 + 63) / 64;    // 16-byte units per lane per offset
    p.opw = WG2_T / p.tpo;
    p.kg = (K + p.opw - 1) / p.opw;
    p.passes = (ntl + p.tpo - 1) / p.tpo;
    const int nchunks = (Ms + 31) / 32;
    // row splits: ~2048 waves in flight, at least 2 chunks per wave, partial buffer <= 8 MB
    const long long wsz = (long long)K * Cin * Cout * 4;
    int R = WG2_TARGET_WGS / (p.kg * p.passes); if (R < 1) R = 1;
    const int maxR_rows = (nchunks + 7) / 8; if (R > maxR_rows) R = maxR_rows;
    const long long maxR_mem = ((long long)WG2_PART_MB << 20) / wsz; if (R > maxR_mem) R = (int)maxR_mem;
    if (R < 1) R = 1;
    p.cpw = (nchunks + R - 1) / R;
    p.cpw = (p.cpw + 3) / 4 * 4;
    p.R = (nchunks + p.cpw - 1) / p.cpw;
    const size_t wave_bytes = (p.tr ? (size_t)p.imgg + p.imgs : (size_t)(mt + nt) * 16 * WG2_LDT * 2) + (size_t)32 * C2_MAXK * 4;
    p.lds = 4 * wave_bytes; if (p.lds < 32 * 1024) p.lds = 32 * 1024;
    p.ws_bytes = p.R > 1 ? (size_t)p.R * wsz : 0;
    return p;
}

static Wg2Plan wg2_plan_flags(int Min, int Mout, int K, int Cin, int Cout, int flags) {
    const bool xstat = (flags & D3_CONV_XSTAT) != 0, xbf = (flags & D3_CONV_XBF16) != 0, dybf = (flags & D3_CONV_DYBF16) != 0;
    if (flags & D3_CONV_F32) {       // spconv_wgrad_f32_kernel: (row ranges) x (offset groups) x (tile blocks), always through the partials
        Wg2Plan p;
        memset(&p, 0, sizeof(p));
        const int Ms = xstat ? Min : Mout;
        const WgfCfg cf = wgf_cfg(xstat ? Cout : Cin, xstat ? Cin : Cout, K);
        const int per_range = ((K + cf.ow - 1) / cf.ow) * ((((xstat ? Cout : Cin) + 15) / 16 + cf.bg - 1) / cf.bg) * ((((xstat ? Cin : Cout) + 15) / 16 + cf.bs - 1) / cf.bs);
        int R = (1536 + per_range - 1) / per_range;
        const int maxR = (Ms + 255) / 256; if (R > maxR) R = maxR;
        const long long wsz = (long long)K * Cin * Cout * 4;
        const long long maxR_mem = (64ll << 20) / wsz; if (R > maxR_mem) R = (int)maxR_mem;
        if (R < 1) R = 1;
        p.cpw = ((Ms + R - 1) / R + 15) / 16 * 16;          // rows per range
        if (p.cpw < 16) p.cpw = 16;
        p.R = (Ms + p.cpw - 1) / p.cpw; if (p.R < 1) p.R = 1;
        p.ws_bytes = (size_t)p.R * wsz;
        return p;
    }
    return xstat ? wg2_plan(Min, Mout, K, Cout, Cin, Cin, Cout, false, dybf, xbf) : wg2_plan(Mout, Min, K, Cin, Cout, Cin, Cout, true, xbf, dybf);
}

// flags: the D3_CONV_XSTAT / D3_CONV_XBF16 / D3_CONV_DYBF16 bits of the d3_spconv_wgrad2 call (the kernel choice depends on them)
extern "C" size_t d3_spconv_wgrad2_ws_bytes(int Min, int Mout, int K, int Cin, int Cout, int flags) {
    return wg2_plan_flags(Min, Mout, K, Cin, Cout, flags).ws_bytes;
}

// number of row splits d3_spconv_wgrad2 uses for this shape (its partials: splits x K*CinW*Cout floats in ws)
extern "C" int d3_spconv_wgrad2_splits(int Min, int Mout, int K, int Cin, int Cout, int flags) {
    return wg2_plan_flags(Min, Mout, K, Cin, Cout, flags).R;
}

template <int MT, int NT, int KV, int NW, int OW, int KG, int S, bool GX, bool DYBF, bool T16>
static int launch_wg3_i(const Wg3Args &a, const Wg2Plan &p, hipStream_t s) {
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev))
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_wgrad3_kernel<MT, NT, KV, NW, OW, KG, S, GX, DYBF, T16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    spconv_wgrad3_kernel<MT, NT, KV, NW, OW, KG, S, GX, DYBF, T16><<<dim3(p.R, KG), NW * 64, p.lds, s>>>(a);
    return 0;
}
template <int MT, int NT, int KV, int NW, int OW, int KG, int S, bool GX>
static int launch_wg3(const Wg3Args &a, const Wg2Plan &p, bool dybf, hipStream_t s) {
    static_assert(NW * OW * KG >= KV, "offsets not covered");
    int rc;
    if constexpr (KV == 27) {
        if (a.tbl16) rc = dybf ? launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, true, true>(a, p, s) : launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, false, true>(a, p, s);
        else rc = dybf ? launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, true, false>(a, p, s) : launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, false, false>(a, p, s);
    } else {
        rc = dybf ? launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, true, false>(a, p, s) : launch_wg3_i<MT, NT, KV, NW, OW, KG, S, GX, false, false>(a, p, s);
    }
    if (rc) return rc;
    D3_LAUNCH_CHECK();
    return 0;
}

template <int TPO, int NU>
static int launch_wg2(const Wg2Args &a, const Wg2Plan &p, hipStream_t s) {
    static bool attr_done_dev[64] = {false};
    if (c2_attr_needed(attr_done_dev)) {
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_wgrad2_kernel<TPO, NU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_wgrad2_kernel<TPO, NU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    }
    if (p.tr) spconv_wgrad2_kernel<TPO, NU, true><<<dim3(p.R, p.kg, p.passes), 256, p.lds, s>>>(a);
    else spconv_wgrad2_kernel<TPO, NU, false><<<dim3(p.R, p.kg, p.passes), 256, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}

// x (Min, ldx) and dy (Mout, ldy), each fp32 or bf16 (D3_CONV_XBF16 / D3_CONV_DYBF16); tbl as for d3_spconv_wgrad
// (the forward map, or with D3_CONV_XSTAT the transposed map); dW (K,CinW,Cout) fp32 (CinW <= Cin: x may carry
// zero-padded channels), written (or accumulated into
// with D3_CONV_ACCUM).  ws >= d3_spconv_wgrad2_ws_bytes().  Cin % 8 == 0 and Cout % 8 == 0, else D3_ERR_ARG.
extern "C" int d3_spconv_wgrad2(const void *x, int ldx, const int *tbl, const void *dy, int ldy, float *dW, int Min,
                                int Mout, int K, int Cin, int Cout, int CinW, int flags, void *ws, size_t ws_bytes,
                                void *stream) {
    D3_CLEAR();
    const void *tbl16 = g_next_tbl16; const int *ok16 = nullptr;      // (the hint of d3_spconv_next_tbl16 belongs to this call)
    g_next_tbl16 = nullptr; g_next_tblq = nullptr;
    if (K < 1 || K > C2_MAXK || Cin < 8 || Cout < 8 || (Cin & 7) || (Cout & 7) || Cin > 224 || Cout > 224) return D3_ERR_ARG;
    if (tbl == nullptr && K != 1) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const int xstat = (flags & D3_CONV_XSTAT) ? 1 : 0, accum = (flags & D3_CONV_ACCUM) ? 1 : 0;
    const int xbf = (flags & D3_CONV_XBF16) ? 1 : 0, dybf = (flags & D3_CONV_DYBF16) ? 1 : 0;
    if ((xbf ? (ldx & 7) : (ldx & 3)) || (dybf ? (ldy & 7) : (ldy & 3))) return D3_ERR_ARG;
    if (CinW < 1 || CinW > Cin) return D3_ERR_ARG;
    const long long wn = (long long)K * CinW * Cout;   // dW is (K, CinW, Cout): x may carry zero-padded channels
    const int Ms = xstat ? Min : Mout;
    if (Ms <= 0) { if (!accum) D3_CHECK(hipMemsetAsync(dW, 0, wn * 4, s)); return 0; }
    if (flags & D3_CONV_F32) {
        if (xbf || dybf) return D3_ERR_ARG;
        const Wg2Plan p = wg2_plan_flags(Min, Mout, K, Cin, Cout, flags);
        if ((size_t)p.R * wn * 4 > ws_bytes) return D3_ERR_WORKSPACE;
        WgfArgs f;
        if (xstat) { f.Sm = (const float *)x; f.lds = ldx; f.G = (const float *)dy; f.ldg = ldy; f.gx = 0; f.Cs = Cin; f.Cg = Cout; }
        else { f.Sm = (const float *)dy; f.lds = ldy; f.G = (const float *)x; f.ldg = ldx; f.gx = 1; f.Cs = Cout; f.Cg = Cin; }
        f.tbl = tbl; f.dst = (float *)ws; f.Ms = Ms; f.K = K; f.CinW = CinW; f.Cout = Cout;
        f.flipk = (flags & D3_CONV_FLIPK) ? 1 : 0; f.rows_per = p.cpw;
        const WgfCfg cf = wgf_cfg(f.Cg, f.Cs, K);
        f.ngb = ((f.Cg + 15) / 16 + cf.bg - 1) / cf.bg; f.nsb = ((f.Cs + 15) / 16 + cf.bs - 1) / cf.bs;
        const double bytes32 = 4.0 * (double)Min * Cin + 4.0 * (double)Mout * Cout + 4.0 * (double)wn + (tbl ? 4.0 * (double)Ms * K : 0.0);
        void *pr32 = d3_prof_begin(1, bytes32, 0.0, s);
        { const int dims[6] = {Min, Mout, K, Cin, Cout, 32}; for (int i = 0; i < 6; i++) d3_prof_tag(pr32, i, dims[i]); }
        { const int lrc = launch_wgf(f, cf, p.R, s); if (lrc) return lrc; }
        if (!(flags & D3_CONV_NOREDUCE)) {
            wgrad2_reduce_kernel<<<(int)((wn + 31) / 32), 256, 0, s>>>((const float *)ws, dW, wn, p.R, accum);
            D3_LAUNCH_CHECK();
        }
        d3_prof_end(pr32, s);
        return 0;
    }
    Wg2Args a;
    if (xstat) { a.Sm = x; a.lds = ldx; a.sbf16 = xbf; a.G = dy; a.ldg = ldy; a.gbf16 = dybf; a.gx = 0; }
    else { a.Sm = dy; a.lds = ldy; a.sbf16 = dybf; a.G = x; a.ldg = ldx; a.gbf16 = xbf; a.gx = 1; }
    const int Cg = xstat ? Cout : Cin, Cs = xstat ? Cin : Cout;
    const Wg2Plan p = wg2_plan_flags(Min, Mout, K, Cin, Cout, flags);
    if (p.ws_bytes > ws_bytes) return D3_ERR_WORKSPACE;
    const bool direct = (p.R == 1 && !accum) && !p.wide && !p.w3;
    const bool noreduce = (flags & D3_CONV_NOREDUCE) != 0;   // the caller sums the partials (batched over its layers)
    a.tbl = tbl; a.dst = direct ? dW : (float *)ws;
    if (!direct && p.R == 1 && ws_bytes < (size_t)wn * 4) return D3_ERR_WORKSPACE;
    a.Ms = Ms; a.K = K; a.mt = (Cg + 15) / 16; a.nt = (Cs + 15) / 16; a.Cg8 = Cg / 8; a.Cs8 = Cs / 8;
    a.invg = (65536u + a.Cg8 - 1) / a.Cg8; a.invs = (65536u + a.Cs8 - 1) / a.Cs8;
    a.cpw = p.cpw; a.flipk = (flags & D3_CONV_FLIPK) ? 1 : 0; a.Cin = CinW; a.Cout = Cout;
    a.rsg = p.rsg; a.dg = p.dg; a.imgg = p.imgg; a.rss = p.rss; a.dss = p.dss; a.imgs = p.imgs;
    const double bytes = (xbf ? 2.0 : 4.0) * (double)Min * Cin + (dybf ? 2.0 : 4.0) * (double)Mout * Cout + 4.0 * (double)wn +
                         (tbl ? 4.0 * (double)Ms * K : 0.0);
    void *pr = d3_prof_begin(1, bytes, 0.0, s);
    { const int dims[6] = {Min, Mout, K, Cin, Cout, p.w3 ? 3 : (p.wide ? 1 : 2)}; for (int i = 0; i < 6; i++) d3_prof_tag(pr, i, dims[i]); }
    int rc = D3_ERR_ARG;
    if (p.w3) {
        const Wg3Cfg &c = *p.w3;
        const int Mg = xstat ? Mout : Min;
        const long long gb = ((long long)(Mg - 1) * a.ldg + Cg) * (a.gbf16 ? 2 : 4), sb = ((long long)(Ms - 1) * a.lds + Cs) * (a.sbf16 ? 2 : 4);
        if (gb >= (1ll << 31) || sb >= (1ll << 31) || Mg < 1) return D3_ERR_ARG;
        Wg3Args b;
        b.G = a.G; b.Sm = a.Sm; b.tbl = tbl; b.dst = (float *)ws;
        b.gbytes = (unsigned int)gb; b.sbytes = (unsigned int)sb; b.tbytes = (unsigned int)((long long)Ms * K * 4);
        b.tbl16 = (tbl16 && K == 27) ? tbl16 : nullptr; b.ok16 = ok16; b.t16bytes = (unsigned int)((long long)Ms * K * 2);
        if (b.tbl16) g_t16_launches++;
        b.growb = a.ldg * (a.gbf16 ? 2 : 4); b.srowb = a.lds * (a.sbf16 ? 2 : 4);
        b.Ms = Ms; b.Cs8 = Cs / 8; b.cpw = p.cpw; b.flipk = a.flipk; b.Cin = CinW; b.Cout = Cout; b.K = K;
        b.invs = a.invs; b.rss = p.rss; b.dss = p.dss; b.imgs = p.imgs;
#define WG3_CASE(MT, NT, KV, GXV, NW, OW, KG, SV)                                                                \
        if (c.mt == MT && c.nt == NT && c.k == KV && c.gx == GXV && c.kg == KG && c.ow == OW && c.s == SV) rc = launch_wg3<MT, NT, KV, NW, OW, KG, SV, (GXV != 0)>(b, p, dybf != 0, s);
        WG3_CONFIGS(WG3_CASE)
#undef WG3_CASE
        if (rc == 0 && !noreduce) {
            wgrad2_reduce_kernel<<<(int)((wn + 31) / 32), 256, 0, s>>>((const float *)ws, dW, wn, p.R, accum);
            D3_LAUNCH_CHECK();
        }
        d3_prof_end(pr, s);
        return rc;
    }
    if (p.wide) {
        if (ws_bytes < p.ws_bytes) return D3_ERR_WORKSPACE;
        a.dst = (float *)ws;
        static bool wide_attr[64] = {false};
        const bool set = c2_attr_needed(wide_attr);
#define WGW_CASE(NTV)                                                                                                              \
        case NTV:                                                                                                                      \
            if (set) {                                                                                                                 \
                D3_CHECK(hipFuncSetAttribute((const void *)spconv_wgrad2_wide_kernel<NTV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
                D3_CHECK(hipFuncSetAttribute((const void *)spconv_wgrad2_wide_kernel<NTV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
            }                                                                                                                          \
            if (p.tr) spconv_wgrad2_wide_kernel<NTV, true><<<p.R, WGW_WAVES * 64, p.lds, s>>>(a);                                      \
            else spconv_wgrad2_wide_kernel<NTV, false><<<p.R, WGW_WAVES * 64, p.lds, s>>>(a);                                          \
            break;
        switch (a.nt) { WGW_CASE(5) WGW_CASE(6) WGW_CASE(7) WGW_CASE(8) WGW_CASE(9) default: return D3_ERR_ARG; }
#undef WGW_CASE
        D3_LAUNCH_CHECK();
        if (!noreduce) {
            wgrad2_reduce_kernel<<<(int)((wn + 31) / 32), 256, 0, s>>>((const float *)ws, dW, wn, p.R, accum);
            D3_LAUNCH_CHECK();
        }
        d3_prof_end(pr, s);
        return 0;
    }
#define WG2_NU(TPOV)                                                                          \
    (p.nu <= 1 ? launch_wg2<TPOV, 1>(a, p, s) : p.nu <= 2 ? launch_wg2<TPOV, 2>(a, p, s)          \
     : p.nu <= 4 ? launch_wg2<TPOV, 4>(a, p, s) : p.nu <= 7 ? launch_wg2<TPOV, 7>(a, p, s)        \
                                                            : launch_wg2<TPOV, 14>(a, p, s))
    switch (p.tpo) {
        case 1: rc = WG2_NU(1); break;
        case 2: rc = WG2_NU(2); break;
        case 4: rc = WG2_NU(4); break;
        case 8: rc = WG2_NU(8); break;
        default: rc = WG2_NU(16); break;
    }
#undef WG2_NU
    if (rc == 0 && !direct && !noreduce) {
        wgrad2_reduce_kernel<<<(int)((wn + 31) / 32), 256, 0, s>>>((const float *)ws, dW, wn, p.R, accum);
        D3_LAUNCH_CHECK();
    }
    d3_prof_end(pr, s);
    return rc;
}
