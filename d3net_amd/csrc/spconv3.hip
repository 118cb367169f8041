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
// tq[(tile * 64 + g * 16 + r) * 8 + q] = nbr[tile * 16 + r][4 q + g] - tile * 16 + 32768   (uint16; 0xFFFF: absent / pad)
// *okq is cleared when an entry does not fit [1, 65534].
__global__ void c3_packq_kernel(const int *__restrict__ nbr, int M, unsigned short *__restrict__ tq, int *okq) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (tile, lane, slot)
    const long long total = (long long)((M + 15) / 16) * 64 * 8;
    bool bad = false;
    if (e < total) {
        const int q = (int)(e & 7), lane = (int)((e >> 3) & 63), tile = (int)(e >> 9);
        const int r = lane & 15, g = lane >> 4, k = 4 * q + g, u = tile * 16 + r;
        unsigned int v = 0xFFFFu;
        if (q < 7 && k < 27 && u < M) {
            const int nb = nbr[(long long)u * 27 + k];
            if (nb >= 0) {
                const int d = nb - tile * 16 + 32768;
                if (d < 1 || d > 65534) bad = true; else v = (unsigned int)d;
            }
        }
        tq[e] = (unsigned short)v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *okq = 0;
}
__global__ void c3_set1_kernel(int *p) { *p = 1; }

extern "C" size_t d3_kmap_k3_q16_bytes(int M) { return (size_t)((M + 15) / 16) * 1024; }

extern "C" int d3_kmap_k3_packq(const int *nbr, int M, void *tq, int *okq, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (!nbr || !tq || !okq) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const long long total = (long long)((M + 15) / 16) * 512;
    c3_set1_kernel<<<1, 1, 0, s>>>(okq);
    c3_packq_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(nbr, M, (unsigned short *)tq, okq);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ forward / data gradient
struct Conv3Args {
    const void *x;              // (Min, ldx) bf16
    const uint4 *tq;            // lane table: [ntiles][64] x 16 bytes
    const uint4 *Wp;            // packed bf16 fragments (d3_spconv_pack order), 27 offsets
    void *out;                  // (Mout, ldo) fp32 or bf16
    const float *res;           // EPI 1: residual (Mout, ldr) fp32
    float *part;                // optional BatchNorm partials [grid][2][NT*16]
    double *part2;              // optional second-level table [C3_P2_ROWS][2][NT*16] (zeroed by the caller)
    int ldx, ldo, ldr, Mout, ntiles;
    unsigned long long xbytes;  // extent of x in bytes: ((Min - 1) * ldx + Cin) * 2
    unsigned long long plane_bytes;   // PLANAR probe: bytes of one channel-group plane (Min * 16)
    // EPI 2 (BatchNorm-backward epilogue): the stored value is g = acc * relu'(bn(bnx)), partials (sum g, sum g * xhat)
    const void *bnx; const float *bn_mean, *bn_var, *bn_gamma, *bn_beta;
    int ldbx, bn_relu; float bn_eps;
};

enum { C3_EPI_PLAIN = 0, C3_EPI_RES = 1, C3_EPI_BNBWD = 2 };

template <int ST> struct C3Chunk { static constexpr int QC = ST <= 2 ? 7 : (8 / ST > 0 ? 8 / ST : 1); };

// ST = Cin / 8, NT = Cout / 16, EPI as above, OBF: bf16 output, BXBF: bnx is bf16, NW waves per workgroup
template <int ST, int NT, int EPI, bool OBF, bool BXBF, int NW, bool PLANAR = false, bool MASKED = false>
__global__ __launch_bounds__(NW * 64) void spconv_fwd3_kernel(const Conv3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WREAL = 27 * ST * NT * 16, WELEMS = 28 * ST * NT * 16;     // 16-byte fragments (offset 27: zeros)
    constexpr int QC = C3Chunk<ST>::QC, NCH = (7 + QC - 1) / QC;
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
    const unsigned int rowb = PLANAR ? 16u : (unsigned int)a.ldx * 2u;
    const int wbase = g * ST * NT * 16 + r;            // fragment of (offset g, c8 0, n 0) for this lane
    const uint4 absent = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    int tile = tg0 * NW + wave;
    uint4 tw = absent;
    if (tg0 < tg1 && tile < a.ntiles) tw = a.tq[(size_t)tile * 64 + lane];
    for (int tg = tg0; tg < tg1; tg += tstride) {
        tile = tg * NW + wave;
        if (tile >= a.ntiles) break;                   // (wave-uniform; later groups only have larger tiles)
        const int tile_u = __builtin_amdgcn_readfirstlane(tile);
        const int row0 = tile_u * 16, urow = row0 + r;
        // the tile's window of x: rows [row0 - 32768, row0 + 32767)
        const long long base_b = ((long long)row0 - 32768) * (long long)rowb;
        const long long avail = (long long)a.xbytes - base_b;
        const long long win = 65535ll * (long long)rowb;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.x + base_b), 0, (int)(avail < win ? avail : win), C3_RSRC_FLAGS);
        __amdgpu_buffer_rsrc_t rxp[ST];      // PLANAR (probe): x is [ST][Min][8] bf16 -- one descriptor per channel-group plane
        if (PLANAR) {
            const long long pav = (long long)a.plane_bytes - base_b;
#pragma unroll
            for (int c8 = 0; c8 < ST; c8++)
                rxp[c8] = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.x + (long long)c8 * a.plane_bytes + base_b), 0, (int)(pav < win ? pav : win), C3_RSRC_FLAGS);
        }
        const unsigned int words[4] = {tw.x, tw.y, tw.z, tw.w};
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
        c3_u32x4 buf[2][QC * ST];
        auto issue = [&](const int c, c3_u32x4 (&dst)[QC * ST]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < QC; j++) {
                const int q = c * QC + j;
                if (q >= 7) continue;
                const unsigned int e = (q & 1) ? (words[q >> 1] >> 16) : (words[q >> 1] & 0xFFFFu);
                const unsigned int off = __umul24(e, rowb);
#pragma unroll
                for (int c8 = 0; c8 < ST; c8++) {
                    if (MASKED) {      // probe: absent lanes leave the instruction's EXEC mask instead of addressing beyond the extent
                        c3_u32x4 v = {0u, 0u, 0u, 0u};
                        if (e != 0xFFFFu) v = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned int)c8 * 16u, 0, 0);
                        dst[j * ST + c8] = v;
                        continue;
                    }
                    if (PLANAR) dst[j * ST + c8] = __builtin_amdgcn_raw_buffer_load_b128(rxp[c8], off, 0, 0);
                    else dst[j * ST + c8] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned int)c8 * 16u, 0, 0);
                }
            }
        };
        auto products = [&](const int c, const c3_u32x4 (&src)[QC * ST]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < QC; j++) {
                const int q = c * QC + j;
                if (q >= 7) continue;
#pragma unroll
                for (int c8 = 0; c8 < ST; c8++) {
                    const c3_bf16x8 B = __builtin_bit_cast(c3_bf16x8, src[j * ST + c8]);
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        const uint4 w = wS[wbase + ((q * 4 * ST + c8) * NT + n) * 16];
                        // transposed product: D[m = channel][n = row] += W^T[channel][k] * X^T[k][row]
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c3_bf16x8, w), B, acc[n], 0, 0, 0);
                    }
                }
            }
        };
        issue(0, buf[0]);
        if (NCH == 1) {      // the next tile's table row rides behind the gathers
            const int nt = tile + tstride * NW;
            tw = absent;
            if (tg + tstride < tg1 && nt < a.ntiles) tw = a.tq[(size_t)nt * 64 + lane];
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (c + 1 < NCH) issue(c + 1, buf[(c + 1) & 1]);
            if (NCH > 1 && c + 1 == NCH - 1) {
                const int nt = tile + tstride * NW;
                tw = absent;
                if (tg + tstride < tg1 && nt < a.ntiles) tw = a.tq[(size_t)nt * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            products(c, buf[c & 1]);
        }
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
struct Conv3Plan { int ok, nw, grid; size_t lds; };

static size_t c3_lds_bytes(int ST, int NT, int nw) { return (size_t)28 * ST * NT * 256 + (size_t)nw * 2 * NT * 16 * 4 + (size_t)NT * 16 * 16; }

// shapes with an instance below
static bool c3_shape(int Cin, int Cout) {
    const int ST = Cin / 8, NT = Cout / 16;
    if ((Cin & 7) || (Cout & 15)) return false;
    return (ST == 2 && (NT == 1 || NT == 2)) || (ST == 4 && (NT == 1 || NT == 2 || NT == 4)) || (ST == 6 && NT == 3) || (ST == 8 && NT == 2);
}
static Conv3Plan conv3_plan(int Mout, int Cin, int Cout) {
    Conv3Plan p{0, 4, 1, 0};
    if (Mout <= 0 || !c3_shape(Cin, Cout)) return p;
    const int ST = Cin / 8, NT = Cout / 16, ntiles = (Mout + 15) / 16;
    const size_t wbytes = (size_t)28 * ST * NT * 256;
    // waves per workgroup: one LDS copy of the weights per workgroup, 16 waves per CU
    const int nw = wbytes <= 20 * 1024 ? 4 : wbytes <= 72 * 1024 ? 8 : 16;
    const int wg_per_cu = 16 / nw;
    p.nw = nw; p.lds = c3_lds_bytes(ST, NT, nw);
    if (p.lds > 160 * 1024) return p;
    const int ntg = (ntiles + nw - 1) / nw;
    int cap = c3_ncu() * wg_per_cu;
    const int per = (ntg + cap - 1) / cap;
    int grid = (ntg + per - 1) / per;
    if (grid >= 8) grid = (grid + 7) / 8 * 8;          // (XCD-interleaved ranges want a multiple of 8; surplus workgroups find no tile)
    p.grid = grid; p.ok = 1;
    return p;
}

extern "C" int d3_spconv_fwd3_nparts(int Mout, int Cin, int Cout) {
    const Conv3Plan p = conv3_plan(Mout, Cin, Cout);
    return p.ok ? p.grid : 0;
}

static std::atomic<long long> g_c3_launches{0};
static int g_c3_planar_probe = 0;
extern "C" void d3x_c3_planar_probe(int on) { g_c3_planar_probe = on; }
extern "C" long long d3_spconv_fwd3_launches(void) { return g_c3_launches.load(); }

static int c3_launch_masked_probe(const Conv3Args &a, const Conv3Plan &p, hipStream_t s) {
    D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd3_kernel<2, 1, C3_EPI_PLAIN, false, false, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    spconv_fwd3_kernel<2, 1, C3_EPI_PLAIN, false, false, 4, false, true><<<p.grid, 256, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}
static int c3_launch_planar_probe(const Conv3Args &a, const Conv3Plan &p, hipStream_t s) {
    D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd3_kernel<2, 1, C3_EPI_PLAIN, false, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    spconv_fwd3_kernel<2, 1, C3_EPI_PLAIN, false, false, 4, true><<<p.grid, 256, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}
template <int ST, int NT, int EPI, bool OBF, bool BXBF, int NW>
static int c3_launch_inst(const Conv3Args &a, const Conv3Plan &p, hipStream_t s) {
    static bool attr_done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !attr_done[dev]) {
        D3_CHECK(hipFuncSetAttribute((const void *)spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW><<<p.grid, NW * 64, p.lds, s>>>(a);
    D3_LAUNCH_CHECK();
    g_c3_launches++;
    return 0;
}
template <int ST, int NT, int NW>
static int c3_launch_shape(const Conv3Args &a, const Conv3Plan &p, int epi, bool obf, bool bxbf, hipStream_t s) {
    if (epi == C3_EPI_PLAIN) return obf ? c3_launch_inst<ST, NT, C3_EPI_PLAIN, true, false, NW>(a, p, s) : c3_launch_inst<ST, NT, C3_EPI_PLAIN, false, false, NW>(a, p, s);
    if (epi == C3_EPI_RES) return obf ? D3_ERR_ARG : c3_launch_inst<ST, NT, C3_EPI_RES, false, false, NW>(a, p, s);
    if (obf) return bxbf ? c3_launch_inst<ST, NT, C3_EPI_BNBWD, true, true, NW>(a, p, s) : c3_launch_inst<ST, NT, C3_EPI_BNBWD, true, false, NW>(a, p, s);
    return bxbf ? c3_launch_inst<ST, NT, C3_EPI_BNBWD, false, true, NW>(a, p, s) : c3_launch_inst<ST, NT, C3_EPI_BNBWD, false, false, NW>(a, p, s);
}

struct Conv3Bn { const void *x; const float *mean, *var, *gamma, *beta; int ldx, relu, xbf16; float eps; };

// internal entry (spconv2.hip's dispatcher and the C ABI below).  Returns D3_ERR_ARG for shapes without an instance.
int d3_conv3_run(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, const float *res, int ldr, float *part,
                 double *part2, int Min, int Mout, int Cin, int Cout, int obf16, const Conv3Bn *bn, int *nparts_out, hipStream_t s) {
    const Conv3Plan p = conv3_plan(Mout, Cin, Cout);
    if (!p.ok || !x || !tq || !Wp || !out) return D3_ERR_ARG;
    if ((ldx & 7) || ldx < Cin || ldo < Cout || (ldo & 3) || (res && (ldr & 3)) || (res && bn) || (res && obf16)) return D3_ERR_ARG;
    if (bn && ((bn->xbf16 ? (bn->ldx & 3) : (bn->ldx & 3)))) return D3_ERR_ARG;
    const unsigned long long rowb = (unsigned long long)ldx * 2ull;
    const unsigned long long xb = Min > 0 ? ((unsigned long long)(Min - 1) * ldx + Cin) * 2ull : 0ull;
    if (Min <= 0 || Min >= (1 << 24) || rowb > 16384ull || xb > 0x7FFFFFFFull) return D3_ERR_RANGE;
    Conv3Args a;
    a.x = x; a.tq = (const uint4 *)tq; a.Wp = (const uint4 *)Wp; a.out = out; a.res = res; a.part = part; a.part2 = part ? part2 : nullptr;
    a.ldx = ldx; a.ldo = ldo; a.ldr = ldr; a.Mout = Mout; a.ntiles = (Mout + 15) / 16; a.xbytes = xb;
    a.bnx = nullptr; a.bn_mean = a.bn_var = a.bn_gamma = a.bn_beta = nullptr; a.ldbx = 0; a.bn_relu = 0; a.bn_eps = 0.f;
    if (bn) { a.bnx = bn->x; a.bn_mean = bn->mean; a.bn_var = bn->var; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta; a.ldbx = bn->ldx; a.bn_relu = bn->relu; a.bn_eps = bn->eps; }
    const int epi = bn ? C3_EPI_BNBWD : (res ? C3_EPI_RES : C3_EPI_PLAIN);
    const bool obf = obf16 != 0, bxbf = bn && bn->xbf16;
    if (nparts_out) *nparts_out = p.grid;
    const int ST = Cin / 8, NT = Cout / 16;
    const double bytes = 2.0 * (double)Min * Cin + (obf ? 2.0 : 4.0) * (double)Mout * Cout + 2.0 * 27.0 * Cin * Cout + 64.0 * (double)Mout + (res ? 4.0 * (double)Mout * Cout : 0.0);
    void *pr = d3_prof_begin(0, bytes, 0.0, s);
    int rc = D3_ERR_ARG;
    a.plane_bytes = (unsigned long long)Min * 16ull;
    if (g_c3_planar_probe == 2 && ST == 2 && NT == 1 && epi == C3_EPI_PLAIN && !obf) { rc = c3_launch_masked_probe(a, p, s); d3_prof_end(pr, s); return rc; }
    if (g_c3_planar_probe == 1 && ST == 2 && NT == 1 && epi == C3_EPI_PLAIN && !obf) { rc = c3_launch_planar_probe(a, p, s); d3_prof_end(pr, s); return rc; }
#define C3_SHAPE(STV, NTV, NWV) if (ST == STV && NT == NTV && p.nw == NWV) rc = c3_launch_shape<STV, NTV, NWV>(a, p, epi, obf, bxbf, s);
    C3_SHAPE(2, 1, 4) C3_SHAPE(2, 2, 8) C3_SHAPE(4, 1, 8) C3_SHAPE(4, 2, 8) C3_SHAPE(4, 4, 16) C3_SHAPE(6, 3, 16) C3_SHAPE(8, 2, 16)
#undef C3_SHAPE
    if (pr) {
        const int dims[12] = {Min, Mout, 27, Cin, Cout, NT, 1, 1, p.nw, 0, 27, ST + 8000};      // (+ 8000: spconv_fwd3_kernel, see bench.py's kernel naming)
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
    return d3_conv3_run(x, ldx, tq, Wp, out, ldo, res, ldr, part, part2, Min, Mout, Cin, Cout, (flags & D3_CONV_OUTBF16) ? 1 : 0, nullptr, nullptr, d3_stream(stream));
}
// data gradient of a BatchNorm -> ReLU -> convolution unit (as d3_spconv_fwd2_bnbwd); bnx: the BatchNorm input, fp32, or bf16 with
// D3_CONV_XBF16 in flags
extern "C" int d3_spconv_fwd3_bnbwd(const void *x, int ldx, const void *tq, const void *Wp, void *out, int ldo, float *part, double *part2,
                                    const void *bnx, int ldbx, const float *mean, const float *var, const float *gamma, const float *beta,
                                    float eps, int relu, int Min, int Mout, int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (Mout <= 0) return 0;
    Conv3Bn bn{bnx, mean, var, gamma, beta, ldbx, relu, (flags & D3_CONV_XBF16) ? 1 : 0, eps};
    return d3_conv3_run(x, ldx, tq, Wp, out, ldo, nullptr, 0, part, part2, Min, Mout, Cin, Cout, (flags & D3_CONV_OUTBF16) ? 1 : 0, &bn, nullptr, d3_stream(stream));
}
