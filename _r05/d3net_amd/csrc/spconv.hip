// spconv.hip -- sparse 3-D convolution as output-stationary gather -> LDS -> MFMA (gfx950).
//
// Stands in for MinkowskiConvolution / MinkowskiConvolutionTranspose forward, data gradient and
// weight gradient (reference call sites: model/common.py:32,38,41,66,90,98; model/pointgroup.py:70).
//
//   out[u,:] = sum_k x[tbl[u,k],:] @ W[k]            tbl: dense (Mout,K) kernel map (coordmap.hip)
//
// Every convolution of the U-Net is this one contraction with a different table:
//   kernel-3 fwd: tbl=nbr(27)          dgrad: tbl=nbr, W[26-k]^T            wgrad: tbl=nbr
//   down k2s2   : tbl=child(8)         dgrad: tbl=up,    W[k]^T             wgrad: tbl=child
//   up   k2s2^T : tbl=up(8)            dgrad: tbl=child, W[k]^T             wgrad: tbl=up
//   1x1         : tbl=NULL (identity)
// Output-stationary: a workgroup owns 64 output rows and all Cout channels, walks the K offsets,
// gathers the 64 input rows of the offset into LDS as bf16 (fp32 in HBM), multiplies by W[k] on the
// matrix cores (v_mfma_f32_16x16x16_bf16, fp32 accumulate) and writes each output row once: no
// scatter, no atomics, deterministic.  Offsets no row of the tile uses are skipped.
// Roofline: HBM.  Algorithmic bytes per launch = 4*(Min*Cin + Mout*Cout) + 4*K*Cin*Cout + 4*Mout*K
// (features once, weights once, table once); FLOPs = 2*pairs*Cin*Cout, AI 8..56 FLOP/B << 300.
#include "common.h"
#include "prof.h"

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
    return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
}

// ------------------------------------------------------------------------------ launch timing (bench.py)
// When enabled, every MFMA convolution launch is bracketed by two HIP events on its own stream and tagged
// with its algorithmic byte / flop count; d3_prof_collect() resolves them after the timed region.
#include <deque>
#include <vector>
struct ProfRec { hipEvent_t a, b; int family; double bytes, flops; int tag[D3_PROF_TAGS]; int dev_slot; double dev_scale; };
static std::deque<ProfRec> g_prof;   // stable element addresses
static size_t g_prof_used = 0;
static int g_prof_on = 0;
#define PROF_MAX 200000

#define PROF_DEV_SLOTS 4096
static double *g_prof_dev = nullptr;
static int g_prof_dev_used = 0;
static int g_prof_stride = 1;
static unsigned long long g_prof_seq = 0;
// on = 0: off; on = n >= 1: bracket every n-th convolution launch (an event pair is a queue barrier plus a timestamp
// write: bracketing all ~260 launches of a step costs the step ~2 ms; a stride coprime with the launches per step
// rotates through the layers, so over the timed region every layer is sampled)
extern "C" int d3_prof_enable(int on) {
    D3_CLEAR();
    g_prof_on = on > 0 ? 1 : 0;
    g_prof_stride = on > 1 ? on : 1;
    g_prof_used = 0;
    g_prof_seq = 0;
    g_prof_dev_used = 0;
    return 0;
}
#include <mutex>
static std::mutex g_prof_mu;
static ProfRec *prof_begin(int family, double bytes, double flops, hipStream_t s) {
    if (!g_prof_on) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((g_prof_seq++ % (unsigned long long)g_prof_stride) != 0 || g_prof_used >= PROF_MAX) return nullptr;
    if (g_prof_used == g_prof.size()) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
        g_prof.push_back(r);
    }
    ProfRec *r = &g_prof[g_prof_used++];
    r->family = family; r->bytes = bytes; r->flops = flops;
    for (int i = 0; i < D3_PROF_TAGS; i++) r->tag[i] = 0;
    r->dev_slot = -1; r->dev_scale = 0.0;
    hipEventRecord(r->a, s);
    return r;
}
static void prof_end(ProfRec *r, hipStream_t s) { if (r) hipEventRecord(r->b, s); }
void *d3_prof_begin(int family, double bytes, double flops, hipStream_t s) { return prof_begin(family, bytes, flops, s); }
void d3_prof_end(void *rec, hipStream_t s) { prof_end((ProfRec *)rec, s); }
double *d3_prof_dev_slot(void *rec, double scale) {
    if (!rec) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_dev && hipMalloc((void **)&g_prof_dev, PROF_DEV_SLOTS * sizeof(double)) != hipSuccess) { g_prof_dev = nullptr; return nullptr; }
    if (g_prof_dev_used >= PROF_DEV_SLOTS) return nullptr;
    ProfRec *r = (ProfRec *)rec;
    r->dev_slot = g_prof_dev_used++; r->dev_scale = scale;
    return g_prof_dev + r->dev_slot;
}
void d3_prof_tag(void *rec, int idx, int value) { if (rec && idx >= 0 && idx < D3_PROF_TAGS) ((ProfRec *)rec)->tag[idx] = value; }
// family: 0 = spconv_fwd2 / spconv_fwd_mfma (forward + data gradient), 1 = weight gradient, 2 = spconv_fwd2_split.
// The elapsed time of an EMPTY event pair on the same stream (median of 32) is subtracted from every sample: it is the
// cost of the bracket itself, not of the kernel (rocprofv3's kernel durations carry no such term).
extern "C" int d3_prof_collect(int family, long long *launches, double *total_ms, double *total_bytes,
                               double *total_flops) {
    D3_CLEAR();
    *launches = 0; *total_ms = 0; *total_bytes = 0; *total_flops = 0;
    static double empty_ms = -1.0;
    if (empty_ms < 0.0) {
        hipEvent_t a, b;
        D3_CHECK(hipEventCreate(&a)); D3_CHECK(hipEventCreate(&b));
        float v[32];
        for (int i = 0; i < 32; i++) {
            hipEventRecord(a, 0); hipEventRecord(b, 0);
            D3_CHECK(hipEventSynchronize(b));
            v[i] = 0.f; hipEventElapsedTime(&v[i], a, b);
        }
        for (int i = 0; i < 32; i++) for (int j = i + 1; j < 32; j++) if (v[j] < v[i]) { float t = v[i]; v[i] = v[j]; v[j] = t; }
        empty_ms = v[16];
        hipEventDestroy(a); hipEventDestroy(b);
    }
    for (size_t i = 0; i < g_prof_used; i++) {
        ProfRec &r = g_prof[i];
        if (r.family != family) continue;
        D3_CHECK(hipEventSynchronize(r.b));
        float ms = 0.f;
        D3_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        double d = (double)ms - empty_ms;
        if (d < 0.0005) d = 0.0005;
        *launches += 1; *total_ms += d; *total_bytes += r.bytes; *total_flops += r.flops;
    }
    return 0;
}

// every sampled launch of a family: rows of (3 + D3_PROF_TAGS) doubles = {ms (empty event pair subtracted), bytes, flops, tags...};
// *n = records of the family (rows beyond `cap` are counted, not written).  Synchronises like d3_prof_collect.
extern "C" int d3_prof_dump(int family, double *rows, int cap, int *n) {
    D3_CLEAR();
    hipEvent_t ea, eb;
    D3_CHECK(hipEventCreate(&ea)); D3_CHECK(hipEventCreate(&eb));
    float v[32];
    for (int i = 0; i < 32; i++) {
        hipEventRecord(ea, 0); hipEventRecord(eb, 0);
        D3_CHECK(hipEventSynchronize(eb));
        v[i] = 0.f; hipEventElapsedTime(&v[i], ea, eb);
    }
    for (int i = 0; i < 32; i++) for (int j = i + 1; j < 32; j++) if (v[j] < v[i]) { float t = v[i]; v[i] = v[j]; v[j] = t; }
    const double empty_ms = v[16];
    hipEventDestroy(ea); hipEventDestroy(eb);
    int k = 0;
    const int W = 3 + D3_PROF_TAGS;
    std::vector<double> devv((size_t)(g_prof_dev_used > 0 ? g_prof_dev_used : 1), 0.0);
    if (g_prof_dev && g_prof_dev_used > 0) D3_CHECK(hipMemcpy(devv.data(), g_prof_dev, (size_t)g_prof_dev_used * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < g_prof_used; i++) {
        ProfRec &r = g_prof[i];
        if (r.family != family) continue;
        if (k < cap && rows) {
            D3_CHECK(hipEventSynchronize(r.b));
            float ms = 0.f;
            D3_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
            double d = (double)ms - empty_ms;
            if (d < 0.0005) d = 0.0005;
            double *o = rows + (size_t)k * W;
            o[0] = d; o[1] = r.bytes + (r.dev_slot >= 0 ? r.dev_scale * devv[(size_t)r.dev_slot] : 0.0); o[2] = r.flops;
            for (int t = 0; t < D3_PROF_TAGS; t++) o[3 + t] = (double)r.tag[t];
        }
        k++;
    }
    *n = k;
    return 0;
}

// ------------------------------------------------------------------------------ exact fp32 kernels
// One thread per output element; used for D3_CONV_EXACT (validation / fp32 mode).
__global__ void spconv_fwd_exact_kernel(const float *__restrict__ x, const int *__restrict__ tbl,
                                        const float *__restrict__ W, float *__restrict__ out, int Mout, int K,
                                        int Cin, int Cout, int flipk, int transw) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)Mout * Cout) return;
    const int u = (int)(e / Cout), co = (int)(e % Cout);
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
        const int idx = tbl ? tbl[(long long)u * K + k] : u;
        if (idx < 0) continue;
        const int wk = flipk ? (K - 1 - k) : k;
        const float *xr = x + (long long)idx * Cin;
        const float *w = W + (long long)wk * Cin * Cout;
        if (transw) { for (int ci = 0; ci < Cin; ci++) acc = fmaf(xr[ci], w[(long long)co * Cin + ci], acc); }
        else { for (int ci = 0; ci < Cin; ci++) acc = fmaf(xr[ci], w[(long long)ci * Cout + co], acc); }
    }
    out[e] = acc;
}
// one thread per weight element, serial over rows (validation only)
__global__ void spconv_wgrad_exact_kernel(const float *__restrict__ x, const int *__restrict__ tbl,
                                          const float *__restrict__ dy, float *__restrict__ dW, int rows, int K,
                                          int Cin, int Cout, int xstat, int flipk) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)K * Cin * Cout) return;
    const int k = (int)(e / ((long long)Cin * Cout)), ci = (int)((e / Cout) % Cin), co = (int)(e % Cout);
    float acc = 0.f;
    for (int r = 0; r < rows; r++) {
        const int idx = tbl ? tbl[(long long)r * K + k] : r;
        if (idx < 0) continue;
        const int xr = xstat ? r : idx, dr = xstat ? idx : r;
        acc = fmaf(x[(long long)xr * Cin + ci], dy[(long long)dr * Cout + co], acc);
    }
    const int wk = flipk ? (K - 1 - k) : k;
    dW[(long long)wk * Cin * Cout + (long long)ci * Cout + co] += acc;
}

// ------------------------------------------------------------------------------ MFMA forward / dgrad
#define CV_BM 64      // output rows per workgroup
#define CV_KC 32      // reduction chunk (input channels) per stage
#define CV_LD 40      // LDS row stride in bf16 (80 B: conflict-free ds_read_b64 fragments)
#define CV_MAXK 27

// Software-pipelined main loop: the (kernel offset, 32-channel chunk) stages that the tile actually uses are walked
// with double-buffered LDS tiles; the global loads of stage s+1 (gathered rows + weight chunk) are issued into
// registers before the MFMAs of stage s and written to the other LDS buffer afterwards -- one barrier per stage, and
// the gather latency hides behind the matrix work and the LDS reads of the current stage.
template <int NT, bool TRANSW>
__global__ __launch_bounds__(256) void spconv_fwd_mfma_kernel(const float *__restrict__ x,
                                                             const int *__restrict__ tbl,
                                                             const float *__restrict__ W, float *__restrict__ out,
                                                             int Mout, int K, int Cin, int Cout, int flipk, int kper,
                                                             int xbf16) {
    constexpr int CoutP = NT * 16;  // Cout rounded up to the MFMA tile; columns >= Cout are zero / not stored
    __shared__ int tblS[CV_BM * CV_MAXK];
    __shared__ unsigned int kmaskS;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][CV_BM * CV_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bt[2][NT * 16 * CV_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * CV_BM;
    if (t == 0) kmaskS = 0u;
    __syncthreads();
    // kernel-map rows of this tile, coalesced; and the set of offsets at least one row of the tile uses
    {
        unsigned int bits = 0u;
        for (int e = t; e < CV_BM * K; e += 256) {
            const int r = e / K, k = e % K;
            const int u = row0 + r;
            const int v = (u < Mout) ? (tbl ? tbl[(long long)u * K + k] : u) : -1;
            tblS[e] = v;
            if (v >= 0) bits |= 1u << k;
        }
        if (bits) atomicOr(&kmaskS, bits);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; n++) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // gridDim.y > 1: the K offsets are split over workgroups (few-row levels: the serial offset loop is pure
    // latency) and the partial sums are added atomically into a zero-filled output
    const int k_begin = blockIdx.y * kper, k_end = min(K, k_begin + kper);
    unsigned int kmask = kmaskS;
    kmask &= (k_end >= 32 ? 0xFFFFFFFFu : ((1u << k_end) - 1u)) & ~((1u << k_begin) - 1u);
    const int arow = t >> 2, aq = t & 3;  // staging role: row, 8-channel group
    const bool cin4 = (Cin & 3) == 0;

    float va[8];          // staged A values of the next stage (fp32 input)
    uint4 vpk = make_uint4(0u, 0u, 0u, 0u);   // ... or 8 packed bf16 when the input is stored as bf16 (D3_CONV_XBF16)
    float vb[NT][2];      // staged B values of the next stage
    auto load_stage = [&](int k, int c0) {
        const int idx = tblS[arow * K + k];
#pragma unroll
        for (int j = 0; j < 8; j++) va[j] = 0.f;
        const int c = c0 + aq * 8;
        if (xbf16) {   // Cin % 8 == 0 (checked on the host): one 16-byte gather per thread, no conversion
            vpk = make_uint4(0u, 0u, 0u, 0u);
            if (idx >= 0 && c + 8 <= Cin) vpk = *(const uint4 *)((const unsigned short *)x + (long long)idx * Cin + c);
        } else if (idx >= 0) {
            const float *src = x + (long long)idx * Cin + c;
            if (cin4) {
                if (c + 4 <= Cin) { float4 f = *(const float4 *)src; va[0] = f.x; va[1] = f.y; va[2] = f.z; va[3] = f.w; }
                if (c + 8 <= Cin) { float4 f = *(const float4 *)(src + 4); va[4] = f.x; va[5] = f.y; va[6] = f.z; va[7] = f.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c + 2 * j + 2 <= Cin) { float2 f = *(const float2 *)(src + 2 * j); va[2 * j] = f.x; va[2 * j + 1] = f.y; }
            }
        }
        const int wk = flipk ? (K - 1 - k) : k;
        const float *Wk = W + (long long)wk * Cin * Cout;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int e = t + i * 256;       // 16 * CoutP == NT * 256 elements, one (n, k-pair) each
            float w0 = 0.f, w1 = 0.f;
            if (TRANSW) {  // W laid out (K, Cout, Cin): contiguous along the reduction index
                const int n = e >> 4, kp = e & 15, cc = c0 + 2 * kp;
                if (n < Cout && cc + 2 <= Cin) { float2 f = *(const float2 *)(Wk + (long long)n * Cin + cc); w0 = f.x; w1 = f.y; }
            } else {       // W laid out (K, Cin, Cout): coalesced along n
                const int kp = e / CoutP, n = e % CoutP, cc = c0 + 2 * kp;
                if (n < Cout && cc < Cin) w0 = Wk[(long long)cc * Cout + n];
                if (n < Cout && cc + 1 < Cin) w1 = Wk[(long long)(cc + 1) * Cout + n];
            }
            vb[i][0] = w0; vb[i][1] = w1;
        }
    };
    auto store_stage = [&](int buf) {
        uint4 pk = vpk;
        if (!xbf16) { pk.x = pack2bf(va[0], va[1]); pk.y = pack2bf(va[2], va[3]); pk.z = pack2bf(va[4], va[5]); pk.w = pack2bf(va[6], va[7]); }
        *(uint4 *)&As[buf][arow * CV_LD + aq * 8] = pk;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int e = t + i * 256;
            int n, kp;
            if (TRANSW) { n = e >> 4; kp = e & 15; } else { kp = e / CoutP; n = e % CoutP; }
            *(unsigned int *)&Bt[buf][n * CV_LD + 2 * kp] = pack2bf(vb[i][0], vb[i][1]);
        }
    };

    if (kmask != 0u) {
        int k = (int)__builtin_ctz(kmask), c0 = 0;
        unsigned int rest = kmask & (kmask - 1u);
        load_stage(k, c0);
        store_stage(0);
        __syncthreads();
        int buf = 0;
        for (;;) {
            // next stage
            int nk = k, nc0 = c0 + CV_KC;
            bool more = true;
            if (nc0 >= Cin) {
                nc0 = 0;
                if (rest == 0u) more = false;
                else { nk = (int)__builtin_ctz(rest); rest &= rest - 1u; }
            }
            if (more) load_stage(nk, nc0);
            const int ksteps = (Cin - c0 > 16) ? 2 : 1;
            for (int ks = 0; ks < ksteps; ks++) {
                const bf16x4 a = *(const bf16x4 *)&As[buf][(wave * 16 + (lane & 15)) * CV_LD + ks * 16 + (lane >> 4) * 4];
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const bf16x4 b = *(const bf16x4 *)&Bt[buf][(n * 16 + (lane & 15)) * CV_LD + ks * 16 + (lane >> 4) * 4];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[n], 0, 0, 0);
                }
            }
            if (!more) break;
            store_stage(buf ^ 1);
            __syncthreads();
            buf ^= 1; k = nk; c0 = nc0;
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int n = 0; n < NT; n++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = row0 + wave * 16 + (lane >> 4) * 4 + r;
            const int col = n * 16 + (lane & 15);
            if (u < Mout && col < Cout) {
                if (gridDim.y == 1) out[(long long)u * Cout + col] = acc[n][r];
                else if (kmask != 0u) atomicAdd(&out[(long long)u * Cout + col], acc[n][r]);
            }
        }
    }
}

template <bool TRANSW>
static int launch_fwd_mfma(const float *x, const int *tbl, const float *W, float *out, int Mout, int K, int Cin,
                           int Cout, int flipk, int xbf16, hipStream_t s) {
    const int tiles = (Mout + CV_BM - 1) / CV_BM;
    // enough workgroups to cover the chip: split the offsets when there are few row tiles
    int ksplit = 1;
    if (K > 1 && tiles < 256) { ksplit = 512 / tiles; if (ksplit > K) ksplit = K; if (ksplit < 1) ksplit = 1; }
    const int kper = (K + ksplit - 1) / ksplit;
    ksplit = (K + kper - 1) / kper;
    if (ksplit > 1) D3_CHECK(hipMemsetAsync(out, 0, (size_t)Mout * Cout * sizeof(float), s));
    const dim3 grid(tiles, ksplit);
#define CV_CASE(NTV)                                                                                              \
    case NTV:                                                                                                     \
        spconv_fwd_mfma_kernel<NTV, TRANSW><<<grid, 256, 0, s>>>(x, tbl, W, out, Mout, K, Cin, Cout, flipk, kper, xbf16); \
        break;
    switch ((Cout + 15) / 16) {
        CV_CASE(1) CV_CASE(2) CV_CASE(3) CV_CASE(4) CV_CASE(5) CV_CASE(6) CV_CASE(7) CV_CASE(8) CV_CASE(9)
        CV_CASE(10) CV_CASE(11) CV_CASE(12) CV_CASE(13) CV_CASE(14)
        default: return D3_ERR_ARG;
    }
#undef CV_CASE
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_spconv_fwd(const float *x, const int *tbl, const float *W, float *out, int Min, int Mout, int K,
                             int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (Mout <= 0) return 0;
    if (K < 1 || K > CV_MAXK || Cin < 1 || Cout < 1) return D3_ERR_ARG;
    if (tbl == nullptr && K != 1) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const int flipk = (flags & D3_CONV_FLIPK) ? 1 : 0, transw = (flags & D3_CONV_TRANSW) ? 1 : 0;
    if ((flags & D3_CONV_EXACT) && (flags & D3_CONV_XBF16)) return D3_ERR_ARG;
    if (flags & D3_CONV_EXACT) {
        long long total = (long long)Mout * Cout;
        spconv_fwd_exact_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(x, tbl, W, out, Mout, K, Cin, Cout, flipk,
                                                                         transw);
        D3_LAUNCH_CHECK();
        return 0;
    }
    const int xbf16 = (flags & D3_CONV_XBF16) ? 1 : 0;
    if ((Cin & 1) != 0 || Cout > 224 || (xbf16 && (Cin & 7) != 0)) return D3_ERR_ARG;
    // algorithmic traffic: features in once (2 B/elem when stored as bf16), out once, weights once, one table entry
    // per (row, offset)
    const double bytes = (xbf16 ? 2.0 : 4.0) * (double)Min * Cin + 4.0 * ((double)Mout * Cout + (double)K * Cin * Cout) +
                         (tbl ? 4.0 * (double)Mout * K : 0.0);
    ProfRec *pr = prof_begin(0, bytes, 0.0, s);
    int rc = transw ? launch_fwd_mfma<true>(x, tbl, W, out, Mout, K, Cin, Cout, flipk, xbf16, s)
                    : launch_fwd_mfma<false>(x, tbl, W, out, Mout, K, Cin, Cout, flipk, xbf16, s);
    prof_end(pr, s);
    return rc;
}

// ------------------------------------------------------------------------------ MFMA weight gradient
// dW[k][ci][co] += sum_u x[tbl[u,k]][ci] * dy[u][co]:  M-dim = ci, N-dim = co, reduction = rows.
// Two equivalent row orders: dy-stationary (rows u of dy are contiguous, x rows gathered through tbl) or, with
// D3_CONV_XSTAT, x-stationary (rows v of x contiguous, dy rows gathered through the TRANSPOSED map, dW index
// flipped for a kernel-3 conv): the wider operand is the one read contiguously.
// grid = (row blocks, K, tile passes).  Each wave owns an equal share of the block's rows, stages 32 rows at a time
// TRANSPOSED into its private LDS region (Xt[ci][row], DYt[co][row]) so that both MFMA operands are
// contiguous 8-byte reads, keeps up to WG_MAXT 16x16 accumulators, and the workgroup's four
// partial results are added to dW with fp32 atomics (order-dependent rounding in the last bits).
#define WG_ROWS_MAX 4096   // rows per workgroup (chosen by the host so that the grid covers the chip)
#define WG_RC 32       // rows per stage
#define WG_MAXT 16     // accumulator tiles per pass

__global__ __launch_bounds__(256) void spconv_wgrad_mfma_kernel(const float *__restrict__ x,
                                                               const int *__restrict__ tbl,
                                                               const float *__restrict__ dy, float *__restrict__ dW,
                                                               int Mout, int K, int Cin, int Cout, int rows_per_block,
                                                               int xstat, int flipk, int xbf16) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int CinP = (Cin + 15) & ~15, CoutP = (Cout + 15) & ~15;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;
    unsigned short *Xt = smem + (size_t)wave * (CinP + CoutP) * CV_LD;
    unsigned short *DYt = Xt + (size_t)CinP * CV_LD;
    const int k = blockIdx.y;
    const int rb0 = blockIdx.x * rows_per_block;       // Mout = number of stationary rows
    const int rows_blk = min(rows_per_block, Mout - rb0);
    const int per_wave = (rows_blk + nwaves - 1) / nwaves;
    const int w0 = rb0 + wave * per_wave;                     // this wave's rows [w0, w1)
    const int w1 = min(rb0 + rows_blk, w0 + per_wave);
    const int nchunks = (per_wave + WG_RC - 1) / WG_RC;       // uniform over the block
    const int mt = CinP / 16, nt = CoutP / 16, ntiles = mt * nt;
    const int srow = lane >> 1, shalf = lane & 1;             // staging role: row of the chunk, channel phase
    float *dWk = dW + (long long)(flipk ? (K - 1 - k) : k) * Cin * Cout;

    {   // one pass of up to WG_MAXT accumulator tiles per workgroup; passes are spread over gridDim.z
        const int tile0 = blockIdx.z * WG_MAXT;
        f32x4 acc[WG_MAXT];
#pragma unroll
        for (int i = 0; i < WG_MAXT; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool any_valid = false;
        for (int ch = 0; ch < nchunks; ch++) {
            const int u = w0 + ch * WG_RC + srow;
            int idx = -1;
            if (u < w1) idx = tbl ? tbl[(long long)u * K + k] : u;
            const bool chunk_valid = __syncthreads_or(idx >= 0) != 0;
            if (!chunk_valid) continue;   // uniform
            any_valid = true;
            const int xrow = xstat ? u : idx, dyrow = xstat ? idx : u;
            // stage x (gathered) and dy, transposed: element (c, row) at [c*CV_LD + row]
            if (xbf16) {   // x stored as bf16: copy the pair as is
                const unsigned short *xb = (const unsigned short *)x;
                for (int c = shalf * 2; c < CinP; c += 4) {
                    unsigned int pr = 0u;
                    if (idx >= 0 && c + 2 <= Cin) pr = *(const unsigned int *)(xb + (long long)xrow * Cin + c);
                    Xt[c * CV_LD + srow] = (unsigned short)(pr & 0xFFFFu);
                    Xt[(c + 1) * CV_LD + srow] = (unsigned short)(pr >> 16);
                }
            } else
            for (int c = shalf * 2; c < CinP; c += 4) {
                float a = 0.f, b = 0.f;
                if (idx >= 0 && c + 2 <= Cin) { float2 f = *(const float2 *)(x + (long long)xrow * Cin + c); a = f.x; b = f.y; }
                else if (idx >= 0 && c < Cin) { a = x[(long long)xrow * Cin + c]; }
                Xt[c * CV_LD + srow] = f2bf(a);
                Xt[(c + 1) * CV_LD + srow] = f2bf(b);
            }
            for (int c = shalf * 2; c < CoutP; c += 4) {
                float a = 0.f, b = 0.f;
                if (idx >= 0 && c + 2 <= Cout) { float2 f = *(const float2 *)(dy + (long long)dyrow * Cout + c); a = f.x; b = f.y; }
                else if (idx >= 0 && c < Cout) { a = dy[(long long)dyrow * Cout + c]; }
                DYt[c * CV_LD + srow] = f2bf(a);
                DYt[(c + 1) * CV_LD + srow] = f2bf(b);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < WG_MAXT; i++) {
                const int tile = tile0 + i;
                if (tile < ntiles) {
                    const int mi = tile / nt, ni = tile % nt;
#pragma unroll
                    for (int ks = 0; ks < WG_RC / 16; ks++) {
                        const bf16x4 a = *(const bf16x4 *)&Xt[(mi * 16 + (lane & 15)) * CV_LD + ks * 16 + (lane >> 4) * 4];
                        const bf16x4 b = *(const bf16x4 *)&DYt[(ni * 16 + (lane & 15)) * CV_LD + ks * 16 + (lane >> 4) * 4];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[i], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        if (any_valid) {
#pragma unroll
            for (int i = 0; i < WG_MAXT; i++) {
                const int tile = tile0 + i;
                if (tile < ntiles) {
                    const int mi = tile / nt, ni = tile % nt;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int ci = mi * 16 + (lane >> 4) * 4 + r, co = ni * 16 + (lane & 15);
                        if (ci < Cin && co < Cout) atomicAdd(&dWk[(long long)ci * Cout + co], acc[i][r]);
                    }
                }
            }
        }
    }
}

extern "C" int d3_spconv_wgrad(const float *x, const int *tbl, const float *dy, float *dW, int Min, int Mout, int K,
                               int Cin, int Cout, int flags, void *stream) {
    D3_CLEAR();
    if (Mout <= 0) return 0;
    if (K < 1 || K > CV_MAXK || Cin < 1 || Cout < 1) return D3_ERR_ARG;
    if (tbl == nullptr && K != 1) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const int xstat = (flags & D3_CONV_XSTAT) ? 1 : 0, flipk = (flags & D3_CONV_FLIPK) ? 1 : 0;
    const int rows = xstat ? Min : Mout;   // stationary rows == rows of tbl
    if (!(flags & D3_CONV_ACCUM)) D3_CHECK(hipMemsetAsync(dW, 0, (size_t)K * Cin * Cout * sizeof(float), s));
    if ((flags & D3_CONV_EXACT) && (flags & D3_CONV_XBF16)) return D3_ERR_ARG;
    if (flags & D3_CONV_EXACT) {
        long long total = (long long)K * Cin * Cout;
        spconv_wgrad_exact_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(x, tbl, dy, dW, rows, K, Cin, Cout, xstat,
                                                                           flipk);
        D3_LAUNCH_CHECK();
        return 0;
    }
    const int xbf16 = (flags & D3_CONV_XBF16) ? 1 : 0;
    if ((Cin & 1) != 0 || (Cout & 1) != 0) return D3_ERR_ARG;
    const int CinP = (Cin + 15) & ~15, CoutP = (Cout + 15) & ~15;
    // 64 KB of dynamic LDS per workgroup: wide layers run with fewer waves (each wave stages its own rows)
    int nwaves = 4;
    while (nwaves > 1 && (size_t)nwaves * (CinP + CoutP) * CV_LD * sizeof(unsigned short) > 64 * 1024) nwaves >>= 1;
    size_t lds = (size_t)nwaves * (CinP + CoutP) * CV_LD * sizeof(unsigned short);
    if (lds > 64 * 1024) return D3_ERR_ARG;
    const int passes = ((CinP / 16) * (CoutP / 16) + WG_MAXT - 1) / WG_MAXT;
    // rows per workgroup: aim at ~3000 workgroups (latency hiding by occupancy), 64..1024 rows per wave
    long long want = ((long long)rows * K * passes + 2999) / 3000;
    int rpb = (int)((want + nwaves * WG_RC - 1) / (nwaves * WG_RC)) * nwaves * WG_RC;
    if (rpb < nwaves * 2 * WG_RC) rpb = nwaves * 2 * WG_RC;
    if (rpb > WG_ROWS_MAX) rpb = WG_ROWS_MAX;
    dim3 grid((rows + rpb - 1) / rpb, K, passes);
    const double bytes = (xbf16 ? 2.0 : 4.0) * (double)Min * Cin + 4.0 * ((double)Mout * Cout + (double)K * Cin * Cout) +
                         (tbl ? 4.0 * (double)Mout * K : 0.0);
    ProfRec *pr = prof_begin(1, bytes, 0.0, s);
    spconv_wgrad_mfma_kernel<<<grid, nwaves * 64, lds, s>>>(x, tbl, dy, dW, rows, K, Cin, Cout, rpb, xstat, flipk, xbf16);
    prof_end(pr, s);
    D3_LAUNCH_CHECK();
    return 0;
}
