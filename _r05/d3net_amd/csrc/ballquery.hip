// ballquery.hip -- radius neighbour lists within each batch item (gfx950).
//
// Replaces PG_OP.ballquery_batch_p (reference: lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cu:15-90):
// one CUDA thread per point scanning ALL points of its batch item (O(n^2/B) distance tests, a
// 4 KB per-thread scratch array, cudaMalloc + two blocking memcpys per call, host retry loop).
//
// Contract kept bit-exact: for every point, the indices k (ascending) of the points of the same
// batch item with  (ox-x)*(ox-x)+(oy-y)*(oy-y)+(oz-z)*(oz-z) < r*r  (strict, evaluated exactly as
// written: separately rounded mul/add, no FMA contraction), stopping after 1000 hits.
//
// Design: ordered scan with bounding-box culling.  Points are grouped, in their given order,
// into chunks of 64 (one wave-load) and super-chunks of 64 chunks; a pre-pass computes each
// group's AABB.  One wave owns one query point: lanes test the 64 chunk boxes of a super-chunk
// in parallel (ballot), then for each surviving chunk the 64 lanes test its 64 points in
// parallel; `ballot` + `mbcnt` turn the hit mask into ascending output positions, so the
// neighbour list comes out sorted with no sort, and the 1000-hit early exit of the reference is
// kept (dense, collapsed clusters stop after ~16 chunk visits).  Scene point orders (mesh vertex
// order, raster order) are spatially coherent, so only a handful of chunks survive the culling;
// an incoherent order degrades towards the reference's brute force but stays exact.
// Two passes (count -> exclusive scan -> fill) give deterministic segment starts and let the
// caller allocate the exact output (the reference guesses n*meanActive and retries).
// HBM bound: bytes = 12*n (coords) + 8*n (start_len) + 4*nActive (lists); box tables are L2-resident.
#include "common.h"
#include <string.h>

#define BQ_CHUNK 64
#define BQ_SUPER 64
#define BQ_CAP 1000  // reference: int idx_temp[1000] (bfs_cluster.cu:20,38-44)

struct BqWs {
    float *clo, *chi;  // chunk boxes   (nchunks,3) each
    float *slo, *shi;  // super boxes   (nsuper,3) each
    int *len;          // n
    int *start;        // n
    int *total;        // 1
    void *temp; size_t temp_bytes;
    int nchunks, nsuper;
};

struct BqGrid;
static void bqg_carve(D3Carver &c, int n, BqGrid &g);
static bool bq_carve(void *ws, size_t ws_bytes, int n, BqWs &w, BqGrid *g = nullptr);
static bool bq_carve(void *ws, size_t ws_bytes, int n, BqWs &w, BqGrid *g) {
    D3Carver c(ws, ws_bytes);
    size_t nn = (size_t)(n > 0 ? n : 1);
    w.nchunks = (int)((nn + BQ_CHUNK - 1) / BQ_CHUNK);
    w.nsuper = (w.nchunks + BQ_SUPER - 1) / BQ_SUPER;
    w.clo = c.take<float>((size_t)w.nchunks * 3);
    w.chi = c.take<float>((size_t)w.nchunks * 3);
    w.slo = c.take<float>((size_t)w.nsuper * 3);
    w.shi = c.take<float>((size_t)w.nsuper * 3);
    w.len = c.take<int>(nn);
    w.start = c.take<int>(nn);
    w.total = c.take<int>(64);
    w.temp_bytes = d3_scan_temp_bytes(n);
    w.temp = c.take<char>(w.temp_bytes);
    if (g) bqg_carve(c, n, *g);        // the cell grid of the padded form sits behind the scan's tables
    return ws != nullptr && c.ok();
}

// the optional hit stash (n * BQ_CAP ints, 400 MB at n = 100k: HBM is 288 GB) sits behind the base workspace; both phases
// use it iff the caller's workspace is big enough (d3_ballquery_ws_bytes_single_pass)
extern "C" size_t d3_ballquery_ws_bytes(int n);
static int *bq_stash(void *ws, size_t ws_bytes, int n) {
    const size_t base = d3_align(d3_ballquery_ws_bytes(n), 4096);
    const size_t need = base + (size_t)(n > 0 ? n : 1) * BQ_CAP * sizeof(int);
    return (ws != nullptr && ws_bytes >= need) ? (int *)((char *)ws + base) : nullptr;
}
extern "C" size_t d3_ballquery_ws_bytes_single_pass(int n) {
    return d3_align(d3_ballquery_ws_bytes(n), 4096) + (size_t)(n > 0 ? n : 1) * BQ_CAP * sizeof(int);
}

static size_t bqg_cap_host(int n) { size_t c = 1024; while (c < (size_t)n * 2) c <<= 1; return c; }
extern "C" size_t d3_ballquery_ws_bytes(int n) {
    BqWs w;
    D3Carver c(nullptr, 0);
    size_t nn = (size_t)(n > 0 ? n : 1);
    int nchunks = (int)((nn + BQ_CHUNK - 1) / BQ_CHUNK), nsuper = (nchunks + BQ_SUPER - 1) / BQ_SUPER;
    c.take<float>((size_t)nchunks * 3); c.take<float>((size_t)nchunks * 3);
    c.take<float>((size_t)nsuper * 3); c.take<float>((size_t)nsuper * 3);
    c.take<int>(nn); c.take<int>(nn); c.take<int>(64);
    c.take<char>(d3_scan_temp_bytes(n));
    (void)w;
    // + the cell grid (d3_ballquery_padded)
    const size_t cap = bqg_cap_host(n);
    c.take<unsigned long long>(nn); c.take<unsigned long long>(nn); c.take<char>(cap * 16);
    c.take<int>(nn); c.take<int>(nn); c.take<int>(nn); c.take<int>(nn); c.take<int>(nn + 1); c.take<int>(nn); c.take<int>(cap); c.take<int>(nn);
    c.take<int>(64); c.take<int>(nn); c.take<float>(nn * 3); c.take<float>(nn * 6);
    size_t tb = d3_sort_pairs_u64_temp_bytes(n);
    if (d3_scan_temp_bytes(n) > tb) tb = d3_scan_temp_bytes(n);
    c.take<char>(tb);
    return c.off + 256;
}

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// one wave per chunk: AABB of its (up to 64) points
__global__ __launch_bounds__(256) void bq_chunk_box_kernel(const float *__restrict__ xyz, int n, float *clo,
                                                          float *chi, int nchunks) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= nchunks) return;
    const int k = wave * BQ_CHUNK + d3_lane();
    float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
    if (k < n) { x = X = xyz[k * 3 + 0]; y = Y = xyz[k * 3 + 1]; z = Z = xyz[k * 3 + 2]; }
    x = wave_min(x); y = wave_min(y); z = wave_min(z);
    X = wave_max(X); Y = wave_max(Y); Z = wave_max(Z);
    if (d3_lane() == 0) {
        clo[wave * 3 + 0] = x; clo[wave * 3 + 1] = y; clo[wave * 3 + 2] = z;
        chi[wave * 3 + 0] = X; chi[wave * 3 + 1] = Y; chi[wave * 3 + 2] = Z;
    }
}
// one wave per super-chunk: AABB of its (up to 64) chunk boxes
__global__ __launch_bounds__(256) void bq_super_box_kernel(const float *__restrict__ clo,
                                                          const float *__restrict__ chi, int nchunks, float *slo,
                                                          float *shi, int nsuper) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= nsuper) return;
    const int c = wave * BQ_SUPER + d3_lane();
    float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
    if (c < nchunks) {
        x = clo[c * 3 + 0]; y = clo[c * 3 + 1]; z = clo[c * 3 + 2];
        X = chi[c * 3 + 0]; Y = chi[c * 3 + 1]; Z = chi[c * 3 + 2];
    }
    x = wave_min(x); y = wave_min(y); z = wave_min(z);
    X = wave_max(X); Y = wave_max(Y); Z = wave_max(Z);
    if (d3_lane() == 0) {
        slo[wave * 3 + 0] = x; slo[wave * 3 + 1] = y; slo[wave * 3 + 2] = z;
        shi[wave * 3 + 0] = X; shi[wave * 3 + 1] = Y; shi[wave * 3 + 2] = Z;
    }
}

// conservative cull: a box can hold a neighbour only if the query is within rc (> r) of it on
// every axis.  rc = 1.01*r (+ a few ulps of the box coordinate) absorbs every rounding of the
// exact fp32 test below.
__device__ __forceinline__ bool bq_axis_near(float o, float lo, float hi, float rc) {
    // widen by a few ulps of the box coordinate so the fp32 subtraction below cannot cull a true hit
    return (o >= lo - (rc + 1e-6f * fabsf(lo))) && (o <= hi + (rc + 1e-6f * fabsf(hi)));
}
__device__ __forceinline__ bool bq_box_near(float ox, float oy, float oz, const float *lo, const float *hi, int i,
                                            float rc) {
    // all six loads first (a short-circuit && would wait for each axis before requesting the next)
    const float l0 = lo[i * 3 + 0], l1 = lo[i * 3 + 1], l2 = lo[i * 3 + 2];
    const float h0 = hi[i * 3 + 0], h1 = hi[i * 3 + 1], h2 = hi[i * 3 + 2];
    const bool a = bq_axis_near(ox, l0, h0, rc), b = bq_axis_near(oy, l1, h1, rc), c = bq_axis_near(oz, l2, h2, rc);
    return a & b & c;
}

// MODE 0: count; 1: fill idx at the scanned starts; 2: count AND stash the hits at stash[q*BQ_CAP + pos] (single pass:
// the fill then only compacts the stash instead of repeating the search)
#ifndef BQ_NB
#define BQ_NB 4     // candidate chunks whose points are requested together
#endif
template <int MODE>
__global__ __launch_bounds__(256) void bq_scan_kernel(const float *__restrict__ xyz,
                                                     const int *__restrict__ batch_idxs,
                                                     const int *__restrict__ batch_offsets, int n, float radius,
                                                     const float *__restrict__ clo, const float *__restrict__ chi,
                                                     const float *__restrict__ slo, const float *__restrict__ shi,
                                                     int nchunks, int *__restrict__ len_out,
                                                     const int *__restrict__ start_in, int *__restrict__ idx,
                                                     long long idx_capacity) {
    const int q = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (q >= n) return;
    const int lane = d3_lane();
    const unsigned long long lt = d3_lanemask_lt();
    const float radius2 = __fmul_rn(radius, radius);
    const float rc = radius * 1.01f + 1e-30f;
    const float ox = xyz[q * 3 + 0], oy = xyz[q * 3 + 1], oz = xyz[q * 3 + 2];
    const int b = batch_idxs[q];
    const int start = batch_offsets[b], end = batch_offsets[b + 1];
    long long base = 0;
    if (MODE == 1) base = start_in[q];
    if (MODE == 2) base = (long long)q * BQ_CAP;
    int cnt = 0;
    if (end > start) {
        const int c_first = start / BQ_CHUNK, c_last = (end - 1) / BQ_CHUNK;
        const int s_first = c_first / BQ_SUPER, s_last = c_last / BQ_SUPER;
        // the super boxes are tested 64 at a time, one per lane (a serial walk paid one round trip per super box)
        for (int sb = s_first; sb <= s_last && cnt < BQ_CAP; sb += 64) {
          const int sl = min(sb + lane, s_last);
          unsigned long long sm = __ballot(sb + lane <= s_last && bq_box_near(ox, oy, oz, slo, shi, sl, rc));
          while (sm != 0ull && cnt < BQ_CAP) {
            const int sc = sb + (int)__builtin_ctzll(sm);
            sm &= sm - 1ull;
            const int c = sc * BQ_SUPER + lane;
            const bool crange = (c >= c_first) && (c <= c_last);
            const bool pass = bq_box_near(ox, oy, oz, clo, chi, crange ? c : c_first, rc) && crange;
            unsigned long long cm = __ballot(pass);
            // BQ_NB candidate chunks per round trip: their points are requested together (branch-free addresses: a
            // branch between the loads would make each one wait for the previous), then tested in chunk order
            while (cm != 0ull && cnt < BQ_CAP) {
                int kk[BQ_NB];
                bool in[BQ_NB], valid[BQ_NB];
                float x[BQ_NB], y[BQ_NB], z[BQ_NB];
#pragma unroll
                for (int j = 0; j < BQ_NB; j++) {
                    valid[j] = cm != 0ull;
                    const int cc = sc * BQ_SUPER + (valid[j] ? (int)__builtin_ctzll(cm) : 0);
                    if (valid[j]) cm &= cm - 1ull;
                    kk[j] = cc * BQ_CHUNK + lane;
                    in[j] = valid[j] && kk[j] >= start && kk[j] < end;
                }
#pragma unroll
                for (int j = 0; j < BQ_NB; j++) {
                    const int ka = in[j] ? kk[j] : q;
                    x[j] = xyz[ka * 3 + 0]; y[j] = xyz[ka * 3 + 1]; z[j] = xyz[ka * 3 + 2];
                }
#pragma unroll
                for (int j = 0; j < BQ_NB; j++) {
                    if (!valid[j] || cnt >= BQ_CAP) continue;   // wave-uniform
                    const float dx = __fsub_rn(ox, x[j]), dy = __fsub_rn(oy, y[j]), dz = __fsub_rn(oz, z[j]);
                    // ((dx*dx + dy*dy) + dz*dz), every operation rounded separately (bfs_cluster.cu:36)
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    const bool hit = in[j] && d2 < radius2;
                    const unsigned long long hm = __ballot(hit);
                    if (MODE != 0 && hit) {
                        const int pos = cnt + (int)__popcll(hm & lt);
                        // cap (bfs_cluster.cu:38-44) and buffer truncation (bfs_cluster.cu:51-59)
                        if (pos < BQ_CAP && (MODE == 2 || base + pos < idx_capacity)) idx[base + pos] = kk[j];
                    }
                    cnt += (int)__popcll(hm);
                }
            }
          }
        }
    }
    if (cnt > BQ_CAP) cnt = BQ_CAP;
    if (MODE != 1 && lane == 0) len_out[q] = cnt;
}

// stash -> idx at the scanned starts (one wave per point; ascending order is preserved)
__global__ __launch_bounds__(256) void bq_compact_kernel(const int *__restrict__ stash, const int *__restrict__ len,
                                                        const int *__restrict__ start, int n, int *__restrict__ idx,
                                                        long long idx_capacity) {
    const int q = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (q >= n) return;
    const int ln = len[q];
    const long long st = start[q];
    for (int e = d3_lane(); e < ln; e += 64)
        if (st + e < idx_capacity) idx[st + e] = stash[(long long)q * BQ_CAP + e];
}

__global__ void bq_pack_kernel(const int *len, const int *start, int *start_len, int n, int *total) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start_len[i * 2 + 0] = start[i];
    start_len[i * 2 + 1] = len[i];
    if (i == n - 1) total[0] = start[i] + len[i];
}

// 64-bit total of the list lengths (only launched when n * 1000 could pass INT_MAX): total[2..3] as one unsigned long long.
// The reference's nActive is an int (lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cpp:17-22); past 2^31 - 1 entries the compact
// form has no representation -- the count then reports D3_ERR_RANGE instead of scanning a wrapped prefix.
__global__ __launch_bounds__(256) void bq_total64_kernel(const int *__restrict__ len, int n, unsigned long long *total64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = i < n ? (unsigned long long)len[i] : 0ull;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (d3_lane() == 0 && v) atomicAdd(total64, v);
}

static int bq_boxes(const float *xyz, int n, BqWs &w, hipStream_t s) {
    bq_chunk_box_kernel<<<(w.nchunks + 3) / 4, 256, 0, s>>>(xyz, n, w.clo, w.chi, w.nchunks);
    bq_super_box_kernel<<<(w.nsuper + 3) / 4, 256, 0, s>>>(w.clo, w.chi, w.nchunks, w.slo, w.shi, w.nsuper);
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_ballquery_count(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n,
                                  float radius, int *start_len, void *ws, size_t ws_bytes, int *nActive_host,
                                  void *stream) {
    D3_CLEAR();
    *nActive_host = 0;
    if (n <= 0) return 0;
    BqWs w;
    if (!bq_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int rc = bq_boxes(xyz, n, w, s);
    if (rc) return rc;
    int *stash = bq_stash(ws, ws_bytes, n);
    if (stash)
        bq_scan_kernel<2><<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, batch_offsets, n, radius, w.clo, w.chi, w.slo,
                                                     w.shi, w.nchunks, w.len, nullptr, stash, 0);
    else
        bq_scan_kernel<0><<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, batch_offsets, n, radius, w.clo, w.chi, w.slo,
                                                     w.shi, w.nchunks, w.len, nullptr, nullptr, 0);
    rc = d3_exclusive_scan_i32(w.len, w.start, n, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    bq_pack_kernel<<<(n + 255) / 256, 256, 0, s>>>(w.len, w.start, start_len, n, w.total);
    const bool may_wrap = (long long)n * BQ_CAP > 0x7FFFFFFFll;          // (lists are capped at BQ_CAP entries)
    if (may_wrap) {
        D3_CHECK(hipMemsetAsync(w.total + 2, 0, sizeof(unsigned long long), s));
        bq_total64_kernel<<<(n + 255) / 256, 256, 0, s>>>(w.len, n, (unsigned long long *)(w.total + 2));
    }
    D3_LAUNCH_CHECK();
    int h[4] = {0, 0, 0, 0};
    D3_CHECK(hipMemcpyAsync(h, w.total, may_wrap ? sizeof(h) : sizeof(int), hipMemcpyDeviceToHost, s));
    D3_CHECK(hipStreamSynchronize(s));
    if (may_wrap) {
        unsigned long long t64;
        memcpy(&t64, h + 2, sizeof(t64));
        if (t64 > 0x7FFFFFFFull) return D3_ERR_RANGE;      // nActive does not fit the reference's int
    }
    *nActive_host = h[0];
    return 0;
}

extern "C" int d3_ballquery_fill(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n,
                                 float radius, const int *start_len, const void *ws, size_t ws_bytes, int *idx,
                                 long long idx_capacity, void *stream) {
    D3_CLEAR();
    (void)start_len;
    if (n <= 0) return 0;
    BqWs w;
    if (!bq_carve((void *)ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    // boxes and starts are still in the workspace from the count phase; with the big workspace so are the hits
    int *stash = bq_stash((void *)ws, ws_bytes, n);
    if (stash) {
        bq_compact_kernel<<<(n + 3) / 4, 256, 0, s>>>(stash, w.len, w.start, n, idx, idx_capacity);
        D3_LAUNCH_CHECK();
        return 0;
    }
    bq_scan_kernel<1><<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, batch_offsets, n, radius, w.clo, w.chi, w.slo,
                                                    w.shi, w.nchunks, nullptr, w.start, idx, idx_capacity);
    D3_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------ cell grid
// The padded ball query (the form the model's two clustering branches use) searches a UNIFORM CELL GRID instead of scanning
// the points in their given order: cell edge = 1.001 * r, so every neighbour of a point lies in the 27 cells around its
// own (two coordinates closer than r differ by less than 1 - 1e-3 cell edges: the floors differ by at most one, whatever
// the fp32 rounding of x * (1 / edge)).  The ordered chunk scan loses its grip exactly where PointGroup needs it most -- the
// SHIFTED coordinates, where a chunk of 64 consecutive points spans many instance centres and nothing is culled.
//   1. key = (batch, cell) per point; a STABLE radix sort of (key, point index) orders the points by cell and keeps the
//      indices of a cell ascending; cells = runs of equal keys, (key -> run) goes into an open-addressing hash.
//   2. one wave per query: 27 hash probes (one per lane) give the candidate runs; candidates are read contiguously from
//      the sorted copies (index + coordinates), tested with the reference's expression (bfs_cluster.cu:36, every
//      operation rounded separately), and the hits are put in ascending index order: <= 64 candidates by a bitonic
//      network across the lanes, more through a per-wave LDS buffer (bitonic sort; beyond 1024 hits the buffer is cut
//      back to its 1000 smallest and later candidates must beat the 1000th) -- the reference's "first 1000 in index
//      order" (bfs_cluster.cu:38-44) without any order assumption on the input.
//   3. CLIQUE CELLS.  If the bounding box of ALL points in the 27 cells around cell c has a diagonal shorter than r, every
//      query of c finds exactly those points: identical lists.  Then only the cell's smallest index (its "leader")
//      searches; the other members' start_len entries point at the leader's slot.  A collapsed instance (what accurate
//      offset predictions produce: thousands of points within a millimetre of their centre) is one such cell -- its
//      list is written once instead of once per member (4 KB each), and the clustering kernels that walk it afterwards
//      (union / label push / star) read one cached copy.  Consumers only ever index idx[start + e], e < len.
#define BQG_EMPTY 0xFFFFFFFFFFFFFFFFull
#define BQG_BUF 2048            // per-wave LDS hit buffer (ints) of the dense kernel
#define BQG_BIAS 16384

// hash slot: (key, first sorted position of the cell, points in the cell) -- ONE 16-byte load answers a probe
struct __attribute__((aligned(16))) BqSlot { unsigned long long key; int start, count; };

struct BqGrid {
    unsigned long long *key;
    int *slot32, *skey;               // a point's cell slot (the sort key) / the sorted slots
    BqSlot *tbl;
    int *pid, *sidx, *head, *rid, *cstart, *cslot, *tlead, *leader_of, *scal, *dense;
    float *sxyz, *cbox;
    void *temp; size_t temp_bytes;
    size_t cap;
};
static size_t bqg_cap(int n) { size_t c = 1024; while (c < (size_t)n * 2) c <<= 1; return c; }
static void bqg_carve(D3Carver &c, int n, BqGrid &g) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    g.cap = bqg_cap(n);
    g.key = c.take<unsigned long long>(nn); g.slot32 = c.take<int>(nn); g.skey = c.take<int>(nn); g.tbl = c.take<BqSlot>(g.cap);
    g.pid = c.take<int>(nn); g.sidx = c.take<int>(nn); g.head = c.take<int>(nn); g.rid = c.take<int>(nn);
    g.cstart = c.take<int>(nn + 1); g.cslot = c.take<int>(nn); g.tlead = c.take<int>(g.cap); g.leader_of = c.take<int>(nn);
    g.scal = c.take<int>(64); g.dense = c.take<int>(nn);
    g.sxyz = c.take<float>(nn * 3); g.cbox = c.take<float>(nn * 6);
    g.temp_bytes = d3_sort_pairs_temp_bytes(n);
    const size_t sb = d3_scan_temp_bytes(n);
    if (sb > g.temp_bytes) g.temp_bytes = sb;
    g.temp = c.take<char>(g.temp_bytes);
}

__device__ __forceinline__ unsigned long long bqg_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
// cell coordinate of x: clamped (a monotone, non-expanding map keeps neighbours within +-1 cell); NaN lands in cell 0 and
// fails every distance test
__device__ __forceinline__ int bqg_cell(float x, float inv) {
    const float f = floorf(__fmul_rn(x, inv));
    return (f >= (float)(BQG_BIAS - 2)) ? BQG_BIAS - 2 : (f <= (float)(-BQG_BIAS + 2)) ? -BQG_BIAS + 2 : (f == f ? (int)f : 0);
}
__device__ __forceinline__ unsigned long long bqg_pack(int b, int cx, int cy, int cz) {
    return ((unsigned long long)(unsigned)(b & 0x7FFFF) << 45) | ((unsigned long long)(unsigned)(cx + BQG_BIAS) << 30) |
           ((unsigned long long)(unsigned)(cy + BQG_BIAS) << 15) | (unsigned long long)(unsigned)(cz + BQG_BIAS);
}
// slot of `key` (its start / count in `out`), or -1.  The first probe's slot and its leader entry are loaded together.
__device__ __forceinline__ int bqg_find(const BqSlot *__restrict__ tbl, size_t mask, unsigned long long key, BqSlot &out) {
    size_t slot = bqg_hash(key) & mask;
    for (;;) {
        const int4 raw = *(const int4 *)&tbl[slot];
        const unsigned long long k = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned)raw.x;
        if (k == key) { out.key = k; out.start = raw.z; out.count = raw.w; return (int)slot; }
        if (k == BQG_EMPTY) return -1;
        slot = (slot + 1) & mask;
    }
}

__global__ void bqg_key_kernel(const float *__restrict__ xyz, const int *__restrict__ batch_idxs, int n, float inv,
                               unsigned long long *key, int *pid, BqSlot *tbl, int *tlead, size_t cap, int *scal) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { tbl[i].key = BQG_EMPTY; tbl[i].start = 0; tbl[i].count = 0; tlead[i] = -1; }
    if (i < 8) scal[i] = 0;
    if (i >= (size_t)n) return;
    key[i] = bqg_pack(batch_idxs[i], bqg_cell(xyz[i * 3 + 0], inv), bqg_cell(xyz[i * 3 + 1], inv), bqg_cell(xyz[i * 3 + 2], inv));
    pid[i] = (int)i;
}
// Round 5: the points are sorted by the HASH SLOT of their cell instead of by the 64-bit cell key.  The cell order is irrelevant
// (a cell is found through the table), only the grouping and the ascending point order inside a cell matter -- a stable sort by any
// injective image of the key gives both, and the slot number is one with log2(cap) <= 22 significant bits in a 32-bit word: 3 radix
// passes over 8-byte pairs instead of 8 over 12-byte pairs (35 -> ~12 library launches, ~270 -> ~100 us per clustering branch).
// Every point claims / finds its cell's slot here; equal neighbours in a wave (collapsed instances: tens of thousands of points in
// one cell) send one lane to the table.
__global__ void bqg_slot_kernel(const unsigned long long *__restrict__ key, int n, BqSlot *tbl, size_t mask, int *slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)d3_lane();
    const bool live = i < n;
    const unsigned long long k = live ? key[i] : BQG_EMPTY;
    const unsigned long long kp = __shfl_up(k, 1);
    const bool lead = live && (lane == 0 || k != kp);
    int slot = -1;
    if (lead) {
        size_t sl = bqg_hash(k) & mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&tbl[sl].key, BQG_EMPTY, k);
            if (prev == BQG_EMPTY || prev == k) break;
            sl = (sl + 1) & mask;
        }
        slot = (int)sl;
    }
    // a follower takes the slot of the nearest leader below it
    const unsigned long long leads = __ballot(lead);
    const unsigned long long below = leads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    const int src = below ? 63 - (int)__builtin_clzll(below) : lane;
    slot = __shfl(slot, src);
    if (live) slot_of[i] = slot;
}

__global__ void bqg_head_kernel(const int *__restrict__ skey, const int *__restrict__ sidx,
                                const float *__restrict__ xyz, int n, int *head, float *sxyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
    const int p = sidx[i];
    sxyz[i * 3 + 0] = xyz[p * 3 + 0]; sxyz[i * 3 + 1] = xyz[p * 3 + 1]; sxyz[i * 3 + 2] = xyz[p * 3 + 2];
}
__global__ void bqg_cells_kernel(const int *__restrict__ skey, const int *__restrict__ head,
                                 const int *__restrict__ rid, int n, int *cstart, int *cslot, BqSlot *tbl, size_t mask, int *scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) { const int nc = rid[i] + head[i]; scal[0] = nc; cstart[nc] = n; }
    if (!head[i]) return;
    const int c = rid[i];
    cstart[c] = i;
    const int slot = skey[i];          // (the sort key IS the cell's slot: bqg_slot_kernel put the key there)
    tbl[slot].start = i; cslot[c] = slot;
    (void)mask;
}
// point count into the cell's hash slot, bounding box of its points (stored at the cell's first sorted position).
// Round 5: EIGHT lanes per cell (a surface's cells hold ~4 points: a wave per cell idled 60 lanes and the launch covered n waves for
// ~n / 4 cells: 104 -> 29 us); a cell of more than 64 points (a collapsed instance: tens of thousands) is then taken by the whole wave.
// (The same treatment of bqg_clique_kernel -- 32 lanes per cell, one probe each -- measured no gain: 94 -> 104 us; not kept.)
#define BQG_CB 8
__global__ __launch_bounds__(256) void bqg_cellbox_kernel(const int *__restrict__ cstart, const int *__restrict__ cslot,
                                                         const int *__restrict__ scal, const float *__restrict__ sxyz,
                                                         BqSlot *tbl, float *cbox) {
    const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = d3_lane(), sub = lane & (BQG_CB - 1);
    const int c = gid / BQG_CB, ncell = scal[0];
    if ((gid - lane) / BQG_CB >= ncell) return;               // (whole waves)
    const bool live = c < ncell;
    const int s0 = live ? cstart[c] : 0, s1 = live ? cstart[c + 1] : 0;
    const bool big = s1 - s0 > 64;
    if (live && !big) {
        float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
        for (int i = s0 + sub; i < s1; i += BQG_CB) {
            const float a = sxyz[i * 3 + 0], b = sxyz[i * 3 + 1], d = sxyz[i * 3 + 2];
            x = fminf(x, a); X = fmaxf(X, a); y = fminf(y, b); Y = fmaxf(Y, b); z = fminf(z, d); Z = fmaxf(Z, d);
        }
#pragma unroll
        for (int o = BQG_CB / 2; o > 0; o >>= 1) {
            x = fminf(x, __shfl_xor(x, o, BQG_CB)); y = fminf(y, __shfl_xor(y, o, BQG_CB)); z = fminf(z, __shfl_xor(z, o, BQG_CB));
            X = fmaxf(X, __shfl_xor(X, o, BQG_CB)); Y = fmaxf(Y, __shfl_xor(Y, o, BQG_CB)); Z = fmaxf(Z, __shfl_xor(Z, o, BQG_CB));
        }
        if (sub == 0) {
            float *o = cbox + (size_t)s0 * 6; o[0] = x; o[1] = y; o[2] = z; o[3] = X; o[4] = Y; o[5] = Z;
            tbl[cslot[c]].count = s1 - s0;
        }
    }
    unsigned long long todo = __ballot(live && big && sub == 0);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int b0 = __shfl(s0, src), b1 = __shfl(s1, src), bc = __shfl(c, src);
        float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
        for (int i = b0 + lane; i < b1; i += 64) {
            const float a = sxyz[i * 3 + 0], b = sxyz[i * 3 + 1], d = sxyz[i * 3 + 2];
            x = fminf(x, a); X = fmaxf(X, a); y = fminf(y, b); Y = fmaxf(Y, b); z = fminf(z, d); Z = fmaxf(Z, d);
        }
        x = wave_min(x); y = wave_min(y); z = wave_min(z); X = wave_max(X); Y = wave_max(Y); Z = wave_max(Z);
        if (lane == 0) {
            float *o = cbox + (size_t)b0 * 6; o[0] = x; o[1] = y; o[2] = z; o[3] = X; o[4] = Y; o[5] = Z;
            tbl[cslot[bc]].count = b1 - b0;
        }
    }
}
// one thread per cell: candidates and bounding box of its 27-cell neighbourhood -> leader (smallest member) of a clique cell
__global__ void bqg_clique_kernel(const int *__restrict__ skey, const int *__restrict__ sidx,
                                  const int *__restrict__ cstart, const int *__restrict__ cslot, const int *__restrict__ scal,
                                  const BqSlot *__restrict__ tbl, size_t mask, const float *__restrict__ cbox, float radius2,
                                  int *tlead) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= scal[0]) return;
    const int s0 = cstart[c], own_n = cstart[c + 1] - s0;
    if (own_n < 2) return;
    const unsigned long long key = tbl[skey[s0]].key;
    const unsigned long long kb = key & ~((1ull << 45) - 1);
    const int cx = (int)((key >> 30) & 0x7FFF), cy = (int)((key >> 15) & 0x7FFF), cz = (int)(key & 0x7FFF);
    int T = 0;
    float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
    for (int j = 0; j < 27; j++) {
        const int nx = cx + j / 9 - 1, ny = cy + (j / 3) % 3 - 1, nz = cz + j % 3 - 1;
        if (nx < 0 || ny < 0 || nz < 0 || nx > 0x7FFF || ny > 0x7FFF || nz > 0x7FFF) continue;
        BqSlot sl;
        if (bqg_find(tbl, mask, kb | ((unsigned long long)nx << 30) | ((unsigned long long)ny << 15) | (unsigned long long)nz, sl) < 0) continue;
        T += sl.count;
        const float *o = cbox + (size_t)sl.start * 6;
        x = fminf(x, o[0]); y = fminf(y, o[1]); z = fminf(z, o[2]); X = fmaxf(X, o[3]); Y = fmaxf(Y, o[4]); Z = fmaxf(Z, o[5]);
    }
    // every pair inside the box is closer than the diagonal; 1e-5 covers the roundings of both this expression and the
    // reference's distance expression (differences of nearby fp32 numbers are exact, the squares and sums round at 2^-24)
    const float ex = X - x, ey = Y - y, ez = Z - z;
    const float diag2 = ex * ex + ey * ey + ez * ez;
    if (T > 64 && diag2 * 1.00001f < radius2 && diag2 == diag2) tlead[cslot[c]] = sidx[s0];
}

__device__ __forceinline__ int bqg_bitonic64(int v, int lane) {   // ascending across the 64 lanes
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}
// ascending bitonic sort of buf[0..P) (P a power of two >= 64) by ONE wave
__device__ __forceinline__ void bqg_sort_lds(int *buf, int P, int lane) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const int a = buf[i], b = buf[p];
                if ((a > b) == ((i & k) == 0)) { buf[i] = b; buf[p] = a; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// the 27 probes of query q (lane j < 27 owns neighbour cell j): run start / length per lane, their exclusive prefix in LDS,
// the total T, and the leader entry of the query's own cell
struct BqProbe { int T, leader; float ox, oy, oz; };
__device__ __forceinline__ BqProbe bqg_probe(const float *__restrict__ xyz, const int *__restrict__ batch_idxs, int q, float inv,
                                             const BqSlot *__restrict__ tbl, const int *__restrict__ tlead, size_t mask,
                                             int lane, int *preS, int *cstS) {
    BqProbe r;
    r.ox = xyz[q * 3 + 0]; r.oy = xyz[q * 3 + 1]; r.oz = xyz[q * 3 + 2];
    const int b = batch_idxs[q];
    const int cx = bqg_cell(r.ox, inv) + BQG_BIAS, cy = bqg_cell(r.oy, inv) + BQG_BIAS, cz = bqg_cell(r.oz, inv) + BQG_BIAS;
    int rs = 0, rn = 0, ld = -1;
    if (lane < 27) {
        const int nx = cx + lane / 9 - 1, ny = cy + (lane / 3) % 3 - 1, nz = cz + lane % 3 - 1;
        if (nx >= 0 && ny >= 0 && nz >= 0 && nx <= 0x7FFF && ny <= 0x7FFF && nz <= 0x7FFF) {
            const unsigned long long nk = ((unsigned long long)(unsigned)(b & 0x7FFFF) << 45) | ((unsigned long long)nx << 30) |
                                          ((unsigned long long)ny << 15) | (unsigned long long)nz;
            // first probe: slot and leader entry requested together (one round trip when the key sits in its home slot)
            size_t slot = bqg_hash(nk) & mask;
            int4 raw = *(const int4 *)&tbl[slot];
            int l0 = tlead[slot];
            for (;;) {
                const unsigned long long k = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned)raw.x;
                if (k == nk) { rs = raw.z; rn = raw.w; ld = l0; break; }
                if (k == BQG_EMPTY) break;
                slot = (slot + 1) & mask;
                raw = *(const int4 *)&tbl[slot];
                l0 = tlead[slot];
            }
        }
    }
    r.leader = __shfl(ld, 13);
    int pre = rn;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(pre, o); if (lane >= o) pre += t; }
    r.T = __shfl(pre, 26);
    pre -= rn;
    if (lane < 27) { preS[lane] = pre; cstS[lane] = rs; }
    if (lane >= 27 && lane < 32) { preS[lane] = 0x7FFFFFFF; cstS[lane] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return r;
}
// candidate t of the flattened candidate list -> (point index, hit)
__device__ __forceinline__ bool bqg_test(const BqProbe &r, int t, const int *preS, const int *cstS, const int *__restrict__ sidx,
                                         const float *__restrict__ sxyz, float radius2, int &k) {
    const bool live = t < r.T;
    int j = 0;   // run of candidate t: the last j with pre[j] <= t (empty runs share their successor's prefix and are skipped)
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) if (j + s < 27 && preS[j + s] <= t) j += s;
    const int pos = live ? cstS[j] + (t - preS[j]) : 0;
    k = sidx[pos];
    const float x = sxyz[pos * 3 + 0], y = sxyz[pos * 3 + 1], z = sxyz[pos * 3 + 2];
    const float dx = __fsub_rn(r.ox, x), dy = __fsub_rn(r.oy, y), dz = __fsub_rn(r.oz, z);
    // ((dx*dx + dy*dy) + dz*dz), every operation rounded separately (bfs_cluster.cu:36)
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return live && d2 < radius2;
}

// one query on one wave: followers of a clique leader and queries with <= 64 candidates finish here, the others are queued for the
// dense pass (which needs an 8 KB hit buffer per wave -- kept out of this kernel's occupancy)
__device__ __forceinline__ void bqg_query_one(int q, const float *__restrict__ xyz, const int *__restrict__ batch_idxs, float radius, float inv,
                                              const int *__restrict__ sidx, const float *__restrict__ sxyz, const BqSlot *__restrict__ tbl,
                                              const int *__restrict__ tlead, size_t mask, int *__restrict__ leader_of,
                                              int *__restrict__ len_out, int *__restrict__ idx, int *dense, int *scal, int lane, int *preS, int *cstS) {
    const BqProbe r = bqg_probe(xyz, batch_idxs, q, inv, tbl, tlead, mask, lane, preS, cstS);
    if (r.leader >= 0 && r.leader != q) { if (lane == 0) leader_of[q] = r.leader; return; }   // shares the leader's list
    if (lane == 0) leader_of[q] = q;
    if (r.T > 64) { if (lane == 0) dense[atomicAdd(&scal[1], 1)] = q; return; }
    int k;
    const bool hit = bqg_test(r, lane, preS, cstS, sidx, sxyz, __fmul_rn(radius, radius), k);
    const int v = bqg_bitonic64(hit ? k : 0x7FFFFFFF, lane);
    const int cnt = (int)__popcll(__ballot(hit));
    if (lane < cnt) idx[(long long)q * BQ_CAP + lane] = v;
    if (lane == 0) len_out[q] = cnt;
}
// sparse pass, one wave per query (rounds 3 - 4; D3_BQ_HALF=0)
__global__ __launch_bounds__(256) void bqg_query_kernel(const float *__restrict__ xyz, const int *__restrict__ batch_idxs, int n,
                                                       float radius, float inv, const int *__restrict__ sidx,
                                                       const float *__restrict__ sxyz, const BqSlot *__restrict__ tbl,
                                                       const int *__restrict__ tlead, size_t mask, int *__restrict__ leader_of,
                                                       int *__restrict__ len_out, int *__restrict__ idx, int *dense, int *scal) {
    __shared__ int preS[4][32], cstS[4][32];
    const int q = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (q >= n) return;                      // (whole waves: no workgroup barrier below)
    const int lane = d3_lane(), wave = (int)(threadIdx.x >> 6);
    bqg_query_one(q, xyz, batch_idxs, radius, inv, sidx, sxyz, tbl, tlead, mask, leader_of, len_out, idx, dense, scal, lane, preS[wave], cstS[wave]);
}
// Round 5 -- TWO queries per wave.  The wave-per-query kernel is bound by waves x latency (600 k waves of four dependent round trips
// at full occupancy: ~390 us for the 4-scene batch) and a query of a surface has ~25 candidates in its 27 cells: 32 lanes hold the 27
// probes, up to 32 candidates and a 32-lane bitonic network.  A half owns lanes [32 h, 32 h + 32); every cross-lane step is
// width-limited to it.  33..64 candidates: a second element per lane; more -> the dense queue.  Same lists, same order.
__device__ __forceinline__ int bqg_bitonic32(int v, int sub) {     // ascending across the 32 lanes of a half
#pragma unroll
    for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 32);
            const bool up = (sub & k) == 0, lower = (sub & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}
__global__ __launch_bounds__(256) void bqg_query32_kernel(const float *__restrict__ xyz, const int *__restrict__ batch_idxs, int n,
                                                         float radius, float inv, const int *__restrict__ sidx,
                                                         const float *__restrict__ sxyz, const BqSlot *__restrict__ tbl,
                                                         const int *__restrict__ tlead, size_t mask, int *__restrict__ leader_of,
                                                         int *__restrict__ len_out, int *__restrict__ idx, int *dense, int *scal) {
    __shared__ int preS[8][32], cstS[8][32];
    const int gw = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (gw * 2 >= n) return;                 // (whole waves)
    const int lane = d3_lane(), half = lane >> 5, sub = lane & 31, hs = (int)(threadIdx.x >> 6) * 2 + half;
    const int q = gw * 2 + half;
    const bool live = q < n;
    const int qq = live ? q : n - 1;
    // ---- probe (bqg_probe on 32 lanes)
    BqProbe r;
    r.ox = xyz[qq * 3 + 0]; r.oy = xyz[qq * 3 + 1]; r.oz = xyz[qq * 3 + 2];
    const int b = batch_idxs[qq];
    const int cx = bqg_cell(r.ox, inv) + BQG_BIAS, cy = bqg_cell(r.oy, inv) + BQG_BIAS, cz = bqg_cell(r.oz, inv) + BQG_BIAS;
    int rs = 0, rn = 0, ld = -1;
    if (sub < 27) {
        const int nx = cx + sub / 9 - 1, ny = cy + (sub / 3) % 3 - 1, nz = cz + sub % 3 - 1;
        if (nx >= 0 && ny >= 0 && nz >= 0 && nx <= 0x7FFF && ny <= 0x7FFF && nz <= 0x7FFF) {
            const unsigned long long nk = ((unsigned long long)(unsigned)(b & 0x7FFFF) << 45) | ((unsigned long long)nx << 30) |
                                          ((unsigned long long)ny << 15) | (unsigned long long)nz;
            size_t slot = bqg_hash(nk) & mask;
            int4 raw = *(const int4 *)&tbl[slot];
            int l0 = tlead[slot];
            for (;;) {
                const unsigned long long k = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned)raw.x;
                if (k == nk) { rs = raw.z; rn = raw.w; ld = l0; break; }
                if (k == BQG_EMPTY) break;
                slot = (slot + 1) & mask;
                raw = *(const int4 *)&tbl[slot];
                l0 = tlead[slot];
            }
        }
    }
    r.leader = __shfl(ld, 13, 32);
    int pre = rn;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(pre, o, 32); if (sub >= o) pre += t; }
    r.T = __shfl(pre, 26, 32);
    pre -= rn;
    preS[hs][sub] = sub < 27 ? pre : 0x7FFFFFFF;
    cstS[hs][sub] = sub < 27 ? rs : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- classify
    const bool follower = r.leader >= 0 && r.leader != q;
    if (live && sub == 0) {
        leader_of[q] = follower ? r.leader : q;
        if (!follower && r.T > 64) dense[atomicAdd(&scal[1], 1)] = q;
    }
    const bool active = live && !follower && r.T <= 64;
    // (a wave of followers / queued queries is done here -- and an inactive half must not run the test either: its 32 lanes would all
    // read candidate 0, and a collapsed instance is tens of thousands of such halves hammering ONE cache line's L2 channel)
    if (!__any(active)) return;
    const float radius2 = __fmul_rn(radius, radius);
    // ---- test, order, store.  Candidates 32..63 of a half are a second element per lane (a queue for them -- one atomic per
    // query on one counter -- cost more than the whole kernel: 40 % of a surface's queries have 33..64 candidates)
    int k0 = 0, k1 = 0;
    bool h0 = false, h1 = false;
    if (active) h0 = bqg_test(r, sub, preS[hs], cstS[hs], sidx, sxyz, radius2, k0);
    const bool two = __any(active && r.T > 32);          // wave-uniform
    if (two && active && r.T > 32) h1 = bqg_test(r, sub + 32, preS[hs], cstS[hs], sidx, sxyz, radius2, k1);
    int v0 = h0 ? k0 : 0x7FFFFFFF, v1 = h1 ? k1 : 0x7FFFFFFF;
    const unsigned long long b0 = __ballot(h0);
    int cnt = (int)__popc((unsigned int)(b0 >> (half * 32)));
    if (!two) {
        v0 = bqg_bitonic32(v0, sub);
    } else {
        const unsigned long long b1 = __ballot(h1);
        cnt += (int)__popc((unsigned int)(b1 >> (half * 32)));
        // bitonic network over the 64 elements e = 32 r + sub of a half (r = register): partner e ^ j is the other lane for j < 32,
        // the other register for j = 32
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
            for (int j = kk >> 1; j > 0; j >>= 1) {
                if (j == 32) {
                    const int lo = min(v0, v1), hi = max(v0, v1);      // (kk = 64: ascending everywhere)
                    v0 = lo; v1 = hi;
                } else {
                    const int o0 = __shfl_xor(v0, j, 32), o1 = __shfl_xor(v1, j, 32);
                    const bool lower = (sub & j) == 0;
                    const bool up0 = (sub & kk) == 0, up1 = ((sub + 32) & kk) == 0;
                    v0 = (lower == up0) ? min(v0, o0) : max(v0, o0);
                    v1 = (lower == up1) ? min(v1, o1) : max(v1, o1);
                }
            }
        }
    }
    if (active) {
        if (sub < cnt) idx[(long long)q * BQ_CAP + sub] = v0;
        if (sub + 32 < cnt) idx[(long long)q * BQ_CAP + sub + 32] = v1;
        if (sub == 0) len_out[q] = cnt;
    }
}
// dense pass: persistent waves over the queued queries
__global__ __launch_bounds__(256) void bqg_dense_kernel(const float *__restrict__ xyz, const int *__restrict__ batch_idxs,
                                                       float radius, float inv, const int *__restrict__ sidx,
                                                       const float *__restrict__ sxyz, const BqSlot *__restrict__ tbl,
                                                       const int *__restrict__ tlead, size_t mask, int *__restrict__ len_out,
                                                       int *__restrict__ idx, const int *__restrict__ dense,
                                                       const int *__restrict__ scal) {
    __shared__ int bufS[4][BQG_BUF];
    __shared__ int preS[4][32], cstS[4][32];
    const int lane = d3_lane(), wave = (int)(threadIdx.x >> 6);
    const unsigned long long lt = d3_lanemask_lt();
    const float radius2 = __fmul_rn(radius, radius);
    int *buf = bufS[wave];
    const int nd = scal[1], nw = (int)(gridDim.x * 4);
    for (int w = (int)(blockIdx.x * 4) + wave; w < nd; w += nw) {
        const int q = dense[w];
        const BqProbe r = bqg_probe(xyz, batch_idxs, q, inv, tbl, tlead, mask, lane, preS[wave], cstS[wave]);
        int cnt = 0, thr = 0x7FFFFFFF;
        for (int t0 = 0; t0 < r.T; t0 += 64) {
            int k;
            const bool hit = bqg_test(r, t0 + lane, preS[wave], cstS[wave], sidx, sxyz, radius2, k) && k < thr;
            const unsigned long long hm = __ballot(hit);
            if (hit) buf[cnt + (int)__popcll(hm & lt)] = k;
            cnt += (int)__popcll(hm);
            if (cnt > BQG_BUF - 64) {   // (wave-uniform) cut back to the 1000 smallest so far; later candidates must beat the 1000th
                for (int e = cnt + lane; e < BQG_BUF; e += 64) buf[e] = 0x7FFFFFFF;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bqg_sort_lds(buf, BQG_BUF, lane);
                cnt = BQ_CAP;
                thr = buf[BQ_CAP - 1];
            }
        }
        int P = 64;
        while (P < cnt) P <<= 1;
        for (int e = cnt + lane; e < P; e += 64) buf[e] = 0x7FFFFFFF;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bqg_sort_lds(buf, P, lane);
        if (cnt > BQ_CAP) cnt = BQ_CAP;
        const long long base = (long long)q * BQ_CAP;
        for (int e = lane; e < cnt; e += 64) idx[base + e] = buf[e];
        if (lane == 0) len_out[q] = cnt;
        __builtin_amdgcn_wave_barrier();     // buf / preS are rewritten by the next query
    }
}
__global__ void bqg_pack_kernel(const int *__restrict__ len, const int *__restrict__ leader_of, int *start_len, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = leader_of[i];
    start_len[i * 2 + 0] = l * BQ_CAP;
    start_len[i * 2 + 1] = len[l];
}

static int bqg_padded(const float *xyz, const int *batch_idxs, int n, float radius, int *start_len, BqWs &w, BqGrid &g,
                      int *idx_padded, hipStream_t s) {
    const float inv = 1.0f / (radius * 1.001f);
    const size_t span = g.cap > (size_t)n ? g.cap : (size_t)n;
    bqg_key_kernel<<<(int)((span + 255) / 256), 256, 0, s>>>(xyz, batch_idxs, n, inv, g.key, g.pid, g.tbl, g.tlead, g.cap, g.scal);
    bqg_slot_kernel<<<(n + 255) / 256, 256, 0, s>>>(g.key, n, g.tbl, g.cap - 1, g.slot32);
    int cap_bits = 1;
    while (((size_t)1 << cap_bits) < g.cap) cap_bits++;
    int rc = d3_sort_pairs_i32(g.slot32, g.skey, g.pid, g.sidx, n, cap_bits, g.temp, g.temp_bytes, s);
    if (rc) return rc;
    bqg_head_kernel<<<(n + 255) / 256, 256, 0, s>>>(g.skey, g.sidx, xyz, n, g.head, g.sxyz);
    rc = d3_exclusive_scan_i32(g.head, g.rid, n, g.temp, g.temp_bytes, s);
    if (rc) return rc;
    bqg_cells_kernel<<<(n + 255) / 256, 256, 0, s>>>(g.skey, g.head, g.rid, n, g.cstart, g.cslot, g.tbl, g.cap - 1, g.scal);
    bqg_cellbox_kernel<<<(int)(((long long)n * BQG_CB + 255) / 256), 256, 0, s>>>(g.cstart, g.cslot, g.scal, g.sxyz, g.tbl, g.cbox);          // (<= n cells)
    bqg_clique_kernel<<<(n + 255) / 256, 256, 0, s>>>(g.skey, g.sidx, g.cstart, g.cslot, g.scal, g.tbl, g.cap - 1, g.cbox,
                                                     radius * radius, g.tlead);
    if (d3_tune(D3T_BQ_HALF) != 0) {
        bqg_query32_kernel<<<(n + 7) / 8, 256, 0, s>>>(xyz, batch_idxs, n, radius, inv, g.sidx, g.sxyz, g.tbl, g.tlead, g.cap - 1,
                                                      g.leader_of, w.len, idx_padded, g.dense, g.scal);
    } else
    bqg_query_kernel<<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, n, radius, inv, g.sidx, g.sxyz, g.tbl, g.tlead, g.cap - 1,
                                                g.leader_of, w.len, idx_padded, g.dense, g.scal);
    // dense pass: as many workgroups as the chip holds (LDS: 5 per CU), each wave walks the queue
    int nblk = (n + 3) / 4;
    if (nblk > 1280) nblk = 1280;
    bqg_dense_kernel<<<nblk, 256, 0, s>>>(xyz, batch_idxs, radius, inv, g.sidx, g.sxyz, g.tbl, g.tlead, g.cap - 1, w.len,
                                         idx_padded, g.dense, g.scal);
    bqg_pack_kernel<<<(n + 255) / 256, 256, 0, s>>>(w.len, g.leader_of, start_len, n);
    D3_LAUNCH_CHECK();
    return 0;
}

// Padded (sync-free) form: every point owns a fixed slot of d3_ballquery_cap() entries, start_len[q] = (q * cap, len).
// No scan, no compaction, no host round trip for nActive: the consumers (d3_bfs_cluster_*) only ever index
// idx[start + e], e < len, so the padded layout is a valid (idx, start_len) pair for them.  idx_padded: n * cap ints.
__global__ void bq_pack_padded_kernel(const int *len, int *start_len, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start_len[i * 2 + 0] = i * BQ_CAP;
    start_len[i * 2 + 1] = len[i];
}
extern "C" int d3_ballquery_cap(void) { return BQ_CAP; }
extern "C" int d3_ballquery_padded(const float *xyz, const int *batch_idxs, const int *batch_offsets, int n, float radius,
                                   int *start_len, void *ws, size_t ws_bytes, int *idx_padded, void *stream) {
    D3_CLEAR();
    if (n <= 0) return 0;
    if ((long long)n * BQ_CAP > 0x7FFFFFFFLL) return D3_ERR_ARG;
    BqWs w;
    BqGrid g;
    if (!bq_carve(ws, ws_bytes, n, w, &g)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    if (d3_tune(D3T_BQ_GRID) != 0 && radius > 0.f) return bqg_padded(xyz, batch_idxs, n, radius, start_len, w, g, idx_padded, s);
    int rc = bq_boxes(xyz, n, w, s);
    if (rc) return rc;
    bq_scan_kernel<2><<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, batch_offsets, n, radius, w.clo, w.chi, w.slo, w.shi,
                                                 w.nchunks, w.len, nullptr, idx_padded, 0);
    bq_pack_padded_kernel<<<(n + 255) / 256, 256, 0, s>>>(w.len, start_len, n);
    D3_LAUNCH_CHECK();
    return 0;
}

