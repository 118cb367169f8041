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

static bool bq_carve(void *ws, size_t ws_bytes, int n, BqWs &w) {
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
    D3_LAUNCH_CHECK();
    D3_CHECK(hipMemcpyAsync(nActive_host, w.total, sizeof(int), hipMemcpyDeviceToHost, s));
    D3_CHECK(hipStreamSynchronize(s));
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
    if (!bq_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int rc = bq_boxes(xyz, n, w, s);
    if (rc) return rc;
    bq_scan_kernel<2><<<(n + 3) / 4, 256, 0, s>>>(xyz, batch_idxs, batch_offsets, n, radius, w.clo, w.chi, w.slo, w.shi,
                                                 w.nchunks, w.len, nullptr, idx_padded, 0);
    bq_pack_padded_kernel<<<(n + 255) / 256, 256, 0, s>>>(w.len, start_len, n);
    D3_LAUNCH_CHECK();
    return 0;
}

