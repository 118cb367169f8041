// voxelize.hip -- point<->voxel pooling and the point->voxel index build (gfx950).
//
// Replaces PG_OP.voxelize_fp/bp, point_recover_fp/bp (reference:
// lib/pointgroup_ops/src/voxelize/voxelize.cu:10-53, voxelize.cpp:155-202) and the CPU
// hash pass PG_OP.voxelize_idx (voxelize.cpp:10-152), which the reference runs single-threaded
// on the host inside the forward (model/pointgroup.py:166-169).
//
// Pooling kernels: one thread per (voxel row, channel) element of the flat (M*C) output, so
// stores are fully coalesced and each gathered point row is read as a contiguous C-float run.
// The sum over a voxel's points is kept in rule order with separately rounded mul and add
// (the reference does atomicAdd(out, multiplier*inp) with one thread per channel, i.e. the
// same serial order), so the result is bit-exact against the oracle.
// HBM bound: bytes = 4*N*C (points) + 4*M*(maxActive+1) (rules) + 4*M*C (voxels).
#include "common.h"

// ------------------------------------------------------------------------- pooling kernels
__global__ void voxelize_fp_kernel(const float *__restrict__ feats, float *__restrict__ out,
                                   const int *__restrict__ rules, long long total, int maxActive, int nPlanes,
                                   bool average) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int row = (int)(e / nPlanes), plane = (int)(e % nPlanes);
    const int *r = rules + (long long)row * (maxActive + 1);
    // A voxel holds one to a few points.  The count, the first four point ids and the accumulator are requested together,
    // then the four gathers together (branch-free: selects instead of a loop whose every step waits for two dependent
    // loads); the adds keep the reference's serial order.
    const int nActive = r[0];
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; j++) id[j] = r[j < maxActive ? 1 + j : 0];
    float acc = out[e];
    const float multiplier = (average && nActive > 0) ? __fdiv_rn(1.0f, (float)nActive) : 1.0f;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = feats[(long long)(j < nActive ? id[j] : 0) * nPlanes + plane];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < nActive) acc = __fadd_rn(acc, __fmul_rn(multiplier, v[j]));
    // crowded voxels (cluster grids hold tens of points per voxel): eight ids, then their eight rows, per round trip
    for (int i0 = 5; i0 <= nActive; i0 += 8) {
        int idn[8];
        float vn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) idn[j] = r[i0 + j <= nActive ? i0 + j : 0];
#pragma unroll
        for (int j = 0; j < 8; j++) vn[j] = feats[(long long)(i0 + j <= nActive ? idn[j] : 0) * nPlanes + plane];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i0 + j <= nActive) acc = __fadd_rn(acc, __fmul_rn(multiplier, vn[j]));
    }
    out[e] = acc;
}

// The input pooling of PointGroup.feed (model/pointgroup.py:468-471): voxelization(cat(feats, locs), v2p_map) without the
// concatenated (N, Ca+Cb) copy and without the zero-filled accumulator read back (0.8 GB of traffic for the 4-scene batch):
// plane < Ca reads feats_a, else feats_b; the sum starts at +0 like the zero-initialised output of the reference's wrapper.
__global__ void voxelize_fp2_kernel(const float *__restrict__ fa, int Ca, const float *__restrict__ fb, int Cb, float *__restrict__ out,
                                    const int *__restrict__ rules, long long total, int maxActive, bool average) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int nPlanes = Ca + Cb;
    const int row = (int)(e / nPlanes), plane = (int)(e % nPlanes);
    const int *r = rules + (long long)row * (maxActive + 1);
    const bool ina = plane < Ca;
    const float *src = ina ? fa : fb;
    const int ld = ina ? Ca : Cb, col = ina ? plane : plane - Ca;
    const int nActive = r[0];
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; j++) id[j] = r[j < maxActive ? 1 + j : 0];
    float acc = 0.f;
    const float multiplier = (average && nActive > 0) ? __fdiv_rn(1.0f, (float)nActive) : 1.0f;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = src[(long long)(j < nActive ? id[j] : 0) * ld + col];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < nActive) acc = __fadd_rn(acc, __fmul_rn(multiplier, v[j]));
    for (int i0 = 5; i0 <= nActive; i0 += 8) {
        int idn[8];
        float vn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) idn[j] = r[i0 + j <= nActive ? i0 + j : 0];
#pragma unroll
        for (int j = 0; j < 8; j++) vn[j] = src[(long long)(i0 + j <= nActive ? idn[j] : 0) * ld + col];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i0 + j <= nActive) acc = __fadd_rn(acc, __fmul_rn(multiplier, vn[j]));
    }
    out[e] = acc;
}

// The same pooling with one WAVE per voxel, lanes along the channels: the rule row is read once per wave (the thread-per-element
// form pays a 64-bit division by 134 and re-reads the rule row per element), the <= 4 x 3 gathers of a lane are in flight
// together.  Same operations per output in the same order: bit-equal.
__global__ __launch_bounds__(256) void voxelize_fp2_rows_kernel(const float *__restrict__ fa, int Ca, const float *__restrict__ fb, int Cb,
                                                               float *__restrict__ out, const int *__restrict__ rules, int M, int maxActive,
                                                               bool average) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nPlanes = Ca + Cb;
    const int *r = rules + row * (maxActive + 1);
    const int rv = (lane <= maxActive && lane < 5) ? r[lane] : 0;
    const int nActive = __shfl(rv, 0);
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; j++) id[j] = __shfl(rv, 1 + j);
    const float multiplier = (average && nActive > 0) ? __fdiv_rn(1.0f, (float)nActive) : 1.0f;
    for (int p0 = 0; p0 < nPlanes; p0 += 192) {
        float v[3][4];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int plane = p0 + q * 64 + lane;
            const bool on = plane < nPlanes;
            const bool ina = plane < Ca || !on;   // lanes past the last plane read (and discard) element 0 of `fa`: `fb` may be empty (Cb == 0)
            const float *src = ina ? fa : fb;
            const int ld = ina ? Ca : Cb, col = on ? (ina ? plane : plane - Ca) : 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v[q][j] = src[(long long)((on && j < nActive) ? id[j] : 0) * ld + col];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int plane = p0 + q * 64 + lane;
            if (plane >= nPlanes) continue;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < nActive) acc = __fadd_rn(acc, __fmul_rn(multiplier, v[q][j]));
            if (nActive > 4) {       // crowded voxel: the remaining points one by one, in rule order
                const bool ina = plane < Ca;
                const float *src = ina ? fa : fb;
                const int ld = ina ? Ca : Cb, col = ina ? plane : plane - Ca;
                for (int i = 5; i <= nActive; i++) acc = __fadd_rn(acc, __fmul_rn(multiplier, src[(long long)r[i] * ld + col]));
            }
            out[row * nPlanes + plane] = acc;
        }
    }
}

// scatter: d_feats[r[i], plane] += multiplier * d_out[row, plane]   (voxelize.cu:35-53)
__global__ void voxelize_bp_kernel(const float *__restrict__ d_out, float *__restrict__ d_feats,
                                   const int *__restrict__ rules, long long total, int maxActive, int nPlanes,
                                   bool average) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int row = (int)(e / nPlanes), plane = (int)(e % nPlanes);
    const int *r = rules + (long long)row * (maxActive + 1);
    const int nActive = r[0];
    const float multiplier = (average && nActive > 0) ? __fdiv_rn(1.0f, (float)nActive) : 1.0f;
    const float g = __fmul_rn(multiplier, d_out[e]);
    for (int i0 = 1; i0 <= nActive; i0 += 8) {   // eight point ids per round trip (crowded cluster voxels), then the atomics
        int idn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) idn[j] = r[i0 + j <= nActive ? i0 + j : 0];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i0 + j <= nActive) atomicAdd(&d_feats[(long long)idn[j] * nPlanes + plane], g);
    }
}

static int launch_fp(const float *feats, float *out, const int *rules, int nActive, int maxActive, int nPlane,
                     bool average, void *stream) {
    long long total = (long long)nActive * nPlane;
    if (total <= 0) return 0;
    voxelize_fp_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(feats, out, rules, total, maxActive,
                                                                                nPlane, average);
    D3_LAUNCH_CHECK();
    return 0;
}
static int launch_bp(const float *d_out, float *d_feats, const int *rules, int nActive, int maxActive, int nPlane,
                     bool average, void *stream) {
    long long total = (long long)nActive * nPlane;
    if (total <= 0) return 0;
    voxelize_bp_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(d_out, d_feats, rules, total,
                                                                                maxActive, nPlane, average);
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_voxelize_fp(const float *feats, float *output_feats, const int *output_map, int mode, int nActive,
                              int maxActive, int nPlane, void *stream) {
    D3_CLEAR();
    return launch_fp(feats, output_feats, output_map, nActive, maxActive, nPlane, mode == 4, stream);
}
extern "C" int d3_voxelize_fp2(const float *feats_a, int Ca, const float *feats_b, int Cb, float *output_feats, const int *output_map,
                               int mode, int nActive, int maxActive, void *stream) {
    D3_CLEAR();
    if (Ca < 1 || Cb < 0) return D3_ERR_ARG;
    const long long total = (long long)nActive * (Ca + Cb);
    if (total <= 0) return 0;
    const int rows_form = d3_tune(D3T_VOX_ROWS);
    if (rows_form && Ca + Cb >= 48)      // wide rows: a wave per voxel
        voxelize_fp2_rows_kernel<<<(nActive + 3) / 4, 256, 0, d3_stream(stream)>>>(feats_a, Ca, feats_b, Cb, output_feats, output_map, nActive,
                                                                                  maxActive, mode == 4);
    else
        voxelize_fp2_kernel<<<(int)((total + 255) / 256), 256, 0, d3_stream(stream)>>>(feats_a, Ca, feats_b, Cb, output_feats, output_map, total,
                                                                                     maxActive, mode == 4);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_voxelize_bp(const float *d_output_feats, float *d_feats, const int *output_map, int mode,
                              int nActive, int maxActive, int nPlane, void *stream) {
    D3_CLEAR();
    return launch_bp(d_output_feats, d_feats, output_map, nActive, maxActive, nPlane, mode == 4, stream);
}
// point_recover_fp == voxelize_bp(average=false); point_recover_bp == voxelize_fp(average=false)
// (reference: voxelize.cpp:181-202)
extern "C" int d3_point_recover_fp(const float *feats, float *output_feats, const int *idx_map, int nActive,
                                   int maxActive, int nPlane, void *stream) {
    D3_CLEAR();
    return launch_bp(feats, output_feats, idx_map, nActive, maxActive, nPlane, false, stream);
}
extern "C" int d3_point_recover_bp(const float *d_output_feats, float *d_feats, const int *idx_map, int nActive,
                                   int maxActive, int nPlane, void *stream) {
    D3_CLEAR();
    return launch_fp(d_output_feats, d_feats, idx_map, nActive, maxActive, nPlane, false, stream);
}

// ------------------------------------------------------------------------------ voxelize_idx
// Device restatement of voxelize_idx<3>: voxel ids in FIRST-OCCURRENCE order of the points.
//   1. open-addressing hash insert of a packed 64-bit (batch,x,y,z) key; each slot keeps the
//      minimum point index that hit it (= the voxel's first point);
//   2. flag the first points, exclusive-scan the flags -> voxel id (ascending first point =
//      first-occurrence order, exactly what `nActive++` produces in voxelize.cpp:78,99);
//   3. p2v[i] = id of i's slot; per-voxel counts; maxActive = max count;
//   4. (fill) stable radix sort of (voxel id, point id) -> point lists in ascending point order
//      (= push_back order, voxelize.cpp:81,103), then the rule rows and the voxel coordinates
//      (coordinate row of the first listed point, voxelize.cpp:34-49).
#define VI_EMPTY 0xFFFFFFFFFFFFFFFFull

struct VoxIdxWs {
    unsigned long long *keys;  // cap
    int *first;                // cap   min point index per slot
    int *slot_vid;             // cap   voxel id per slot
    int *slot_of;              // n     slot of each point
    int *flag;                 // n     1 if point is the first of its voxel
    int *scan;                 // n     exclusive scan of flag
    int *cnt;                  // n     points per voxel (first M entries used)
    int *vstart;               // n     exclusive scan of cnt
    int *sorted_pts;           // n
    int *sorted_keys;          // n
    int *scalars;              // [0]=M-helper [1]=maxActive [2]=range error
    void *temp; size_t temp_bytes;
    size_t cap;
};

static size_t vi_cap(int n) { size_t c = 1024; while (c < (size_t)n * 2) c <<= 1; return c; }

static bool vi_carve(void *ws, size_t ws_bytes, int n, VoxIdxWs &w) {
    D3Carver c(ws, ws_bytes);
    size_t nn = (size_t)(n > 0 ? n : 1);
    w.cap = vi_cap(n);
    w.keys = c.take<unsigned long long>(w.cap);
    w.first = c.take<int>(w.cap);
    w.slot_vid = c.take<int>(w.cap);
    w.slot_of = c.take<int>(nn);
    w.flag = c.take<int>(nn);
    w.scan = c.take<int>(nn);
    w.cnt = c.take<int>(nn);
    w.vstart = c.take<int>(nn);
    w.sorted_pts = c.take<int>(nn);
    w.sorted_keys = c.take<int>(nn);
    w.scalars = c.take<int>(64);
    size_t t1 = d3_scan_temp_bytes(n), t2 = d3_sort_pairs_temp_bytes(n);
    w.temp_bytes = t1 > t2 ? t1 : t2;
    w.temp = c.take<char>(w.temp_bytes);
    return ws == nullptr ? false : c.ok();
}

extern "C" size_t d3_voxelize_idx_ws_bytes(int n) {
    VoxIdxWs w;
    D3Carver c(nullptr, 0);
    size_t nn = (size_t)(n > 0 ? n : 1);
    size_t cap = vi_cap(n);
    c.take<unsigned long long>(cap); c.take<int>(cap); c.take<int>(cap);
    for (int i = 0; i < 7; i++) c.take<int>(nn);
    c.take<int>(64);
    size_t t1 = d3_scan_temp_bytes(n), t2 = d3_sort_pairs_temp_bytes(n);
    c.take<char>(t1 > t2 ? t1 : t2);
    (void)w;
    return c.off + 256;
}

__device__ __forceinline__ unsigned long long vi_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// key layout: batch 19 bits | x 15 | y 15 | z 15 (x,y,z biased by 2^14)
__device__ __forceinline__ bool vi_pack(const int64_t *c, int ncols, unsigned long long &key) {
    int b = 0, x, y, z;
    if (ncols == 4) { b = (int)c[0]; x = (int)c[1]; y = (int)c[2]; z = (int)c[3]; }  // long -> Int truncation
    else { x = (int)c[0]; y = (int)c[1]; z = (int)c[2]; }
    const int B = 1 << 14;
    bool ok = (b >= 0 && b < (1 << 19)) && (x >= -B && x < B) && (y >= -B && y < B) && (z >= -B && z < B);
    key = ((unsigned long long)(unsigned)b << 45) | ((unsigned long long)(unsigned)(x + B) << 30) |
          ((unsigned long long)(unsigned)(y + B) << 15) | (unsigned long long)(unsigned)(z + B);
    return ok;
}

__global__ void vi_init_kernel(unsigned long long *keys, int *first, size_t cap, int *cnt, int n, int *scalars) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = VI_EMPTY; first[i] = 0x7FFFFFFF; }
    if (i < (size_t)n) cnt[i] = 0;
    if (i < 8) scalars[i] = 0;
}

__global__ void vi_insert_kernel(const int64_t *__restrict__ coords, int n, int ncols, unsigned long long *keys,
                                 int *first, size_t cap, int *slot_of, int *scalars) {
    // Round 5: points of one voxel often follow each other (cluster members arrive in BFS order, scene points in scan order): only the
    // FIRST lane of a run of equal keys inside the wave probes the table -- its index is the run's smallest, so first[] gets the same
    // minimum -- and hands the slot to the rest of the run by a shuffle (~4x fewer CAS / atomicMin pairs on the same cache lines).
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = i < n;
    unsigned long long key = VI_EMPTY;
    if (live && !vi_pack(coords + (size_t)i * ncols, ncols, key)) { scalars[2] = 1; key &= ~(1ull << 63); }
    const unsigned long long pk = __shfl_up(key, 1);
    const bool head = live && (lane == 0 || pk != key);
    const unsigned long long heads = __ballot(head);
    int myslot = 0;
    if (head) {
        size_t slot = vi_hash(key) & (cap - 1);
        bool done = false;
        for (size_t probe = 0; probe < cap; probe++) {
            unsigned long long prev = atomicCAS(&keys[slot], VI_EMPTY, key);
            if (prev == VI_EMPTY || prev == key) { atomicMin(&first[slot], i); myslot = (int)slot; done = true; break; }
            slot = (slot + 1) & (cap - 1);
        }
        if (!done) scalars[2] = 2;  // table full (cannot happen with cap >= 2n)
    }
    // the head of my run: the highest head lane at or below mine (a dead lane's key is VI_EMPTY, which no live key equals)
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int hl = below ? 63 - __builtin_clzll(below) : lane;
    myslot = __shfl(myslot, hl);
    if (live) slot_of[i] = myslot;
}

__global__ void vi_flag_kernel(const int *first, const int *slot_of, int *flag, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (first[slot_of[i]] == i) ? 1 : 0;
}
__global__ void vi_assign_kernel(const int *flag, const int *scan, const int *slot_of, int *slot_vid, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) slot_vid[slot_of[i]] = scan[i];
}
__global__ void vi_p2v_kernel(const int *slot_of, const int *slot_vid, int *input_map, int *cnt, int n,
                              int *scalars) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = i < n;
    int v = -1;
    if (live) { v = slot_vid[slot_of[i]]; input_map[i] = v; }
    // One NON-returning atomicAdd per distinct voxel of a run of equal neighbours (points of a voxel often follow each other);
    // the largest count is taken from the finished counters by vi_max_kernel.  (A returning atomic per distinct voxel of the
    // wave, looped over the wave's groups, cost a memory round trip per group: 64 groups deep on cluster-ordered points,
    // 144 us for 400 k points.)
    const int prev = __shfl_up(v, 1);
    const bool head = live && (lane == 0 || prev != v);
    const unsigned long long heads = __ballot(head), lives = __ballot(live);
    if (head) {
        const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1)) << (lane + 1);
        const int end = above ? (int)__builtin_ctzll(above) : 64;              // next run head (or the end of the wave)
        const unsigned long long span = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
        atomicAdd(&cnt[v], (int)__popcll(span & lives));
    }
    (void)scalars;
}
// maxActive = the largest voxel population: the maximum of cnt[0 .. M) once vi_p2v_kernel has counted and vi_total_kernel has written
// M (scalars[0]) -- one contiguous read per voxel (round 5: up to round 4 every voxel's FIRST POINT chased flag -> slot -> voxel id ->
// count, four dependent random reads per point: 92 us for 400 k points, on the critical path).  One word for the whole launch: a
// wave whose maximum is not above the word's current value (read at device scope) has nothing to add.
__global__ void vi_max_kernel(const int *__restrict__ cnt, int n, int *scalars) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int M = scalars[0];
    if ((int)(blockIdx.x * blockDim.x) >= M) return;
    int c = (i < M && i < n) ? cnt[i] : 0;
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o));
    if (lane == 0 && c > 0 && c > __hip_atomic_load(&scalars[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&scalars[1], c);
}
__global__ void vi_total_kernel(const int *flag, const int *scan, int n, int *scalars) {
    if (threadIdx.x == 0 && blockIdx.x == 0) scalars[0] = n > 0 ? scan[n - 1] + flag[n - 1] : 0;
}

extern "C" int d3_voxelize_idx_count(const int64_t *coords, int n, int ncols, int mode, int *input_map, void *ws,
                                     size_t ws_bytes, int *M_host, int *maxActive_host, void *stream) {
    D3_CLEAR();
    if (ncols != 3 && ncols != 4) return D3_ERR_ARG;
    if (mode < 0 || mode > 4) return D3_ERR_ARG;
    *M_host = 0; *maxActive_host = 1;
    if (n <= 0) return 0;
    VoxIdxWs w;
    if (!vi_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256;
    size_t initn = w.cap > (size_t)n ? w.cap : (size_t)n;
    vi_init_kernel<<<(int)((initn + T - 1) / T), T, 0, s>>>(w.keys, w.first, w.cap, w.cnt, n, w.scalars);
    const int nb = (n + T - 1) / T;
    vi_insert_kernel<<<nb, T, 0, s>>>(coords, n, ncols, w.keys, w.first, w.cap, w.slot_of, w.scalars);
    vi_flag_kernel<<<nb, T, 0, s>>>(w.first, w.slot_of, w.flag, n);
    int rc = d3_exclusive_scan_i32(w.flag, w.scan, n, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    vi_assign_kernel<<<nb, T, 0, s>>>(w.flag, w.scan, w.slot_of, w.slot_vid, n);
    vi_p2v_kernel<<<nb, T, 0, s>>>(w.slot_of, w.slot_vid, input_map, w.cnt, n, w.scalars);
    vi_total_kernel<<<1, 64, 0, s>>>(w.flag, w.scan, n, w.scalars);
    vi_max_kernel<<<nb, T, 0, s>>>(w.cnt, n, w.scalars);
    D3_LAUNCH_CHECK();
    int h[3];
    D3_CHECK(hipMemcpyAsync(h, w.scalars, sizeof(h), hipMemcpyDeviceToHost, s));
    D3_CHECK(hipStreamSynchronize(s));
    if (h[2] == 1) return D3_ERR_RANGE;
    if (h[2] == 2) return D3_ERR_OVERFLOW;
    *M_host = h[0];
    *maxActive_host = (mode == 3 || mode == 4) ? (h[1] > 1 ? h[1] : 1) : 1;  // voxelize.cpp:141-145
    return 0;
}

// The (M, maxActive + 1) rule table: a GROUP of G lanes per voxel row (G = the power of two that covers the row, at most 64), the
// group's lanes stride over the row -- the voxel's count and list start are read once per row, the point ids and the stores are
// contiguous.  (Round 5: one thread per ELEMENT with a 64-bit e / w, e % w per thread took 102 us for the 2 M elements of the
// cluster voxelisation -- on the critical path between the clustering and ScoreNet.)
__global__ __launch_bounds__(256) void vi_rules_kernel(int mode, const int *__restrict__ cnt, const int *__restrict__ vstart,
                                                       const int *__restrict__ sorted_pts, int *__restrict__ out_map, int M,
                                                       int maxActive, int gshift) {
    const int G = 1 << gshift, w = maxActive + 1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long v = t >> gshift;
    if (v >= M) return;
    const int j0 = (int)(t & (G - 1));
    const int c = cnt[v];
    const int *pts = sorted_pts + vstart[v];
    int *row = out_map + v * w;
    for (int j = j0; j < w; j += G) {
        int val;
        if (mode == 3 || mode == 4) val = (j == 0) ? c : (j - 1 < c ? pts[j - 1] : 0);  // zero padded (voxelize.cpp:151)
        else val = (j == 0) ? 1 : ((mode == 2) ? pts[c - 1] : pts[0]);  // mode 1: front(), mode 2: back() (:130-140)
        row[j] = val;
    }
}
// voxel coordinate = coordinate row of the first listed point (voxelize_outputmap, voxelize.cpp:34-49)
__global__ void vi_coords_kernel(const int64_t *__restrict__ coords, int ncols, int mode,
                                 const int *__restrict__ cnt, const int *__restrict__ vstart,
                                 const int *__restrict__ sorted_pts, int64_t *__restrict__ out_coords, int M) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * ncols) return;
    const int v = e / ncols, j = e % ncols;
    const int *pts = sorted_pts + vstart[v];
    const int firstListed = (mode == 2) ? pts[cnt[v] - 1] : pts[0];
    out_coords[e] = coords[(size_t)firstListed * ncols + j];
}

__global__ void vi_iota_kernel(int *a, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

extern "C" int d3_voxelize_idx_fill(const int64_t *coords, int n, int ncols, int mode, const int *input_map,
                                    void *ws, size_t ws_bytes, int64_t *output_coords, int *output_map, int M,
                                    int maxActive, void *stream) {
    D3_CLEAR();
    if (n <= 0 || M <= 0) return 0;
    VoxIdxWs w;
    if (!vi_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256;
    int rc = d3_exclusive_scan_i32(w.cnt, w.vstart, M, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    // stable sort of point ids by voxel id; w.flag is reused as the iota payload
    vi_iota_kernel<<<(n + T - 1) / T, T, 0, s>>>(w.flag, n);
    int bits = 1; while ((1ll << bits) < (long long)M && bits < 31) bits++;
    rc = d3_sort_pairs_i32(input_map, w.sorted_keys, w.flag, w.sorted_pts, n, bits, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    int gshift = 0; while ((1 << gshift) < maxActive + 1 && gshift < 6) gshift++;
    const long long rthreads = (long long)M << gshift;
    vi_rules_kernel<<<(int)((rthreads + T - 1) / T), T, 0, s>>>(mode, w.cnt, w.vstart, w.sorted_pts, output_map, M, maxActive, gshift);
    vi_coords_kernel<<<(M * ncols + T - 1) / T, T, 0, s>>>(coords, ncols, mode, w.cnt, w.vstart, w.sorted_pts,
                                                         output_coords, M);
    D3_LAUNCH_CHECK();
    return 0;
}
