// coordmap.hip -- coordinate hash + kernel maps for the sparse convolutions (gfx950).
//
// Stands in for the part of MinkowskiEngine's CoordinateManager the reference exercises
// (reference call sites: model/common.py:38,41,66,90,98; model/pointgroup.py:70,176,268): the
// neighbour table of a kernel-3 stride-1 convolution and the parent/child maps of the kernel-2
// stride-2 convolution and its transpose.  MinkowskiEngine is not vendored or pinned by the
// reference, so the semantics are the build's own (oracle/sparse_oracle.py, pinned to dense conv3d).
//
// Kernel maps are stored output-stationary as dense int32 tables (Mout, K): the convolution then
// needs no scatter and no atomics (d3net_amd/csrc/spconv.hip).  Table builds are hash-probe bound:
// bytes = 16*M (coords) + 4*M*K (table) against an L2-resident 12-byte-per-slot hash.
#include "common.h"
#include <mutex>
#include <vector>

#define CM_EMPTY 0xFFFFFFFFFFFFFFFFull

struct CmWs {
    unsigned long long *keys;  // cap
    int *first;                // cap   min row index per slot (= the row for unique coordinates)
    int *slot_vid;             // cap
    int *slot_of;              // M
    int *flag;                 // M
    int *scan;                 // M
    int *scalars;              // 8
    void *temp; size_t temp_bytes;
    size_t cap;
};
static size_t cm_cap(int n) { size_t c = 1024; while (c < (size_t)n * 2) c <<= 1; return c; }
static size_t cm_layout(void *ws, size_t ws_bytes, int M, CmWs &w) {
    D3Carver c(ws, ws_bytes);
    size_t nn = (size_t)(M > 0 ? M : 1);
    w.cap = cm_cap(M);
    w.keys = c.take<unsigned long long>(w.cap);
    w.first = c.take<int>(w.cap);
    w.slot_vid = c.take<int>(w.cap);
    w.slot_of = c.take<int>(nn);
    w.flag = c.take<int>(nn);
    w.scan = c.take<int>(nn);
    w.scalars = c.take<int>(64);
    w.temp_bytes = d3_scan_temp_bytes(M);
    w.temp = c.take<char>(w.temp_bytes);
    return c.off;
}
extern "C" size_t d3_coordmap_ws_bytes(int M) {
    CmWs w;
    return cm_layout(nullptr, 0, M, w) + 256;
}

__device__ __forceinline__ unsigned long long cm_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
// key layout: batch 19 bits | x 15 | y 15 | z 15 (x,y,z biased by 2^14)
__device__ __forceinline__ bool cm_pack(int b, int x, int y, int z, unsigned long long &key) {
    const int B = 1 << 14;
    bool ok = (b >= 0 && b < (1 << 19)) && (x >= -B && x < B) && (y >= -B && y < B) && (z >= -B && z < B);
    key = ((unsigned long long)(unsigned)b << 45) | ((unsigned long long)(unsigned)(x + B) << 30) |
          ((unsigned long long)(unsigned)(y + B) << 15) | (unsigned long long)(unsigned)(z + B);
    return ok;
}
__device__ __forceinline__ int floor_div(int a, int s) { return (a >= 0) ? a / s : -((-a + s - 1) / s); }

__global__ void cm_init_kernel(unsigned long long *keys, int *first, size_t cap, int *scalars) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = CM_EMPTY; first[i] = 0x7FFFFFFF; }
    if (i < 8) scalars[i] = 0;
}
__global__ void cm_init2_kernel(unsigned long long *keys, int *first, size_t cap, int *scalars, int *ok16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = CM_EMPTY; first[i] = 0x7FFFFFFF; }
    if (i < 8) scalars[i] = 0;
    if (i == 0) *ok16 = 1;
}
// insert coords[i] quantised to step q (q == 0: as is)
__global__ void cm_insert_kernel(const int *__restrict__ coords, int M, int q, unsigned long long *keys, int *first,
                                 size_t cap, int *slot_of, int *scalars) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int b = coords[i * 4], x = coords[i * 4 + 1], y = coords[i * 4 + 2], z = coords[i * 4 + 3];
    if (q > 0) { x = floor_div(x, q) * q; y = floor_div(y, q) * q; z = floor_div(z, q) * q; }
    unsigned long long key;
    if (!cm_pack(b, x, y, z, key)) { scalars[2] = 1; key &= ~(1ull << 63); }
    size_t slot = cm_hash(key) & (cap - 1);
    for (size_t probe = 0; probe < cap; probe++) {
        unsigned long long prev = atomicCAS(&keys[slot], CM_EMPTY, key);
        if (prev == CM_EMPTY || prev == key) { atomicMin(&first[slot], i); slot_of[i] = (int)slot; return; }
        slot = (slot + 1) & (cap - 1);
    }
    scalars[2] = 2;
    slot_of[i] = 0;
}
__device__ __forceinline__ int cm_lookup(const unsigned long long *keys, const int *vals, size_t cap,
                                         unsigned long long key) {
    size_t slot = cm_hash(key) & (cap - 1);
    for (size_t probe = 0; probe < cap; probe++) {
        unsigned long long k = keys[slot];
        if (k == key) return vals[slot];
        if (k == CM_EMPTY) return -1;
        slot = (slot + 1) & (cap - 1);
    }
    return -1;
}

// one thread per (row, offset): coalesced table stores
// nbr16 / ok16 (optional, round 4): the int16-delta form of the table in the same pass (see cm_pack16_kernel below);
// *ok16 was set to 1 by the hash build's init kernel
__global__ void cm_k3_kernel(const int *__restrict__ coords, int M, int ts, const unsigned long long *keys,
                             const int *first, size_t cap, int *__restrict__ nbr, short *__restrict__ nbr16, int *ok16) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)M * 27;
    bool bad = false;
    if (e < total) {
        const int u = (int)(e / 27), k = (int)(e % 27);
        const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
        unsigned long long key;
        int r = -1;
        if (cm_pack(coords[u * 4], coords[u * 4 + 1] + ox * ts, coords[u * 4 + 2] + oy * ts, coords[u * 4 + 3] + oz * ts, key))
            r = cm_lookup(keys, first, cap, key);
        nbr[e] = r;
        if (nbr16) {
            int d = -32768;
            if (r >= 0) { d = r - u; if (d < -32767 || d > 32767) { bad = true; d = -32768; } }
            nbr16[e] = (short)d;
        }
    } else if (nbr16 && e < total + 2) nbr16[e] = (short)-32768;
    if (nbr16 && __any(bad) && (threadIdx.x & 63) == 0) *ok16 = 0;
}

static int cm_build_hash(const int *coords, int M, int q, CmWs &w, hipStream_t s, int *ok16 = nullptr) {
    const int T = 256;
    if (ok16) cm_init2_kernel<<<(int)((w.cap + T - 1) / T), T, 0, s>>>(w.keys, w.first, w.cap, w.scalars, ok16);
    else
    cm_init_kernel<<<(int)((w.cap + T - 1) / T), T, 0, s>>>(w.keys, w.first, w.cap, w.scalars);
    cm_insert_kernel<<<(M + T - 1) / T, T, 0, s>>>(coords, M, q, w.keys, w.first, w.cap, w.slot_of, w.scalars);
    D3_LAUNCH_CHECK();
    return 0;
}

static int cm_k3_run(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *nbr, short *nbr16, int *ok16, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (ts <= 0 || ((nbr16 == nullptr) != (ok16 == nullptr))) return D3_ERR_ARG;
    CmWs w;
    if (ws == nullptr || cm_layout(ws, ws_bytes, M, w) > ws_bytes) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int rc = cm_build_hash(coords, M, 0, w, s, ok16);
    if (rc) return rc;
    long long total = (long long)M * 27 + (nbr16 ? 2 : 0);
    cm_k3_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(coords, M, ts, w.keys, w.first, w.cap, nbr, nbr16, ok16);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_kmap_k3(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *nbr, void *stream) {
    return cm_k3_run(coords, M, ts, ws, ws_bytes, nbr, nullptr, nullptr, stream);
}
// d3_kmap_k3 + the int16-delta form of the table and its validity flag in the same pass (see d3_kmap_k3_pack16)
extern "C" int d3_kmap_k3_16(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *nbr, void *nbr16, int *ok16, void *stream) {
    if (!nbr16 || !ok16) return D3_ERR_ARG;
    return cm_k3_run(coords, M, ts, ws, ws_bytes, nbr, (short *)nbr16, ok16, stream);
}

// ---- 16-bit form of a K = 27 neighbour table (round 4).  A stride-1 table is read by the forward, the data gradient and the
// weight gradient of every convolution of its level (33 launches at level 0 of the backbone): 108 bytes per row each time.
// Neighbours of row u sit near u in any spatially coherent row order, so the entries are stored as int16 deltas nbr - u
// (54 bytes per row; -32768 = absent).  *ok16 = 1 when every delta fits; otherwise the consumers keep the dense table.
// nbr16 holds M * 27 shorts (+ 2 pad shorts: the kernels read it in 32-bit words).
__global__ void cm_pack16_kernel(const int *__restrict__ nbr, long long total, short *__restrict__ nbr16, int *ok16) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (e < total) {
        const int v = nbr[e];
        int d = -32768;
        if (v >= 0) {
            d = v - (int)(e / 27);
            if (d < -32767 || d > 32767) { bad = true; d = -32768; }
        }
        nbr16[e] = (short)d;
    } else if (e < total + 2) nbr16[e] = (short)-32768;
    if (__any(bad) && (threadIdx.x & 63) == 0) *ok16 = 0;
}
__global__ void cm_set1_kernel(int *p) { *p = 1; }
extern "C" int d3_kmap_k3_pack16(const int *nbr, int M, void *nbr16, int *ok16, void *stream) {
    D3_CLEAR();
    if (M <= 0) return 0;
    if (!nbr || !nbr16 || !ok16) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const long long total = (long long)M * 27;
    cm_set1_kernel<<<1, 1, 0, s>>>(ok16);
    cm_pack16_kernel<<<(int)((total + 2 + 255) / 256), 256, 0, s>>>(nbr, total, (short *)nbr16, ok16);
    D3_LAUNCH_CHECK();
    return 0;
}

__global__ void cm_flag_kernel(const int *first, const int *slot_of, int *flag, int M) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flag[i] = (first[slot_of[i]] == i) ? 1 : 0;
}
__global__ void cm_assign_kernel(const int *flag, const int *scan, const int *slot_of, int *slot_vid, int M,
                                 int *scalars) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (flag[i]) slot_vid[slot_of[i]] = scan[i];
    if (i == M - 1) scalars[0] = scan[i] + flag[i];
}
__global__ void cm_parent_kernel(const int *__restrict__ coords, int M, int ts, const int *slot_of,
                                 const int *slot_vid, int *parent, int *kidx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    parent[i] = slot_vid[slot_of[i]];
    const int s2 = 2 * ts;
    const int x = coords[i * 4 + 1], y = coords[i * 4 + 2], z = coords[i * 4 + 3];
    const int dx = (x - floor_div(x, s2) * s2) / ts, dy = (y - floor_div(y, s2) * s2) / ts,
              dz = (z - floor_div(z, s2) * s2) / ts;
    kidx[i] = dx + 2 * dy + 4 * dz;
}

extern "C" int d3_kmap_down_count(const int *coords, int M, int ts, void *ws, size_t ws_bytes, int *parent,
                                  int *kidx, int *Mout_host, void *stream) {
    D3_CLEAR();
    *Mout_host = 0;
    if (M <= 0) return 0;
    if (ts <= 0) return D3_ERR_ARG;
    CmWs w;
    if (ws == nullptr || cm_layout(ws, ws_bytes, M, w) > ws_bytes) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256, nb = (M + T - 1) / T;
    int rc = cm_build_hash(coords, M, 2 * ts, w, s);
    if (rc) return rc;
    cm_flag_kernel<<<nb, T, 0, s>>>(w.first, w.slot_of, w.flag, M);
    rc = d3_exclusive_scan_i32(w.flag, w.scan, M, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    cm_assign_kernel<<<nb, T, 0, s>>>(w.flag, w.scan, w.slot_of, w.slot_vid, M, w.scalars);
    cm_parent_kernel<<<nb, T, 0, s>>>(coords, M, ts, w.slot_of, w.slot_vid, parent, kidx);
    D3_LAUNCH_CHECK();
    int h[3];
    D3_CHECK(hipMemcpyAsync(h, w.scalars, sizeof(h), hipMemcpyDeviceToHost, s));
    D3_CHECK(hipStreamSynchronize(s));
    if (h[2] == 1) return D3_ERR_RANGE;
    if (h[2] == 2) return D3_ERR_OVERFLOW;
    *Mout_host = h[0];
    return 0;
}

__global__ void cm_fill_neg_kernel(int *a, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = -1;
}
__global__ void cm_down_fill_kernel(const int *__restrict__ coords, int M, int ts, const int *__restrict__ parent,
                                    const int *__restrict__ kidx, const int *__restrict__ flag, int *out_coords,
                                    int *child, int *up) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int p = parent[i], k = kidx[i];
    child[p * 8 + k] = i;
    up[i * 8 + k] = p;
    if (flag && flag[i]) {  // first row of its parent cell: defines the output coordinate
        const int s2 = 2 * ts;
        out_coords[p * 4 + 0] = coords[i * 4];
        out_coords[p * 4 + 1] = floor_div(coords[i * 4 + 1], s2) * s2;
        out_coords[p * 4 + 2] = floor_div(coords[i * 4 + 2], s2) * s2;
        out_coords[p * 4 + 3] = floor_div(coords[i * 4 + 3], s2) * s2;
    }
}

extern "C" int d3_kmap_down_fill(const int *coords, int M, int ts, void *ws, size_t ws_bytes, const int *parent,
                                 const int *kidx, int *out_coords, int *child, int *up, int Mout, void *stream) {
    D3_CLEAR();
    if (M <= 0 || Mout <= 0) return 0;
    CmWs w;
    if (ws == nullptr || cm_layout(ws, ws_bytes, M, w) > ws_bytes) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256;
    long long nc = (long long)Mout * 8, nu = (long long)M * 8;
    cm_fill_neg_kernel<<<(int)((nc + T - 1) / T), T, 0, s>>>(child, nc);
    cm_fill_neg_kernel<<<(int)((nu + T - 1) / T), T, 0, s>>>(up, nu);
    cm_down_fill_kernel<<<(M + T - 1) / T, T, 0, s>>>(coords, M, ts, parent, kidx, w.flag, out_coords, child, up);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// All stride-2 levels of a U-Net with ONE host round trip.  d3_kmap_down_count synchronises once per level to return
// the row count the caller allocates with (6 round trips for the 7-level backbone, each draining the stream).  Here
// the row counts stay on the device while the whole coordinate pyramid is built -- every kernel takes its row count
// from device memory and is launched over the level-0 bound -- and one copy returns all of them; the kernel-map
// tables are then filled with exact sizes (d3_kmap_k3, d3_kmap_down_fill2), which needs no further synchronisation.
__global__ void cmp_insert_kernel(const int *__restrict__ coords, const int *Mdev, int q, unsigned long long *keys,
                                  int *first, size_t cap, int *slot_of, int *scalars) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *Mdev) return;
    int b = coords[i * 4], x = coords[i * 4 + 1], y = coords[i * 4 + 2], z = coords[i * 4 + 3];
    x = floor_div(x, q) * q; y = floor_div(y, q) * q; z = floor_div(z, q) * q;
    unsigned long long key;
    if (!cm_pack(b, x, y, z, key)) { scalars[2] = 1; key &= ~(1ull << 63); }
    size_t slot = cm_hash(key) & (cap - 1);
    for (size_t probe = 0; probe < cap; probe++) {
        unsigned long long prev = atomicCAS(&keys[slot], CM_EMPTY, key);
        if (prev == CM_EMPTY || prev == key) { atomicMin(&first[slot], i); slot_of[i] = (int)slot; return; }
        slot = (slot + 1) & (cap - 1);
    }
    scalars[2] = 2;
    slot_of[i] = 0;
}
__global__ void cmp_flag_kernel(const int *first, const int *slot_of, int *flag, const int *Mdev, int bound) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bound) flag[i] = (i < *Mdev && first[slot_of[i]] == i) ? 1 : 0;
}
// parent / kernel index of every row, the coarse coordinates, and the coarse row count
__global__ void cmp_level_kernel(const int *__restrict__ coords, const int *Mdev, int ts, const int *flag, const int *scan,
                                 const int *slot_of, int *slot_vid_unused, int *parent, int *kidx, int *out_coords,
                                 int *Mnext) {
    const int M = *Mdev;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *Mnext = (M > 0) ? scan[M - 1] + flag[M - 1] : 0;
    if (i >= M) return;
    (void)slot_vid_unused; (void)slot_of;
    const int s2 = 2 * ts;
    const int x = coords[i * 4 + 1], y = coords[i * 4 + 2], z = coords[i * 4 + 3];
    kidx[i] = (x - floor_div(x, s2) * s2) / ts + 2 * ((y - floor_div(y, s2) * s2) / ts) + 4 * ((z - floor_div(z, s2) * s2) / ts);
    if (flag[i]) {
        const int p = scan[i];
        out_coords[p * 4 + 0] = coords[i * 4];
        out_coords[p * 4 + 1] = floor_div(x, s2) * s2;
        out_coords[p * 4 + 2] = floor_div(y, s2) * s2;
        out_coords[p * 4 + 3] = floor_div(z, s2) * s2;
    }
}
// parent[i] = id of the first row of i's cell (two passes: the ids are the scan values of the flagged rows)
__global__ void cmp_vid_kernel(const int *flag, const int *scan, const int *slot_of, int *slot_vid, const int *Mdev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *Mdev && flag[i]) slot_vid[slot_of[i]] = scan[i];
}
__global__ void cmp_parent_kernel(const int *slot_of, const int *slot_vid, int *parent, const int *Mdev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *Mdev) parent[i] = slot_vid[slot_of[i]];
}

__global__ void cmp_set_kernel(int *p, int v) { *p = v; }

// coords0 (M0,4); levels 1..nlevels-1 are written to coords_out[(l-1)*M0*4 ...], parent / kidx / flag of level l
// (rows of level l) to [l*M0 ...]; rows_host[l] = row count of level l.  ws >= d3_coordmap_ws_bytes(M0).
// The host round trip of the pyramid, split in two so that the caller can put independent device work between the row-count
// copy and the wait for it (PointGroup.feed enqueues the input voxelisation there: the device pools the features while the
// host reads the level sizes and enqueues the table fills -- the 60-200 us the stream used to idle at this point are hidden).
// A ticket owns a small pinned buffer and an event; tickets are pooled.
struct PyrTicket { int *pinned; hipEvent_t ev; int nlevels; };
static std::mutex g_pyr_mu;
static std::vector<PyrTicket *> g_pyr_free;
#define PYR_MAXLEV 16

extern "C" int d3_kmap_pyramid_begin(const int *coords0, int M0, int nlevels, void *ws, size_t ws_bytes, int *coords_out,
                                     int *parent, int *kidx, int *flag, int *rows_dev, void **ticket, void *stream) {
    D3_CLEAR();
    if (!ticket || nlevels < 1 || nlevels > PYR_MAXLEV) return D3_ERR_ARG;
    *ticket = nullptr;
    if (M0 <= 0) return 0;
    CmWs w;
    if (ws == nullptr || cm_layout(ws, ws_bytes, M0, w) > ws_bytes) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256, nb = (M0 + T - 1) / T;
    D3_CHECK(hipMemsetAsync(w.scalars, 0, 8 * sizeof(int), s));
    cmp_set_kernel<<<1, 1, 0, s>>>(rows_dev, M0);
    const int *cur = coords0;
    int ts = 1;
    for (int l = 0; l + 1 < nlevels; l++) {
        int *pl = parent + (size_t)l * M0, *kl = kidx + (size_t)l * M0, *fl = flag + (size_t)l * M0;
        int *nxt = coords_out + (size_t)l * M0 * 4;
        cm_init_kernel<<<(int)((w.cap + T - 1) / T), T, 0, s>>>(w.keys, w.first, w.cap, w.scalars + 8);   // scalars[2] (error flag) is kept
        cmp_insert_kernel<<<nb, T, 0, s>>>(cur, rows_dev + l, 2 * ts, w.keys, w.first, w.cap, w.slot_of, w.scalars);
        cmp_flag_kernel<<<nb, T, 0, s>>>(w.first, w.slot_of, fl, rows_dev + l, M0);
        int rc = d3_exclusive_scan_i32(fl, w.scan, M0, w.temp, w.temp_bytes, s);
        if (rc) return rc;
        cmp_vid_kernel<<<nb, T, 0, s>>>(fl, w.scan, w.slot_of, w.slot_vid, rows_dev + l);
        cmp_parent_kernel<<<nb, T, 0, s>>>(w.slot_of, w.slot_vid, pl, rows_dev + l);
        cmp_level_kernel<<<nb, T, 0, s>>>(cur, rows_dev + l, ts, fl, w.scan, w.slot_of, w.slot_vid, pl, kl, nxt, rows_dev + l + 1);
        cur = nxt; ts *= 2;
    }
    D3_LAUNCH_CHECK();
    PyrTicket *t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pyr_mu);
        if (!g_pyr_free.empty()) { t = g_pyr_free.back(); g_pyr_free.pop_back(); }
    }
    if (!t) {
        t = new PyrTicket{nullptr, nullptr, 0};
        hipError_t he = hipHostMalloc((void **)&t->pinned, (PYR_MAXLEV + 4) * sizeof(int));
        if (he == hipSuccess) he = hipEventCreateWithFlags(&t->ev, hipEventDisableTiming);
        if (he != hipSuccess) { delete t; return (int)he; }
    }
    t->nlevels = nlevels;
    D3_CHECK(hipMemcpyAsync(t->pinned, rows_dev, sizeof(int) * nlevels, hipMemcpyDeviceToHost, s));
    D3_CHECK(hipMemcpyAsync(t->pinned + PYR_MAXLEV, w.scalars, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    D3_CHECK(hipEventRecord(t->ev, s));
    *ticket = t;
    return 0;
}

// waits for the counts of d3_kmap_pyramid_begin (rows_host: nlevels ints) and returns the ticket to the pool
extern "C" int d3_kmap_pyramid_end(void *ticket, int *rows_host, int nlevels) {
    D3_CLEAR();
    for (int l = 0; l < nlevels; l++) rows_host[l] = 0;
    if (!ticket) return 0;                      // (an empty level 0: nothing was enqueued)
    PyrTicket *t = (PyrTicket *)ticket;
    if (t->nlevels != nlevels) return D3_ERR_ARG;
    const hipError_t e = hipEventSynchronize(t->ev);
    int h2 = 0;
    if (e == hipSuccess) {
        for (int l = 0; l < nlevels; l++) rows_host[l] = t->pinned[l];
        h2 = t->pinned[PYR_MAXLEV + 2];
    }
    {
        std::lock_guard<std::mutex> lk(g_pyr_mu);
        g_pyr_free.push_back(t);
    }
    D3_CHECK(e);
    if (h2 == 1) return D3_ERR_RANGE;
    if (h2 == 2) return D3_ERR_OVERFLOW;
    return 0;
}

extern "C" int d3_kmap_pyramid(const int *coords0, int M0, int nlevels, void *ws, size_t ws_bytes, int *coords_out,
                               int *parent, int *kidx, int *flag, int *rows_dev, int *rows_host, void *stream) {
    void *t = nullptr;
    for (int l = 0; l < nlevels; l++) rows_host[l] = 0;
    if (M0 <= 0 || nlevels < 1) return 0;
    int rc = d3_kmap_pyramid_begin(coords0, M0, nlevels, ws, ws_bytes, coords_out, parent, kidx, flag, rows_dev, &t, stream);
    if (rc) return rc;
    return d3_kmap_pyramid_end(t, rows_host, nlevels);
}

// d3_kmap_down_fill with the first-row flags passed in (the pyramid keeps them per level)
extern "C" int d3_kmap_down_fill2(int M, int Mout, const int *parent, const int *kidx, int *child, int *up, void *stream) {
    D3_CLEAR();
    if (M <= 0 || Mout <= 0) return 0;
    hipStream_t s = d3_stream(stream);
    const int T = 256;
    long long nc = (long long)Mout * 8, nu = (long long)M * 8;
    cm_fill_neg_kernel<<<(int)((nc + T - 1) / T), T, 0, s>>>(child, nc);
    cm_fill_neg_kernel<<<(int)((nu + T - 1) / T), T, 0, s>>>(up, nu);
    cm_down_fill_kernel<<<(M + T - 1) / T, T, 0, s>>>(nullptr, M, 1, parent, kidx, nullptr, nullptr, child, up);
    D3_LAUNCH_CHECK();
    return 0;
}
