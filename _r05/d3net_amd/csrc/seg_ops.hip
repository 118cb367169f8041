// seg_ops.hip -- segment reductions over cluster point lists (gfx950).
//
// Replaces PG_OP.sec_mean/sec_min/sec_max, roipool_fp/bp and get_iou
// (reference: lib/pointgroup_ops/src/sec_mean/sec_mean.cu:12-86, src/roipool/roipool.cu:12-57,
//  src/get_iou/get_iou.cu:12-38).  The reference launches min(C,32) threads per segment and
// walks the segment serially (3 threads for the only sec_* caller, C=3).  Here one 256-thread
// workgroup owns a segment and reads it as one flat, fully coalesced stream of rows*C floats;
// lane t always sees channel t % C (the active thread count is rounded down to a multiple of C).
// All of these are HBM-bound: algorithmic bytes = 4*(rows*C) in + 4*(P*C) out + 4*(P+1).
#include "common.h"

#define SEG_THREADS 256

// ----------------------------------------------------------------------------- min / max
template <bool IS_MAX>
__global__ __launch_bounds__(SEG_THREADS) void sec_minmax_kernel(const float *__restrict__ inp,
                                                                const int *__restrict__ offsets,
                                                                float *__restrict__ out, int nProposal, int C) {
    __shared__ float red[SEG_THREADS];
    const int active = (SEG_THREADS / C) * C;  // C <= SEG_THREADS checked on the host
    const int t = threadIdx.x;
    for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
        const int start = offsets[p], end = offsets[p + 1];
        const long long base = (long long)start * C;
        const long long total = (long long)(end - start) * C;
        float v = IS_MAX ? -INFINITY : INFINITY;  // reference: +-1e50 -> +-inf in float
        if (t < active) {
            for (long long f = t; f < total; f += active) {
                float x = inp[base + f];
                if (IS_MAX ? (x > v) : (x < v)) v = x;  // same comparison as the reference (NaN never wins)
            }
        }
        red[t] = v;
        __syncthreads();
        if (t < C) {
            float r = red[t];
            for (int k = t + C; k < active; k += C) {
                float x = red[k];
                if (IS_MAX ? (x > r) : (x < r)) r = x;
            }
            out[(long long)p * C + t] = r;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------- mean
// The reference accumulates inp[i]/count term by term in row order (sec_mean.cu:19-23); fp32
// addition is not associative, so to stay BIT-EXACT the additions are kept serial per channel:
// the workgroup streams a chunk of the segment into LDS (coalesced, divisions in parallel) and
// C lanes then add their column in row order.  Cost: one dependent fp32 add per row.
#define MEAN_CHUNK 2048  // floats staged per round
__global__ __launch_bounds__(SEG_THREADS) void sec_mean_kernel(const float *__restrict__ inp,
                                                              const int *__restrict__ offsets,
                                                              float *__restrict__ out, int nProposal, int C) {
    __shared__ float stage[MEAN_CHUNK];
    const int t = threadIdx.x;
    const int rows_per_chunk = MEAN_CHUNK / C;
    for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
        const int start = offsets[p], end = offsets[p + 1];
        const float count = (float)(end - start);
        float mean = 0.f;
        for (int r0 = start; r0 < end; r0 += rows_per_chunk) {
            const int rows = min(rows_per_chunk, end - r0);
            const int nflt = rows * C;
            const long long base = (long long)r0 * C;
            for (int f = t; f < nflt; f += SEG_THREADS) stage[f] = __fdiv_rn(inp[base + f], count);  // IEEE divide
            __syncthreads();
            if (t < C) {  // serial add chain; the LDS reads are issued 8 at a time ahead of it
                int r = 0;
                for (; r + 8 <= rows; r += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = stage[(r + j) * C + t];
#pragma unroll
                    for (int j = 0; j < 8; j++) mean = __fadd_rn(mean, v[j]);
                }
                for (; r < rows; r++) mean = __fadd_rn(mean, stage[r * C + t]);
            }
            __syncthreads();
        }
        if (t < C) out[(long long)p * C + t] = mean;
    }
}

// Producer / consumer workgroup per segment: the add chain (the only serial part -- one dependent fp32 add per row;
// bit-exactness forbids reordering it) runs back to back on wave 0 while waves 1-3 load, divide (IEEE division is
// ~40 instructions per element) and stage the chunks of the NEXT round in LDS.  The workgroup-per-segment kernel
// above idles 253 of 256 lanes during the chain and the chain during the staging.
#define MEANW_CHUNK 1024   // floats per chunk (16 per producer lane)
#define MEANW_PB 16        // elements of a producer lane's 16 whose loads are in flight together (the 512-thread instance has 256 registers per lane: no spills; at 1024 threads / 128 registers every variant spilled, and a spill's reload waits on vmcnt(0), i.e. for the loads in flight)
// waves of an instance: the chain wave + NPROD producers; beyond 3 producers the waves that would share the chain wave's SIMD
// (wave % 4 == 0: the hardware deals a workgroup's waves round the four SIMDs) stay idle -- the chain owns its SIMD's issue slots
#define MEANW_WAVES(NP) ((NP) <= 3 ? (NP) + 1 : (NP) + 1 + ((NP) - 1) / 3)
#define MEANW_PROD 3       // producer waves = chunks per round (the 256-thread instance; the 1024-thread one runs 15)
// The chunk is staged TRANSPOSED (channel-major, rows padded to a multiple of 4), so the chain lane of a channel reads
// four consecutive rows with one ds_read_b128 and keeps 32 rows in flight behind the 32 dependent adds: the chain runs
// at the issue rate of v_add_f32 instead of waiting for LDS.
// gidx != nullptr: row r of the input is inp[gidx[2r + 1]] (the (cluster, point) pairs of `clusters_idx`: the mean of the
// clusters' point coordinates without materialising the gathered (S, 3) copy)
// NPROD (round 5): with 3 producers a round is bound by the PRODUCERS' latency (two dependent gathers + IEEE divisions: ~11 us for
// the ~1020 rows the chain then adds in ~2 us -- 370 us for the 33,721-point floor of the canonical scene, on the step's critical
// path); 6 producer waves (512 threads, 48 KB of LDS) with all 16 elements of a lane in flight stage ~1,900 rows per round, which the
// chain adds in ~5 us -- more than the producers' two memory round trips + divisions -- on a SIMD of its own.  The chain itself:
// tools/probes/addchain.hip measures 2.4 ns per dependent v_add_f32 from registers (~80 us for 33 k rows: the floor of this kernel).
template <int NPROD>
__global__ __launch_bounds__(MEANW_WAVES(NPROD) * 64) void sec_mean_pc_kernel(const float *__restrict__ inp, const int *__restrict__ offsets,
                                                         float *__restrict__ out, int nProposal, int C, const int *__restrict__ gidx, int prediv) {
    extern __shared__ __attribute__((aligned(16))) float meanw_smem[];
    float (*stage)[NPROD][MEANW_CHUNK] = (float (*)[NPROD][MEANW_CHUNK])meanw_smem;      // [2][NPROD][MEANW_CHUNK]
    constexpr int MEANW_PROD_ = NPROD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = blockIdx.x;
    const int start = offsets[p], end = offsets[p + 1];
    const float count = (float)(end - start);
    // rows / floats per chunk: a multiple of 256 rows where a chunk holds that many (the chain's straight-line turn takes 256
    // rows; the 20-row tail of a 340-row chunk at C = 3 ran one LDS round trip per row and cost more than the 320 rows in front
    // of it), else of 64, else of 4 (or 0)
    const int rpc0 = MEANW_CHUNK / C, rpc = rpc0 >= 256 ? (rpc0 & ~255) : (rpc0 >= 64 ? (rpc0 & ~63) : (rpc0 & ~3)), cf = rpc * C;
    if (rpc == 0) return;                                    // (C <= 64 on this path: rpc >= 16)
    const long long base = (long long)start * C, total = (long long)(end - start) * C;
    const int nchunk = (int)((total + cf - 1) / cf);
    const int nround = (nchunk + MEANW_PROD_ - 1) / MEANW_PROD;
    const unsigned int invC = (65536u + C - 1) / C;          // f / C for f < 1024 (exact: f * invC < 2^26, C <= 64)
    float mean = 0.f;
    if (NPROD > 3 && wave == 0) __builtin_amdgcn_s_setprio(3);
    const int pslot = NPROD <= 3 ? wave - 1 : ((wave & 3) ? wave - 1 - (wave >> 2) : -1);      // producer slot of this wave (-1: chain / idle)
    for (int rd = 0; rd <= nround; rd++) {
        if (pslot >= 0) {   // produce chunk (rd, pslot) of round rd into buffer rd & 1
            const int k = rd * MEANW_PROD_ + pslot;
            if (rd < nround && k < nchunk) {
                float *st = stage[rd & 1][pslot];
                const long long cb = (long long)k * cf;
                // Branch-free, every load of a stage issued before the first use (round 5: written with a branch per element the
                // compiler emitted, per element, index load -> wait -> value load -> wait: 32 dependent memory round trips per
                // chunk and producer wave, ~30 us -- the whole kernel was bound by it, whatever the number of producers)
                // (batches of MEANW_PB elements: 16 x (index, 64-bit address, value) spilled at the 128 registers of a 1024-thread block;
                // `lz` = lane behind an opaque move: the row / channel of an element are loop invariants the compiler otherwise
                // hoists out of the round loop -- 48 registers held across it, spilled, and every reload's s_waitcnt vmcnt(0)
                // serialised the global loads again)
                int lz;
                asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
#pragma unroll
                for (int hf = 0; hf < 16 / MEANW_PB; hf++) {
                    float v[MEANW_PB];
                    int rowj[MEANW_PB], cj[MEANW_PB];
                    bool okj[MEANW_PB];
                    unsigned int src[MEANW_PB];
#pragma unroll
                    for (int j = 0; j < MEANW_PB; j++) {
                        const int f = (hf * MEANW_PB + j) * 64 + lz;
                        okj[j] = f < cf && cb + f < total;
                        rowj[j] = (int)(((unsigned int)f * invC) >> 16); cj[j] = f - rowj[j] * C;
                    }
                    if (gidx) {
                        int pt[MEANW_PB];
                        const int *gi = gidx + ((long long)start + (long long)k * rpc) * 2 + 1;
#pragma unroll
                        for (int j = 0; j < MEANW_PB; j++) pt[j] = gi[(okj[j] ? rowj[j] : 0) * 2];      // (row 0 of a chunk always exists)
#pragma unroll
                        for (int j = 0; j < MEANW_PB; j++) src[j] = (unsigned int)pt[j] * (unsigned int)C + (unsigned int)cj[j];
#pragma unroll
                        for (int j = 0; j < MEANW_PB; j++) v[j] = inp[src[j]];
                    } else {
                        const float *ib = inp + base + cb;
#pragma unroll
                        for (int j = 0; j < MEANW_PB; j++) v[j] = ib[okj[j] ? (hf * MEANW_PB + j) * 64 + lz : 0];
                    }
#pragma unroll
                    for (int j = 0; j < MEANW_PB; j++) {
                        const int f = (hf * MEANW_PB + j) * 64 + lz;
                        if (f < cf) st[cj[j] * rpc + rowj[j]] = okj[j] ? (prediv ? v[j] : __fdiv_rn(v[j], count)) : 0.f;
                    }
                }
            }
        } else if (wave == 0 && rd > 0) {   // consume round rd-1
            for (int q = 0; q < MEANW_PROD_; q++) {
                const int k = (rd - 1) * MEANW_PROD_ + q;
                if (k >= nchunk) break;
                const float *st = stage[(rd - 1) & 1][q] + lane * rpc;
                const long long left = total - (long long)k * cf;
                const int rows = (int)((left < cf ? left : cf) / C);
                if (lane < C) {
                    int r = 0;
                    // two register sets with FIXED roles (rows r .. r+31 / r+32 .. r+63 of the current 64): a set is reloaded
                    // for the next turn right behind the adds that read it -- no rotation, hence no register copies in the
                    // chain wave's issue stream (round 5: the rotating form spent a third of its VALU slots on v_mov)
                    // 256 rows per turn as STRAIGHT-LINE code: eight groups of 32 rows, group g + 2 requested from LDS behind the
                    // adds of group g (scheduling barriers pin that order) -- no register set lives across the loop's back edge.
                    // (Round 5: both software-pipelined loop forms made the compiler copy every prefetched register at the back
                    // edge -- 64 moves per 64 adds -- and wait for LDS with nothing in flight: 6 ns per row against 2.4 ns for the
                    // bare dependent-add chain, tools/probes/addchain.hip.)
#define MEANW_LD(W, R0)  _Pragma("unroll") for (int j = 0; j < 8; j++) W[j] = *(const float4 *)(st + (R0) + j * 4)
#define MEANW_ADD(W)     _Pragma("unroll") for (int j = 0; j < 8; j++) { mean = __fadd_rn(mean, W[j].x); mean = __fadd_rn(mean, W[j].y); mean = __fadd_rn(mean, W[j].z); mean = __fadd_rn(mean, W[j].w); }
                    for (; r + 256 <= rows; r += 256) {
                        float4 g0[8], g1[8], g2[8], g3[8], g4[8], g5[8], g6[8], g7[8];
                        MEANW_LD(g0, r); MEANW_LD(g1, r + 32);
                        __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g0); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g2, r + 64); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g1); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g3, r + 96); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g2); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g4, r + 128); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g3); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g5, r + 160); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g4); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g6, r + 192); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g5); __builtin_amdgcn_sched_barrier(0); MEANW_LD(g7, r + 224); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g6); __builtin_amdgcn_sched_barrier(0);
                        MEANW_ADD(g7); __builtin_amdgcn_sched_barrier(0);
                    }
                    for (; r + 32 <= rows; r += 32) {
                        float4 g0[8];
                        MEANW_LD(g0, r);
                        MEANW_ADD(g0);
                    }
#undef MEANW_LD
#undef MEANW_ADD
                    for (; r + 4 <= rows; r += 4) {      // (the tail of a segment's last chunk: its real rows only, four per LDS read)
                        const float4 w = *(const float4 *)(st + r);
                        mean = __fadd_rn(mean, w.x); mean = __fadd_rn(mean, w.y); mean = __fadd_rn(mean, w.z); mean = __fadd_rn(mean, w.w);
                    }
                    for (; r < rows; r++) mean = __fadd_rn(mean, st[r]);
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane < C) out[(long long)p * C + lane] = mean;
}

// -------------------------------------------------------------------------------- roipool
// max + FIRST argmax per (proposal, channel): strict '>' in ascending row order in the
// reference (roipool.cu:20-25).  Per-thread partials keep the earliest row of their maximum;
// the cross-thread combine breaks value ties towards the smaller row index.
__global__ __launch_bounds__(SEG_THREADS) void roipool_fp_kernel(const float *__restrict__ feats,
                                                                const int *__restrict__ offsets,
                                                                float *__restrict__ out_feats,
                                                                int *__restrict__ out_maxidx, int nProposal, int C) {
    __shared__ float redv[SEG_THREADS];
    __shared__ int redi[SEG_THREADS];
    const int active = (SEG_THREADS / C) * C;
    const int rows_per_pass = active / C;
    const int t = threadIdx.x;
    for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
        const int start = offsets[p], end = offsets[p + 1];
        float v = -INFINITY;
        int a = -1;
        if (t < active) {
            const int c = t % C;
            for (int r = start + t / C; r < end; r += rows_per_pass) {
                float x = feats[(long long)r * C + c];
                if (x > v) { v = x; a = r; }
            }
        }
        redv[t] = v; redi[t] = a;
        __syncthreads();
        if (t < C) {
            float bv = redv[t]; int bi = redi[t];
            for (int k = t + C; k < active; k += C) {
                float x = redv[k]; int xi = redi[k];
                if (x > bv || (x == bv && xi < bi)) { bv = x; bi = xi; }
            }
            out_feats[(long long)p * C + t] = bv;
            out_maxidx[(long long)p * C + t] = bi;
        }
        __syncthreads();
    }
}

__global__ void roipool_bp_kernel(float *__restrict__ d_feats, const int *__restrict__ maxidx,
                                  const float *__restrict__ d_out, long long total, int C) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int a = maxidx[e];
    if (a >= 0) atomicAdd(&d_feats[(long long)a * C + (int)(e % C)], d_out[e]);
}

// -------------------------------------------------------------------------------- get_iou
// Reference: every (proposal, instance) thread re-reads the whole proposal: O(S*nInst) loads.
// Here a workgroup histograms its proposal's instance labels in LDS once (O(S) loads), then
// forms the ratios.  `+ 1e-5` is a double literal in the reference, so the quotient is
// evaluated in double and rounded to float (get_iou.cu:25).
#define IOU_BINS 8192
__global__ __launch_bounds__(SEG_THREADS) void get_iou_kernel(const int *__restrict__ proposals_idx,
                                                             const int *__restrict__ offsets,
                                                             const int64_t *__restrict__ instance_labels,
                                                             const int *__restrict__ instance_pointnum,
                                                             float *__restrict__ iou, int nInstance, int nProposal) {
    __shared__ int hist[IOU_BINS];
    const int t = threadIdx.x;
    for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
        const int start = offsets[p], end = offsets[p + 1];
        const int proposal_total = end - start;
        for (int w0 = 0; w0 < nInstance; w0 += IOU_BINS) {
            const int wn = min(IOU_BINS, nInstance - w0);
            for (int b = t; b < wn; b += SEG_THREADS) hist[b] = 0;
            __syncthreads();
            // eight points per thread per round trip (point ids, then their labels): a big proposal is ~130 steps of two
            // dependent loads otherwise
            for (int i0 = start + t; i0 < end; i0 += 8 * SEG_THREADS) {
                int pi[8];
                int64_t lb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const int i = i0 + j * SEG_THREADS; pi[j] = proposals_idx[i < end ? i : start]; }
#pragma unroll
                for (int j = 0; j < 8; j++) lb[j] = instance_labels[pi[j]];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int lab = (int)lb[j] - w0;  // (int) cast as in the reference
                    if (i0 + j * SEG_THREADS < end && lab >= 0 && lab < wn) atomicAdd(&hist[lab], 1);
                }
            }
            __syncthreads();
            for (int b = t; b < wn; b += SEG_THREADS) {
                int inter = hist[b];
                int instance_total = instance_pointnum[w0 + b];
                double denom = (double)(float)(proposal_total + instance_total - inter) + 1e-5;
                iou[(long long)p * nInstance + w0 + b] = (float)((double)(float)inter / denom);
            }
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------- flat (row-parallel) min / max / roipool
// One workgroup per SEGMENT leaves the chip idle when a few clusters hold most of the points (the canonical scene:
// 21 clusters, the two largest ~50k points each -> 0.55 ms for roipool on one CU).  These kernels split the ROWS
// evenly over the grid instead (total = offsets[nProposal] is read on the device, so the C ABI is unchanged): a thread
// keeps the running extremum of its channel for the segment it is in and flushes it with one atomic when the segment
// changes.  min / max are order independent -> bit-exact; the FIRST arg-max is recovered exactly by a second pass
// (atomicMin of the row index over the rows that attain the maximum).
#define SEG_FLAT_GRID 1024

__device__ __forceinline__ void seg_atomic_max_f32(float *addr, float v) {
    if (v >= 0.f) atomicMax((int *)addr, __float_as_int(v));
    else atomicMin((unsigned int *)addr, __float_as_uint(v));
}
__device__ __forceinline__ void seg_atomic_min_f32(float *addr, float v) {
    if (v >= 0.f) atomicMin((int *)addr, __float_as_int(v));
    else atomicMax((unsigned int *)addr, __float_as_uint(v));
}
__device__ __forceinline__ int seg_find(const int *__restrict__ offsets, int nProposal, int r) {
    int lo = 0, hi = nProposal - 1;   // largest p with offsets[p] <= r
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offsets[mid] <= r) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void seg_init_kernel(float *f, int *i, long long n, float fval, int ival) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    f[e] = fval;
    if (i) i[e] = ival;
}
template <bool IS_MAX>
__global__ __launch_bounds__(SEG_THREADS) void seg_minmax_flat_kernel(const float *__restrict__ inp,
                                                                     const int *__restrict__ offsets,
                                                                     float *__restrict__ out, int nProposal, int C) {
    __shared__ float red[SEG_THREADS];
    const int active = (SEG_THREADS / C) * C, rpp = active / C, t = threadIdx.x;
    const int c = t % C;
    const int total = offsets[nProposal], first = offsets[0];
    int per = (total - first + gridDim.x - 1) / gridDim.x;
    per = (per + rpp - 1) / rpp * rpp;
    const int R0 = first + blockIdx.x * per, R1 = min(total, R0 + per);
    if (R0 >= R1) return;                                   // uniform
    const int p0 = seg_find(offsets, nProposal, R0);
    if (offsets[p0 + 1] >= R1) {
        // the whole range lies in one segment (the common case: a few clusters hold most rows): reduce in the
        // workgroup and issue C atomics instead of one per thread -- hundreds of thousands of atomics on the same
        // P*C addresses serialise in L2
        float v = IS_MAX ? -INFINITY : INFINITY;
        if (t < active)
            for (int r = R0 + t / C; r < R1; r += rpp) {
                const float x = inp[(long long)r * C + c];
                if (IS_MAX ? (x > v) : (x < v)) v = x;
            }
        red[t] = v;
        __syncthreads();
        if (t < C) {
            float b = red[t];
            for (int k = t + C; k < active; k += C) { const float x = red[k]; if (IS_MAX ? (x > b) : (x < b)) b = x; }
            if (IS_MAX ? (b > -INFINITY) : (b < INFINITY)) {
                if (IS_MAX) seg_atomic_max_f32(&out[(long long)p0 * C + t], b); else seg_atomic_min_f32(&out[(long long)p0 * C + t], b);
            }
        }
        return;
    }
    if (t >= active) return;
    int r = R0 + t / C;
    if (r >= R1) return;
    int p = seg_find(offsets, nProposal, r);
    int pend = offsets[p + 1];
    float v = IS_MAX ? -INFINITY : INFINITY;
    bool any = false;
    for (; r < R1; r += rpp) {
        if (r >= pend) {
            if (any) { if (IS_MAX) seg_atomic_max_f32(&out[(long long)p * C + c], v); else seg_atomic_min_f32(&out[(long long)p * C + c], v); }
            while (r >= pend) { p++; pend = offsets[p + 1]; }
            v = IS_MAX ? -INFINITY : INFINITY; any = false;
        }
        const float x = inp[(long long)r * C + c];
        if (IS_MAX ? (x > v) : (x < v)) { v = x; any = true; }
    }
    if (any) { if (IS_MAX) seg_atomic_max_f32(&out[(long long)p * C + c], v); else seg_atomic_min_f32(&out[(long long)p * C + c], v); }
}
// both extrema of the rows inp[gidx[2r + 1]] per segment in one pass (same structure as above)
__global__ __launch_bounds__(SEG_THREADS) void seg_minmax_gather_kernel(const float *__restrict__ inp, const int *__restrict__ gidx,
                                                                       const int *__restrict__ offsets, float *__restrict__ omin,
                                                                       float *__restrict__ omax, int nProposal, int C) {
    __shared__ float rmin[SEG_THREADS], rmax[SEG_THREADS];
    const int active = (SEG_THREADS / C) * C, rpp = active / C, t = threadIdx.x;
    const int c = t % C;
    const int total = offsets[nProposal], first = offsets[0];
    int per = (total - first + gridDim.x - 1) / gridDim.x;
    per = (per + rpp - 1) / rpp * rpp;
    const int R0 = first + blockIdx.x * per, R1 = min(total, R0 + per);
    if (R0 >= R1) return;                                   // uniform
    const int p0 = seg_find(offsets, nProposal, R0);
    if (offsets[p0 + 1] >= R1) {
        float lo = INFINITY, hi = -INFINITY;
        if (t < active)
            for (int r = R0 + t / C; r < R1; r += rpp) {
                const float x = inp[(long long)gidx[(long long)r * 2 + 1] * C + c];
                if (x < lo) lo = x;
                if (x > hi) hi = x;
            }
        rmin[t] = lo; rmax[t] = hi;
        __syncthreads();
        if (t < C) {
            float a = rmin[t], b = rmax[t];
            for (int k = t + C; k < active; k += C) { if (rmin[k] < a) a = rmin[k]; if (rmax[k] > b) b = rmax[k]; }
            if (a < INFINITY) seg_atomic_min_f32(&omin[(long long)p0 * C + t], a);
            if (b > -INFINITY) seg_atomic_max_f32(&omax[(long long)p0 * C + t], b);
        }
        return;
    }
    if (t >= active) return;
    int r = R0 + t / C;
    if (r >= R1) return;
    int p = seg_find(offsets, nProposal, r);
    int pend = offsets[p + 1];
    float lo = INFINITY, hi = -INFINITY;
    bool any = false;
    for (; r < R1; r += rpp) {
        if (r >= pend) {
            if (any) { seg_atomic_min_f32(&omin[(long long)p * C + c], lo); seg_atomic_max_f32(&omax[(long long)p * C + c], hi); }
            while (r >= pend) { p++; pend = offsets[p + 1]; }
            lo = INFINITY; hi = -INFINITY; any = false;
        }
        const float x = inp[(long long)gidx[(long long)r * 2 + 1] * C + c];
        if (x < lo) lo = x;
        if (x > hi) hi = x;
        any = true;
    }
    if (any) { seg_atomic_min_f32(&omin[(long long)p * C + c], lo); seg_atomic_max_f32(&omax[(long long)p * C + c], hi); }
}
// out[r] = [cluster, trunc((coords[point] - mean[cluster]) * scale[cluster] + offset[cluster])] as int64: the per-point part of
// `clusters_voxelization` (model/pointgroup.py:141-166: subtract the mean, scale, shift, `.long()`, concatenate the cluster id),
// every fp32 operation rounded separately like the elementwise library kernels it replaces
__global__ void cluster_transform_kernel(const float *__restrict__ coords, const int *__restrict__ cidx, const float *__restrict__ mean,
                                         const float *__restrict__ scale, const float *__restrict__ offset, long long *__restrict__ out,
                                         long long S) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S) return;
    const int cl = cidx[r * 2], pt = cidx[r * 2 + 1];
    const float sc = scale[cl];
    long long o[4];
    o[0] = cl;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v = __fadd_rn(__fmul_rn(__fsub_rn(coords[(long long)pt * 3 + k], mean[(long long)cl * 3 + k]), sc), offset[(long long)cl * 3 + k]);
        o[k + 1] = (long long)v;
    }
    long long *op = out + r * 4;
    op[0] = o[0]; op[1] = o[1]; op[2] = o[2]; op[3] = o[3];
}

// The per-cluster arithmetic of `clusters_voxelization` between the coordinate statistics and the per-point transform
// (model/pointgroup.py:146-165): size, centre, the scale that fits the cluster into fullscale^3 (capped), and the random
// placement offset -- ~30 elementwise library launches on (P,3) tensors, each ~4.7 us of pure launch latency.  Every fp32
// operation is rounded separately and in the library's form: `x / python_scalar` is a multiplication by the fp32 reciprocal
// of the scalar, `1 / x` a correctly rounded division, `python_scalar - x` one subtraction.
__global__ void cluster_norm_kernel(const float *__restrict__ mean, const float *__restrict__ raw_min, const float *__restrict__ raw_max,
                                    int P, float fullscale, float scale_cap, float r00, float r01, float r02, float r10, float r11,
                                    float r12, float *__restrict__ size, float *__restrict__ center, float *__restrict__ cscale,
                                    float *__restrict__ offset) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float inv_full = __fdiv_rn(1.f, fullscale);
    const float r0[3] = {r00, r01, r02}, r1[3] = {r10, r11, r12};
    float cmin[3], cmax[3], mx = -INFINITY;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float m = mean[p * 3 + k];
        cmin[k] = __fsub_rn(raw_min[p * 3 + k], m);
        cmax[k] = __fsub_rn(raw_max[p * 3 + k], m);
        const float ext = __fsub_rn(cmax[k], cmin[k]);
        size[p * 3 + k] = ext;
        center[p * 3 + k] = __fadd_rn(__fmul_rn(__fadd_rn(cmax[k], cmin[k]), 0.5f), m);
        const float d = __fmul_rn(ext, inv_full);
        nan |= d != d;
        mx = fmaxf(mx, d);
    }
    if (nan) mx = NAN;                                       // torch.max propagates NaN
    float sc = __fsub_rn(__fdiv_rn(1.f, mx), 0.01f);          // 1 / x - 0.01
    sc = (sc != sc) ? sc : fminf(sc, scale_cap);               // clamp(max=scale) keeps NaN
    cscale[p] = sc;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = __fmul_rn(cmin[k], sc), hi = __fmul_rn(cmax[k], sc);
        const float rng = __fsub_rn(hi, lo);
        const float room = __fsub_rn(fullscale, rng);
        float a = __fsub_rn(room, 0.001f), b = __fadd_rn(room, 0.001f);
        a = (a != a) ? a : fmaxf(a, 0.f);                      // clamp(min=0)
        b = (b != b) ? b : fminf(b, 0.f);                      // clamp(max=0)
        offset[p * 3 + k] = __fadd_rn(__fadd_rn(-lo, __fmul_rn(a, r0[k])), __fmul_rn(b, r1[k]));
    }
}
extern "C" int d3_cluster_norm_params(const float *mean, const float *raw_min, const float *raw_max, int P, float fullscale,
                                      float scale_cap, const float *rand6, float *size, float *center, float *cscale, float *offset,
                                      void *stream) {
    D3_CLEAR();
    if (P <= 0) return 0;
    if (!rand6) return D3_ERR_ARG;
    cluster_norm_kernel<<<(P + 255) / 256, 256, 0, d3_stream(stream)>>>(mean, raw_min, raw_max, P, fullscale, scale_cap, rand6[0], rand6[1],
                                                                         rand6[2], rand6[3], rand6[4], rand6[5], size, center, cscale, offset);
    D3_LAUNCH_CHECK();
    return 0;
}

// first row attaining the (already final) maximum: strict '>' in ascending row order == smallest such row
__global__ __launch_bounds__(SEG_THREADS) void roipool_arg_flat_kernel(const float *__restrict__ feats,
                                                                      const int *__restrict__ offsets,
                                                                      const float *__restrict__ out_feats,
                                                                      int *__restrict__ out_maxidx, int nProposal, int C) {
    __shared__ int redi[SEG_THREADS];
    const int active = (SEG_THREADS / C) * C, rpp = active / C, t = threadIdx.x;
    const int c = t % C;
    const int total = offsets[nProposal], first = offsets[0];
    int per = (total - first + gridDim.x - 1) / gridDim.x;
    per = (per + rpp - 1) / rpp * rpp;
    const int R0 = first + blockIdx.x * per, R1 = min(total, R0 + per);
    if (R0 >= R1) return;
    const int p0 = seg_find(offsets, nProposal, R0);
    if (offsets[p0 + 1] >= R1) {   // one segment: workgroup reduction, C atomics
        int a = 0x7FFFFFFF;
        if (t < active) {
            const float mx = out_feats[(long long)p0 * C + c];
            for (int r = R0 + t / C; r < R1; r += rpp)
                if (feats[(long long)r * C + c] == mx) { a = r; break; }
        }
        redi[t] = a;
        __syncthreads();
        if (t < C) {
            int b = redi[t];
            for (int k = t + C; k < active; k += C) b = min(b, redi[k]);
            if (b != 0x7FFFFFFF) atomicMin((unsigned int *)&out_maxidx[(long long)p0 * C + t], (unsigned int)b);
        }
        return;
    }
    if (t >= active) return;
    int r = R0 + t / C;
    if (r >= R1) return;
    int p = seg_find(offsets, nProposal, r);
    int pend = offsets[p + 1];
    float mx = out_feats[(long long)p * C + c];
    int a = -1;
    for (; r < R1; r += rpp) {
        if (r >= pend) {
            if (a >= 0) atomicMin((unsigned int *)&out_maxidx[(long long)p * C + c], (unsigned int)a);
            while (r >= pend) { p++; pend = offsets[p + 1]; }
            mx = out_feats[(long long)p * C + c]; a = -1;
        }
        if (a < 0 && feats[(long long)r * C + c] == mx) a = r;
    }
    if (a >= 0) atomicMin((unsigned int *)&out_maxidx[(long long)p * C + c], (unsigned int)a);
}

// ------------------------------------------------------------------------------ C entry points
static inline int seg_grid(int nProposal) { return nProposal < 65535 ? nProposal : 65535; }

extern "C" int d3_sec_mean(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    if (C <= 0 || C > SEG_THREADS) return D3_ERR_ARG;
    if (C <= 64) sec_mean_pc_kernel<MEANW_PROD><<<nProposal, 256, 2 * MEANW_PROD * MEANW_CHUNK * sizeof(float) + 256, d3_stream(stream)>>>(inp, offsets, out, nProposal, C, nullptr, 0);
    else sec_mean_kernel<<<seg_grid(nProposal), SEG_THREADS, 0, d3_stream(stream)>>>(inp, offsets, out, nProposal, C);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_sec_min(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    if (C <= 0 || C > SEG_THREADS) return D3_ERR_ARG;
    const long long n = (long long)nProposal * C;
    seg_init_kernel<<<(int)((n + 255) / 256), 256, 0, d3_stream(stream)>>>(out, nullptr, n, INFINITY, 0);
    seg_minmax_flat_kernel<false><<<SEG_FLAT_GRID, SEG_THREADS, 0, d3_stream(stream)>>>(inp, offsets, out, nProposal, C);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_sec_max(const float *inp, const int *offsets, float *out, int nProposal, int C, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    if (C <= 0 || C > SEG_THREADS) return D3_ERR_ARG;
    const long long n = (long long)nProposal * C;
    seg_init_kernel<<<(int)((n + 255) / 256), 256, 0, d3_stream(stream)>>>(out, nullptr, n, -INFINITY, 0);
    seg_minmax_flat_kernel<true><<<SEG_FLAT_GRID, SEG_THREADS, 0, d3_stream(stream)>>>(inp, offsets, out, nProposal, C);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_roipool_fp(const float *feats, const int *proposals_offset, float *output_feats,
                             int *output_maxidx, int nProposal, int C, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    if (C <= 0 || C > SEG_THREADS) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    const long long n = (long long)nProposal * C;
    seg_init_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(output_feats, output_maxidx, n, -INFINITY, -1);
    seg_minmax_flat_kernel<true><<<SEG_FLAT_GRID, SEG_THREADS, 0, s>>>(feats, proposals_offset, output_feats, nProposal, C);
    roipool_arg_flat_kernel<<<SEG_FLAT_GRID, SEG_THREADS, 0, s>>>(feats, proposals_offset, output_feats, output_maxidx, nProposal, C);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_roipool_bp(float *d_feats, const int *proposals_offset, const int *output_maxidx,
                             const float *d_output_feats, int nProposal, int C, void *stream) {
    D3_CLEAR();
    (void)proposals_offset;
    long long total = (long long)nProposal * C;
    if (total <= 0) return 0;
    int blocks = (int)((total + 255) / 256);
    roipool_bp_kernel<<<blocks, 256, 0, d3_stream(stream)>>>(d_feats, output_maxidx, d_output_feats, total, C);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_get_iou(const int *proposals_idx, const int *proposals_offset, const int64_t *instance_labels,
                          const int *instance_pointnum, float *proposals_iou, int nInstance, int nProposal,
                          void *stream) {
    D3_CLEAR();
    if (nProposal <= 0 || nInstance <= 0) return 0;
    get_iou_kernel<<<seg_grid(nProposal), SEG_THREADS, 0, d3_stream(stream)>>>(
        proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance, nProposal);
    D3_LAUNCH_CHECK();
    return 0;
}

// ---- cluster normalisation helpers of PointGroup.clusters_voxelization (model/pointgroup.py:125-178): the three passes over
// the S (cluster, point) pairs without the gathered / subtracted / scaled (S, 3) temporaries of the library-op form.
// clusters_idx (S,2) int32 [cluster, point]; offsets (P+1); coords (N,3).
// q[e, ch] = coords[point of pair e, ch] / (points of e's cluster): the addends of the clusters' mean chains, gathered and divided
// (IEEE) by the whole chip -- the chain kernel then streams contiguous rows.  Gathering inside it (round 3: "without the (S, 3)
// copy") made its few workgroups wait for their compute unit's miss queue: a chunk of 320 rows is 320 scattered 12-byte reads, ~15 us
// per round of six chunks against the ~5 us the chain needs for them (258 -> ~130 us per launch for the canonical 33 k-point floors).
__global__ void cluster_quot_kernel(const float *__restrict__ coords, const int *__restrict__ clusters_idx, const int *__restrict__ offsets,
                                    float *__restrict__ q, long long S, int nProposal) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * 3) return;
    const long long e = t / 3;
    const int ch = (int)(t - e * 3);
    const int pt = clusters_idx[e * 2 + 1];
    int lo = 0, hi = nProposal;                    // the segment of pair e: largest c with offsets[c] <= e (the offsets define the
    while (hi - lo > 1) { const int m = (lo + hi) >> 1; if ((long long)offsets[m] <= e) lo = m; else hi = m; }      // segments, as for sec_mean)
    const int c = lo;
    const float count = (float)(offsets[c + 1] - offsets[c]);
    q[t] = __fdiv_rn(coords[(long long)pt * 3 + ch], count);
}

static int cluster_mean_launch(const float *inp, const int *offsets, float *mean, int nProposal, const int *gidx, int prediv, hipStream_t s) {
    // (segments of tens of thousands of points -- a floor, a wall -- set this launch's time: the 6-producer instance: 8 waves, wave 4 idle)
    constexpr int NP = 6;
    const size_t lds = (size_t)2 * NP * MEANW_CHUNK * sizeof(float) + 256;
    static bool attr_done[64] = {false};
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64 || !attr_done[dev_id]) {
        D3_CHECK(hipFuncSetAttribute((const void *)sec_mean_pc_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev_id >= 0 && dev_id < 64) attr_done[dev_id] = true;
    }
    sec_mean_pc_kernel<NP><<<nProposal, MEANW_WAVES(NP) * 64, lds, s>>>(inp, offsets, mean, nProposal, 3, gidx, prediv);
    return 0;
}

static int cluster_minmax_launch(const float *coords, const int *clusters_idx, const int *offsets, float *cmin, float *cmax, int nProposal, hipStream_t s);

extern "C" int d3_cluster_coords_stats(const float *coords, const int *clusters_idx, const int *offsets, float *mean, float *cmin,
                                       float *cmax, int nProposal, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    hipStream_t s = d3_stream(stream);
    int rc = cluster_mean_launch(coords, offsets, mean, nProposal, clusters_idx, 0, s);
    if (rc) return rc;
    return cluster_minmax_launch(coords, clusters_idx, offsets, cmin, cmax, nProposal, s);
}

// the same with S = the number of (cluster, point) pairs and S * 3 floats of scratch: the chains' addends are staged by
// cluster_quot_kernel (bit-identical results: the same IEEE quotients added in the same order)
extern "C" size_t d3_cluster_coords_stats_ws_bytes(long long S) { return (size_t)(S > 0 ? S : 1) * 3 * sizeof(float); }
extern "C" int d3_cluster_coords_stats2(const float *coords, const int *clusters_idx, const int *offsets, long long S, float *mean, float *cmin,
                                        float *cmax, int nProposal, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (nProposal <= 0) return 0;
    if (S < 0 || ws == nullptr || ws_bytes < d3_cluster_coords_stats_ws_bytes(S)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    float *q = (float *)ws;
    if (S > 0) cluster_quot_kernel<<<(int)((S * 3 + 255) / 256), 256, 0, s>>>(coords, clusters_idx, offsets, q, S, nProposal);
    int rc = cluster_mean_launch(q, offsets, mean, nProposal, nullptr, 1, s);
    if (rc) return rc;
    return cluster_minmax_launch(coords, clusters_idx, offsets, cmin, cmax, nProposal, s);
}

static int cluster_minmax_launch(const float *coords, const int *clusters_idx, const int *offsets, float *cmin, float *cmax, int nProposal, hipStream_t s) {
    const long long n = (long long)nProposal * 3;
    seg_init_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(cmin, nullptr, n, INFINITY, 0);
    seg_init_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(cmax, nullptr, n, -INFINITY, 0);
    seg_minmax_gather_kernel<<<SEG_FLAT_GRID, SEG_THREADS, 0, s>>>(coords, clusters_idx, offsets, cmin, cmax, nProposal, 3);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_cluster_transform(const float *coords, const int *clusters_idx, const float *mean, const float *scale,
                                    const float *offset, long long *out, long long S, void *stream) {
    D3_CLEAR();
    if (S <= 0) return 0;
    cluster_transform_kernel<<<(int)((S + 255) / 256), 256, 0, d3_stream(stream)>>>(coords, clusters_idx, mean, scale, offset, out, S);
    D3_LAUNCH_CHECK();
    return 0;
}
