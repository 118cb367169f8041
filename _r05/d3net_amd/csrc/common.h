// common.h -- shared helpers for the gfx950 kernels of libd3hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/d3hip.h"

#define D3_WAVE 64

#define D3_CHECK(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return (int)_e;            \
    } while (0)

#define D3_LAUNCH_CHECK()                                \
    do {                                                 \
        hipError_t _e = hipGetLastError();               \
        if (_e != hipSuccess) return (int)_e;            \
    } while (0)

// drop a stale (sticky) error left by an earlier, unrelated HIP call so that launch checks report our own
#define D3_CLEAR() (void)hipGetLastError()

static inline hipStream_t d3_stream(void *s) { return (hipStream_t)s; }

static inline size_t d3_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve a typed region out of a workspace; returns nullptr when it does not fit
struct D3Carver {
    char *base; size_t cap; size_t off;
    D3Carver(void *ws, size_t bytes) : base((char *)ws), cap(bytes), off(0) {}
    template <typename T> T *take(size_t count) {
        size_t bytes = d3_align(count * sizeof(T));
        T *p = (T *)(base + off);
        off += bytes;
        return p;
    }
    bool ok() const { return base != nullptr && off <= cap; }
};

__device__ __forceinline__ int d3_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ unsigned long long d3_lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// measurement / test switches: ONE table parsed once from the environment (tuning.hip); launch paths read a slot
enum D3Tune { D3T_ATTN_SCALAR, D3T_BFS_NO_STAR, D3T_BFS_DEBUG, D3T_EC_KSPLIT, D3T_HG_TILED, D3T_C2_GRIDCAP, D3T_WG3, D3T_WG3_R, D3T_WG3_S,
              D3T_WG2_TR, D3T_LASTBLOCK_FINALIZE, D3T_GRAD_BF16, D3T_SIDE_PRIO, D3T_SIDE_MIN_ROWS, D3T_RED_TAIL, D3T_VOX_ROWS,
              D3T_C2_WLDS_KB, D3T_C2_NW16_KB, D3T_BQ_GRID, D3T_SIDE_OP_ROWS, D3T_LASTBLOCK_ROWS, D3T_C2_STATIC, D3T_GRU_RT1, D3T_HG_RT1, D3T_GRU4, D3T_C2_INTERLEAVE, D3T_BN_FUSED_ROWS, D3T_KMAP16, D3T_HG_BF16X3, D3T_BN_FUSED_BIG, D3T_HG_CLASS_SPLIT, D3T_HG_SPLITK, D3T_BFS3, D3T_BN_PART2, D3T_CL_HOOK, D3T_ACT_GRAD_BF16, D3T_CL_SPEC, D3T_TD_FUSE_GATES, D3T_UNSAFE_NO_HAZARD_WAIT, D3T_SIDE2, D3T_SORT_ONESWEEP_MIN, D3T_BQ_HALF, D3T_C2_KSPLIT, D3T_C2_COMPACT, D3T_COUNT };
int d3_tune(int key);

// exclusive int32 scan / total via rocPRIM (implemented in scan_sort.hip)
size_t d3_scan_temp_bytes(int n);
int d3_exclusive_scan_i32(const int *in, int *out, int n, void *temp, size_t temp_bytes, hipStream_t s);
size_t d3_sort_pairs_temp_bytes(int n);
// stable ascending sort of (key,val) int32 pairs on the low `bits` bits of key
int d3_sort_pairs_i32(const int *kin, int *kout, const int *vin, int *vout, int n, int bits, void *temp,
                      size_t temp_bytes, hipStream_t s);
size_t d3_sort_pairs_u64_temp_bytes(int n);
int d3_sort_pairs_u64(const unsigned long long *kin, unsigned long long *kout, const int *vin, int *vout, int n, void *temp,
                      size_t temp_bytes, hipStream_t s);

// internal (C++ linkage, spconv2.hip): the 16-bit delta form of the kernel map handed to the NEXT d3_spconv_fwd2* / d3_spconv_wgrad2
// call of this thread (consumed by that call; tbl16 = NULL: dense table only).  The caller has VALIDATED the table (every delta
// fits: d3_kmap_k3_pack16's flag read on the host); ok16 is unused by the kernels.  csrc/unet.hip sets it per K = 27 launch.
void d3_spconv_next_tbl16(const void *tbl16, const int *ok16);
void d3_spconv_next_part2(double *part2);      // spconv2.hip: second-level BatchNorm partial table of this thread's next forward / data-gradient call
