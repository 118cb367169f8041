// scan_sort.hip -- device-wide scan / stable radix sort used by the indexing ops.
// rocPRIM (AMD's native primitives library) is used directly.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

size_t d3_scan_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const int *)nullptr, (int *)nullptr, 0, (size_t)(n > 0 ? n : 1),
                            rocprim::plus<int>());
    return d3_align(bytes + 256);
}

int d3_exclusive_scan_i32(const int *in, int *out, int n, void *temp, size_t temp_bytes, hipStream_t s) {
    if (n <= 0) return 0;
    size_t need = 0;
    D3_CHECK(rocprim::exclusive_scan(nullptr, need, in, out, 0, (size_t)n, rocprim::plus<int>(), s));
    if (need > temp_bytes) return D3_ERR_WORKSPACE;
    D3_CHECK(rocprim::exclusive_scan(temp, need, in, out, 0, (size_t)n, rocprim::plus<int>(), s));
    return 0;
}

// rocPRIM's default sorts up to 2^20 items by block sort + merge passes whatever the key width (~35 launches for the 600 k points
// of a clustering branch, ~200 us); the Onesweep radix path walks only the requested bits, 8 per pass (round 5: 22-bit cell-slot
// keys = 3 passes behind one histogram launch).  Merge sort stays for the small inputs, where its few passes are cheaper.
using D3SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;      // (never merge: the caller decides by size, D3_SORT_ONESWEEP_MIN)

size_t d3_sort_pairs_temp_bytes(int n) {
    size_t bytes = 0, b2 = 0;
    (void)rocprim::radix_sort_pairs<D3SortConfig>(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                              (int *)nullptr, (size_t)(n > 0 ? n : 1), 0, 32);
    (void)rocprim::radix_sort_pairs(nullptr, b2, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                              (int *)nullptr, (size_t)(n > 0 ? n : 1), 0, 32);
    return d3_align((bytes > b2 ? bytes : b2) + 256);
}

int d3_sort_pairs_i32(const int *kin, int *kout, const int *vin, int *vout, int n, int bits, void *temp,
                      size_t temp_bytes, hipStream_t s) {
    if (n <= 0) return 0;
    if (bits < 1) bits = 1;
    if (bits > 31) bits = 31;
    size_t need = 0;
    if (n >= d3_tune(D3T_SORT_ONESWEEP_MIN)) {
        D3_CHECK(rocprim::radix_sort_pairs<D3SortConfig>(nullptr, need, kin, kout, vin, vout, (size_t)n, 0, (unsigned)bits, s));
        if (need > temp_bytes) return D3_ERR_WORKSPACE;
        D3_CHECK(rocprim::radix_sort_pairs<D3SortConfig>(temp, need, kin, kout, vin, vout, (size_t)n, 0, (unsigned)bits, s));
        return 0;
    }
    D3_CHECK(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, (size_t)n, 0, (unsigned)bits, s));
    if (need > temp_bytes) return D3_ERR_WORKSPACE;
    D3_CHECK(rocprim::radix_sort_pairs(temp, need, kin, kout, vin, vout, (size_t)n, 0, (unsigned)bits, s));
    return 0;
}

// stable ascending sort of (64-bit key, int32 value) pairs (the ball query's cell grid: points ordered by cell, ascending
// point index inside a cell because the sort is stable)
size_t d3_sort_pairs_u64_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int *)nullptr, (int *)nullptr, (size_t)(n > 0 ? n : 1), 0, 64);
    return d3_align(bytes + 256);
}

int d3_sort_pairs_u64(const unsigned long long *kin, unsigned long long *kout, const int *vin, int *vout, int n, void *temp,
                      size_t temp_bytes, hipStream_t s) {
    if (n <= 0) return 0;
    size_t need = 0;
    D3_CHECK(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, (size_t)n, 0, 64, s));
    if (need > temp_bytes) return D3_ERR_WORKSPACE;
    D3_CHECK(rocprim::radix_sort_pairs(temp, need, kin, kout, vin, vout, (size_t)n, 0, 64, s));
    return 0;
}
