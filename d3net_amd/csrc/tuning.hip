// tuning.hip -- the ONE place where libd3hip.so reads its environment.
//
// Every measurement / test switch of the library lives in this table.  The environment is parsed
// exactly once, at the first d3_tune() of the process; launch paths read an array slot, never getenv().  Tests and the
// A/B tools flip a switch at run time through d3_tuning_set() (include/d3hip.h) instead of mutating the environment.
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {
struct Entry { const char *name; int dflt; };
// order == enum D3Tune (common.h)
const Entry kTable[D3T_COUNT] = {
    {"D3_ATTN_SCALAR", 0},          // 1: round-1 scalar attention kernels (cross-check)
    {"D3_BFS_NO_STAR", 0},          // 1: force the BFS level loop for every cluster (tests)
    {"D3_BFS_DEBUG", 0},            // 1: clustering debug dumps
    {"D3_EC_KSPLIT", 1},            // 0: EdgeConv weight gradients as one problem
    {"D3_WG3", 1},                  // 0: second-generation weight-gradient kernel
    {"D3_WG2_TR", 1},               // 0: first LDS staging scheme of the second-generation weight gradient
    {"D3_GRAD_BF16", 1},            // 0: every gradient buffer in fp32
    {"D3_SIDE_MIN_ROWS", 32768},    // level-0 rows from which the weight gradients run on the side stream
    {"D3_RED_TAIL", 5},             // flush the batched weight-gradient reduction when this many convolutions are left
    {"D3_VOX_ROWS", 1},             // 0: thread-per-element input voxelisation
    {"D3_C2_NW16_KB", 24},          // packed weights of at least this many KB: 16 waves share one LDS copy (1 << 20: never)
    {"D3_BQ_GRID", 1},              // 0: padded ball query by the ordered chunk scan instead of the cell grid
    {"D3_C2_STATIC", 1},            // 0: never the statically shaped convolution instances (K = 27, 16 / 32 / 48 / 64 input channels)
    {"D3_GRU_RT1", 1},              // 0: fused GRU cell with two 16-row tiles per workgroup for 17..64 rows
    {"D3_HG_RT1", 1},               // 0: K-split GEMM with two 16-row tiles per workgroup for 17..32 rows
    {"D3_GRU4", 1},                 // 0: fused GRU cell on 16 hidden units x 3 gate tiles per workgroup (rounds 1-2)
    {"D3_C2_INTERLEAVE", 1},        // 0: every convolution workgroup walks its own contiguous tile range (rounds 1-3) instead of the XCD's workgroups sweeping one window together
    {"D3_BN_FUSED_ROWS", 16384},    // BatchNorm over at most this many rows: finalize + apply (forward) / final + apply (backward) in one launch (0: never)
    {"D3_KMAP16", 1},               // 0: the executor's K = 27 convolutions read the dense int32 kernel maps only
    {"D3_BN_FUSED_BIG", 1},         // 0: only BatchNorms of at most D3_BN_FUSED_ROWS rows run as one launch; the big levels keep finalize + apply
    {"D3_HG_CLASS_SPLIT", 1},       // 0: a batched heads GEMM launch always runs the kernel its largest problem asks for (rounds 1-3)
    {"D3_HG_SPLITK", 256},          // largest number of 16 x 16 output tiles of a deep (K >= 8192) heads GEMM whose reduction is cut over 4 workgroups; 0: never
    {"D3_BN_PART2", 1},             // 0: BatchNorm launches reduce the producer's whole per-workgroup partial table (rounds 1-4) instead of the 16-row fp64 second-level table
    {"D3_CL_HOOK", 2},              // 0: the clustering's union-find starts from singletons (rounds 1-4); 1: one hook per node under a smaller-index neighbour first (ECL-CC init); 2 (default): + the hooked trees flattened before the unions, so that most edges find parent[i] == parent[j] with two loads and no walk (speaker step 16.59 -> 16.42 ms in-process, gpurun_out/r05_j32)
    {"D3_CL_SPEC", 1},              // 0: d3_bfs_cluster_run waits for the cluster counts before it enqueues the fill (count_ex + fill2); 1: the fill is enqueued behind the count kernels with its sizes read on the device, the host waits for the counts while the fill already runs
    {"D3_TD_FUSE_GATES", 1},        // 0: the captioner's backward step keeps its two GRU gate kernels (rounds 2-4: 6 dependent launches per step) instead of running them as epilogues of the GEMMs that complete their input (4 launches)
    {"D3_SIDE2", 2},                // 1: weight gradients whose dy buffer is later accumulated into in place (the caller's stream has to wait for them) run on a SECOND side stream: they no longer queue behind the other weight gradients; 2 (default): all weight gradients alternate between the two streams (speaker step 17.71 -> 17.49 ms in-process, mode 1: 17.57; detector step inside the noise: gpurun_out/r05_j17); 0: one side stream (rounds 1-4)
    {"D3_SORT_ONESWEEP_MIN", 65536}, // pair sorts of at least this many items take rocPRIM's Onesweep radix path (requested bits only, 8 per pass) instead of its default block sort + merge passes (~35 launches up to 2^20 items whatever the key width); 0x7fffffff: never
    {"D3_BQ_HALF", 1},              // 0: the cell-grid ball query runs one WAVE per query point (rounds 3-4); 1: two queries per wave (32 lanes each: 27 probes, up to 64 candidates as two elements per lane, bitonic order inside the half)
    {"D3_C3", 1},                   // 0: the K = 27 convolutions of the big levels never run spconv_fwd3_kernel (lane table, round 6) -- A/B against spconv_fwd2_kernel
};
std::atomic<int> g_val[D3T_COUNT];
std::once_flag g_once;

void parse_once() {
    for (int i = 0; i < D3T_COUNT; i++) {
        const char *e = getenv(kTable[i].name);
        g_val[i].store((e && e[0]) ? atoi(e) : kTable[i].dflt, std::memory_order_relaxed);
    }
}
int find(const char *name) {
    for (int i = 0; name && i < D3T_COUNT; i++)
        if (!strcmp(name, kTable[i].name)) return i;
    return -1;
}
}  // namespace

int d3_tune(int key) {
    std::call_once(g_once, parse_once);
    return g_val[key].load(std::memory_order_relaxed);
}

extern "C" int d3_tuning_set(const char *name, int value) {
    std::call_once(g_once, parse_once);
    const int i = find(name);
    if (i < 0) return D3_ERR_ARG;
    g_val[i].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int d3_tuning_get(const char *name, int *value) {
    std::call_once(g_once, parse_once);
    const int i = find(name);
    if (i < 0 || !value) return D3_ERR_ARG;
    *value = g_val[i].load(std::memory_order_relaxed);
    return 0;
}

extern "C" int d3_tuning_count(void) { return D3T_COUNT; }

extern "C" const char *d3_tuning_name(int i) { return (i >= 0 && i < D3T_COUNT) ? kTable[i].name : nullptr; }
