// cluster.hip -- same-label connected components over ball-query lists, in the reference's
// FIFO-BFS order, entirely on the device (gfx950).
//
// Replaces PG_OP.bfs_cluster (reference: lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cpp:28-112),
// which runs single-threaded on the host behind a D2H copy of the neighbour lists (up to
// n*300*4 B), twice per forward (model/pointgroup.py:296-305).
//
// Reference semantics: for i = 0..n-1, if i is unvisited, FIFO-BFS from i over the DIRECTED
// edges i -> list(i) restricted to equal semantic labels; keep the visited set if it has
// >= threshold points; emit (cluster_id, point) in visitation order.  Lists are capped at
// 1000 entries (smallest indices first), so edges are symmetric except where a list was cut.
//
// Parallel formulation (checked against the sequential oracle on the CPU by
// tests/bfs_parallel_model.py, the numpy model of exactly these steps):
//   1. owner(j) = smallest index that reaches j.  (The smallest ancestor is never claimed by an
//      earlier seed, so it is a seed, and it is the first seed that reaches j.)
//      a. lock-free min-hooking union-find over edges whose two lists are both complete
//         (len < 1000) -- such edges are mutual, so a tree is a strongly connected set and its
//         root is its smallest index;
//      b. push labels root(i) -> root(j) over ALL edges with atomicMin until a fixpoint
//         (needed only across truncated lists; 1-3 passes in practice).
//   2. sizes by owner, keep >= threshold, cluster ids / offsets by exclusive scans in seed order.
//   3. one 1024-thread workgroup per kept cluster replays the BFS level-synchronously: the queue
//      segment of the level is expanded in three passes -- A: first discoverer of every node
//      (atomicMin of the parent's queue position), B: children per parent, scan, C: children
//      written in (parent position, list order) -- which is the FIFO order.
// All passes stream the neighbour lists: bytes = 4*nActive per pass + 12*n, HBM/L2 bound.
#include "common.h"
#include "prof.h"
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#define CL_CAP 1000
#define CL_BFS_THREADS 1024
#define CL_INF 0x7FFFFFFF

struct ClWs {
    int *parent;   // n  union-find forest (phase 1), then root id per node
    int *lab;      // n  label per root
    int *own;      // n  owner (seed) per node
    int *sizes;    // n  points per owner
    int *flag;     // n  1 if owner kept
    int *cid;      // n  exclusive scan of flag
    int *ksz;      // n  kept size
    int *koff;     // n  exclusive scan of ksz
    int *seeds;    // n  seed of cluster c
    int *par;      // n  BFS: queue position of the first discoverer
    int *queue;    // n  BFS queues (cluster c at koff[seed])
    int *fcnt;     // n  BFS: list start of every queued node
    int *klen;     // n  list length of a node of a kept cluster, else 0
    int *estart;   // n  exclusive scan of klen: the node's list start in the (compact) record array
    int *qln;      // n  BFS: list length of every queued node
    int *lid;      // n  BFS (record kernel): dense id of a node inside its cluster (any bijection)
    int *lcnt;     // n  per-owner counter behind lid
    int *star;     // n  star[o] = 1: the kept cluster of owner o is the seed's own list (no level loop needed)
    int *scalars;  // [0]=changed [1]=nCluster [2]=sumNPoint
    int4 *ninfo;   // n  (owner or -1 when not in a level-loop cluster, dense id, record start, list length): ONE gather per list entry
    void *temp; size_t temp_bytes;
};

static bool cl_carve(void *ws, size_t ws_bytes, int n, ClWs &w) {
    D3Carver c(ws, ws_bytes);
    size_t nn = (size_t)(n > 0 ? n : 1);
    w.parent = c.take<int>(nn); w.lab = c.take<int>(nn); w.own = c.take<int>(nn); w.sizes = c.take<int>(nn);
    w.flag = c.take<int>(nn); w.cid = c.take<int>(nn); w.ksz = c.take<int>(nn); w.koff = c.take<int>(nn);
    w.seeds = c.take<int>(nn); w.par = c.take<int>(nn); w.queue = c.take<int>(nn); w.fcnt = c.take<int>(nn);
    w.qln = c.take<int>(nn);
    w.lid = c.take<int>(nn); w.lcnt = c.take<int>(nn); w.star = c.take<int>(nn);
    w.klen = c.take<int>(nn); w.estart = c.take<int>(nn);
    w.scalars = c.take<int>(64);
    w.ninfo = c.take<int4>(nn);
    w.temp_bytes = d3_scan_temp_bytes(n);
    w.temp = c.take<char>(w.temp_bytes);
    return ws != nullptr && c.ok();
}
extern "C" size_t d3_bfs_cluster_ws_bytes(int n) {
    D3Carver c(nullptr, 0);
    size_t nn = (size_t)(n > 0 ? n : 1);
    for (int i = 0; i < 18; i++) c.take<int>(nn);
    c.take<int>(64);
    c.take<int4>(nn);
    c.take<char>(d3_scan_temp_bytes(n));
    return c.off + 256;
}

// L1-bypassing load/store for words other waves update inside the same launch
__device__ __forceinline__ int ld_dev(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_dev(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void cl_init_kernel(int *parent, int *lab, int *sizes, int *par, int *pushed, int *lpush, int n, int *scalars) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = i; lab[i] = i; sizes[i] = 0; par[i] = CL_INF; pushed[i] = CL_INF; lpush[i] = CL_INF; }
    if (i < 8) scalars[i] = 0;
}

// Reads go through the cache: parent pointers only ever move to smaller ancestors, so a stale value is still an
// ancestor (the walk just takes an older path), and a stale "root" is caught by the atomicMin in cl_union, which
// returns the current parent.  L2-bypassing loads here made the walk a chain of memory-side round trips.
__device__ __forceinline__ int cl_find(int *parent, int x) {
    int p = parent[x];
    while (p != x) {
        int gp = parent[p];
        if (gp != p) parent[x] = gp;  // path halving: any value ever stored in parent[x] is an ancestor of x, and x is not a
                                      // root here, so a plain (racy) store can only trade one valid ancestor for another
        x = p; p = gp;
    }
    return x;
}
__device__ __forceinline__ void cl_union(int *parent, int a, int b) {
    for (;;) {
        a = cl_find(parent, a); b = cl_find(parent, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        int old = atomicMin(&parent[b], a);  // hook the larger root under the smaller
        if (old == b) return;                // b was still a root: done
        b = old;                             // b had been hooked meanwhile: keep uniting with its old parent
    }
}

// phase 1a, opening move (round 5; the initialisation of ECL-CC, Jaiganesh & Burtscher 2018): every node hooks itself under ONE
// smaller-index neighbour across a mutual edge before any union runs -- no atomics (a thread writes only its own entry), parent < child
// keeps the forest acyclic and every tree inside a true component.  On surfaces in scan order that alone builds most of each tree; the
// union pass below then finds most edges already inside one tree (two short walks, no atomic) instead of hooking root by root.
// Lists need not be sorted: any of the first CL_HOOK_TRY entries that qualifies will do (in ascending lists the smallest come first).
#define CL_HOOK_TRY 4
__global__ __launch_bounds__(256) void cl_hook_kernel(const int *__restrict__ sem, const int *__restrict__ idx,
                                                     const int *__restrict__ start_len, int n, int *parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int st = start_len[i * 2], ln = start_len[i * 2 + 1];
    if (ln >= CL_CAP) return;                      // a capped list's edges need not be mutual: left to the label push
    const int si = sem[i];
    int j[CL_HOOK_TRY];
#pragma unroll
    for (int e = 0; e < CL_HOOK_TRY; e++) j[e] = e < ln ? idx[st + e] : i;
    int sj[CL_HOOK_TRY], lj[CL_HOOK_TRY];
#pragma unroll
    for (int e = 0; e < CL_HOOK_TRY; e++) { sj[e] = sem[j[e]]; lj[e] = start_len[j[e] * 2 + 1]; }
#pragma unroll
    for (int e = 0; e < CL_HOOK_TRY; e++)
        if (j[e] < i && sj[e] == si && lj[e] < CL_CAP) { parent[i] = j[e]; return; }
}

// phase 1a: eight lanes per node walk its (complete, hence short) list, four edges per lane in flight: every edge is a
// chain of dependent gathers (neighbour id -> its label / list length -> the two finds), and a thread per node walked
// that chain once per edge.  scalars[3] is raised when any list is capped: only then does phase 1b have work.
#define CL_UG 8
__global__ __launch_bounds__(256) void cl_union_kernel(const int *__restrict__ sem, const int *__restrict__ idx,
                                                      const int *__restrict__ start_len, int n, int *parent, int *scalars) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid / CL_UG, sub = gid % CL_UG;
    const bool live = i < n;
    const int st = live ? start_len[i * 2] : 0, ln = live ? start_len[i * 2 + 1] : 0;
    if (__any(ln >= CL_CAP) && d3_lane() == 0) scalars[3] = 1;
    if (!live || ln >= CL_CAP) return;
    const int si = sem[i];
    for (int e0 = sub; e0 < ln; e0 += 4 * CL_UG) {
        int j[4], sj[4], lj[4], pj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = e0 + q * CL_UG; j[q] = e < ln ? idx[st + e] : i; }
        // (the neighbours' parent entries travel with their class / list length: after the hook + flatten opening almost every
        // edge joins two nodes that already point at the same root -- two equal words, no walk, no atomic.  Equal parents are the same
        // tree whatever other threads do meanwhile: an entry only ever moves to another ancestor of its node.)
        const int pi = parent[i];
#pragma unroll
        for (int q = 0; q < 4; q++) { sj[q] = sem[j[q]]; lj[q] = start_len[j[q] * 2 + 1]; pj[q] = parent[j[q]]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // the edge is mutual (both lists complete): handle it once, from its smaller endpoint
            if (j[q] <= i || sj[q] != si || lj[q] >= CL_CAP || pj[q] == pi) continue;
            cl_union(parent, i, j[q]);
        }
    }
}
__global__ void cl_flatten_kernel(int *parent, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // Cached loads (round 3; device-scope loads made every hop a memory-side round trip: 158 us for 600 k nodes): the unions
    // finished with the previous kernel, a root keeps parent[r] == r for the whole launch, and whatever another thread stores
    // meanwhile into a non-root entry is that entry's root -- a stale read is an older ancestor, the walk still ends at the root.
    int r = i;
    for (;;) { int p = parent[r]; if (p == r) break; r = p; }
    // every thread only writes its own entry with its root; roots keep parent[r]==r
    if (r != i) parent[i] = r;
}

__device__ __forceinline__ int cl_chase(const int *lab, int l) {
    for (;;) { int m = ld_dev(&lab[l]); if (m >= l) return l; l = m; }
}

// phase 1b: push labels over all edges; root[] == parent[] after flatten
// Round 5: the FILTERS run one THREAD per node, the list walks one WAVE per surviving node.  A wave per node (rounds 1-4) spent a
// wave launch and five dependent round trips (list extent, class, root, label chase, last pushed label) on every node only to find
// that almost none has anything to push -- the second and third sweep of a collapsed instance: 240 + 111 us for 600 k nodes.
#define CL_PUSH_SHORT 48
__global__ __launch_bounds__(256) void cl_push_kernel(const int *__restrict__ sem, const int *__restrict__ idx,
                                                     const int *__restrict__ start_len, int n,
                                                     const int *__restrict__ root, int *lab, int *pushed, int *lpush,
                                                     int *changed_flag, const int *__restrict__ capped_flag, int ascending, int minima_only) {
    if (*capped_flag == 0) return;   // no capped list: every edge is mutual and already united, the labels stay the roots
    const int lane = d3_lane();
    const int nthreads = (int)(gridDim.x * blockDim.x);
    bool changed = false;
    for (int base = (int)(blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += nthreads) {      // (wave-uniform)
        const int i = base + lane;
        // ---- filters, one node per lane
        bool want = false, cand = false;
        int st = 0, ln = 0, si = 0, ri = 0, li = 0, slot = 0;
        if (i < n) {
            st = start_len[i * 2]; ln = start_len[i * 2 + 1];
            // Opening sweep (ascending lists only): just the nodes without a smaller-index neighbour push -- the future seeds.  In a
            // collapsed instance (every list = its first 1000 members) that is ONE node, whose push settles all 1000 labels without
            // contention; the full sweep behind it then finds them settled through its cached filter read.  Without it every member
            // pushed its own index at all later members at once: 500 k contended atomicMin per instance, most of the sweep's time
            // (profiles/r02_q_cluster_timeline.txt: 2.1 ms).  Any sweep order reaches the same fixpoint.
            const bool skip = minima_only && (ln == 0 || idx[st] < i);
            if (!skip) {
                si = sem[i];
                ri = root[i];
                li = cl_chase(lab, ld_dev(&lab[ri]));
                if (li < ld_dev(&lab[ri])) atomicMin(&lab[ri], li);
                // Worklist: a node pushes again only when its own label got smaller since its last push -- every neighbour's label
                // was <= that value then and labels only decrease.  The verification sweep therefore walks only the lists of the
                // nodes the previous sweep changed (the first 1000 points of a collapsed instance, not all of them).
                if (pushed[i] != li) {
                    pushed[i] = li;
                    want = true;
                    // Shared lists (round 3): the cell-grid ball query hands every member of a clique cell the SAME list (start =
                    // leader * 1000).  A push of label l over a list by a node of class c settles every target of that class at <= l,
                    // so another node with the same list, the same class and a label >= l has nothing to add -- in an instance
                    // collapsed onto its centre that is all but one of its first 1000 members.  lpush[slot] = smallest label pushed
                    // so far over the list starting at slot * 1000 by a node of the slot owner's class.  (Keyed by the exact start:
                    // private lists have private keys, whatever the layout.)
                    if (ln > 0 && st % CL_CAP == 0) {
                        slot = st / CL_CAP;
                        cand = slot < n && sem[slot] == si;
                    }
                }
            }
        }
        // (the shared-list filter's atomic, aggregated: consecutive members of a collapsed instance carry the same (list, class, label)
        // -- the first group of equal lanes sends ONE lane to the counter; its other members would find its label there and skip, so
        // they skip.  64 same-address atomics per wave instruction on ~160 hot words were most of the productive sweep's time.)
        {
            const unsigned long long cm = __ballot(cand);
            if (cm) {
                const int L = (int)__builtin_ctzll(cm);
                const int ls = __shfl(slot, L), ll = __shfl(li, L), ss = __shfl(si, L);
                if (cand && slot == ls && li == ll && si == ss) {
                    if (lane == L) { if (atomicMin(&lpush[slot], li) <= li) want = false; }
                    else want = false;
                    cand = false;
                }
            }
            if (cand && atomicMin(&lpush[slot], li) <= li) want = false;
        }
        // ---- short lists (a surface's ~9 entries): every surviving lane walks its own, four entries in flight -- 64 nodes at once.
        // (One after the other on the whole wave, the 135 k floor nodes of the first full sweep cost 64 x three dependent round trips
        // per wave: 184 us.)  A target j <= li cannot be lowered whatever the list order (see below): checked per entry here.
        if (want && ln <= CL_PUSH_SHORT) {
            for (int e0 = 0; e0 < ln; e0 += 4) {
                int j[4], rj[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { ok[q] = e0 + q < ln; j[q] = ok[q] ? idx[st + e0 + q] : 0; }
#pragma unroll
                for (int q = 0; q < 4; q++) { ok[q] = ok[q] && j[q] > li && sem[j[q]] == si; rj[q] = ok[q] ? root[j[q]] : ri; }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!ok[q] || rj[q] == ri) continue;
                    if (lab[rj[q]] > li) { if (atomicMin(&lab[rj[q]], li) > li) changed = true; }
                }
            }
            want = false;
        }
        // ---- the surviving nodes' long lists (capped: 1000 entries), one after the other, on the whole wave
        unsigned long long todo = __ballot(want);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int wst = __shfl(st, src), wln = __shfl(ln, src), wsi = __shfl(si, src), wri = __shfl(ri, src), wli = __shfl(li, src);
            // A push can only lower the label of a target j > li: lab[root(j)] <= root(j) <= j at all times (a label starts as the
            // node's own index, a root is the smallest index of its tree, labels only decrease).  The lists are ascending
            // (ball query order), so the useless targets j <= li are a PREFIX: found with two 64-way probes instead of walked --
            // in a collapsed instance of m points whose lists all are its first 1000 members, a member's own rank of them.
            int e_first = 0;
            if (ascending && wln > 0) {      // (the caller vouches for ascending lists: D3_BFS_ASCENDING)
                const int p = (int)(((long long)lane * wln) >> 6);               // 64 probes, probe 0 = entry 0
                const unsigned long long gt = __ballot(idx[wst + p] > wli);
                if (gt == 0ull) {                                               // every probe <= li: only the tail behind the last probe is left
                    const int p63 = (int)((63ll * wln) >> 6);
                    const int q = p63 + lane;
                    const unsigned long long g2 = __ballot(q < wln && idx[wst + (q < wln ? q : 0)] > wli);
                    // (the last segment is at most ln/64 + 1 <= 17 entries long)
                    e_first = g2 ? p63 + (int)__builtin_ctzll(g2) : wln;
                } else {
                    const int f = (int)__builtin_ctzll(gt);                     // first probe > li; the boundary lies in (probe f-1, probe f]
                    const int lo = f == 0 ? 0 : (int)(((long long)(f - 1) * wln) >> 6);
                    const int hi = (int)(((long long)f * wln) >> 6);
                    const int q = lo + lane;
                    const unsigned long long g2 = __ballot(q <= hi && idx[wst + (q <= hi ? q : lo)] > wli);
                    e_first = g2 ? lo + (int)__builtin_ctzll(g2) : hi;
                }
            }
            // four edges per lane in flight: every edge is a chain of three dependent gathers (neighbour id -> its root ->
            // the root's label) and a capped list is 16 passes long
            for (int e0 = e_first + lane; e0 < wln; e0 += 256) {
                int j[4], rj[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const int e = e0 + q * 64; ok[q] = e < wln; j[q] = ok[q] ? idx[wst + e] : 0; }
#pragma unroll
                for (int q = 0; q < 4; q++) { ok[q] = ok[q] && sem[j[q]] == wsi; rj[q] = ok[q] ? root[j[q]] : wri; }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!ok[q] || rj[q] == wri) continue;
                    // The filter reads through the cache: labels only ever decrease, so a stale (larger) value can at worst let an
                    // atomicMin through that changes nothing -- it can never hide a needed update.  (An L2-bypassing load here,
                    // once per edge of a capped list, was most of this kernel's time.)
                    if (lab[rj[q]] > wli) { if (atomicMin(&lab[rj[q]], wli) > wli) changed = true; }
                }
            }
        }
    }
    if (__any(changed) && lane == 0) *changed_flag = 1;
}

__global__ void cl_owner_kernel(const int *root, const int *lab, int *own, int *sizes, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;   // tail lanes simply drop out of the ballots below
    int o = cl_chase(lab, lab[root[i]]);
    own[i] = o;
    // a wave holds only a few distinct owners: one atomic per distinct owner (leader = lowest lane of each group)
    unsigned long long todo = __ballot(1);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const int ol = __shfl(o, leader);
        const unsigned long long grp = __ballot(o == ol) & todo;
        if (lane == leader) atomicAdd(&sizes[ol], (int)__popcll(grp));
        todo &= ~grp;
    }
}
// scalars[5] <- 1 when some list is longer than cl_bfs3_kernel's key can number (B3_MAXLIST entries; the reference's ball query
// stops at 1000): read back with the counts, the fill then keeps the edge-parallel replay
__global__ void cl_keep_kernel(const int *sizes, int *flag, int *ksz, int n, int threshold, const int *__restrict__ start_len, int *scalars) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int k = sizes[i] >= threshold && sizes[i] > 0;
    flag[i] = k; ksz[i] = k ? sizes[i] : 0;
    if (start_len[i * 2 + 1] > 2047) scalars[5] = 1;
}
__global__ void cl_totals_kernel(const int *flag, const int *cid, const int *ksz, const int *koff, int n,
                                 int *scalars) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        scalars[1] = cid[n - 1] + flag[n - 1];
        scalars[2] = koff[n - 1] + ksz[n - 1];
    }
}

static int cl_count(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold, void *ws,
                    size_t ws_bytes, int *sumNPoint_host, int *nCluster_host, int flags, void *stream);
// The count phase reads back, with the counts, whether every list fits cl_bfs3_kernel's key (<= 2047 entries).  The fill uses
// that replay only for a workspace this thread's last count vouched for; anything else (a fill on another thread, a count that
// saw a longer list) keeps cl_bfs2_kernel, which has no such limit.
static thread_local const void *g_cl_checked_ws = nullptr;
static thread_local bool g_cl_short_lists = false;

extern "C" int d3_bfs_cluster_count(const int *semantic_label, const int *ball_query_idxs, const int *start_len,
                                    int n, int threshold, void *ws, size_t ws_bytes, int *sumNPoint_host,
                                    int *nCluster_host, void *stream) {
    return cl_count(semantic_label, ball_query_idxs, start_len, n, threshold, ws, ws_bytes, sumNPoint_host, nCluster_host, 0, stream);
}
// flags: D3_BFS_ASCENDING -- every list is in ascending index order (what ballquery_batch_p produces): the label push then
// skips, per node, the prefix of targets that cannot change
extern "C" int d3_bfs_cluster_count_ex(const int *semantic_label, const int *ball_query_idxs, const int *start_len,
                                       int n, int threshold, void *ws, size_t ws_bytes, int *sumNPoint_host,
                                       int *nCluster_host, int flags, void *stream) {
    return cl_count(semantic_label, ball_query_idxs, start_len, n, threshold, ws, ws_bytes, sumNPoint_host, nCluster_host, flags, stream);
}

// One iteration of the count phase, enqueued only: a pair of label-push sweeps (it == 0: also the union-find in front of them), owners,
// sizes, kept flags, cluster ids / offsets, and the copy of the scalars to `h` (6 ints; pinned memory for the asynchronous form).
static int cl_count_enqueue(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold, ClWs &w,
                            int asc, int it, int *h, hipStream_t s) {
    const int T = 256, nb = (n + T - 1) / T, nwb = (n + 3) / 4;
    if (it == 0) {
        cl_init_kernel<<<nb, T, 0, s>>>(w.parent, w.lab, w.sizes, w.par, w.klen, w.qln, n, w.scalars);   // (klen, qln: scratch until the fill)
        if (d3_tune(D3T_CL_HOOK) != 0) cl_hook_kernel<<<nb, T, 0, s>>>(semantic_label, ball_query_idxs, start_len, n, w.parent);
        if (d3_tune(D3T_CL_HOOK) == 2) cl_flatten_kernel<<<nb, T, 0, s>>>(w.parent, n);      // (trees flattened before the unions: most edges then compare two roots without a walk)
        cl_union_kernel<<<(int)(((long long)n * CL_UG + T - 1) / T), T, 0, s>>>(semantic_label, ball_query_idxs, start_len, n, w.parent, w.scalars);
        cl_flatten_kernel<<<nb, T, 0, s>>>(w.parent, n);
        D3_LAUNCH_CHECK();
    }
    // Label pushes in pairs, and the sizes / ids / offsets computed right behind them, all read back with ONE host round
    // trip: the usual case is one productive sweep plus the sweep that finds nothing left to do (the second one reports
    // through its own flag, scalars[4]); only when both sweeps still changed labels is the tail recomputed after more.
    const int npb = nb < 2048 ? nb : 2048;        // label push: a bounded grid, one thread per node for the filters, a wave per surviving list
    if (it > 0) {          // (the first pair of sweeps finds both flags zeroed by cl_init_kernel: two 4-byte fill launches less per clustering)
        D3_CHECK(hipMemsetAsync(w.scalars, 0, sizeof(int), s));
        D3_CHECK(hipMemsetAsync(w.scalars + 4, 0, sizeof(int), s));
    }
    if (it == 0 && asc)
        cl_push_kernel<<<npb, T, 0, s>>>(semantic_label, ball_query_idxs, start_len, n, w.parent, w.lab, w.klen, w.qln, w.scalars, w.scalars + 3, asc, 1);
    cl_push_kernel<<<npb, T, 0, s>>>(semantic_label, ball_query_idxs, start_len, n, w.parent, w.lab, w.klen, w.qln, w.scalars, w.scalars + 3, asc, 0);
    cl_push_kernel<<<npb, T, 0, s>>>(semantic_label, ball_query_idxs, start_len, n, w.parent, w.lab, w.klen, w.qln, w.scalars + 4, w.scalars + 3, asc, 0);
    if (it > 0) D3_CHECK(hipMemsetAsync(w.sizes, 0, (size_t)n * sizeof(int), s));   // (cl_owner_kernel accumulates)
    cl_owner_kernel<<<nb, T, 0, s>>>(w.parent, w.lab, w.own, w.sizes, n);
    cl_keep_kernel<<<nb, T, 0, s>>>(w.sizes, w.flag, w.ksz, n, threshold, start_len, w.scalars);
    int rc = d3_exclusive_scan_i32(w.flag, w.cid, n, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    rc = d3_exclusive_scan_i32(w.ksz, w.koff, n, w.temp, w.temp_bytes, s);
    if (rc) return rc;
    cl_totals_kernel<<<1, 64, 0, s>>>(w.flag, w.cid, w.ksz, w.koff, n, w.scalars);
    D3_LAUNCH_CHECK();
    D3_CHECK(hipMemcpyAsync(h, w.scalars, 6 * sizeof(int), hipMemcpyDeviceToHost, s));
    return 0;
}

static int cl_count(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold, void *ws,
                    size_t ws_bytes, int *sumNPoint_host, int *nCluster_host, int flags, void *stream) {
    D3_CLEAR();
    const int asc = (flags & D3_BFS_ASCENDING) ? 1 : 0;
    *sumNPoint_host = 0; *nCluster_host = 0;
    if (n <= 0) return 0;
    ClWs w;
    if (!cl_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    int h[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0;; it += 2) {
        int rc = cl_count_enqueue(semantic_label, ball_query_idxs, start_len, n, threshold, w, asc, it, h, s);
        if (rc) return rc;
        D3_CHECK(hipStreamSynchronize(s));
        if (!h[0] || !h[4] || it >= n + 2) break;
    }
    *nCluster_host = h[1];
    *sumNPoint_host = h[2];
    // what the fill may assume about THIS workspace's lists (same thread: count and fill are one operator call)
    g_cl_checked_ws = ws; g_cl_short_lists = h[5] == 0;
    return 0;
}

// z0 / z1 (optional, n ints each): zeroed here instead of by two fill launches in front of the record pass (3 MB each at the bench
// batch: 13 + 29 us of fill kernels and their launch gaps on the clustering's critical path)
// cnt (optional): the count phase's device scalars ([1] = nCluster, [2] = sumNPoint) -- the speculative fill of d3_bfs_cluster_run is
// enqueued before the host has read them
__global__ void cl_seed_kernel(const int *flag, const int *cid, const int *koff, int n, int *seeds,
                               int *cluster_offsets, int nCluster, int sumNPoint, int *z0, int *z1, const int *__restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cluster_offsets[cnt ? cnt[1] : nCluster] = cnt ? cnt[2] : sumNPoint;
    if (i < n && z0) { z0[i] = 0; z1[i] = 0; }
    if (i >= n || !flag[i]) return;
    seeds[cid[i]] = i;
    cluster_offsets[cid[i]] = koff[i];
}

// block-wide exclusive scan of one int per thread; `total` = sum over the block.  wsum: >= 18 ints of LDS.
__device__ __forceinline__ int cl_blk_scan(int v, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    if (wv == 0) {
        int w = (lane < nw) ? wsum[lane] : 0, ws = w;
        for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(ws, o); if (lane >= o) ws += y; }
        if (lane < nw) wsum[lane] = ws - w;
        if (lane == 63) wsum[nw] = ws;
    }
    __syncthreads();
    const int r = wsum[wv] + x - v;
    total = wsum[nw];
    __syncthreads();
    return r;
}

// phase 3: one workgroup per kept cluster replays the FIFO BFS level by level, EDGE-parallel:
// the frontier's list lengths are prefix-summed in LDS, every thread takes flat edge ids (binary search for
// the owning frontier entry), so a level of F nodes / E edges costs ~E/1024 iterations whatever the list
// lengths are.  Flat edge order == (parent queue position, list position) == the FIFO discovery order, so
// pass C is a plain ordered compaction of the "first discoverer" edges.
// The level loop is latency bound (a 4 m floor is ~200 levels deep), so dependent global round trips are cut:
// the queue stores (node, list start, list length) records written when a node is appended; a component has a
// single semantic label; pass A keeps its edge candidates in registers for pass C (frontiers up to 4096 edges).
#define CL_FCH 1024
#define CL_KEEP 4     // edge candidates kept per thread between pass A and pass C
__global__ __launch_bounds__(CL_BFS_THREADS) void cl_bfs_kernel(const int *__restrict__ sem,
                                                               const int *__restrict__ idx,
                                                               const int *__restrict__ start_len,
                                                               const int *__restrict__ own,
                                                               const int *__restrict__ seeds,
                                                               const int *__restrict__ koff,
                                                               const int *__restrict__ sizes, int *par, int *queue,
                                                               int *qst_all, int *qln_all, int *cluster_idxs, int min_size) {
    __shared__ int s_st[CL_FCH], s_off[CL_FCH + 1], s_w[24];
    const int c = blockIdx.x;
    const int s = seeds[c];
    const int base = koff[s];
    const int size = sizes[s];
    (void)sem;
    if (size <= min_size) return;   // handled by cl_bfs2_kernel
    int *q = queue + base, *qst = qst_all + base, *qln = qln_all + base;
    const int tid = threadIdx.x;
    if (tid == 0) { st_dev(&q[0], s); st_dev(&qst[0], start_len[s * 2]); st_dev(&qln[0], start_len[s * 2 + 1]); st_dev(&par[s], -1); }
    __syncthreads();
    int lo = 0, hi = 1;
    while (lo < hi && hi <= size) {
        const bool single = (hi - lo) <= CL_FCH;
        int E = 0;
        int keep_j[CL_KEEP], keep_gp[CL_KEEP];
        int2 keep_sl[CL_KEEP];
#pragma unroll
        for (int r = 0; r < CL_KEEP; r++) { keep_j[r] = -1; keep_gp[r] = 0; keep_sl[r] = make_int2(0, 0); }
        // ---- pass A: first discoverer of every neighbour = smallest parent queue position
        for (int fb = lo; fb < hi; fb += CL_FCH) {
            const int nf = min(CL_FCH, hi - fb);
            int ln = 0;
            if (tid < nf) { s_st[tid] = ld_dev(&qst[fb + tid]); ln = ld_dev(&qln[fb + tid]); }
            const int off = cl_blk_scan(ln, s_w, E);
            if (tid < nf) s_off[tid] = off;
            if (tid == 0) s_off[nf] = E;
            __syncthreads();
            const bool keep = single && E <= CL_KEEP * CL_BFS_THREADS;
#pragma unroll
            for (int r = 0; r < CL_KEEP; r++) {
                const int e = tid + r * CL_BFS_THREADS;
                if (e < E) {
                    int a = 0, b = nf;  // largest f with s_off[f] <= e
                    while (b - a > 1) { const int m = (a + b) >> 1; if (s_off[m] <= e) a = m; else b = m; }
                    const int j = idx[s_st[a] + e - s_off[a]];
                    if (own[j] == s) {   // owned by this seed => same semantic label (a component has one label)
                        const int gp = fb + a;
                        if (ld_dev(&par[j]) > gp) atomicMin(&par[j], gp);
                        if (keep) { keep_j[r] = j; keep_gp[r] = gp; keep_sl[r] = *(const int2 *)&start_len[j * 2]; }
                    }
                }
            }
            for (int e = tid + CL_KEEP * CL_BFS_THREADS; e < E; e += CL_BFS_THREADS) {
                int a = 0, b = nf;
                while (b - a > 1) { const int m = (a + b) >> 1; if (s_off[m] <= e) a = m; else b = m; }
                const int j = idx[s_st[a] + e - s_off[a]];
                if (own[j] == s) { const int gp = fb + a; if (ld_dev(&par[j]) > gp) atomicMin(&par[j], gp); }
            }
            __syncthreads();
        }
        // ---- pass C: children in flat edge order
        int tail = hi;
        if (single && E <= CL_KEEP * CL_BFS_THREADS) {
#pragma unroll
            for (int r = 0; r < CL_KEEP; r++) {
                if (r * CL_BFS_THREADS >= E) break;   // uniform
                const int j = keep_j[r];
                int child = 0, cst = 0, cln = 0;
                if (j >= 0 && ld_dev(&par[j]) == keep_gp[r]) { child = 1; cst = keep_sl[r].x; cln = keep_sl[r].y; }
                if (!__syncthreads_or(child)) continue;
                int tot;
                const int pos = cl_blk_scan(child, s_w, tot);
                if (child && tail + pos < size) { st_dev(&q[tail + pos], j); st_dev(&qst[tail + pos], cst); st_dev(&qln[tail + pos], cln); }
                tail += tot;
            }
            __syncthreads();
        } else {
            for (int fb = lo; fb < hi; fb += CL_FCH) {
                const int nf = min(CL_FCH, hi - fb);
                if (!single) {  // several frontier chunks: rebuild this chunk's LDS tables
                    int ln = 0;
                    if (tid < nf) { s_st[tid] = ld_dev(&qst[fb + tid]); ln = ld_dev(&qln[fb + tid]); }
                    const int off = cl_blk_scan(ln, s_w, E);
                    if (tid < nf) s_off[tid] = off;
                    if (tid == 0) s_off[nf] = E;
                    __syncthreads();
                }
                for (int e0 = 0; e0 < E; e0 += CL_BFS_THREADS) {
                    const int e = e0 + tid;
                    int child = 0, j = 0, cst = 0, cln = 0;
                    if (e < E) {
                        int a = 0, b = nf;
                        while (b - a > 1) { const int m = (a + b) >> 1; if (s_off[m] <= e) a = m; else b = m; }
                        j = idx[s_st[a] + e - s_off[a]];
                        child = (own[j] == s && ld_dev(&par[j]) == fb + a) ? 1 : 0;
                        if (child) { cst = start_len[j * 2]; cln = start_len[j * 2 + 1]; }
                    }
                    if (!__syncthreads_or(child)) continue;
                    int tot;
                    const int pos = cl_blk_scan(child, s_w, tot);
                    if (child && tail + pos < size) { st_dev(&q[tail + pos], j); st_dev(&qst[tail + pos], cst); st_dev(&qln[tail + pos], cln); }
                    tail += tot;
                }
                __syncthreads();
            }
        }
        lo = hi; hi = tail;
    }
    for (int p = tid; p < size; p += blockDim.x) {
        cluster_idxs[(size_t)(base + p) * 2 + 0] = c;
        cluster_idxs[(size_t)(base + p) * 2 + 1] = ld_dev(&q[p]);
    }
}

extern "C" int d3_bfs_cluster_fill(const int *semantic_label, const int *ball_query_idxs, const int *start_len,
                                   int n, void *ws, size_t ws_bytes, int *cluster_idxs, int *cluster_offsets,
                                   int sumNPoint, int nCluster, void *stream) {
    D3_CLEAR();
    if (n <= 0) return 0;
    ClWs w;
    if (!cl_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256, nb = (n + T - 1) / T;
    cl_seed_kernel<<<nb, T, 0, s>>>(w.flag, w.cid, w.koff, n, w.seeds, cluster_offsets, nCluster, sumNPoint, nullptr, nullptr, nullptr);
    if (nCluster > 0)
        cl_bfs_kernel<<<nCluster, CL_BFS_THREADS, 0, s>>>(semantic_label, ball_query_idxs, start_len, w.own, w.seeds,
                                                         w.koff, w.sizes, w.par, w.queue, w.fcnt, w.qln, cluster_idxs, 0);
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// phase 3, record form.  The level loop above pays three to four dependent L2 round trips per level (frontier
// records, neighbour ids, owner / visited state of the neighbours, winner read-back) and a 4 m floor is ~200 levels
// deep: 2.3 ms for the canonical scene (profiles/r01_i).  Here everything a level needs about a neighbour travels
// WITH the edge: a fully parallel pre-pass rewrites every list entry of a kept cluster as the record
// (node, dense id inside its cluster, list start, list length) -- or node = -1 for a neighbour of another
// component -- and the BFS keeps its state in LDS: a visited bitmap over the dense ids, the frontier's list
// extents, and a small hash that picks, among the edges of a batch that reach the same unvisited node, the one
// with the smallest flat edge id (= the FIFO discoverer).  Batches are processed in flat edge order and the bitmap
// is updated between them, so "first discoverer" is preserved exactly.  One global round trip per batch.
#ifndef B2_THREADS
#define B2_THREADS 1024
#endif
#ifndef B2_EPT
#define B2_EPT 3       // edges per thread per batch (a level of the canonical floor has 2-3.5k edges)
#endif
#define B2_BATCH (B2_THREADS * B2_EPT)
#define B2_HASH 8192
#define B2_FMAX 1024                            // frontier nodes whose list extents are kept in LDS
#define B2_IPT (B2_FMAX / B2_THREADS)
#ifndef B2_HGRID
#define B2_HGRID 16                             // owner hints: one per B2_HGRID edges of the next level
#endif
#define B2_HINTS 8192
#define B2_BITWORDS 8192                        // 32 KB: clusters up to 262144 points; larger ones use cl_bfs_kernel
#define B2_MAXSIZE (B2_BITWORDS * 32)
#define B2_LDS_INTS (B2_BITWORDS + 2 * B2_HASH + 2 * B2_FMAX + 2 * (B2_FMAX + 8) + B2_HINTS + 160)

// Star clusters: when every member of a kept cluster is in its seed's own list, the reference's FIFO BFS is the seed
// followed by those members in list order and ends after the first level (every later pop finds only visited nodes).  That is
// what the shifted coordinates produce -- an instance collapses onto its centre, every list is the instance's first 1000
// members -- and it needs neither edge records (16 B per list entry of every member: 4 GB for 4 x 40 collapsed instances)
// nor the level loop.  One wave per kept cluster: count the seed's same-owner entries, and if they are the whole cluster
// write it out in list order (ballot compaction).
__global__ __launch_bounds__(256) void cl_star_kernel(const int *__restrict__ idx, const int *__restrict__ start_len,
                                                     const int *__restrict__ own, const int *__restrict__ seeds,
                                                     const int *__restrict__ koff, const int *__restrict__ sizes, int nCluster,
                                                     int *__restrict__ star, int *__restrict__ cluster_idxs, const int *__restrict__ dcnt) {
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = d3_lane();
    if (c >= (dcnt ? dcnt[1] : nCluster)) return;
    const int s = seeds[c], st = start_len[s * 2], ln = start_len[s * 2 + 1], size = sizes[s];
    int cnt = 0;
    for (int e0 = 0; e0 < ln; e0 += 64) {
        const int e = e0 + lane;
        const int j = e < ln ? idx[st + e] : -1;
        cnt += (int)__popcll(__ballot(j >= 0 && j != s && own[j] == s));
    }
    const bool is_star = cnt + 1 == size;
    if (lane == 0) star[s] = is_star ? 1 : 0;
    if (!is_star) return;
    const size_t base = (size_t)koff[s];
    if (lane == 0) { cluster_idxs[base * 2] = c; cluster_idxs[base * 2 + 1] = s; }
    int pos = 1;
    for (int e0 = 0; e0 < ln; e0 += 64) {
        const int e = e0 + lane;
        const int j = e < ln ? idx[st + e] : -1;
        const bool m = j >= 0 && j != s && own[j] == s;
        const unsigned long long bal = __ballot(m);
        if (m) {
            const size_t o = base + pos + __popcll(bal & d3_lanemask_lt());
            cluster_idxs[o * 2] = c; cluster_idxs[o * 2 + 1] = j;
        }
        pos += (int)__popcll(bal);
    }
}

__global__ void cl_lid_kernel(const int *__restrict__ own, const int *__restrict__ flag, const int *__restrict__ star,
                              const int *__restrict__ start_len, int *lcnt, int *lid, int *klen, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;   // tail lanes simply drop out of the ballots below
    const int o = own[i];
    const bool kept = flag[o] != 0 && star[o] == 0;
    klen[i] = kept ? start_len[i * 2 + 1] : 0;
    // a wave holds only a few distinct owners: one atomic per distinct owner (leader = lowest lane of each group);
    // one atomic per NODE serialises tens of thousands of updates of the same counter in L2
    int id = -1;
    unsigned long long todo = __ballot(kept);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const int ol = __shfl(o, leader);
        const unsigned long long grp = __ballot(kept && o == ol) & todo;
        int b = 0;
        if (lane == leader) b = atomicAdd(&lcnt[ol], (int)__popcll(grp));
        b = __shfl(b, leader);
        if (kept && o == ol) id = b + (int)__popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    lid[i] = id;
}
// one wave per node of a kept cluster: its list -> edge records
// The records are COMPACT whatever the layout of idx (the padded ball query gives every node a 1000-entry slot: records at
// the same sparse positions cost 4x the write time and scatter the BFS's loads over 16 KB strides): node i's records
// start at estart[i], the exclusive scan of the kept nodes' list lengths, and a record carries its target's estart.
__global__ void cl_ninfo_kernel(const int *__restrict__ own, const int *__restrict__ lid, const int *__restrict__ estart,
                                const int *__restrict__ start_len, int4 *__restrict__ ninfo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ninfo[i] = make_int4(own[i], lid[i], estart[i], start_len[i * 2 + 1]);
}
// Round 5: EIGHT lanes per node (a wave per node left 55 of 64 lanes idle on the ~9-entry lists of a surface and cost one wave
// launch + three dependent round trips per node: 200 - 260 us for the 600 k nodes of the 4-scene batch); a node's lanes take its
// entries 8 apart, four per lane in flight, so a list of up to 32 entries is one pass and the eight 16-byte records of a pass are
// one contiguous 128-byte store.
#define CL_EG 8
__global__ __launch_bounds__(256) void cl_erec_kernel(const int *__restrict__ idx, const int *__restrict__ start_len,
                                                     const int *__restrict__ own, const int *__restrict__ flag,
                                                     const int *__restrict__ star, const int4 *__restrict__ ninfo,
                                                     const int *__restrict__ estart, int4 *__restrict__ erec, int n) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(gid / CL_EG), sub = (int)(gid % CL_EG);
    if (i >= n) return;
    const int oi = own[i];
    if (!flag[oi] || star[oi]) return;
    const int st = start_len[i * 2], ln = start_len[i * 2 + 1];
    const long long es = estart[i];
    // four entries per lane per round trip pair (ids; then owner / dense id / record start / length of all four together as ONE
    // 16-byte record per neighbour -- round 3; four 4-byte gathers per entry before)
    for (int e0 = sub; e0 < ln; e0 += 4 * CL_EG) {
        int j[4];
        int4 nj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = e0 + q * CL_EG; j[q] = idx[st + (e < ln ? e : 0)]; }
#pragma unroll
        for (int q = 0; q < 4; q++) nj[q] = ninfo[j[q]];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = e0 + q * CL_EG;
            if (e < ln) erec[es + e] = (nj[q].x == oi) ? make_int4(j[q], nj[q].y, nj[q].z, nj[q].w) : make_int4(-1, 0, 0, 0);
        }
    }
}

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for every outstanding GLOBAL access of the wave
// (s_waitcnt vmcnt(0)), and each level issues write-through stores (cluster_idxs, queue records) whose completion
// nobody in the level loop depends on: ~2 us per barrier, several barriers per level, ~200 levels.  All cross-wave
// traffic of the level loop goes through LDS, so only lgkmcnt has to drain.
__device__ __forceinline__ void b2_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Workgroup exclusive scan of a pair of ints with ONE barrier: every wave reads all wave totals and scans them itself;
// the totals live in two alternating LDS buffers, so the next call needs no barrier before overwriting them.
__device__ __forceinline__ void b2_scan2(int v0, int v1, int *wsum, int &phase, int &p0, int &p1, int &t0, int &t1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int nw = B2_THREADS / 64;
    int x0 = v0, x1 = v1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y0 = __shfl_up(x0, o), y1 = __shfl_up(x1, o);
        if (lane >= o) { x0 += y0; x1 += y1; }
    }
    int *buf = wsum + (phase & 1) * 64;
    phase++;
    if (lane == 63) { buf[wv] = x0; buf[32 + wv] = x1; }
    b2_barrier();
    int w0 = (lane < nw) ? buf[lane] : 0, w1 = (lane < nw) ? buf[32 + lane] : 0;
    int s0 = w0, s1 = w1;
#pragma unroll
    for (int o = 1; o < nw; o <<= 1) {
        const int y0 = __shfl_up(s0, o), y1 = __shfl_up(s1, o);
        if (lane >= o) { s0 += y0; s1 += y1; }
    }
    t0 = __shfl(s0, nw - 1); t1 = __shfl(s1, nw - 1);
    p0 = __shfl(s0 - w0, wv) + x0 - v0;
    p1 = __shfl(s1 - w1, wv) + x1 - v1;
}

#ifdef B2_TIMING
#define B2_TICK(k) { const long long t_ = (long long)__builtin_readcyclecounter(); tacc[k] += t_ - tprev; tprev = t_; }
#define B2_TDUMP if (dbg && tid == 0 && c < 20) for (int k = 0; k < 8; k++) dbg[60 + c * 8 + k] = (int)(tacc[k] >> 4);
#else
#define B2_TICK(k)
#define B2_TDUMP
#endif
// One workgroup per kept cluster.  Profiled per level (cycle counters, profiles/r01_n): the global round trip for the
// edge records is only ~15 % of a level; the rest is workgroup barriers and dependent LDS chains.  So the level loop
// is organised to need few of both:
//   * the winners of a batch are ranked by ONE scan that carries (count, list length): the list offsets of the next
//     frontier come out of the enqueue step and the next level starts without a scan of its own;
//   * a winner also writes an owner hint for every B2_HGRID-th edge of its list, so a thread of the next level finds
//     the frontier node of its first edge with one LDS read and a step or two instead of a binary search;
//   * the scan needs one barrier (b2_scan2), a batch three in total.
__global__ __launch_bounds__(B2_THREADS) void cl_bfs2_kernel(const int4 *__restrict__ erec, const int *__restrict__ start_len,
                                                            const int *__restrict__ estart,
                                                            const int *__restrict__ lid, const int *__restrict__ seeds,
                                                            const int *__restrict__ koff, const int *__restrict__ sizes,
                                                            const int *__restrict__ star, int *qst_all, int *qln_all,
                                                            int *cluster_idxs, int *dbg, int min_size, const int *__restrict__ cnt, int c0) {
    extern __shared__ __attribute__((aligned(16))) int b2_smem[];
    if (cnt && (int)blockIdx.x + c0 >= cnt[1]) return;       // (speculative launch on an upper-bound grid: no such cluster)
    int n_levels = 0, n_batches = 0;
#ifdef B2_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = (long long)__builtin_readcyclecounter();
#endif
    unsigned int *bitmap = (unsigned int *)b2_smem;                 // B2_BITWORDS
    int *hkey = b2_smem + B2_BITWORDS;                              // B2_HASH
    int *hval = hkey + B2_HASH;                                     // B2_HASH
    int *fst = hval + B2_HASH;                                      // 2 * B2_FMAX: list starts of the frontier nodes
    int *foff = fst + 2 * B2_FMAX;                                  // 2 * (B2_FMAX + 8): exclusive prefix of their lengths
    unsigned short *hint = (unsigned short *)(foff + 2 * (B2_FMAX + 8));   // 2 * B2_HINTS
    int *s_w = (int *)(hint + 2 * B2_HINTS);                        // 128
    const int c = blockIdx.x + c0, tid = threadIdx.x;
    const int s = seeds[c], base = koff[s], size = sizes[s];
    if (size > B2_MAXSIZE) return;                                  // left to cl_bfs_kernel
    if (size <= min_size) return;                                   // written by cl_bfs3_kernel
    if (star[s]) return;                                            // written by cl_star_kernel
    int *qst = qst_all + base, *qln = qln_all + base;
    const int words = (size + 31) >> 5;
    for (int w = tid; w < words; w += B2_THREADS) bitmap[w] = 0u;
    for (int h = tid; h < B2_HASH; h += B2_THREADS) { hkey[h] = -1; hval[h] = CL_INF; }
    __syncthreads();
    if (tid == 0) {
        const int ls = lid[s];
        bitmap[ls >> 5] = 1u << (ls & 31);
        fst[0] = estart[s]; foff[0] = 0; foff[1] = start_len[s * 2 + 1];
        cluster_idxs[(size_t)base * 2] = c; cluster_idxs[(size_t)base * 2 + 1] = s;
    }
    __syncthreads();
    int lo = 0, hi = 1, cur = 0, phase = 0;
    // Prefetch of the next level's records: a winner touches the first lines of its own list as soon as it knows it has
    // won, so that the lines travel to this XCD's L2 while the level finishes (rank scan, enqueue, barriers) and the
    // next level's record loads hit near.  (The loaded values are never used.)
    bool hints_ok = false;                             // hint[cur] covers every B2_EPT-th edge of this level
    while (lo < hi && hi <= size) {
        const bool small = (hi - lo) <= B2_FMAX;      // the frontier's list extents are already in LDS
        int tail = hi, ltail = 0;                     // next frontier: nodes queued / list entries so far
        int *nst = fst + (cur ^ 1) * B2_FMAX, *noff = foff + (cur ^ 1) * (B2_FMAX + 8);
        unsigned short *nhint = hint + (cur ^ 1) * B2_HINTS;
        const unsigned short *chint = hint + cur * B2_HINTS;
        for (int fb = lo; fb < hi; fb += B2_FMAX) {
            const int nf = min(B2_FMAX, hi - fb);
            int *cst = fst + cur * B2_FMAX, *coff = foff + cur * (B2_FMAX + 8);
            if (!small) {
                // frontier beyond the LDS window: its records come back from the global queue, B2_FMAX at a time
                __syncthreads();   // (global queue records written by other waves: full barrier)
                int ln[B2_IPT], sum = 0;
#pragma unroll
                for (int i = 0; i < B2_IPT; i++) {
                    const int f = tid * B2_IPT + i;
                    ln[i] = 0;
                    if (f < nf) { cst[f] = ld_dev(&qst[fb + f]); ln[i] = ld_dev(&qln[fb + f]); }
                    sum += ln[i];
                }
                int p0, p1, t0, t1;
                b2_scan2(sum, 0, s_w, phase, p0, p1, t0, t1);
#pragma unroll
                for (int i = 0; i < B2_IPT; i++) {
                    const int f = tid * B2_IPT + i;
                    if (f < nf) coff[f] = p0;
                    p0 += ln[i];
                }
                if (tid == 0) coff[nf] = t0;
                b2_barrier();
            }
            B2_TICK(0)
            const int E = coff[nf];
            for (int e0 = 0; e0 < E; e0 += B2_BATCH) {
                // A level is a chain of dependent LDS / L2 latencies, so the per-thread work is written for
                // instruction-level parallelism: the owner of the thread's first edge (hint or binary search; its edges are
                // consecutive, the owners of the following edges are found by stepping), all record loads, then all bitmap
                // tests, then all first hash probes are issued before any of their results is used.
                int4 rec[B2_EPT];
                int slot[B2_EPT], oldk[B2_EPT];
                bool cand[B2_EPT];
                const int ef = e0 + tid * B2_EPT;               // first edge of this thread
                long long addr[B2_EPT];
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) addr[r] = -1;
                if (ef < E) {
                    int a = 0;
                    if (hints_ok) a = chint[ef / B2_HGRID];     // owner of edge (ef / B2_HGRID) * B2_HGRID: a lower bound
                    else {
                        int b = nf;                              // largest f with coff[f] <= ef
                        while (b - a > 1) { const int m = (a + b) >> 1; if (coff[m] <= ef) a = m; else b = m; }
                    }
                    // owners first (LDS-only loops), THEN all record loads back to back: a loop between two global loads makes
                    // the compiler drain vmcnt before it, i.e. one full memory round trip per edge slot
                    int o0 = coff[a], o1 = coff[a + 1], st = cst[a];
#pragma unroll
                    for (int r = 0; r < B2_EPT; r++) {
                        const int e = ef + r;
                        if (e < E) {
                            while (o1 <= e) { a++; o0 = o1; o1 = coff[a + 1]; st = cst[a]; }   // coff[nf] = E > e terminates
                            addr[r] = (long long)st + e - o0;
                        }
                    }
                }
                B2_TICK(2)
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) rec[r] = erec[addr[r] >= 0 ? addr[r] : 0];   // branch-free: one wait for all
                __builtin_amdgcn_sched_barrier(0);   // (keeps the first use, and its wait, behind the last load)
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) if (addr[r] < 0) rec[r] = make_int4(-1, 0, 0, 0);
#ifdef B2_TIMING
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                B2_TICK(3)
                unsigned int bw[B2_EPT];
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) bw[r] = (rec[r].x >= 0) ? bitmap[rec[r].y >> 5] : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) {
                    cand[r] = rec[r].x >= 0 && !((bw[r] >> (rec[r].y & 31)) & 1u);
                    slot[r] = (int)(((unsigned int)rec[r].y * 2654435761u) >> 19);
                    oldk[r] = 0;
                }
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) if (cand[r]) oldk[r] = atomicCAS(&hkey[slot[r]], -1, rec[r].y);
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) {
                    if (cand[r]) {
                        int h = slot[r], old = oldk[r];
                        while (old != -1 && old != rec[r].y) {   // occupied by another node: linear probing
                            h = (h + 1) & (B2_HASH - 1);
                            old = atomicCAS(&hkey[h], -1, rec[r].y);
                        }
                        slot[r] = h;
                        atomicMin(&hval[h], tid * B2_EPT + r);
                    }
                }
                b2_barrier();
                B2_TICK(4)
                unsigned int win = 0u;
                int nwin = 0, lwin = 0;
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) {
                    if (cand[r] && hval[slot[r]] == tid * B2_EPT + r) { win |= 1u << r; nwin++; lwin += rec[r].w; }
                }
#ifndef B2_NO_PREFETCH
                // (inline asm: written as C++ loads the compiler merges the non-winner addresses and waits on each value)
                int pf0[B2_EPT], pf1[B2_EPT], pf2[B2_EPT];
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) {                // a non-winner touches the first record (always valid)
                    const bool wn = (win >> r) & 1u;
                    const int *q0 = (const int *)(erec + (wn ? rec[r].z : 0));   // 8 records per 128-byte line
                    const int *q1 = q0 + ((wn && rec[r].w > 8) ? 32 : 0), *q2 = q0 + ((wn && rec[r].w > 16) ? 64 : 0);
                    asm volatile("global_load_dword %0, %1, off" : "=v"(pf0[r]) : "v"(q0) : "memory");
                    asm volatile("global_load_dword %0, %1, off" : "=v"(pf1[r]) : "v"(q1) : "memory");
                    asm volatile("global_load_dword %0, %1, off" : "=v"(pf2[r]) : "v"(q2) : "memory");
                }
#endif
                int pos, lpos, tot, ltot;
                b2_scan2(nwin, lwin, s_w, phase, pos, lpos, tot, ltot);   // (its barrier: every hval read is done)
                B2_TICK(5)
                int p = tail + pos, lp = ltail + lpos;
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) {
                    if ((win >> r) & 1u) {
                        if (p < size) {
                            *(int2 *)&cluster_idxs[(size_t)(base + p) * 2] = make_int2(c, rec[r].x);
                            const int nx = p - hi;            // position inside the next frontier
                            if (nx < B2_FMAX) {
                                nst[nx] = rec[r].z; noff[nx] = lp;
                                for (int g = (lp + B2_HGRID - 1) / B2_HGRID; g * B2_HGRID < lp + rec[r].w && g < B2_HINTS; g++)
                                    nhint[g] = (unsigned short)nx;
                            }
                            // plain stores: the only reader is this workgroup (same XCD, L2-coherent) through ld_dev
                            qst[p] = rec[r].z; qln[p] = rec[r].w;
                            atomicOr(&bitmap[rec[r].y >> 5], 1u << (rec[r].y & 31));
                        }
                        p++; lp += rec[r].w;
                    }
                    if (cand[r]) { hkey[slot[r]] = -1; hval[slot[r]] = CL_INF; }   // every occupied slot has >= 1 candidate
                }
                tail += tot; ltail += ltot;
                if (tid == 0 && tail - hi <= B2_FMAX) noff[tail - hi] = ltail;   // closes the prefix (rewritten per batch)
#ifndef B2_NO_PREFETCH
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetch registers are free again only now
#pragma unroll
                for (int r = 0; r < B2_EPT; r++) asm volatile("" :: "v"(pf0[r]), "v"(pf1[r]), "v"(pf2[r]));
#endif
                b2_barrier();
                B2_TICK(6)
                n_batches++;
                if (tail >= size) {   // every node of the component is queued: the remaining edges (a dense component has
                                      // ~size^2 of them) cannot discover anything
                    if (dbg && tid == 0 && c < 20) { dbg[c * 3] = size; dbg[c * 3 + 1] = n_levels; dbg[c * 3 + 2] = n_batches; }
                    B2_TDUMP
                    return;
                }
            }
        }
        hints_ok = (tail - hi) <= B2_FMAX && ltail <= B2_HINTS * B2_HGRID;
        lo = hi; hi = tail; cur ^= 1; n_levels++;
    }
    if (dbg && tid == 0 && c < 20) { dbg[c * 3] = size; dbg[c * 3 + 1] = n_levels; dbg[c * 3 + 2] = n_batches; }
    B2_TDUMP
}

// ------------------------------------------------------------------------------------------------------------------
// phase 3, third form (round 5): a 16-lane GROUP per frontier ENTRY, discovery keys in an LDS array over the dense ids.
// Both level loops of this file are bound by VALU issue on ONE compute unit: cl_bfs2_kernel spends ~14,000 cycles per level of the
// canonical floor (~300 nodes / ~2,700 edges per level, 192 levels; cycle counters, gpurun_out r04_j20) = ~800 instructions per
// wave per level on the flat-edge -> owner search, the CAS / probe / atomicMin hash that elects the first discoverer, hint tables,
// their clean-up and a (count, list length) block scan.  What a level needs per edge is much less:
//   * a frontier ENTRY is (record start, <= 16 records): a node with a longer list is queued as consecutive entries of 16.
//     Group g of pass p owns entry a = p * 32 + g, lane l its record l: one contiguous 256-byte read per group, no owner search,
//     no list-offset prefix;
//   * the election is ONE LDS atomic per edge: disc[dense id] = min(disc, key), key = batch number : entry : lane (19 + 9 + 4
//     bits).  A word claimed by an earlier batch is smaller than every key of this one (= visited), 0xFFFFFFFF = never seen;
//     after the barrier the edge whose key is still there is the FIFO discoverer (entries and lanes are in (parent position,
//     list position) order) -- no bitmap, no hash, no clean-up;
//   * inside a wave-pass that order IS the lane order, so a winner's rank is mbcnt(ballot) and the wave's total a scalar
//     popcount; the (pass, wave) totals go through a 128-entry LDS table that every wave scans for itself and reads back with
//     v_readlane.  Winners with more than 16 records (rare) add their extra entries through six more ballots (bit planes of the
//     chunk count).  Three LDS-only barriers per batch; records / outputs through raw buffer instructions (32-bit offsets).
// Earlier attempts of this round, both bit-exact and both SLOWER than cl_bfs2_kernel (944 us): one THREAD per frontier node
// (1,460 us: 64 lanes x 16-byte loads from 64 different lines per instruction, lists beyond eight records walked with dependent
// loads) and this layout with per-group masks / cross-lane reads and 64-bit addressing (1,500 us: ~100 instructions per
// wave-pass, ~80 wave-passes per level).  A wave-pass here is ~35 instructions.
// disc needs 4 B per node: clusters up to B3_MAXNODES; larger ones, inputs with a list beyond 2,047 entries (the reference's ball
// query stops at 1,000: lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cu:45) and record arrays beyond 4 GiB keep
// cl_bfs2_kernel.  Same outputs as the other forms, bit for bit.
#ifndef B3_T
#define B3_T 512
#endif
#define B3_G 16                                // lanes per frontier entry = records per entry
#define B3_NG (B3_T / B3_G)                    // entries per pass
#define B3_FMAX 512                            // frontier entries per batch (= kept in LDS)
#define B3_P (B3_FMAX / B3_NG)                 // passes per batch
#define B3_NW (B3_T / 64)
#define B3_MAXNODES 37632                      // 147 KB of discovery words
#define B3_QMAX 0x7FFFFu
#define B3_LDS_INTS (B3_MAXNODES + 2 * 2 * B3_FMAX + 2 * B3_P * B3_NW + 64 + 64)
#define B3_RSRC_FLAGS 0x00020000               // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
static_assert(B3_P * B3_NW == 128, "the (pass, wave) tables are scanned as two entries per lane");
typedef unsigned int b3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int b3_u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(B3_T) void cl_bfs3_kernel(const int4 *__restrict__ erec, unsigned int erec_bytes,
                                                      const int *__restrict__ start_len, const int *__restrict__ estart,
                                                      const int *__restrict__ lid, const int *__restrict__ seeds,
                                                      const int *__restrict__ koff, const int *__restrict__ sizes,
                                                      const int *__restrict__ star, int *qst_all, int *qln_all, int *cluster_idxs,
                                                      int *dbg) {
    extern __shared__ __attribute__((aligned(16))) int b3_smem[];
    unsigned int *disc = (unsigned int *)b3_smem;                      // B3_MAXNODES
    int2 *ftab = (int2 *)(b3_smem + B3_MAXNODES);                      // 2 x B3_FMAX: (first record, records <= 16) per frontier entry
    int *wtabN = (int *)(ftab + 2 * B3_FMAX);                          // B3_P x B3_NW: winners (nodes) per (pass, wave)
    int *wtabE = wtabN + B3_P * B3_NW;                                 // ... and their frontier entries
    int *misc = wtabE + B3_P * B3_NW;                                  // [0..1]: cut of a read-back batch (nodes, entries); [8..]: block scan
    // landing zone of the list prefetches: a winner's first record line is pulled towards this XCD's L2 by a load that writes to
    // LDS (no register to keep alive across the level's barriers; the values are never read, every wave shares the 256 bytes)
    __attribute__((address_space(3))) void *pfz = (__attribute__((address_space(3))) void *)(misc + 64);
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = tid / B3_G, l = tid % B3_G;
    const int s = seeds[c], base = koff[s], size = sizes[s];
    if (size > B3_MAXNODES) return;                                    // left to cl_bfs2_kernel
    if (star[s]) return;                                               // written by cl_star_kernel
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc((void *)erec, 0, erec_bytes, B3_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(cluster_idxs + (size_t)base * 2), 0, (unsigned int)size * 8u, B3_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rqs = __builtin_amdgcn_make_buffer_rsrc((void *)(qst_all + base), 0, (unsigned int)size * 4u, B3_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rql = __builtin_amdgcn_make_buffer_rsrc((void *)(qln_all + base), 0, (unsigned int)size * 4u, B3_RSRC_FLAGS);
    int *qst = qst_all + base, *qln = qln_all + base;
    for (int w = tid; w < size; w += B3_T) disc[w] = 0xFFFFFFFFu;
    __syncthreads();
    int ne = 0;                                                        // entries of the current frontier held in LDS
    {
        const int sl = start_len[s * 2 + 1], es = estart[s];
        ne = (sl + B3_G - 1) / B3_G;
        if (tid == 0) { disc[lid[s]] = 0u; cluster_idxs[(size_t)base * 2] = c; cluster_idxs[(size_t)base * 2 + 1] = s; qst[0] = es; qln[0] = sl; }
        if (tid < ne && tid < B3_FMAX) ftab[tid] = make_int2(es + tid * B3_G, min(B3_G, sl - tid * B3_G));
    }
    __syncthreads();
    int lo = 0, hi = 1, cur = 0, n_levels = 0, n_batches = 0;
    unsigned int q = 1u;                                               // batch number (the key's high field)
#ifdef B3_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = (long long)__builtin_readcyclecounter();
#define B3_TICK(k) { const long long t_ = (long long)__builtin_readcyclecounter(); tacc[k] += t_ - tprev; tprev = t_; }
#define B3_TDUMP if (dbg && tid == 0 && c < 20) for (int k = 0; k < 8; k++) dbg[60 + c * 8 + k] = (int)(tacc[k] >> 4);
#else
#define B3_TICK(k)
#define B3_TDUMP
#endif
    while (lo < hi && hi < size) {
        const bool in_lds = ne <= B3_FMAX;                             // the frontier's entry table was written by the previous level
        int tail = hi, etail = 0;                                      // nodes queued / entries of the next frontier so far
        int2 *ntab = ftab + (cur ^ 1) * B3_FMAX;
        int2 *ctab = ftab + cur * B3_FMAX;
        for (int fb = lo; fb < hi;) {
            int nb;                                                    // entries of this batch
            if (in_lds) { nb = ne; fb = hi; }
            else {
                // frontier beyond the LDS window: node records (first record, list length) come back from the global queue and are
                // cut into entries again -- as many whole nodes as fit B3_FMAX entries
                __syncthreads();                                       // (queue records written by other waves: full barrier)
                const int NB = min(B3_FMAX, hi - fb);
                int st = 0, ln = 0;
                if (tid < NB) { st = ld_dev(&qst[fb + tid]); ln = ld_dev(&qln[fb + tid]); }
                const int nch = (ln + B3_G - 1) / B3_G;
                if (tid == 0) { misc[0] = 0; misc[1] = 0; }
                int tot_;
                const int eoff = cl_blk_scan(nch, misc + 8, tot_);     // (block scan: __syncthreads inside)
                const bool fits = tid < NB && eoff + nch <= B3_FMAX;
                if (fits) { atomicMax(&misc[0], tid + 1); atomicMax(&misc[1], eoff + nch); }
                if (fits) for (int j = 0; j < nch; j++) ctab[eoff + j] = make_int2(st + j * B3_G, min(B3_G, ln - j * B3_G));
                __syncthreads();
                nb = misc[1]; fb += misc[0];
                __syncthreads();                                       // (misc is rewritten by the next read-back batch)
            }
            const int npass = (nb + B3_NG - 1) / B3_NG;
            b3_u32x4 rec[B3_P];
            int rk[B3_P];                                              // winner: node rank | entry rank << 8 inside the wave-pass; else -1
            const unsigned int kthread = (q << 13) | ((unsigned int)g << 4) | (unsigned int)l;
            // ---- claim: every record bids for its target with (batch, entry, lane)
#pragma unroll
            for (int p = 0; p < B3_P; p++) {
                rec[p] = (b3_u32x4){0xFFFFFFFFu, 0u, 0u, 0u};
                if (p < npass) {
                    const int a = p * B3_NG + g;
                    if (a < nb) {
                        const int2 me = ctab[a];
                        if (l < me.y) rec[p] = __builtin_amdgcn_raw_buffer_load_b128(rrec, (unsigned int)(me.x + l) * 16u, 0, 0);
                    }
                }
            }
            B3_TICK(0)
            __builtin_amdgcn_sched_barrier(0);                         // (all record loads issued before the first use)
#ifdef B3_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            B3_TICK(1)
#endif
#pragma unroll
            for (int p = 0; p < B3_P; p++)
                if (p < npass && (int)rec[p].x >= 0) atomicMin(&disc[rec[p].y], kthread + (unsigned int)(p * B3_NG << 4));
            B3_TICK(2)
            b2_barrier();
            B3_TICK(3)
            // ---- check: the bid that is still there discovered the node first; lane order = FIFO order inside a wave-pass
            int tvN = 0, tvE = 0;                                      // lane p: this wave's totals of pass p
#pragma unroll
            for (int p = 0; p < B3_P; p++) {
                rk[p] = -1;
                if (p < npass) {
                    const bool w0 = (int)rec[p].x >= 0 && disc[rec[p].y] == kthread + (unsigned int)(p * B3_NG << 4);
                    const unsigned long long bal = __ballot(w0);
                    const int nrank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0u));
                    int erank = nrank, ecount = (int)__popcll(bal);
                    const int ncount = ecount;
                    const unsigned long long balL = __ballot(w0 && rec[p].w > (unsigned int)B3_G);
                    if (balL != 0ull) {                                // some winner brings more than one entry: bit planes of (entries - 1)
                        const unsigned int x = w0 ? (rec[p].w + B3_G - 1) / B3_G - 1u : 0u;
#pragma unroll
                        for (int bit = 0; bit < 7; bit++) {
                            const unsigned long long bb = __ballot((x >> bit) & 1u);
                            erank += (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bb, 0u)) << bit;
                            ecount += (int)__popcll(bb) << bit;
                        }
                    }
                    if (w0) rk[p] = nrank | (erank << 8);
                    if (lane == p) { tvN = ncount; tvE = ecount; }
                }
            }
            if (lane < npass) { wtabN[lane * B3_NW + wv] = tvN; wtabE[lane * B3_NW + wv] = tvE; }
            B3_TICK(4)
            b2_barrier();
            // ---- ranks: every wave scans the (pass, wave) totals for itself (128 entries, two per lane), nodes and entries
            int totN, totE, exN0, exN1, exE0, exE1;
            {
                const int npw = npass * B3_NW;
                const int2 vn = (2 * lane < npw) ? *(const int2 *)&wtabN[2 * lane] : make_int2(0, 0);      // (npw is even)
                const int2 ve = (2 * lane < npw) ? *(const int2 *)&wtabE[2 * lane] : make_int2(0, 0);
                int xn = vn.x + vn.y, xe = ve.x + ve.y;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int yn = __shfl_up(xn, o), ye = __shfl_up(xe, o);
                    if (lane >= o) { xn += yn; xe += ye; }
                }
                totN = __builtin_amdgcn_readlane(xn, 63); totE = __builtin_amdgcn_readlane(xe, 63);
                exN0 = xn - vn.x - vn.y; exN1 = exN0 + vn.x;
                exE0 = xe - ve.x - ve.y; exE1 = exE0 + ve.x;
            }
            B3_TICK(5)
            // ---- enqueue in (parent position, list position) order
#pragma unroll
            for (int p = 0; p < B3_P; p++) {
                if (p < npass) {
                    // base of (pass p, this wave): table entry p * NW + wv sits in lane (p * NW + wv) / 2 -- a wave-uniform lane
                    const int ti = p * B3_NW + wv;
                    const int bN = (ti & 1) ? __builtin_amdgcn_readlane(exN1, ti >> 1) : __builtin_amdgcn_readlane(exN0, ti >> 1);
                    const int bE = (ti & 1) ? __builtin_amdgcn_readlane(exE1, ti >> 1) : __builtin_amdgcn_readlane(exE0, ti >> 1);
                    if (rk[p] >= 0) {
                        const int pos = tail + bN + (rk[p] & 0xFF);    // queue position of the node
                        const int e0 = etail + bE + (rk[p] >> 8);      // its first entry in the next frontier
                        __builtin_amdgcn_raw_buffer_store_b64((b3_u32x2){(unsigned int)c, rec[p].x}, rout, (unsigned int)pos * 8u, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(rec[p].z, rqs, (unsigned int)pos * 4u, 0, 0);      // (plain stores: the only reader is this
                        __builtin_amdgcn_raw_buffer_store_b32(rec[p].w, rql, (unsigned int)pos * 4u, 0, 0);      //  workgroup, through ld_dev)
                        if (rec[p].w <= (unsigned int)B3_G) { if (e0 < B3_FMAX) ntab[e0] = make_int2((int)rec[p].z, (int)rec[p].w); }
                        else {
                            const int nch = ((int)rec[p].w + B3_G - 1) / B3_G;
                            for (int j = 0; j < nch && e0 + j < B3_FMAX; j++) ntab[e0 + j] = make_int2((int)rec[p].z + j * B3_G, min(B3_G, (int)rec[p].w - j * B3_G));
                        }
#ifndef B3_NO_PREFETCH
                        // the winner's own records start travelling now (8 per line)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrec, pfz, 4, rec[p].z * 16u, 0, 0, 0);
                        if (rec[p].w > 8u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rrec, pfz, 4, rec[p].z * 16u + 128u, 0, 0, 0);
#endif
                    }
                }
            }
            tail += totN; etail += totE;
            q++; n_batches++;
            B3_TICK(6)
            b2_barrier();
            B3_TICK(7)
            if (tail >= size) {   // every node of the component is queued: the remaining edges cannot discover anything
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no LDS-landing load may outlive the workgroup's LDS allocation)
                if (dbg && tid == 0 && c < 20) { dbg[c * 3] = size; dbg[c * 3 + 1] = n_levels + 1; dbg[c * 3 + 2] = n_batches; }
                B3_TDUMP
                return;
            }
            if (q == B3_QMAX) {   // batch numbers wrap: every visited word becomes "batch 0"
                for (int w = tid; w < size; w += B3_T) if (disc[w] != 0xFFFFFFFFu) disc[w] = 0u;
                q = 1u;
                b2_barrier();
            }
        }
        lo = hi; hi = tail; ne = etail; cur ^= 1; n_levels++;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg && tid == 0 && c < 20) { dbg[c * 3] = size; dbg[c * 3 + 1] = n_levels; dbg[c * 3 + 2] = n_batches; }
    B3_TDUMP
}

// (cnt: the speculative fill does not know sumNPoint on the host -- its 8 S bytes are added here, in units of the slot's factor 4)
__global__ void cl_prof_total_kernel(const int *estart, const int *klen, int n, double *out, const int *cnt) {
    *out = (double)estart[n - 1] + (double)klen[n - 1] + (cnt ? 2.0 * (double)cnt[2] : 0.0);
}

extern "C" size_t d3_bfs_cluster_erec_bytes(long long nActive) { return (size_t)(nActive > 0 ? nActive : 1) * sizeof(int4); }

// d3_bfs_cluster_fill with the record-form level loop; erec: d3_bfs_cluster_erec_bytes(nActive) bytes of scratch
// (nActive = length of ball_query_idxs).  Same outputs, bit for bit.
// dev_counts: the sizes are read on the device (w.scalars, written by the count kernels enqueued in front); sumNPoint / nCluster are
// then UPPER BOUNDS for the grids (the star pass and the replay walk `nCluster` slots and find the real count on the device), and
// the generic level loop for clusters beyond the LDS bitmap is left to the caller, who launches it once it knows sumNPoint.
static int cl_fill2_impl(const int *semantic_label, const int *ball_query_idxs, const int *start_len,
                         int n, void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive,
                         int *cluster_idxs, int *cluster_offsets, int sumNPoint, int nCluster, bool dev_counts, int c0, void *stream) {
    if (n <= 0) return 0;
    ClWs w;
    if (!cl_carve(ws, ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    if (erec == nullptr || erec_bytes < d3_bfs_cluster_erec_bytes(nActive)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const int T = 256, nb = (n + T - 1) / T, nwb = (n + 3) / 4;
    const int *cnt = dev_counts ? w.scalars : nullptr;
    if (c0 == 0)
        cl_seed_kernel<<<nb, T, 0, s>>>(w.flag, w.cid, w.koff, n, w.seeds, cluster_offsets, nCluster, sumNPoint, nCluster > 0 ? w.lcnt : nullptr, w.star, cnt);
    if (nCluster > 0) {
        static bool attr_done_dev[64] = {false};   // the attribute is per device
        const size_t lds = (size_t)B2_LDS_INTS * sizeof(int);
        int dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64 || !attr_done_dev[dev_id]) {
            D3_CHECK(hipFuncSetAttribute((const void *)cl_bfs2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev_id >= 0 && dev_id < 64) attr_done_dev[dev_id] = true;
        }
        const bool no_star = d3_tune(D3T_BFS_NO_STAR) != 0;   // (tests: force the level loop for every cluster)
        if (c0 == 0) {      // (c0 > 0: a second replay launch for the clusters beyond the speculative grid -- the tables are built)
        if (!no_star)
            cl_star_kernel<<<(nCluster + 3) / 4, T, 0, s>>>(ball_query_idxs, start_len, w.own, w.seeds, w.koff, w.sizes, nCluster, w.star,
                                                           cluster_idxs, cnt);
        cl_lid_kernel<<<nb, T, 0, s>>>(w.own, w.flag, w.star, start_len, w.lcnt, w.lid, w.klen, n);
        int rc = d3_exclusive_scan_i32(w.klen, w.estart, n, w.temp, w.temp_bytes, s);
        if (rc) return rc;
        cl_ninfo_kernel<<<nb, T, 0, s>>>(w.own, w.lid, w.estart, start_len, w.ninfo, n);
        cl_erec_kernel<<<(int)(((long long)n * CL_EG + T - 1) / T), T, 0, s>>>(ball_query_idxs, start_len, w.own, w.flag, w.star, w.ninfo, w.estart, (int4 *)erec, n);
        }
        const bool debug = d3_tune(D3T_BFS_DEBUG) != 0;
        // launch timing (bench.py): SURVEY 8(d) "BFS/CC" bytes = 4 nActive + 12 n + 8 S, nActive = the list entries of the kept
        // clusters' nodes (what the replay streams; the padded lists' capacity says nothing) -- known on the device only
        void *pr = d3_prof_begin(5, 12.0 * (double)n + (dev_counts ? 0.0 : 8.0 * (double)sumNPoint), 0.0, s);
        // round 5: clusters whose discovery words fit the LDS (<= B3_MAXNODES nodes) replay on the thread-per-frontier-node kernel;
        // larger ones (and everything with D3_BFS3=0) on the edge-parallel hash form
        const bool use3 = !dev_counts && c0 == 0 && d3_tune(D3T_BFS3) != 0 && g_cl_checked_ws == ws && g_cl_short_lists &&
                          (unsigned long long)(nActive > 0 ? nActive : 1) * sizeof(int4) < 0xFFFFFFFFull;      // (32-bit record offsets)
        if (use3) {
            static bool attr3_done_dev[64] = {false};
            const size_t lds3 = (size_t)B3_LDS_INTS * sizeof(int);
            if (dev_id < 0 || dev_id >= 64 || !attr3_done_dev[dev_id]) {
                D3_CHECK(hipFuncSetAttribute((const void *)cl_bfs3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
                if (dev_id >= 0 && dev_id < 64) attr3_done_dev[dev_id] = true;
            }
            cl_bfs3_kernel<<<nCluster, B3_T, lds3, s>>>((const int4 *)erec, (unsigned int)((size_t)(nActive > 0 ? nActive : 1) * sizeof(int4)), start_len,
                                                       w.estart, w.lid, w.seeds, w.koff, w.sizes, w.star, w.fcnt, w.qln, cluster_idxs,
                                                       debug ? w.lcnt : nullptr);
        }
        if (!use3 || sumNPoint > B3_MAXNODES)
            cl_bfs2_kernel<<<nCluster - c0, B2_THREADS, lds, s>>>((const int4 *)erec, start_len, w.estart, w.lid, w.seeds, w.koff, w.sizes,
                                                            w.star, w.fcnt, w.qln, cluster_idxs, debug ? w.lcnt : nullptr, use3 ? B3_MAXNODES : 0, cnt, c0);
        if (pr) {
            d3_prof_tag(pr, 0, n); d3_prof_tag(pr, 1, nCluster); d3_prof_end(pr, s);
            if (double *slot = d3_prof_dev_slot(pr, 4.0)) cl_prof_total_kernel<<<1, 1, 0, s>>>(w.estart, w.klen, n, slot, cnt);   // (behind the bracket)
        }
        if (debug && !dev_counts) {
            int h[60 + 160];
            hipMemcpyAsync(h, w.lcnt, sizeof(h), hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
            for (int c = 0; c < nCluster && c < 20; c++) {
                fprintf(stderr, "bfs2 cluster %d size %d levels %d batches %d", c, h[c * 3], h[c * 3 + 1], h[c * 3 + 2]);
#if defined(B2_TIMING) || defined(B3_TIMING)
                for (int k = 0; k < 8; k++) fprintf(stderr, " t%d=%d", k, h[60 + c * 8 + k] * 16);
#endif
                fprintf(stderr, "\n");
            }
        }
        // clusters beyond the LDS bitmap: the generic level loop (none can exist when all kept points together fit)
        if (!dev_counts && c0 == 0 && sumNPoint > B2_MAXSIZE)
            cl_bfs_kernel<<<nCluster, CL_BFS_THREADS, 0, s>>>(semantic_label, ball_query_idxs, start_len, w.own, w.seeds,
                                                         w.koff, w.sizes, w.par, w.queue, w.fcnt, w.qln, cluster_idxs, B2_MAXSIZE);
    }
    D3_LAUNCH_CHECK();
    return 0;
}

extern "C" int d3_bfs_cluster_fill2(const int *semantic_label, const int *ball_query_idxs, const int *start_len,
                                    int n, void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive,
                                    int *cluster_idxs, int *cluster_offsets, int sumNPoint, int nCluster, void *stream) {
    D3_CLEAR();
    return cl_fill2_impl(semantic_label, ball_query_idxs, start_len, n, ws, ws_bytes, erec, erec_bytes, nActive, cluster_idxs, cluster_offsets,
                         sumNPoint, nCluster, false, 0, stream);
}

// count + fill as ONE native call (round 5): the caller hands in outputs at their upper bounds (cluster_idxs: cap_points x 2 ints,
// cluster_offsets: cap_clusters + 1 ints) and reads back how much of them was written.  Between the two phases the two-call form
// goes back to its caller for the output allocation; when that caller is a Python thread next to another busy one (the two
// clustering branches of PointGroup.forward), re-acquiring the interpreter lock there cost 80 - 470 us of idle queue per branch
// (gpurun_out/r05_j11/cluster_timeline.txt).  Same results as count_ex + fill2: it IS count_ex + fill2.
// A pinned landing buffer + event per in-flight count (pooled; the clustering branches run on two host threads)
struct ClTicket { int *pinned; hipEvent_t ev; };
static std::mutex g_clt_mu;
static std::vector<ClTicket *> g_clt_free;
#define CL_SPEC_GRID 2048      // cluster slots of the speculative replay launch (more kept clusters: a second launch once the count is known)

// begin / end (round 5): d3_bfs_cluster_run cut at its one host wait.  `begin` enqueues the count kernels, the copy of their scalars,
// an event and the whole speculative fill, and returns a ticket; `end` waits for the event and finishes (the rare cases: more label
// sweeps, kept clusters beyond the speculative grid, clusters beyond the LDS bitmap).  ONE host thread can so keep two clusterings
// (PointGroup's shifted and unshifted branch, on two streams) in flight: begin, begin, end, end -- no helper thread whose wake-up
// sits on the step's critical path (on a slow host the wait for the helper's branch grew from 0.5 to 0.8 ms: profiles r05_d vs r05_e).
// When the speculative form is not available (D3_CL_SPEC=0, debug, cap_points < n) `begin` runs the whole blocking call.
struct ClRun {
    const int *sem, *idx, *start_len; int n, threshold; void *ws; size_t ws_bytes; void *erec; size_t erec_bytes; long long nActive; int flags;
    int *cluster_idxs; long long cap_points; int *cluster_offsets; long long cap_clusters; void *stream;
    ClTicket *t; int slots; int done; int rc; int sumNPoint, nCluster;
};
extern "C" int d3_bfs_cluster_end(void *ticket, int *sumNPoint_host, int *nCluster_host);
extern "C" int d3_bfs_cluster_begin(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold,
                                    void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive, int flags,
                                    int *cluster_idxs, long long cap_points, int *cluster_offsets, long long cap_clusters,
                                    void **ticket, void *stream) {
    if (!ticket) return D3_ERR_ARG;
    *ticket = nullptr;
    ClRun *r = new ClRun{semantic_label, ball_query_idxs, start_len, n, threshold, ws, ws_bytes, erec, erec_bytes, nActive, flags,
                         cluster_idxs, cap_points, cluster_offsets, cap_clusters, stream, nullptr, 0, 0, 0, 0, 0};
    if (n <= 0 || d3_tune(D3T_CL_SPEC) == 0 || d3_tune(D3T_BFS3) != 0 || d3_tune(D3T_BFS_DEBUG) != 0 || cap_points < n) {
        int S = 0, P = 0;
        int rc = cl_count(semantic_label, ball_query_idxs, start_len, n, threshold, ws, ws_bytes, &S, &P, flags, stream);
        if (!rc && n > 0) {
            if ((long long)S > cap_points || (long long)P > cap_clusters) rc = D3_ERR_WORKSPACE;
            else rc = d3_bfs_cluster_fill2(semantic_label, ball_query_idxs, start_len, n, ws, ws_bytes, erec, erec_bytes, nActive, cluster_idxs,
                                           cluster_offsets, S, P, stream);
        }
        r->done = 1; r->rc = rc; r->sumNPoint = S; r->nCluster = P;
        *ticket = r;
        return rc;
    }
    // Speculative form: the count kernels, the copy of their scalars and an event are enqueued, then the WHOLE fill with its sizes
    // read on the device (upper-bound grids); the host waits later (`end`) -- for the event, not for the stream: while it reads the
    // counts the record pass and the level replay are already running.  Should the label push not have converged in its first pair
    // of sweeps (both sweeps still changed labels: capped lists in a chain), the speculative fill's output is overwritten by the
    // regular path in `end`.
    D3_CLEAR();
    int rc = 0;
    ClWs w;
    if (erec == nullptr || erec_bytes < d3_bfs_cluster_erec_bytes(nActive) || !cl_carve(ws, ws_bytes, n, w)) { delete r; return D3_ERR_WORKSPACE; }
    const int asc = (flags & D3_BFS_ASCENDING) ? 1 : 0;
    hipStream_t s = d3_stream(stream);
    ClTicket *t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_clt_mu);
        if (!g_clt_free.empty()) { t = g_clt_free.back(); g_clt_free.pop_back(); }
    }
    if (!t) {
        t = new ClTicket{nullptr, nullptr};
        hipError_t he = hipHostMalloc((void **)&t->pinned, 8 * sizeof(int));
        if (he == hipSuccess) he = hipEventCreateWithFlags(&t->ev, hipEventDisableTiming);
        if (he != hipSuccess) { delete t; delete r; return (int)he; }
    }
    r->t = t;
    auto fail = [&](int code) { { std::lock_guard<std::mutex> lk(g_clt_mu); g_clt_free.push_back(t); } delete r; return code; };
    rc = cl_count_enqueue(semantic_label, ball_query_idxs, start_len, n, threshold, w, asc, 0, t->pinned, s);
    if (rc) return fail(rc);
    if (hipEventRecord(t->ev, s) != hipSuccess) return fail(D3_ERR_ARG);
    r->slots = (int)(cap_clusters < CL_SPEC_GRID ? cap_clusters : CL_SPEC_GRID);
    // (cluster_offsets has cap_clusters + 1 entries and cluster_idxs n rows: whatever the device counts turn out to be, they fit)
    rc = cl_fill2_impl(semantic_label, ball_query_idxs, start_len, n, ws, ws_bytes, erec, erec_bytes, nActive, cluster_idxs, cluster_offsets,
                       n, r->slots > 0 ? r->slots : 1, true, 0, stream);
    if (rc) return fail(rc);
    *ticket = r;
    return 0;
}
extern "C" int d3_bfs_cluster_end(void *ticket, int *sumNPoint_host, int *nCluster_host) {
    if (!ticket || !sumNPoint_host || !nCluster_host) return D3_ERR_ARG;
    ClRun *r = (ClRun *)ticket;
    struct Free { ClRun *r; ~Free() { if (r->t) { std::lock_guard<std::mutex> lk(g_clt_mu); g_clt_free.push_back(r->t); } delete r; } } fr{r};
    *sumNPoint_host = 0; *nCluster_host = 0;
    if (r->done) { *sumNPoint_host = r->sumNPoint; *nCluster_host = r->nCluster; return r->rc; }
    D3_CLEAR();
    const int n = r->n;
    hipStream_t s = d3_stream(r->stream);
    ClWs w;
    if (!cl_carve(r->ws, r->ws_bytes, n, w)) return D3_ERR_WORKSPACE;
    const int asc = (r->flags & D3_BFS_ASCENDING) ? 1 : 0;
    D3_CHECK(hipEventSynchronize(r->t->ev));
    int h[6];
    for (int k = 0; k < 6; k++) h[k] = r->t->pinned[k];
    int rc = 0;
    if (h[0] && h[4]) {      // not converged: more sweeps, then the regular fill over the speculative one
        for (int it = 2;; it += 2) {
            rc = cl_count_enqueue(r->sem, r->idx, r->start_len, n, r->threshold, w, asc, it, h, s);
            if (rc) return rc;
            D3_CHECK(hipStreamSynchronize(s));
            if (!h[0] || !h[4] || it >= n + 2) break;
        }
        *nCluster_host = h[1]; *sumNPoint_host = h[2];
        g_cl_checked_ws = r->ws; g_cl_short_lists = h[5] == 0;
        if ((long long)h[2] > r->cap_points || (long long)h[1] > r->cap_clusters) return D3_ERR_WORKSPACE;
        return d3_bfs_cluster_fill2(r->sem, r->idx, r->start_len, n, r->ws, r->ws_bytes, r->erec, r->erec_bytes, r->nActive, r->cluster_idxs,
                                    r->cluster_offsets, h[2], h[1], r->stream);
    }
    *nCluster_host = h[1]; *sumNPoint_host = h[2];
    g_cl_checked_ws = r->ws; g_cl_short_lists = h[5] == 0;
    if ((long long)h[1] > r->cap_clusters) return D3_ERR_WORKSPACE;      // (cannot happen for cap_clusters >= n / threshold: kept clusters have >= threshold points)
    if (h[1] > r->slots)        // kept clusters beyond the speculative grid: their replay now, on the tables the first launch built
        rc = cl_fill2_impl(r->sem, r->idx, r->start_len, n, r->ws, r->ws_bytes, r->erec, r->erec_bytes, r->nActive, r->cluster_idxs, r->cluster_offsets,
                           h[2], h[1], false, r->slots, r->stream);
    if (rc) return rc;
    if (h[2] > B2_MAXSIZE && h[1] > 0)      // clusters beyond the LDS bitmap: the generic level loop (none can exist when all kept points together fit)
        cl_bfs_kernel<<<h[1], CL_BFS_THREADS, 0, s>>>(r->sem, r->idx, r->start_len, w.own, w.seeds, w.koff, w.sizes, w.par, w.queue,
                                                     w.fcnt, w.qln, r->cluster_idxs, B2_MAXSIZE);
    D3_LAUNCH_CHECK();
    return 0;
}
extern "C" int d3_bfs_cluster_run(const int *semantic_label, const int *ball_query_idxs, const int *start_len, int n, int threshold,
                                  void *ws, size_t ws_bytes, void *erec, size_t erec_bytes, long long nActive, int flags,
                                  int *cluster_idxs, long long cap_points, int *cluster_offsets, long long cap_clusters,
                                  int *sumNPoint_host, int *nCluster_host, void *stream) {
    if (!sumNPoint_host || !nCluster_host) return D3_ERR_ARG;
    void *ticket = nullptr;
    const int rc = d3_bfs_cluster_begin(semantic_label, ball_query_idxs, start_len, n, threshold, ws, ws_bytes, erec, erec_bytes, nActive, flags,
                                        cluster_idxs, cap_points, cluster_offsets, cap_clusters, &ticket, stream);
    if (!ticket) { *sumNPoint_host = 0; *nCluster_host = 0; return rc; }
    const int rc2 = d3_bfs_cluster_end(ticket, sumNPoint_host, nCluster_host);
    return rc ? rc : rc2;
}
