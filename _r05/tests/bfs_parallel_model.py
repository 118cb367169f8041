"""Sequential numpy model of the PARALLEL clustering algorithm used by d3net_amd/csrc/cluster.hip.

It exists to check, on the CPU, that the data-parallel formulation (ownership by minimum
ancestor + level-synchronous ordering by (parent queue position, neighbour index)) reproduces the
reference's sequential FIFO BFS (reference: lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cpp:28-75)
exactly, including on truncated (asymmetric) neighbour lists.  Test helper only.
"""
import numpy as np

CAP = 1000


def owners(sem, idx, start_len):
    """owner[j] = smallest index that reaches j along same-label list edges (the reference's seed)."""
    n = len(sem)
    start, ln = start_len[:, 0], start_len[:, 1]
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    # phase 1: union over edges whose two lists are both complete (len < CAP) -> mutual edges
    for i in range(n):
        if ln[i] >= CAP:
            continue
        for j in idx[start[i]:start[i] + ln[i]]:
            if sem[j] != sem[i] or ln[j] >= CAP:
                continue
            a, b = find(i), find(j)
            if a != b:
                if a < b:
                    parent[b] = a
                else:
                    parent[a] = b
    root = np.array([find(i) for i in range(n)])
    # phase 2: push labels between roots over ALL edges until a fixpoint
    lab = np.arange(n)
    changed = True
    while changed:
        changed = False
        for i in range(n):
            li = lab[root[i]]
            while lab[li] < li:
                li = lab[li]
            for j in idx[start[i]:start[i] + ln[i]]:
                if sem[j] != sem[i]:
                    continue
                rj = root[j]
                if li < lab[rj]:
                    lab[rj] = li
                    changed = True
    own = lab[root]
    for i in range(n):
        while own[i] != lab[root[own[i]]]:
            own[i] = lab[root[own[i]]]
    return own


def bfs_order(seed, own, sem, idx, start_len):
    """level-synchronous BFS of one component; order inside a level = (parent queue pos, neighbour)."""
    start, ln = start_len[:, 0], start_len[:, 1]
    INF = np.iinfo(np.int64).max
    par = {}
    queue = [seed]
    par[seed] = -1
    lo, hi = 0, 1
    while lo < hi:
        # pass A: first discoverer
        for f in range(lo, hi):
            u = queue[f]
            for j in idx[start[u]:start[u] + ln[u]]:
                if sem[j] != sem[u] or own[j] != own[u]:
                    continue
                if par.get(j, INF) > f:
                    par[j] = f
        # pass B/C: children of f in list order
        for f in range(lo, hi):
            u = queue[f]
            for j in idx[start[u]:start[u] + ln[u]]:
                if sem[j] != sem[u] or own[j] != own[u]:
                    continue
                if par[j] == f and j != seed:
                    queue.append(j)
        # de-duplicate is unnecessary: a node has exactly one parent f and appears once in list(f)
        lo, hi = hi, len(queue)
    return queue


def bfs_cluster_parallel_model(sem, idx, start_len, threshold):
    n = len(sem)
    own = owners(sem, idx, start_len)
    sizes = np.bincount(own, minlength=n)
    out_idx, offs = [], [0]
    cid = 0
    for s in range(n):
        if own[s] == s and sizes[s] >= threshold:
            q = bfs_order(s, own, sem, idx, start_len)
            assert len(q) == sizes[s], (len(q), sizes[s])
            out_idx += [(cid, v) for v in q]
            offs.append(offs[-1] + len(q))
            cid += 1
    return np.array(out_idx, np.int32).reshape(-1, 2), np.array(offs, np.int32)


def bfs_order_keys(seed, own, sem, idx, start_len, T=512, qmax=4095):
    """model of cl_bfs3_kernel (round 5): a lane group of a batch owns frontier node a; every list entry k bids for its target
    with key = (batch number, node position a in the batch, list position k) through a min-reduction on disc[target]; the bid that
    is still there after the batch discovered the node (a word claimed by an earlier batch is smaller than every key of this one =
    visited).  Batches hold <= T frontier nodes; winners are ranked in key order = (parent queue position, list position) = the
    reference's FIFO order; batch numbers wrap at qmax (visited words are renumbered to batch 0)."""
    start, ln = start_len[:, 0], start_len[:, 1]
    INF = (1 << 62)
    disc = {seed: (0, 0, 0)}
    queue = [seed]
    lo, hi, q = 0, 1, 1
    size = int((own == own[seed]).sum())
    while lo < hi and hi < size:
        for fb in range(lo, hi, T):                       # batches of the frontier
            nb = min(T, hi - fb)
            keys = []
            for a in range(nb):                           # claim
                u = queue[fb + a]
                for k, j in enumerate(idx[start[u]:start[u] + ln[u]]):
                    if sem[j] == sem[u] and own[j] == own[u]:
                        key = (q, a, k)
                        if disc.get(int(j), (INF, 0, 0)) > key:
                            disc[int(j)] = key
                        keys.append((key, int(j)))
            for key, j in keys:                           # check + enqueue (keys are generated in ascending order)
                if disc[j] == key:
                    queue.append(j)
            q += 1
            if q == qmax:
                disc = {j: (0, 0, 0) for j in disc}
                q = 1
        lo, hi = hi, len(queue)
    return queue


def bfs_cluster_keys_model(sem, idx, start_len, threshold, **kw):
    n = len(sem)
    own = owners(sem, idx, start_len)
    sizes = np.bincount(own, minlength=n)
    out_idx, offs = [], [0]
    cid = 0
    for s in range(n):
        if own[s] == s and sizes[s] >= threshold:
            q = bfs_order_keys(s, own, sem, idx, start_len, **kw)
            assert len(q) == sizes[s], (len(q), sizes[s])
            out_idx += [(cid, v) for v in q]
            offs.append(offs[-1] + len(q))
            cid += 1
    return np.array(out_idx, np.int32).reshape(-1, 2), np.array(offs, np.int32)
