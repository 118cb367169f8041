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
