/*
 * oracle/pg_ops_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's native operator
 * library `PG_OP` (reference: lib/pointgroup_ops/src/ tree).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (d3net_amd/) never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * these operators, its .cu kernels need nvcc + a CUDA GPU, and
 * its two CPU natives (voxelize.cpp, bfs_cluster.cpp) include
 * <google/dense_hash_map> (sparsehash) and <THC/THC.h>, neither of which exists in
 * this image, so they are unbuildable here without writing stand-in headers
 * (not allowed).  Every function below therefore follows the reference SOURCE
 * TEXT line by line and cites it; the only known answers it is checked against
 * are the two toy results recorded in SURVEY.md section 8(c) (tests/test_oracle_pg_ops.py).
 *
 * Build: make -C oracle      (gcc -O2 -ffp-contract=off: no FMA contraction, the
 * arithmetic is exactly the C expressions of the reference source).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int32_t Int; /* reference: src/datatype/datatype.h:9 */

/* ------------------------------------------------------------------ hash map */
/* Stand-in for google::dense_hash_map<Point<3>,Int> per batch item
 * (src/datatype/datatype.h:11-34).  Only find / insert are used by the
 * reference (src/voxelize/voxelize.cpp:75-82,96-103), so any exact map gives the
 * same result; the key here is (batch, x, y, z) after the reference's
 * long -> Int truncation (voxelize.cpp:73,93-95). */
typedef struct { Int b, x, y, z; Int val; int used; } Slot;
typedef struct { Slot *s; size_t cap; } Map;

static uint64_t mix(Int b, Int x, Int y, Int z) {
    uint64_t h = 1469598103934665603ULL;
    uint32_t v[4] = {(uint32_t)b, (uint32_t)x, (uint32_t)y, (uint32_t)z};
    for (int i = 0; i < 4; i++) { h ^= v[i]; h *= 1099511628211ULL; h ^= h >> 29; }
    return h;
}
static Int *map_find_or_insert(Map *m, Int b, Int x, Int y, Int z, int *inserted) {
    size_t i = mix(b, x, y, z) & (m->cap - 1);
    for (;;) {
        Slot *s = &m->s[i];
        if (!s->used) { s->used = 1; s->b = b; s->x = x; s->y = y; s->z = z; *inserted = 1; return &s->val; }
        if (s->b == b && s->x == x && s->y == y && s->z == z) { *inserted = 0; return &s->val; }
        i = (i + 1) & (m->cap - 1);
    }
}

/* ---------------------------------------------------------------- voxelize_idx */
/* reference: src/voxelize/voxelize.cpp:10-152 (voxelize_idx<3>, voxelize_inputmap,
 * voxelize_outputmap) and functions/pointgroup_ops.py:11-39.
 * coords: (n, ncols) int64, ncols in {3,4}; column 0 is the batch index when ncols==4.
 * input_map: (n) int32 out.  Returns M; *max_active_out, *out_coords (M,ncols) int64 and
 * *out_map (M, maxActive+1) int32 are malloc'ed here, free with orc_free. */
int orc_voxelize_idx(const int64_t *coords, int n, int ncols, int mode, Int *input_map,
                     int64_t **out_coords, Int **out_map, int *max_active_out) {
    Map m; m.cap = 16; while (m.cap < (size_t)n * 2 + 16) m.cap <<= 1;
    m.s = (Slot *)calloc(m.cap, sizeof(Slot));
    Int nActive = 0;
    /* outputRows (voxelize.cpp:66): per-voxel point lists in arrival order; built as
     * counts + a second pass so that rows keep ascending point index. */
    Int *cnt = (Int *)calloc((size_t)n + 1, sizeof(Int));
    for (int i = 0; i < n; i++) {
        const int64_t *c = coords + (size_t)i * ncols;
        Int b = 0, p[3];
        if (ncols == 3) { for (int j = 0; j < 3; j++) p[j] = (Int)c[j]; }          /* :72-74 */
        else { b = (Int)c[0]; for (int j = 0; j < 3; j++) p[j] = (Int)c[j + 1]; }  /* :91-95 */
        int ins; Int *v = map_find_or_insert(&m, b, p[0], p[1], p[2], &ins);
        if (ins) *v = nActive++;                                                   /* :76-79,97-100 */
        input_map[i] = *v;                                                         /* :83,105 */
        cnt[*v]++;
    }
    Int *start = (Int *)calloc((size_t)nActive + 1, sizeof(Int));
    for (Int v = 0; v < nActive; v++) start[v + 1] = start[v] + cnt[v];
    Int *rows = (Int *)malloc(sizeof(Int) * (size_t)(n > 0 ? n : 1));
    Int *fill = (Int *)calloc((size_t)nActive + 1, sizeof(Int));
    for (int i = 0; i < n; i++) { Int v = input_map[i]; rows[start[v] + fill[v]++] = i; }

    Int maxActive = 1;                                                             /* :141 */
    if (mode == 3 || mode == 4)
        for (Int v = 0; v < nActive; v++) if (cnt[v] > maxActive) maxActive = cnt[v]; /* :143-145 */
    size_t w = (size_t)maxActive + 1;
    Int *om = (Int *)calloc((size_t)(nActive > 0 ? nActive : 1) * w, sizeof(Int));  /* zero padded :151, :22-23 */
    int64_t *oc = (int64_t *)calloc((size_t)(nActive > 0 ? nActive : 1) * ncols, sizeof(int64_t));
    for (Int v = 0; v < nActive; v++) {
        Int *r = om + (size_t)v * w;
        const Int *pts = rows + start[v];
        if (mode == 0) { r[0] = 1; r[1] = pts[0]; }                                /* :122-129 */
        else if (mode == 1) { r[0] = 1; r[1] = pts[0]; }                           /* front(): :130-135 */
        else if (mode == 2) { r[0] = 1; r[1] = pts[cnt[v] - 1]; }                  /* back(): :136-140 */
        else { r[0] = cnt[v]; for (Int j = 0; j < cnt[v]; j++) r[1 + j] = pts[j]; } /* :146-151 */
        /* voxelize_outputmap (:34-49): coordinate of the FIRST listed point, all columns */
        const int64_t *c = coords + (size_t)r[1] * ncols;
        for (int j = 0; j < ncols; j++) oc[(size_t)v * ncols + j] = c[j];
    }
    free(m.s); free(cnt); free(start); free(rows); free(fill);
    *out_coords = oc; *out_map = om; *max_active_out = maxActive;
    return nActive;
}
void orc_free(void *p) { free(p); }

/* ---------------------------------------------------------- voxelize fp / bp */
/* reference: src/voxelize/voxelize.cu:10-31.  `out` is NOT cleared here (the python
 * wrapper zero-fills it, functions/pointgroup_ops.py:57); accumulation order = rule order. */
void orc_voxelize_fp(const float *feats, float *out, const Int *rules, int nOutputRows,
                     int maxActive, int nPlanes, int average) {
    for (int row = 0; row < nOutputRows; row++) {
        float *o = out + (size_t)row * nPlanes;
        const Int *r = rules + (size_t)row * (maxActive + 1);
        Int nActive = r[0];
        float multiplier = (average && nActive > 0) ? (float)1 / nActive : (float)1;
        for (int i = 1; i <= nActive; i++) {
            const float *inp = feats + (size_t)r[i] * nPlanes;
            for (int plane = 0; plane < nPlanes; plane++) o[plane] += multiplier * inp[plane];
        }
    }
}
/* reference: src/voxelize/voxelize.cu:35-53 */
void orc_voxelize_bp(const float *d_out, float *d_feats, const Int *rules, int nOutputRows,
                     int maxActive, int nPlanes, int average) {
    for (int row = 0; row < nOutputRows; row++) {
        const float *o = d_out + (size_t)row * nPlanes;
        const Int *r = rules + (size_t)row * (maxActive + 1);
        Int nActive = r[0];
        float multiplier = (average && nActive > 0) ? (float)1 / nActive : (float)1;
        for (int i = 1; i <= nActive; i++) {
            float *inp = d_feats + (size_t)r[i] * nPlanes;
            for (int plane = 0; plane < nPlanes; plane++) inp[plane] += multiplier * o[plane];
        }
    }
}

/* ------------------------------------------------------------ ballquery_batch_p */
/* reference: src/bfs_cluster/bfs_cluster.cu:15-60 (one CUDA thread per point) and the
 * retry loop of functions/pointgroup_ops.py:135-142 (done by the caller).
 * The reference reserves output segments with atomicAdd(cumsum,cnt) so `start` depends on
 * thread scheduling; this restatement visits points in ascending index, i.e. start =
 * exclusive prefix sum of len -- one of the orders the reference can produce.
 * Distance expression evaluated exactly as written (no FMA contraction; nvcc's default
 * -fmad=true may contract it on the reference's GPU -- unverifiable here).
 * Returns cumsum (total hits, may exceed n*meanActive => caller must retry). */
/* one point's scan: the reference thread body (bfs_cluster.cu:27-47) */
static int orc_bq_point(int pt, float radius2, const float *xyz, const int *batch_idxs, const int *batch_offsets,
                        int *idx_temp) {
    float o_x = xyz[pt * 3 + 0], o_y = xyz[pt * 3 + 1], o_z = xyz[pt * 3 + 2];
    int b = batch_idxs[pt];
    int start = batch_offsets[b], end = batch_offsets[b + 1];
    int cnt = 0;
    for (int k = start; k < end; k++) {
        float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        float d2 = (o_x - x) * (o_x - x) + (o_y - y) * (o_y - y) + (o_z - z) * (o_z - z);
        if (d2 < radius2) {
            if (cnt < 1000) idx_temp[cnt] = k; else break;
            ++cnt;
        }
    }
    return cnt;
}

/* Points are independent (one CUDA thread each in the reference), so the host restatement may spread them over the
 * cores (OpenMP; bench.py's cpu_baseline states the thread count): count pass, serial exclusive scan, fill pass.  The
 * result is identical to the single-threaded loop for any thread count. */
int orc_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz,
                          const int *batch_idxs, const int *batch_offsets, int *idx, int *start_len) {
    float radius2 = radius * radius;
    long long thre = (long long)n * meanActive;
#pragma omp parallel
    {
        int *idx_temp = (int *)malloc(sizeof(int) * 1000);
#pragma omp for schedule(dynamic, 128)
        for (int pt = 0; pt < n; pt++)
            start_len[pt * 2 + 1] = orc_bq_point(pt, radius2, xyz, batch_idxs, batch_offsets, idx_temp);
        free(idx_temp);
    }
    long long cumsum = 0;
    for (int pt = 0; pt < n; pt++) {
        start_len[pt * 2 + 0] = (int)cumsum;
        cumsum += start_len[pt * 2 + 1];
    }
#pragma omp parallel
    {
        int *idx_temp = (int *)malloc(sizeof(int) * 1000);
#pragma omp for schedule(dynamic, 128)
        for (int pt = 0; pt < n; pt++) {
            long long s = start_len[pt * 2 + 0];
            if (s >= thre) continue;
            int cnt = orc_bq_point(pt, radius2, xyz, batch_idxs, batch_offsets, idx_temp);
            if (s + cnt >= thre) cnt = (int)(thre - s);
            for (int k = 0; k < cnt; k++) idx[s + k] = idx_temp[k];
        }
        free(idx_temp);
    }
    return (int)cumsum;
}

/* ------------------------------------------------------------------ bfs_cluster */
/* reference: src/bfs_cluster/bfs_cluster.cpp:28-112 (find_cc, get_clusters,
 * fill_cluster_idxs_).  Two-phase for a C caller: returns sumNPoint and *nCluster, with
 * cluster_idxs (sumNPoint,2) / cluster_offsets (nCluster+1) malloc'ed (orc_free). */
int orc_bfs_cluster(const int *semantic_label, const Int *ball_query_idxs, const int *start_len,
                    int nPoint, int threshold, int **cluster_idxs_out, int **cluster_offsets_out,
                    int *nCluster_out) {
    int *visited = (int *)calloc((size_t)nPoint + 1, sizeof(int));
    Int *queue = (Int *)malloc(sizeof(Int) * ((size_t)nPoint + 1));
    Int *members = (Int *)malloc(sizeof(Int) * ((size_t)nPoint + 1)); /* kept clusters, concatenated */
    int *offsets = (int *)malloc(sizeof(int) * ((size_t)nPoint + 2));
    int nCluster = 0, sumNPoint = 0;
    offsets[0] = 0;
    for (Int i = 0; i < nPoint; i++) {
        if (visited[i]) continue;
        /* find_cc: FIFO BFS; the queue array doubles as cc.pt_idxs (visitation order) */
        int head = 0, tail = 0;
        queue[tail++] = i; visited[i] = 1;
        while (head < tail) {
            Int cur = queue[head++];
            int start = start_len[cur * 2], len = start_len[cur * 2 + 1];
            int label_cur = semantic_label[cur];
            for (Int k = start; k < start + len; k++) {
                Int j = ball_query_idxs[k];
                if (semantic_label[j] != label_cur) continue;
                if (visited[j] == 1) continue;
                visited[j] = 1;
                queue[tail++] = j;
            }
        }
        if (tail >= threshold) {
            memcpy(members + sumNPoint, queue, sizeof(Int) * (size_t)tail);
            sumNPoint += tail;
            offsets[++nCluster] = sumNPoint;
        }
    }
    int *ci = (int *)calloc((size_t)(sumNPoint > 0 ? sumNPoint : 1) * 2, sizeof(int));
    int *co = (int *)calloc((size_t)nCluster + 1, sizeof(int));
    for (int c = 0; c < nCluster; c++) {
        co[c + 1] = offsets[c + 1];
        for (int j = offsets[c]; j < offsets[c + 1]; j++) { ci[j * 2] = c; ci[j * 2 + 1] = members[j]; }
    }
    free(visited); free(queue); free(members); free(offsets);
    *cluster_idxs_out = ci; *cluster_offsets_out = co; *nCluster_out = nCluster;
    return sumNPoint;
}

/* ---------------------------------------------------------------------- roipool */
/* reference: src/roipool/roipool.cu:12-39.  `float max_val = -1e50` is -inf in float. */
void orc_roipool_fp(int nProposal, int C, const float *feats, const int *proposals_offset,
                    float *output_feats, int *output_maxidx) {
    for (int pp = 0; pp < nProposal; pp++) {
        int start = proposals_offset[pp], end = proposals_offset[pp + 1];
        for (int plane = 0; plane < C; plane++) {
            int argmax_idx = -1;
            float max_val = -INFINITY /* -1e50 -> -inf in float */;
            for (int i = start; i < end; i++) {
                if (feats[(size_t)i * C + plane] > max_val) { argmax_idx = i; max_val = feats[(size_t)i * C + plane]; }
            }
            output_maxidx[pp * C + plane] = argmax_idx;
            output_feats[pp * C + plane] = max_val;
        }
    }
}
/* reference: src/roipool/roipool.cu:42-57 (argmax -1 of an empty proposal would index out of
 * bounds there; skipped here). */
void orc_roipool_bp(int nProposal, int C, float *d_feats, const int *proposals_offset,
                    const int *output_maxidx, const float *d_output_feats) {
    (void)proposals_offset;
    for (int pp = 0; pp < nProposal; pp++)
        for (int plane = 0; plane < C; plane++) {
            int a = output_maxidx[pp * C + plane];
            if (a >= 0) d_feats[(size_t)a * C + plane] += d_output_feats[pp * C + plane];
        }
}

/* ---------------------------------------------------------------------- get_iou */
/* reference: src/get_iou/get_iou.cu:12-38.  `+ 1e-5` is a double literal, so the division
 * is evaluated in double and rounded to float on the store. */
void orc_get_iou(int nInstance, int nProposal, const int *proposals_idx, const int *proposals_offset,
                 const int64_t *instance_labels, const int *instance_pointnum, float *proposals_iou) {
    for (int p = 0; p < nProposal; p++) {
        int start = proposals_offset[p], end = proposals_offset[p + 1];
        int proposal_total = end - start;
        for (int inst = 0; inst < nInstance; inst++) {
            int instance_total = instance_pointnum[inst];
            int intersection = 0;
            for (int i = start; i < end; i++) {
                int idx = proposals_idx[i];
                if ((int)instance_labels[idx] == inst) intersection += 1;
            }
            proposals_iou[(size_t)p * nInstance + inst] =
                (float)((float)intersection / ((float)(proposal_total + instance_total - intersection) + 1e-5));
        }
    }
}

/* --------------------------------------------------------- sec_mean / min / max */
/* reference: src/sec_mean/sec_mean.cu:12-34 -- accumulates inp/count term by term. */
void orc_sec_mean(int nProposal, int C, const float *inp, const int *offsets, float *out) {
    for (int p = 0; p < nProposal; p++) {
        int start = offsets[p], end = offsets[p + 1];
        float count = (float)(end - start);
        for (int plane = 0; plane < C; plane++) {
            float mean = 0;
            for (int i = start; i < end; i++) mean += (inp[(size_t)i * C + plane] / count);
            out[p * C + plane] = mean;
        }
    }
}
/* reference: src/sec_mean/sec_mean.cu:38-60 (init 1e50 -> +inf in float) */
void orc_sec_min(int nProposal, int C, const float *inp, const int *offsets, float *out) {
    for (int p = 0; p < nProposal; p++) {
        int start = offsets[p], end = offsets[p + 1];
        for (int plane = 0; plane < C; plane++) {
            float min_val = INFINITY /* 1e50 -> +inf in float */;
            for (int i = start; i < end; i++)
                if (inp[(size_t)i * C + plane] < min_val) min_val = inp[(size_t)i * C + plane];
            out[p * C + plane] = min_val;
        }
    }
}
/* reference: src/sec_mean/sec_mean.cu:64-86 */
void orc_sec_max(int nProposal, int C, const float *inp, const int *offsets, float *out) {
    for (int p = 0; p < nProposal; p++) {
        int start = offsets[p], end = offsets[p + 1];
        for (int plane = 0; plane < C; plane++) {
            float max_val = -INFINITY /* -1e50 -> -inf in float */;
            for (int i = start; i < end; i++)
                if (inp[(size_t)i * C + plane] > max_val) max_val = inp[(size_t)i * C + plane];
            out[p * C + plane] = max_val;
        }
    }
}
