// clusterprep.hip -- the index plumbing around PointGroup's two clusterings as two launches (gfx950).
//
// Reference: model/pointgroup.py:288-316.  Before the clusterings the object points (semantic class > 0) are compacted:
//     batch_idxs_ = batch_idxs[object_idxs]; coords_ = coords[object_idxs]; pt_offsets_ = pt_offsets[object_idxs];
//     semantic_preds_ = semantic_preds[object_idxs].int(); shifted = coords_ + pt_offsets_
// and after them the (cluster, compact point) pairs of both branches are mapped back to scene point ids, their batch ids
// looked up, the second branch's cluster ids / offsets shifted behind the first's and everything concatenated -- in the
// reference's way, including its one-element-short batch-id concatenation (:316: `proposals_batchId_shift_all[1:]`).
// As library ops that is ~25 launches per step (six 400-750 k-row gathers with 64-bit indices among them, ~0.5 ms);
// integer copies: bit-exact by construction (tests/test_pg_ops_gpu.py compares with the library-op chain).
#include "common.h"

// boff (optional, nb + 1 ints): the offsets of the object points' batch ids (model/pointgroup.py:110-122 get_batch_offsets: boff[b] =
// points with a batch id below b) written at the boundaries of the id column -- which the reference's callers hand over SORTED (the
// batch is a concatenation of scenes, and ballquery_batch_p reads [boff[b], boff[b+1]) as THE points of scene b).  For a column that
// steps down the boundaries are not the reference's counts: such a caller keeps PointGroup.get_batch_offsets
__global__ void cp_select_kernel(const float *__restrict__ locs, const float *__restrict__ offs, const long long *__restrict__ sem,
                                 const int *__restrict__ batch, const long long *__restrict__ obj, int n, int *__restrict__ batch_o,
                                 float *__restrict__ coords_o, float *__restrict__ shifted_o, int *__restrict__ sem_o,
                                 int *__restrict__ boff, int nb) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const long long o = obj[r];
    const int bi = batch[o];
    batch_o[r] = bi;
    if (boff) {
        const int bp = r > 0 ? batch[obj[r - 1]] : -1;
        for (int b = max(bp + 1, 0); b <= bi && b <= nb; b++) boff[b] = r;          // first point with an id >= b
        if (r == n - 1) for (int b = max(bi + 1, 0); b <= nb; b++) boff[b] = n;
    }
    sem_o[r] = (int)sem[o];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = locs[o * 3 + k];
        coords_o[(long long)r * 3 + k] = c;
        shifted_o[(long long)r * 3 + k] = __fadd_rn(c, offs[o * 3 + k]);
    }
}

extern "C" int d3_cluster_select(const float *locs, const float *pt_offsets, const int64_t *semantic_preds, const int *batch_idxs,
                                 const int64_t *object_idxs, int n, int *batch_out, float *coords_out, float *shifted_out,
                                 int *semantic_out, void *stream) {
    D3_CLEAR();
    if (n <= 0) return 0;
    cp_select_kernel<<<(n + 255) / 256, 256, 0, d3_stream(stream)>>>(locs, pt_offsets, (const long long *)semantic_preds, batch_idxs,
                                                                    (const long long *)object_idxs, n, batch_out, coords_out, shifted_out,
                                                                    semantic_out, nullptr, 0);
    D3_LAUNCH_CHECK();
    return 0;
}
// + batch_offsets_out (batch_size + 1 ints, see cp_select_kernel); n >= 1
extern "C" int d3_cluster_select2(const float *locs, const float *pt_offsets, const int64_t *semantic_preds, const int *batch_idxs,
                                  const int64_t *object_idxs, int n, int batch_size, int *batch_out, float *coords_out, float *shifted_out,
                                  int *semantic_out, int *batch_offsets_out, void *stream) {
    D3_CLEAR();
    if (n <= 0 || batch_size < 1 || !batch_offsets_out) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    cp_select_kernel<<<(n + 255) / 256, 256, 0, s>>>(locs, pt_offsets, (const long long *)semantic_preds, batch_idxs, (const long long *)object_idxs, n,
                                                    batch_out, coords_out, shifted_out, semantic_out, batch_offsets_out, batch_size);
    D3_LAUNCH_CHECK();
    return 0;
}

// idx1 (S1,2) / off1 (P1+1): clusters of the original coordinates; idx2 / off2: of the shifted ones (compact point ids).
// out_idx (S1+S2, 2): [cluster id (second set + P1), scene point id]; out_off (P1+P2+1); out_bid (max(S1+S2-1, 0)): batch id of
// every pair, the first pair of the second set dropped (the reference's concatenation).
__global__ void cp_merge_kernel(const int *__restrict__ idx1, int S1, const int *__restrict__ off1, int P1,
                                const int *__restrict__ idx2, int S2, const int *__restrict__ off2, int P2,
                                const long long *__restrict__ obj, const int *__restrict__ batch, int *__restrict__ out_idx,
                                int *__restrict__ out_off, int *__restrict__ out_bid) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < S1 + S2) {
        const bool second = e >= S1;
        const int *src = second ? idx2 + (long long)(e - S1) * 2 : idx1 + (long long)e * 2;
        const int pt = (int)obj[src[1]];
        out_idx[(long long)e * 2] = src[0] + (second ? P1 : 0);
        out_idx[(long long)e * 2 + 1] = pt;
        const int b = batch[pt];
        if (!second) out_bid[e] = b;
        else if (e > S1) out_bid[e - 1] = b;      // (pair S1, the first of the second set, has no slot)
    }
    if (e <= P1) out_off[e] = off1[e];
    else if (e <= P1 + P2) out_off[e] = off2[e - P1] + S1;
}

extern "C" int d3_cluster_merge(const int *idx1, int S1, const int *off1, int P1, const int *idx2, int S2, const int *off2, int P2,
                                const int64_t *object_idxs, const int *batch_idxs, int *out_idx, int *out_off, int *out_bid,
                                void *stream) {
    D3_CLEAR();
    if (S1 < 0 || S2 < 0 || P1 < 0 || P2 < 0) return D3_ERR_ARG;
    int n = S1 + S2; if (P1 + P2 + 1 > n) n = P1 + P2 + 1;
    cp_merge_kernel<<<(n + 255) / 256, 256, 0, d3_stream(stream)>>>(idx1, S1, off1, P1, idx2, S2, off2, P2, (const long long *)object_idxs,
                                                                   batch_idxs, out_idx, out_off, out_bid);
    D3_LAUNCH_CHECK();
    return 0;
}

// Per-proposal bookkeeping between the score head and the proposal selection (model/pointgroup.py:338-372): number of points,
// the score / size threshold mask, the batch id read at the cluster start (with the reference's one-element-short batch-id
// vector: the index is clamped to its last element), and the (P,9) crop box [centre | size | 0 | semantic class of the first
// point | score] -- a dozen library launches on a few hundred proposals.  sig = sigmoid(score), computed by the caller.
__global__ void cp_proposals_kernel(const float *__restrict__ sig, const int *__restrict__ offsets, const int *__restrict__ bid_all,
                                    int nbid, const int *__restrict__ pidx, const long long *__restrict__ sem,
                                    const float *__restrict__ center, const float *__restrict__ size, float score_thr,
                                    float npoint_thr, int P, float *__restrict__ npoint, unsigned char *__restrict__ mask,
                                    int *__restrict__ bid, float *__restrict__ crop) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int st = offsets[p];
    const float np = (float)(offsets[p + 1] - st);
    const float sg = sig[p];
    npoint[p] = np;
    mask[p] = (sg > score_thr && np > npoint_thr) ? 1 : 0;
    int s2 = st;
    if (s2 > nbid - 1) s2 = nbid - 1;
    if (s2 < 0) s2 = 0;
    bid[p] = nbid > 0 ? bid_all[s2] : 0;
    float *c = crop + (long long)p * 9;
    c[0] = center[p * 3]; c[1] = center[p * 3 + 1]; c[2] = center[p * 3 + 2];
    c[3] = size[p * 3]; c[4] = size[p * 3 + 1]; c[5] = size[p * 3 + 2];
    c[6] = 0.f;
    c[7] = crop != nullptr && sem != nullptr ? (float)sem[pidx[(long long)st * 2 + 1]] : 0.f;
    c[8] = sg;
}
extern "C" int d3_proposal_prepare(const float *sig, const int *offsets, const int *batch_id_all, int n_batch_id, const int *proposals_idx,
                                   const int64_t *semantic_preds, const float *center, const float *size, float score_thr,
                                   float npoint_thr, int P, float *npoint, unsigned char *mask, int *batch_id, float *crop, void *stream) {
    D3_CLEAR();
    if (P <= 0) return 0;
    cp_proposals_kernel<<<(P + 255) / 256, 256, 0, d3_stream(stream)>>>(sig, offsets, batch_id_all, n_batch_id, proposals_idx,
                                                                       (const long long *)semantic_preds, center, size, score_thr, npoint_thr,
                                                                       P, npoint, mask, batch_id, crop);
    D3_LAUNCH_CHECK();
    return 0;
}
