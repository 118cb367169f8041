// cider.hip -- CIDEr-D of the self-critical caption reward on the device (gfx950).
//
// Reference: lib/capeval/cider/cider_scorer.py:11-193 (`precook`, `compute_doc_freq`, `counts2vec`, `sim`) as called per
// RL step by lib/captioning/loss_helper.py:15-96 -- python dictionaries of word tuples on the host, twice per step (sampled
// and greedy captions), 27 ms of host time per step in the joint configuration (profiles/r02_p_joint_cprofile.txt) behind
// two device->host transfers of the sampled tokens.
// Here sentences are int32 token ids (a vocabulary id, or a corpus-private id >= V for a reference word outside the
// vocabulary, so string equality == id equality); an n-gram (n <= 4, ids < 65535) is ONE 64-bit key, 16 bits per token
// (id + 1: shorter n-grams keep zero high fields, keys of different orders never collide).  Three launches per call:
//   cd_df_kernel   one workgroup per distinct reference set: the set's distinct n-grams (LDS hash) -> global hash,
//                  count += number of batch entries that use the set (duplicates of a set count each time, :92-103);
//   cd_vec_kernel  one wave per sentence (references of the used sets, then the candidates): n-grams in the reference's
//                  insertion order (order 1..4, first occurrence by position), tf, tf-idf weight, per-order norm, bigram length;
//   cd_sim_kernel  one wave per entry: clipped cosine per (reference, order) with the Gaussian length penalty, summed in the
//                  reference's order.
// float64 like numpy; every accumulation follows the reference's order, so the scores agree with the host restatement
// (d3net_amd/cider.py, itself pinned to the reference's scorer by golden vectors) to the last bits of log / pow
// (tests/test_rl_gpu.py: 1e-12 relative).  Integer / hash work plus a few hundred flops: launch-bound.
#include "common.h"

#define CD_MAXT 160             // tokens per sentence (descriptions: max_lis_len 126 + eos; longer ones: the caller falls back)
#define CD_MAXG (4 * CD_MAXT)   // n-gram slots per sentence
#define CD_SETH 4096            // LDS hash slots per reference set (a set holds <= CD_SETH / 2 distinct n-grams)

__device__ __forceinline__ unsigned long long cd_key(const int *tok, int p, int n) {
    unsigned long long k = 0;
    for (int q = 0; q < n; q++) k |= (unsigned long long)(unsigned)(tok[p + q] + 1) << (16 * q);
    return k;
}
__device__ __forceinline__ unsigned int cd_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}

// corpus rows: tokens (R, ldt) int32, lens (R).  set u of the batch: rows slot_row[u_off[u] .. u_off[u+1])
__global__ __launch_bounds__(256) void cd_df_kernel(const int *__restrict__ tokens, int ldt, const int *__restrict__ lens,
                                                    const int *__restrict__ slot_row, const int *__restrict__ u_off,
                                                    const int *__restrict__ mult, unsigned long long *__restrict__ hkeys,
                                                    int *__restrict__ hcnt, unsigned int hmask, int *__restrict__ overflow) {
    __shared__ unsigned long long skey[CD_SETH];
    const int u = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < CD_SETH; i += 256) skey[i] = 0ull;
    __syncthreads();
    for (int s = u_off[u]; s < u_off[u + 1]; s++) {
        const int row = slot_row[s];
        const int *tok = tokens + (long long)row * ldt;
        const int L = min(lens[row], CD_MAXT);
        for (int i = t; i < 4 * L; i += 256) {
            const int n = i / L + 1, p = i - (n - 1) * L;
            if (p + n > L) continue;
            const unsigned long long k = cd_key(tok, p, n);
            unsigned int h = cd_hash(k) & (CD_SETH - 1);
            for (int probe = 0; probe < CD_SETH; probe++) {
                const unsigned long long old = atomicCAS(&skey[h], 0ull, k);
                if (old == 0ull || old == k) break;
                h = (h + 1) & (CD_SETH - 1);
                if (probe == CD_SETH - 1) *overflow = 1;
            }
        }
    }
    __syncthreads();
    const int m = mult[u];
    for (int i = t; i < CD_SETH; i += 256) {
        const unsigned long long k = skey[i];
        if (k == 0ull) continue;
        unsigned int h = cd_hash(k) & hmask;
        for (unsigned int probe = 0; probe <= hmask; probe++) {
            const unsigned long long old = atomicCAS(&hkeys[h], 0ull, k);
            if (old == 0ull || old == k) { atomicAdd(&hcnt[h], m); break; }
            h = (h + 1) & hmask;
            if (probe == hmask) *overflow = 1;
        }
    }
}

struct CdVec {                   // tf-idf vector of one sentence, entries grouped by order in insertion order
    unsigned long long key[CD_MAXG];
    double w[CD_MAXG];
    double norm[4];
    int off[5];                  // entries of order o: [off[o], off[o+1])
    int length;                  // number of bigrams
};

// slot s < SR: corpus row slot_row[s]; slot SR + e: candidate e (cand (E, ldc), clen (E); "eos" appended when absent)
__global__ __launch_bounds__(64) void cd_vec_kernel(const int *__restrict__ tokens, int ldt, const int *__restrict__ lens,
                                                    const int *__restrict__ slot_row, int SR, const int *__restrict__ cand, int ldc,
                                                    const int *__restrict__ clen, int eos, const unsigned long long *__restrict__ hkeys,
                                                    const int *__restrict__ hcnt, unsigned int hmask, double log_ref_len,
                                                    CdVec *__restrict__ vecs) {
    __shared__ int tok[CD_MAXT + 1];
    __shared__ unsigned long long gk[CD_MAXT];
    const int s = blockIdx.x, lane = threadIdx.x;
    int L;
    if (s < SR) {
        const int row = slot_row[s];
        L = min(lens[row], CD_MAXT);
        for (int i = lane; i < L; i += 64) tok[i] = tokens[(long long)row * ldt + i];
    } else {
        const int e = s - SR;
        L = min(clen[e], CD_MAXT);
        bool has = false;
        for (int i = lane; i < L; i += 64) { const int v = cand[(long long)e * ldc + i]; tok[i] = v; has |= (v == eos); }
        if (!__any(has) && L < CD_MAXT) { if (lane == 0) tok[L] = eos; L++; }   // `if "eos" not in tokens: tokens.append("eos")`
    }
    __syncthreads();
    CdVec &v = vecs[s];
    int total = 0;
    for (int n = 1; n <= 4; n++) {
        if (lane == 0) v.off[n - 1] = total;
        const int P = L - n + 1;                     // positions of this order
        for (int p = lane; p < P; p += 64) gk[p] = cd_key(tok, p, n);
        __syncthreads();
        for (int p0 = 0; p0 < P; p0 += 64) {         // rounds of 64 positions, in position order
            const int p = p0 + lane;
            const unsigned long long k = p < P ? gk[p] : 0ull;
            int tf = 0; bool first = p < P;
            if (p < P)
                for (int q = 0; q < P; q++)
                    if (gk[q] == k) { tf++; if (q < p) first = false; }
            const unsigned long long bal = __ballot(first);
            const int idx = total + __popcll(bal & ((1ull << lane) - 1ull));
            if (first) {
                int df = 0;
                unsigned int h = cd_hash(k) & hmask;
                for (unsigned int probe = 0; probe <= hmask; probe++) {
                    const unsigned long long hk = hkeys[h];
                    if (hk == k) { df = hcnt[h]; break; }
                    if (hk == 0ull) break;
                    h = (h + 1) & hmask;
                }
                const double d = log(fmax(1.0, (double)df));
                v.key[idx] = k;
                v.w[idx] = (double)tf * (log_ref_len - d);
            }
            total += __popcll(bal);
        }
        __syncthreads();
    }
    if (lane == 0) v.off[4] = total;
    __syncthreads();
    // per-order norms and the bigram count, summed in insertion order like the reference's loop over the dict
    if (lane < 4) {
        double sq = 0.0;
        for (int i = v.off[lane]; i < v.off[lane + 1]; i++) sq += v.w[i] * v.w[i];
        v.norm[lane] = sqrt(sq);
    }
    if (lane == 0) v.length = L >= 2 ? L - 1 : 0;
}

// entry e: candidate slot SR + e against the reference slots of set ent_u[e]
__global__ __launch_bounds__(64) void cd_sim_kernel(const CdVec *__restrict__ vecs, int SR, const int *__restrict__ ent_u,
                                                    const int *__restrict__ u_off, double sigma, double *__restrict__ scores) {
    __shared__ double val[16][4];
    const int e = blockIdx.x, lane = threadIdx.x;
    const CdVec &hyp = vecs[SR + e];
    const int u = ent_u[e], r0 = u_off[u], nref = u_off[u + 1] - r0;
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < nref; base += 16) {
        const int j = base + (lane >> 2), o = lane & 3;
        if (j < nref) {
            const CdVec &ref = vecs[r0 + j];
            const double delta = (double)(hyp.length - ref.length);
            const double penalty = pow(2.718281828459045, -(delta * delta) / (2.0 * sigma * sigma));
            double acc = 0.0;
            for (int i = hyp.off[o]; i < hyp.off[o + 1]; i++) {
                const unsigned long long k = hyp.key[i];
                double rv = 0.0;
                for (int q = ref.off[o]; q < ref.off[o + 1]; q++)
                    if (ref.key[q] == k) { rv = ref.w[q]; break; }
                acc += fmin(hyp.w[i], rv) * rv;
            }
            if (hyp.norm[o] != 0.0 && ref.norm[o] != 0.0) acc /= (hyp.norm[o] * ref.norm[o]);
            val[lane >> 2][o] = acc * penalty;
        }
        __syncthreads();
        if (lane == 0)
            for (int jj = 0; jj < min(16, nref - base); jj++)
                for (int oo = 0; oo < 4; oo++) acc4[oo] += val[jj][oo];
        __syncthreads();
    }
    if (lane == 0) {
        double s = ((acc4[0] + acc4[1]) + acc4[2]) + acc4[3];
        s = s / 4.0;
        s /= (double)nref;
        s *= 10.0;
        scores[e] = s;
    }
}

extern "C" size_t d3_cider_ws_bytes(int SR, int E, int hash_slots) {
    return d3_align((size_t)hash_slots * 8) + d3_align((size_t)hash_slots * 4) + d3_align(256) + d3_align((size_t)(SR + E) * sizeof(CdVec));
}

// tokens (R, ldt) / lens (R): the reference corpus on the device.  Batch description (small int32 device arrays):
// slot_row (SR) corpus rows of the reference sentences of the U used sets, grouped by set; u_off (U+1); mult (U) entries per
// set; ent_u (E) set of every entry.  cand (E, ldc) / clen (E): candidate token ids.  hash_slots: a power of two >= 2 x the
// distinct n-grams of the used sets.  scores (E) float64.  *overflow_dev != 0 afterwards: a hash was too small (caller retries
// bigger / falls back).
extern "C" int d3_cider_scores(const int *tokens, int ldt, const int *lens, const int *slot_row, const int *u_off, const int *mult,
                               const int *ent_u, int U, int SR, const int *cand, int ldc, const int *clen, int E, int eos, double sigma,
                               int hash_slots, double *scores, int *overflow_dev, void *ws, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (U < 1 || SR < 1 || E < 1 || hash_slots < 1024 || (hash_slots & (hash_slots - 1))) return D3_ERR_ARG;
    if (ws_bytes < d3_cider_ws_bytes(SR, E, hash_slots)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    D3Carver c(ws, ws_bytes);
    unsigned long long *hkeys = c.take<unsigned long long>(hash_slots);
    int *hcnt = c.take<int>(hash_slots);
    c.take<int>(64);
    CdVec *vecs = c.take<CdVec>((size_t)SR + E);
    D3_CHECK(hipMemsetAsync(hkeys, 0, (size_t)hash_slots * 8, s));
    D3_CHECK(hipMemsetAsync(hcnt, 0, (size_t)hash_slots * 4, s));
    D3_CHECK(hipMemsetAsync(overflow_dev, 0, sizeof(int), s));
    cd_df_kernel<<<U, 256, 0, s>>>(tokens, ldt, lens, slot_row, u_off, mult, hkeys, hcnt, (unsigned int)hash_slots - 1u, overflow_dev);
    cd_vec_kernel<<<SR + E, 64, 0, s>>>(tokens, ldt, lens, slot_row, SR, cand, ldc, clen, eos, hkeys, hcnt, (unsigned int)hash_slots - 1u,
                                        log((double)E), vecs);
    cd_sim_kernel<<<E, 64, 0, s>>>(vecs, SR, ent_u, u_off, sigma, scores);
    D3_LAUNCH_CHECK();
    return 0;
}
