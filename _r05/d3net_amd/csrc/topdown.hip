// topdown.hip -- the top-down attention captioner of the speaker head as native gfx950 code: teacher-forced training
// pass (forward + backward through time) and the step used by the greedy / evaluation decodes.
//
// Reference: model/caption_module.py:72-133 (`TopDownSceneCaptionModule.step`), :510-687 (`_forward_sample_batch`, the XE
// driver).  One decode step there is
//     x1 = map_topdown([emb[word], h2, target]);  h1 = GRUCell1(x1, h1)
//     a  = softmax_k( attend . tanh(map_feat(obj)[k] + map_hidd(h1)) , masked scores := 0 );  att = sum_k a[k] obj[k]
//     x2 = map_lang([att, h1]);  h2 = GRUCell2(x2, h2);  logits = classifier(h2)
// issued as ~25 library kernels per step (31 steps, then ~2x that in the backward: ~4,000 launches per training step at
// batch 32 -- 24 ms of the 74 ms PointGroup+speaker step, profiles/r02_a).  Restructured here without changing a result:
//   * everything that does not depend on the recurrence is batched over the S time steps: the embedding + target part
//     of map_topdown (a 3-segment GEMM with gathered rows), map_feat(obj), and the whole classifier (two GEMMs over
//     S*N = 992 rows instead of 62 over 32 rows) -- teacher forcing makes every input word known up front;
//   * a step is 6 launches: x1 GEMM (h2 segment + static addend), fused GRUCell (both gate GEMMs + gate math in one
//     kernel), map_hidd GEMM, fused attention (tanh scores of the <= num_locals unmasked proposals only, softmax,
//     weighted sum), map_lang GEMM (two segments, no concat), fused GRUCell;
//   * backward through time: per step 8 launches (GRU gate backward, two data-gradient GEMMs sharing a launch, ...);
//     all weight gradients are batched over time afterwards (k-major GEMMs with K = S*N rows), bias gradients are column
//     sums; activations are kept (45 MB -- 288 GB of HBM: nothing is recomputed except the attention tanh);
//   * the host loop is native: one C-ABI call for the forward, one for the backward.
// Arithmetic: fp32 throughout (v_mfma_f32_16x16x4_f32 for the products: exact fp32), expf / tanhf from the device
// library.  Results match the reference's own module to summation order (tests/test_speaker_gpu.py, golden vectors).
// Roofline: launch / latency bound (0.3 GFLOP and ~20 MB of L2-resident weights per step); the measure is launches and
// microseconds per step, reported by bench.py's profile.
#include "common.h"
#include "prof.h"
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef GRU_U
#define GRU_U 2      // k blocks of 16 per wave and batch of loads (812 / 16 = 51 blocks over 8 waves; 4 measured slower: 12.4 -> 13.4 us)
#endif

int hg_launch(const d3_gemm_prob *probs, int nprobs, hipStream_t s);
size_t hg_colsum_ws_bytes(int njobs, int cmax);
int hg_colsum_multi(const float *const *x, const long long *ld, const int *R, const int *C, float *const *out, const int *accum, int n,
                    void *ws, size_t ws_bytes, hipStream_t s);

// ------------------------------------------------------------------------------ fused GRUCell forward
// h' = GRUCell(x, h) (torch.nn.GRUCell semantics: r, z, n gate order; n = tanh(W_in x + b_in + r * (W_hn h + b_hn))).
// A workgroup owns 16 hidden units (the three gate rows j, H+j, 2H+j of both weight matrices) x RT*16 batch rows; its 4
// waves split the K loop over [x | h] and are summed through LDS; the gate math runs on the reduced tile.
struct GruArgs {
    const float *x; long long ldx; int I;
    const float *h; long long ldh;
    const float *Wih, *Whh, *bih, *bhh;
    float *hout; long long ldo;
    float *r, *z, *n, *ghn;      // (N,H) each, kept for the backward (NULL: inference)
    int N, H;
    // packed-sequence form (LangModule): the input-side gates x W_ih^T + b_ih are precomputed for all steps (gi_pre (N, 3H)
    // rows at stride ldgi; x / Wih / bih unused), rows with t_step >= lens[row] keep their state (and report an identity
    // step to the backward), hid_out (row stride ldhid) receives the step's output, zero for finished rows
    const float *gi_pre; long long ldgi;
    const int *lens; int t_step;
    float *hid_out; long long ldhid;
    // second input-side segment and an explicit row stride of Wih (round 3: the captioner's COMPOSED input weights -- GRU-2
    // reads [attended | h1] against Wih2 W_lang, GRU-1 reads h2 against Wih1 W_td[:, h2] on top of gates precomputed for all
    // steps): Wih is (3H, I + Ib) with row stride ldw (0: I); x may be NULL (no first segment); with gi_pre AND x the two add
    const float *xb; long long ldxb; int Ib; long long ldw;
};

__device__ __forceinline__ f32x4 td_load4(const float *row, int k0, int K, bool valid) {
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!valid || k0 >= K) return v;
    if (k0 + 3 < K) return *(const f32x4 *)(row + k0);
#pragma unroll
    for (int s = 0; s < 4; s++) if (k0 + s < K) v[s] = row[k0 + s];
    return v;
}

#define GRU_NW 8     // waves per workgroup: [x | h] is 51 k blocks at (300, 512): two batches of loads per wave
template <int RT>
__global__ __launch_bounds__(GRU_NW * 64) void td_gru_fwd_kernel(const GruArgs a) {
    __shared__ float red[GRU_NW * 4 * RT * 256];   // [wave][acc][rt][q][lane]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int ct = blockIdx.x, rg = blockIdx.y, H = a.H;
    f32x4 accR[RT], accZ[RT], accNI[RT], accNH[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) { accR[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; accZ[r] = accR[r]; accNI[r] = accR[r]; accNH[r] = accR[r]; }
    const int col = ct * 16 + i;           // hidden unit of this lane's B rows (H % 16 == 0: host check)
    int gkb = 0;
#pragma unroll
    for (int seg = 0; seg < 3; seg++) {      // input-side segment(s), then the recurrent one
        if (seg == 0 && !a.x) continue;
        if (seg == 1 && !a.xb) continue;
        const int K = seg == 0 ? a.I : seg == 1 ? a.Ib : H;
        const float *X = seg == 0 ? a.x : seg == 1 ? a.xb : a.h;
        const long long ldx = seg == 0 ? a.ldx : seg == 1 ? a.ldxb : a.ldh;
        const long long ldw = seg == 2 ? (long long)H : (a.ldw ? a.ldw : (long long)a.I);
        const float *W = seg == 0 ? a.Wih : seg == 1 ? a.Wih + a.I : a.Whh;
        const int nkb = (K + 15) >> 4;
        const float *xr[RT];
        bool xv[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int row = (rg * RT + r) * 16 + i;
            xv[r] = row < a.N;
            xr[r] = X + (long long)(xv[r] ? row : 0) * ldx;
        }
        const float *w0 = W + (long long)col * ldw, *w1 = W + (long long)(H + col) * ldw, *w2 = W + (long long)(2 * H + col) * ldw;
        const int first = (wave - gkb) & (GRU_NW - 1);
        gkb += nkb;
        constexpr int U = GRU_U;
        for (int kb0 = first; kb0 < nkb; kb0 += GRU_NW * U) {
            f32x4 xa[U][RT], b0[U], b1[U], b2[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k0 = (kb0 + GRU_NW * u) * 16 + g * 4;
                const bool in = kb0 + GRU_NW * u < nkb;
                b0[u] = td_load4(w0, k0, K, in); b1[u] = td_load4(w1, k0, K, in); b2[u] = td_load4(w2, k0, K, in);
#pragma unroll
                for (int r = 0; r < RT; r++) xa[u][r] = td_load4(xr[r], k0, K, in && xv[r]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (kb0 + GRU_NW * u < nkb) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int r = 0; r < RT; r++) {
                            accR[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][r][q], b0[u][q], accR[r], 0, 0, 0);
                            accZ[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][r][q], b1[u][q], accZ[r], 0, 0, 0);
                            if (seg < 2) accNI[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][r][q], b2[u][q], accNI[r], 0, 0, 0);
                            else accNH[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][r][q], b2[u][q], accNH[r], 0, 0, 0);
                        }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            red[(((wave * 4 + 0) * RT + r) * 4 + q) * 64 + lane] = accR[r][q];
            red[(((wave * 4 + 1) * RT + r) * 4 + q) * 64 + lane] = accZ[r][q];
            red[(((wave * 4 + 2) * RT + r) * 4 + q) * 64 + lane] = accNI[r][q];
            red[(((wave * 4 + 3) * RT + r) * 4 + q) * 64 + lane] = accNH[r][q];
        }
    __syncthreads();
    for (int e = t; e < RT * 256; e += GRU_NW * 64) {
        const int r = e >> 8, q = (e >> 6) & 3, ln = e & 63;
        const int row = (rg * RT + r) * 16 + (ln >> 4) * 4 + q, c = ct * 16 + (ln & 15);
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < GRU_NW; w++) v += red[(((w * 4 + k) * RT + r) * 4 + q) * 64 + ln];
            s[k] = v;
        }
        if (row >= a.N) continue;
        const float hp = a.h[(long long)row * a.ldh + c];
        const long long o = (long long)row * H + c;
        if (a.lens && a.t_step >= a.lens[row]) {     // finished sequence: carry the state, emit zeros (pad_packed_sequence)
            a.hout[(long long)row * a.ldo + c] = hp;
            if (a.hid_out) a.hid_out[(long long)row * a.ldhid + c] = 0.f;
            if (a.r) { a.r[o] = 0.f; a.z[o] = 1.f; a.n[o] = 0.f; a.ghn[o] = 0.f; }   // identity step for the backward
            continue;
        }
        float gr = s[0], gz = s[1], gn = s[2];     // (s[0], s[1]: input + recurrent parts; s[2]: input part, 0 without x segments)
        if (a.gi_pre) {
            const float *gi = a.gi_pre + (long long)row * a.ldgi;
            gr += gi[c]; gz += gi[H + c]; gn += gi[2 * H + c];
        } else { gr += a.bih[c]; gz += a.bih[H + c]; gn += a.bih[2 * H + c]; }
        const float rr = 1.f / (1.f + expf(-(gr + a.bhh[c])));
        const float zz = 1.f / (1.f + expf(-(gz + a.bhh[H + c])));
        const float gh = s[3] + a.bhh[2 * H + c];
        const float nv = tanhf(gn + rr * gh);
        const float hn = (1.f - zz) * nv + zz * hp;
        a.hout[(long long)row * a.ldo + c] = hn;
        if (a.hid_out) a.hid_out[(long long)row * a.ldhid + c] = hn;
        if (a.r) { a.r[o] = rr; a.z[o] = zz; a.n[o] = nv; a.ghn[o] = gh; }
    }
}

// The same cell with the GATES packed into the MFMA columns (round 3): a workgroup owns 4 hidden units and its 16 columns are
// (r, z, n-input, n-hidden) x 4 units -- a lane's weight row is Wih / Whh row gate * H + unit (no row for the n-hidden column in
// the input segments and for the n-input column in the recurrent one), ONE accumulator per row tile.  The cell is bound by the
// fp32 MFMA chain of the few workgroups that hold it (16 units: 12 products per k block and row tile on 32 compute units =
// 5.5 us of matrix time at N = 32, H = 512); here 128 workgroups issue 4 products per k block, and 16 waves split the k blocks.
#define GRU4_NW 16
#ifndef GRU4_U
#define GRU4_U 3
#endif
template <int RT>
__global__ __launch_bounds__(GRU4_NW * 64) void td_gru4_fwd_kernel(const GruArgs a) {
    __shared__ float red[GRU4_NW * RT * 256];   // [wave][rt][q][lane]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int ut = blockIdx.x, rg = blockIdx.y, H = a.H;
    const int gate = i >> 2, unit = ut * 4 + (i & 3);
    f32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int gkb = 0;
#pragma unroll
    for (int seg = 0; seg < 3; seg++) {      // input-side segment(s), then the recurrent one
        if (seg == 0 && !a.x) continue;
        if (seg == 1 && !a.xb) continue;
        const int K = seg == 0 ? a.I : seg == 1 ? a.Ib : H;
        const float *X = seg == 0 ? a.x : seg == 1 ? a.xb : a.h;
        const long long ldx = seg == 0 ? a.ldx : seg == 1 ? a.ldxb : a.ldh;
        const long long ldw = seg == 2 ? (long long)H : (a.ldw ? a.ldw : (long long)a.I);
        const float *W = seg == 0 ? a.Wih : seg == 1 ? a.Wih + a.I : a.Whh;
        const int nkb = (K + 15) >> 4;
        const float *xr[RT];
        bool xv[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int row = (rg * RT + r) * 16 + i;
            xv[r] = row < a.N;
            xr[r] = X + (long long)(xv[r] ? row : 0) * ldx;
        }
        const bool wlive = gate < 2 || (gate == 2 ? seg < 2 : seg == 2);
        const float *w = W + (long long)((gate == 3 ? 2 : gate) * H + unit) * ldw;
        const int first = (wave - gkb) & (GRU4_NW - 1);
        gkb += nkb;
        constexpr int U = GRU4_U;
        for (int kb0 = first; kb0 < nkb; kb0 += GRU4_NW * U) {
            f32x4 xa[U][RT], b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k0 = (kb0 + GRU4_NW * u) * 16 + g * 4;
                const bool in = kb0 + GRU4_NW * u < nkb;
                b[u] = td_load4(w, k0, K, in && wlive);
#pragma unroll
                for (int r = 0; r < RT; r++) xa[u][r] = td_load4(xr[r], k0, K, in && xv[r]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (kb0 + GRU4_NW * u < nkb) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int r = 0; r < RT; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][r][q], b[u][q], acc[r], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) red[((wave * RT + r) * 4 + q) * 64 + lane] = acc[r][q];
    __syncthreads();
    // D layout: column lane & 15, row (lane >> 4) * 4 + q.  One thread per (row, unit): its four gate sums are columns gate * 4 + unit
    for (int e = t; e < RT * 64; e += GRU4_NW * 64) {
        const int r = e >> 6, rt = (e >> 2) & 15, uu = e & 3;
        const int row = (rg * RT + r) * 16 + rt, c = ut * 4 + uu;
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ln = (rt >> 2) * 16 + k * 4 + uu, q = rt & 3;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < GRU4_NW; w2++) v += red[((w2 * RT + r) * 4 + q) * 64 + ln];
            s[k] = v;
        }
        if (row >= a.N) continue;
        const float hp = a.h[(long long)row * a.ldh + c];
        const long long o = (long long)row * H + c;
        if (a.lens && a.t_step >= a.lens[row]) {     // finished sequence: carry the state, emit zeros (pad_packed_sequence)
            a.hout[(long long)row * a.ldo + c] = hp;
            if (a.hid_out) a.hid_out[(long long)row * a.ldhid + c] = 0.f;
            if (a.r) { a.r[o] = 0.f; a.z[o] = 1.f; a.n[o] = 0.f; a.ghn[o] = 0.f; }   // identity step for the backward
            continue;
        }
        float gr = s[0], gz = s[1], gn = s[2];
        if (a.gi_pre) {
            const float *gi = a.gi_pre + (long long)row * a.ldgi;
            gr += gi[c]; gz += gi[H + c]; gn += gi[2 * H + c];
        } else { gr += a.bih[c]; gz += a.bih[H + c]; gn += a.bih[2 * H + c]; }
        const float rr = 1.f / (1.f + expf(-(gr + a.bhh[c])));
        const float zz = 1.f / (1.f + expf(-(gz + a.bhh[H + c])));
        const float gh = s[3] + a.bhh[2 * H + c];
        const float nv = tanhf(gn + rr * gh);
        const float hn = (1.f - zz) * nv + zz * hp;
        a.hout[(long long)row * a.ldo + c] = hn;
        if (a.hid_out) a.hid_out[(long long)row * a.ldhid + c] = hn;
        if (a.r) { a.r[o] = rr; a.z[o] = zz; a.n[o] = nv; a.ghn[o] = gh; }
    }
}

static int td_gru_fwd(const GruArgs &a, hipStream_t s) {
    if (a.H & 15) return D3_ERR_ARG;
    if (a.N <= 0) return 0;
    if (d3_tune(D3T_GRU4) != 0 && a.N <= 256) {   // gate-packed columns: 4 hidden units per workgroup
        // launch timing (bench.py): a GRU cell = two skinny fp32 GEMMs (N x 3H x I, N x 3H x H) + gates; bytes = weights + rows once
        const int I = a.gi_pre ? 0 : a.I;
        void *pr = d3_prof_begin(4, 4.0 * (3.0 * a.H * (I + a.H) + (double)a.N * (I + 6.0 * a.H)), 2.0 * a.N * 3.0 * a.H * (I + a.H), s);
        td_gru4_fwd_kernel<1><<<dim3(a.H / 4, (a.N + 15) / 16), GRU4_NW * 64, 0, s>>>(a);
        if (pr) { d3_prof_tag(pr, 0, 1); d3_prof_tag(pr, 1, a.N); d3_prof_tag(pr, 2, a.H); d3_prof_tag(pr, 3, I); d3_prof_end(pr, s); }
        D3_LAUNCH_CHECK();
        return 0;
    }
    // (17..64 rows: one 16-row tile per workgroup -- twice the workgroups, half the MFMA chain of each; D3_GRU_RT1=0: two tiles)
    if (a.N <= 16 || (a.N <= 64 && d3_tune(D3T_GRU_RT1) != 0)) td_gru_fwd_kernel<1><<<dim3(a.H / 16, (a.N + 15) / 16), GRU_NW * 64, 0, s>>>(a);
    else td_gru_fwd_kernel<2><<<dim3(a.H / 16, (a.N + 31) / 32), GRU_NW * 64, 0, s>>>(a);
    D3_LAUNCH_CHECK();
    return 0;
}

// GRUCell backward, gate part: dh' = d0 + d1 + d2 (up to three contributions, NULL = none) ->
//   dgi (N,3H) = [dr_pre, dz_pre, dn_pre], dgh (N,3H) = [dr_pre, dz_pre, dn_pre * r], dhp (N,H) = dh' * z
__global__ void td_gru_bwd_gates_kernel(const float *d0, long long ld0, const float *d1, long long ld1, const float *d2, long long ld2,
                                        const float *__restrict__ r, const float *__restrict__ z, const float *__restrict__ n,
                                        const float *__restrict__ ghn, const float *__restrict__ hp, long long ldh,
                                        float *__restrict__ dgi, long long lddgi, float *__restrict__ dgh, float *__restrict__ dhp,
                                        int N, int H, const int *__restrict__ lens, int t_step) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * H) return;
    const int row = e / H, c = e - row * H;
    float dh = 0.f;
    if (d0) dh += d0[(long long)row * ld0 + c];
    if (d1 && !(lens && t_step >= lens[row])) dh += d1[(long long)row * ld1 + c];   // (a finished row's output is the constant 0)
    if (d2) dh += d2[(long long)row * ld2 + c];
    const float rr = r[e], zz = z[e], nv = n[e];
    const float dn = dh * (1.f - zz), dz = dh * (hp[(long long)row * ldh + c] - nv);
    const float dnp = dn * (1.f - nv * nv);
    const float drp = dnp * ghn[e] * rr * (1.f - rr);
    const float dzp = dz * zz * (1.f - zz);
    const long long o = (long long)row * 3 * H + c, oi = (long long)row * lddgi + c;
    dgi[oi] = drp; dgi[oi + H] = dzp; dgi[oi + 2 * H] = dnp;
    dgh[o] = drp; dgh[o + H] = dzp; dgh[o + 2 * H] = dnp * rr;
    dhp[e] = dh * zz;
}

// ------------------------------------------------------------------------------ top-down attention
// scores[k] = attend . tanh(map_feat(obj)[k] + map_hidd(h1)); masked proposals get score 0 -- `masked_fill_(mask == 0, 0)`, not
// -inf: caption_module.py:112-114 -- and still take part in the softmax.  With num_locals = 10 only ~10 of the K = 128
// proposals are unmasked, so the kernels work on the ACTIVE list only and treat the masked ones in closed form:
//   every masked k has the same probability a_m = exp(0 - max) / sum;  attended = a_m * (sum of the masked objects'
//   features, constant over the time steps: `msum`) + sum_active a[k] obj[k];
//   backward: sum_k a[k] da[k] = datt . attended (no K x F product), ds[k] != 0 only for active k; the gradient w.r.t. the
//   object features, dobj[k] = sum_t a_t[k] datt_t, is accumulated after the time loop from the saved a / datt.
#define TD_PREP_T 1024
__global__ __launch_bounds__(TD_PREP_T) void td_attn_prep_kernel(const float *__restrict__ mask, const float *__restrict__ obj,
                                                                 int *__restrict__ act, int *__restrict__ nact, float *__restrict__ msum,
                                                                 int K, int F, int obj_div) {
    extern __shared__ float sm[];          // (TD_PREP_T / F) * F partial sums
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const long long ns = n / obj_div;
    if (t < 64) {                          // wave 0: active list in ascending k (ballot + prefix popcount)
        int base = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const bool on = k < K && mask[(long long)n * K + k] != 0.f;
            const unsigned long long bal = __ballot(on);
            if (on) act[(long long)n * K + base + __popcll(bal & ((1ull << lane) - 1ull))] = k;
            base += __popcll(bal);
        }
        if (lane == 0) nact[n] = base;
    }
    // masked objects' feature sum: TD_PREP_T / F slices of the proposals, each summed in ascending k (loads unconditional and
    // batched, the mask is a factor), the slices then added in slice order: fixed order, deterministic
    const int nsl = TD_PREP_T / F, sl = t / F, c = t - sl * F;
    if (sl < nsl) {
        const int per = (K + nsl - 1) / nsl, k0 = sl * per, k1 = min(K, k0 + per);
        float s = 0.f;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            float v[4], m[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { v[q] = obj[(ns * K + k + q) * F + c]; m[q] = mask[(long long)n * K + k + q]; }
#pragma unroll
            for (int q = 0; q < 4; q++) s += (m[q] == 0.f) ? v[q] : 0.f;
        }
        for (; k < k1; k++)
            if (mask[(long long)n * K + k] == 0.f) s += obj[(ns * K + k) * F + c];
        sm[sl * F + c] = s;
    }
    __syncthreads();
    if (t < F) {
        float s = sm[t];
        for (int q = 1; q < nsl; q++) s += sm[q * F + t];
        msum[(long long)n * F + t] = s;
    }
}

#ifndef TD_ATT_T
#define TD_ATT_T 1024          // threads of the per-sample attention workgroups (256: rounds 2-4)
#endif
static inline size_t td_attn_bwd_rsum_floats(int H) { const int CT = H < TD_ATT_T ? H : TD_ATT_T, RS = TD_ATT_T / CT; return RS > 1 ? (size_t)RS * 2 * H : 0; }
// one workgroup per sample; attn_out: (N, K, S) slice t of `topdown_attn` (NULL: not wanted)
// NTH threads (round 5: 1024 -- the N <= 32 workgroups of a step are all the kernel has, and a sample's na x H tanh evaluations on
// four waves were 9.6 us of a 31-step dependent chain; sixteen waves split the proposals: same expressions, same summation order)
template <int NTH>
__global__ __launch_bounds__(NTH) void td_attn_fwd_kernel(const float *__restrict__ fp, const float *__restrict__ q, long long ldq,
                                                          const float *__restrict__ watt, const float *__restrict__ obj,
                                                          const int *__restrict__ act, const int *__restrict__ nact,
                                                          const float *__restrict__ msum, float *__restrict__ a_out,
                                                          float *__restrict__ att, long long ldatt, float *__restrict__ attn_out,
                                                          int t_step, int S, int K, int H, int F, int obj_div) {
    extern __shared__ float sm[];
    float *qs = sm, *ws = qs + H, *sc = ws + H, *ak = sc + K, *part = ak + K;   // part: 2*F
    __shared__ float am_s;
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long ns = n / obj_div;      // obj / fp row block of this sample (evaluation: the K targets of a scene share it)
    const int na = nact[n], nm = K - na;
    const int *al = act + (long long)n * K;
    for (int c = t; c < H; c += NTH) { qs[c] = q[(long long)n * ldq + c]; ws[c] = watt[c]; }
    __syncthreads();
    for (int j = wave; j < na; j += NTH / 64) {
        const float *row = fp + (ns * K + al[j]) * H;
        float s = 0.f;
        for (int c = lane * 4; c < H; c += 256) {
            const f32x4 v = *(const f32x4 *)(row + c);
#pragma unroll
            for (int u = 0; u < 4; u++) s += ws[c + u] * tanhf(v[u] + qs[c + u]);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) sc[j] = s;
    }
    __syncthreads();
    if (wave == 0) {   // softmax over all K proposals: nm of them at score 0
        float mx = nm > 0 ? 0.f : -3.0e38f;
        for (int j = lane; j < na; j += 64) mx = fmaxf(mx, sc[j]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = lane; j < na; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float e0 = expf(-mx);
        sum += (float)nm * e0;
        const float inv = 1.f / sum;
        for (int j = lane; j < na; j += 64) sc[j] *= inv;
        if (lane == 0) am_s = e0 * inv;
    }
    __syncthreads();
    const float am = am_s;
    for (int k = t; k < K; k += NTH) ak[k] = am;
    __syncthreads();
    for (int j = t; j < na; j += NTH) ak[al[j]] = sc[j];
    __syncthreads();
    for (int k = t; k < K; k += NTH) {
        a_out[(long long)n * K + k] = ak[k];
        if (attn_out) attn_out[((long long)n * K + k) * S + t_step] = ak[k];
    }
    // attended[c] = a_m * msum[c] + sum_active a[k] obj[k, c]: two halves of the active list per column, fixed order
    const int half = t / F, c = t - half * F;
    if (half < 2) {
        const int j0 = half * ((na + 1) / 2), j1 = min(na, j0 + (na + 1) / 2);
        float s = 0.f;
        for (int j = j0; j < j1; j++) s += sc[j] * obj[(ns * K + al[j]) * F + c];
        part[half * F + c] = s;
    }
    __syncthreads();
    if (t < F) att[(long long)n * ldatt + t] = am * msum[(long long)n * F + t] + (part[t] + part[F + t]);
}

// backward of one step: datt (N,F) -> dq (N,H), dfp rows of the active proposals +=, dwpart row (H); datt is also copied to
// dattS (N,F) for the accumulation of dobj after the time loop
// NTH threads: CT = min(H, NTH) column threads x RS = NTH / CT row slices (slice r takes the row quads r, r + RS, ...; the slices'
// dq / dw sums are added in slice order through LDS: deterministic; NTH = 256 is the rounds-2-4 kernel)
template <int NTH>
__global__ __launch_bounds__(NTH) void td_attn_bwd_kernel(const float *__restrict__ datt, long long lddatt, const float *__restrict__ a_in,
                                                          const float *__restrict__ att, long long ldatt, const float *__restrict__ fp,
                                                          const float *__restrict__ q, long long ldq, const float *__restrict__ watt,
                                                          const float *__restrict__ obj, const int *__restrict__ act,
                                                          const int *__restrict__ nact, float *__restrict__ dq, long long lddq,
                                                          float *__restrict__ dfp, float *__restrict__ dwpart, float *__restrict__ dattS,
                                                          int K, int H, int F) {
    extern __shared__ float sm[];
    float *dat = sm, *dss = dat + F, *red = dss + K, *rsum = red + 4;   // red: 4; rsum: (RS - 1) * 2 * H (row-slice partials)
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int na = nact[n];
    const int *al = act + (long long)n * K;
    float pd = 0.f;
    for (int c = t; c < F; c += NTH) {
        const float v = datt[(long long)n * lddatt + c];
        dat[c] = v; dattS[(long long)n * F + c] = v;
        pd += v * att[(long long)n * ldatt + c];
    }
    for (int o = 32; o > 0; o >>= 1) pd += __shfl_xor(pd, o);
    if (lane == 0 && wave < 4) red[wave] = pd;      // (F <= 256: the waves beyond hold no element)
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);      // sum_k a[k] da[k] = datt . attended
    for (int j = wave; j < na; j += NTH / 64) {
        const int k = al[j];
        float s = 0.f;
        for (int c = lane; c < F; c += 64) s += dat[c] * obj[((long long)n * K + k) * F + c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) dss[j] = a_in[(long long)n * K + k] * (s - dot);
    }
    __syncthreads();
    const int CT = H < NTH ? H : NTH, RS = NTH / CT, rs = t / CT;
    for (int c = t - rs * CT; c < H && rs < RS; c += CT) {
        const float qc = q[(long long)n * ldq + c], wc = watt[c];
        float dqa = 0.f, dwa = 0.f;
        for (int j0 = rs * 4; j0 < na; j0 += 4 * RS) {          // four rows' loads in flight
            float fv[4], dv[4];
            long long o[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u < na ? j0 + u : j0;
                o[u] = ((long long)n * K + al[j]) * H + c;
                fv[u] = fp[o[u]]; dv[u] = dfp[o[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j0 + u < na) {
                    const float ds = dss[j0 + u];
                    const float th = tanhf(fv[u] + qc);
                    const float dp = ds * wc * (1.f - th * th);
                    dfp[o[u]] = dv[u] + dp;
                    dqa += dp; dwa += ds * th;
                }
            }
        }
        if (rs == 0 && RS == 1) { dq[(long long)n * lddq + c] = dqa; dwpart[(long long)n * H + c] = dwa; }
        else if (rs > 0) { rsum[((rs - 1) * 2) * H + c] = dqa; rsum[((rs - 1) * 2 + 1) * H + c] = dwa; }
        else { rsum[(RS - 1) * 2 * H + c] = dqa; rsum[(RS - 1) * 2 * H + H + c] = dwa; }      // (slice 0 parks its sums behind the others')
    }
    if (RS > 1) {
        __syncthreads();
        for (int c = t; c < H; c += NTH) {
            float dqa = rsum[(RS - 1) * 2 * H + c], dwa = rsum[(RS - 1) * 2 * H + H + c];
            for (int r = 1; r < RS; r++) { dqa += rsum[((r - 1) * 2) * H + c]; dwa += rsum[((r - 1) * 2 + 1) * H + c]; }
            dq[(long long)n * lddq + c] = dqa;
            dwpart[(long long)n * H + c] = dwa;
        }
    }
}

// dobj[n, k, c] = sum_t a[t, n, k] * datt[t, n, c]  (after the time loop; S*(K+F) floats staged per sample)
__global__ __launch_bounds__(256) void td_dobj_kernel(const float *__restrict__ a, const float *__restrict__ dattS, float *__restrict__ dobj,
                                                      int S, int N, int K, int F, int SC) {
    extern __shared__ float sm[];
    float *aS = sm, *dS = aS + (size_t)SC * K;
    const int n = blockIdx.x, t = threadIdx.x;
    const int c = t % F, kb = t / F, kst = 256 / F;       // F divides 256 (host check)
    for (int s0 = 0; s0 < S; s0 += SC) {
        const int sc = min(SC, S - s0);
        __syncthreads();
        for (int e = t; e < sc * K; e += 256) { const int tt = e / K, k = e - tt * K; aS[e] = a[((long long)(s0 + tt) * N + n) * K + k]; }
        for (int e = t; e < sc * F; e += 256) { const int tt = e / F, cc = e - tt * F; dS[e] = dattS[((long long)(s0 + tt) * N + n) * F + cc]; }
        __syncthreads();
        // (the proposals are split over gridDim.y workgroups: one workgroup per sample left 32 CUs walking 32 k outputs each)
        const int kper = (K + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * kper, k1 = min(K, k0 + kper);
        for (int k = k0 + kb; k < k1; k += kst) {
            float v = 0.f;
            for (int tt = 0; tt < sc; tt++) v += aS[tt * K + k] * dS[tt * F + c];
            float *o = dobj + ((long long)n * K + k) * F + c;
            *o = s0 == 0 ? v : *o + v;
        }
    }
}

// ------------------------------------------------------------------------------ small helpers
// (token ids are clamped into the vocabulary: an out-of-range id must not become an out-of-bounds gather)
__global__ void td_rows_kernel(const long long *__restrict__ word_ids, int Tw, int N, int S, int V, int *__restrict__ widx,
                               int *__restrict__ nidx, int *__restrict__ bidx) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;   // time-major row r = t * N + n
    if (r >= S * N) return;
    const int tt = r / N, n = r - tt * N;
    const long long w = word_ids[(long long)n * Tw + (tt < Tw ? tt : Tw - 1)];
    widx[r] = (int)(w < 0 ? 0 : w >= V ? V - 1 : w);
    nidx[r] = n;
    bidx[r] = n * S + tt;                                  // the same row in batch-major order
}
// dst[r, :] = src[idx[r], :]  (time-major copies of batch-major / vocabulary-indexed rows)
__global__ void td_gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ idx, float *__restrict__ dst, long long R, int C) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * C) return;
    const long long r = e / C;
    const int c = (int)(e - r * C);
    dst[e] = src[(long long)idx[r] * C + c];
}
// dst = relu'(c0) * src (in place allowed)
__global__ void td_relu_mask_kernel(float *__restrict__ d, const float *__restrict__ c0, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && c0[e] <= 0.f) d[e] = 0.f;
}
// out[n, c] = sum_t x[(t*N + n), c]
__global__ void td_sum_time_kernel(const float *__restrict__ x, float *__restrict__ out, int S, int N, int C) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * C) return;
    float s = 0.f;
    for (int tt = 0; tt < S; tt++) s += x[(long long)tt * N * C + e];
    out[e] = s;
}

static d3_gemm_seg td_seg(const float *A, long long lda, const float *B, long long ldb, int K, const int *ia = nullptr, int akm = 0, int bkm = 0) {
    d3_gemm_seg s;
    s.A = A; s.ia = ia; s.lda = lda; s.a_kmajor = akm; s.B = B; s.ldb = ldb; s.b_kmajor = bkm; s.K = K;
    return s;
}
static d3_gemm_prob td_prob(int M, int N, float *C, long long ldc) {
    d3_gemm_prob p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.C = C; p.ldc = ldc;
    return p;
}

// ------------------------------------------------------------------------------ workspace layout
struct TdLayout {
    size_t widx, nidx, bidx, act, nact, msum, fp, TD, x1, x2, H1, H2, g1, g2, q, a, att, c0, gi1, Wa, Wb, bc2, total;   // byte offsets; g1/g2: r,z,n,ghn blocks
};
static TdLayout td_layout(int N, int K, int S, int H, int E, int F) {
    TdLayout L;
    size_t o = 0;
    const size_t R = (size_t)S * N;
    auto take = [&](size_t bytes) { size_t at = o; o += d3_align(bytes); return at; };
    L.widx = take(R * 4); L.nidx = take(R * 4); L.bidx = take(R * 4);
    L.act = take((size_t)N * K * 4); L.nact = take((size_t)N * 4); L.msum = take((size_t)N * F * 4);
    L.fp = take((size_t)N * K * H * 4);
    L.TD = take(R * E * 4); L.x1 = take(R * E * 4); L.x2 = take(R * E * 4);
    L.H1 = take((R + N) * H * 4); L.H2 = take((R + N) * H * 4);
    L.g1 = take(4 * R * H * 4); L.g2 = take(4 * R * H * 4);
    L.q = take(R * H * 4); L.a = take(R * K * 4); L.att = take(R * F * 4); L.c0 = take(R * H * 4);
    // composed weights (shared with the backward) and GRU-1's input-side gates for all steps
    L.gi1 = take(R * 3 * H * 4); L.Wa = take((size_t)3 * H * (F + H) * 4); L.Wb = take((size_t)3 * H * H * 4); L.bc2 = take((size_t)3 * H * 4);
    L.total = o;
    return L;
}

extern "C" size_t d3_topdown_ws_bytes(int N, int K, int S, int H, int E, int F) {
    return td_layout(N, K, S, H, E, F).total;
}

static int td_check(const d3_topdown_args *a) {
    if (!a || a->N < 1 || a->K < 1 || a->S < 1 || a->V < 1 || a->S > a->Tw) return D3_ERR_ARG;
    if ((a->H & 15) || (a->E & 3) || (a->F & 3) || a->F > 128 || (256 % a->F) || a->K > 1024) return D3_ERR_ARG;
    if (a->ws_bytes < td_layout(a->N, a->K, a->S, a->H, a->E, a->F).total) return D3_ERR_WORKSPACE;
    return 0;
}

// Teacher-forced forward over S steps (model/caption_module.py:636-668): logits (N,S,V), attn (N,K,S).
extern "C" int d3_topdown_xe_forward(const d3_topdown_args *a, void *stream) {
    D3_CLEAR();
    int rc = td_check(a);
    if (rc) return rc;
    hipStream_t s = d3_stream(stream);
    const int N = a->N, K = a->K, S = a->S, V = a->V, H = a->H, E = a->E, F = a->F, R = S * N;
    const TdLayout L = td_layout(N, K, S, H, E, F);
    char *ws = (char *)a->ws;
    int *widx = (int *)(ws + L.widx), *nidx = (int *)(ws + L.nidx), *bidx = (int *)(ws + L.bidx);
    float *fp = (float *)(ws + L.fp), *TD = (float *)(ws + L.TD), *x1 = (float *)(ws + L.x1), *x2 = (float *)(ws + L.x2);
    float *H1 = (float *)(ws + L.H1), *H2 = (float *)(ws + L.H2), *g1 = (float *)(ws + L.g1), *g2 = (float *)(ws + L.g2);
    float *q = (float *)(ws + L.q), *av = (float *)(ws + L.a), *att = (float *)(ws + L.att), *c0 = (float *)(ws + L.c0);
    int *act = (int *)(ws + L.act), *nact = (int *)(ws + L.nact);
    float *msum = (float *)(ws + L.msum);
    const long long ldtd = H + F + E;          // map_topdown weight: (E, E + H + F) over [emb | h2 | target]
    const long long ldlang = F + H;            // map_lang weight: (E, F + H) over [attended | h1]
    td_rows_kernel<<<(R + 255) / 256, 256, 0, s>>>(a->word_ids, a->Tw, N, S, V, widx, nidx, bidx);
    td_attn_prep_kernel<<<N, TD_PREP_T, (size_t)(TD_PREP_T / F) * F * 4, s>>>(a->mask, a->obj, act, nact, msum, K, F, 1);
    D3_CHECK(hipMemsetAsync(H1, 0, (size_t)N * H * 4, s));
    D3_CHECK(hipMemsetAsync(H2, 0, (size_t)N * H * 4, s));
    {   // batched, recurrence-free parts: map_feat(obj) and the [embedding | target] part of map_topdown (+ bias)
        d3_gemm_prob p[2];
        p[0] = td_prob(N * K, H, fp, H);
        p[0].nseg = 1; p[0].seg[0] = td_seg(a->obj, F, a->W_feat, F, F);
        p[1] = td_prob(R, E, TD, E);
        p[1].nseg = 2;
        p[1].seg[0] = td_seg(a->emb, E, a->W_td, ldtd, E, widx);
        p[1].seg[1] = td_seg(a->target, F, a->W_td + E + H, ldtd, F, nidx);
        p[1].bias = a->b_td;
        if ((rc = hg_launch(&p[0], 1, s))) return rc;
        if ((rc = hg_launch(&p[1], 1, s))) return rc;
    }
    // The recurrence is a chain of dependent ~10 us launches; two of its six links per step are compositions of linear maps and
    // are taken out of it by composing the weights once per call (as the backward does):
    //   GRU-1 input gates = Wih1 (TD[t] + W_td[:, h2] h2) + b = GI1[t] + (Wih1 W_td[:, h2]) h2 = GI1[t] + Wb h2      (GI1 batched over t)
    //   GRU-2 input gates = Wih2 (W_lang [att | h1] + b_lang) + b = (Wih2 W_lang) [att | h1] + (Wih2 b_lang + b) = Wa [att | h1] + bc2
    // x1 / x2 themselves (operands of the weight gradients) are computed after the loop, batched over time: 6 -> 4 launches per step.
    float *gi1 = (float *)(ws + L.gi1), *Wa = (float *)(ws + L.Wa), *Wb = (float *)(ws + L.Wb), *bc2 = (float *)(ws + L.bc2);
    {
        d3_gemm_prob p[4];
        p[0] = td_prob(3 * H, F + H, Wa, F + H);
        p[0].nseg = 1; p[0].seg[0] = td_seg(a->Wih2, E, a->W_lang, ldlang, E, nullptr, 0, 1);
        p[1] = td_prob(3 * H, H, Wb, H);
        p[1].nseg = 1; p[1].seg[0] = td_seg(a->Wih1, E, a->W_td + E, ldtd, E, nullptr, 0, 1);
        p[2] = td_prob(3 * H, 1, bc2, 1);                      // bc2 = Wih2 b_lang + bih2
        p[2].nseg = 1; p[2].seg[0] = td_seg(a->Wih2, E, a->b_lang, E, E);
        p[2].add = a->bih2; p[2].ldadd = 1;
        p[3] = td_prob(R, 3 * H, gi1, 3 * H);                  // GI1 = TD Wih1^T + bih1, all steps
        p[3].nseg = 1; p[3].seg[0] = td_seg(TD, E, a->Wih1, E, E);
        p[3].bias = a->bih1;
        if ((rc = hg_launch(p, 3, s))) return rc;
        if ((rc = hg_launch(&p[3], 1, s))) return rc;
    }
    const size_t RH = (size_t)R * H;
    for (int t = 0; t < S; t++) {
        const size_t rN = (size_t)t * N;
        float *h1p = H1 + rN * H, *h1n = H1 + (rN + N) * H, *h2p = H2 + rN * H, *h2n = H2 + (rN + N) * H;
        {
            GruArgs g{h2p, H, H, h1p, H, Wb, a->Whh1, nullptr, a->bhh1, h1n, H,
                      g1 + rN * H, g1 + RH + rN * H, g1 + 2 * RH + rN * H, g1 + 3 * RH + rN * H, N, H, gi1 + rN * 3 * H, 3 * H, nullptr, 0, nullptr, 0,
                      nullptr, 0, 0, H};
            if ((rc = td_gru_fwd(g, s))) return rc;
        }
        {   // q = map_hidd(h1)
            d3_gemm_prob p = td_prob(N, H, q + rN * H, H);
            p.nseg = 1; p.seg[0] = td_seg(h1n, H, a->W_hidd, H, H);
            if ((rc = hg_launch(&p, 1, s))) return rc;
        }
        td_attn_fwd_kernel<TD_ATT_T><<<N, TD_ATT_T, (size_t)(2 * H + 2 * K + 2 * F) * 4, s>>>(fp, q + rN * H, H, a->w_att, a->obj, act, nact, msum, av + rN * K,
                                                                             att + rN * F, F, a->attn, t, S, K, H, F, 1);
        {
            GruArgs g{att + rN * F, F, F, h2p, H, Wa, a->Whh2, bc2, a->bhh2, h2n, H,
                      g2 + rN * H, g2 + RH + rN * H, g2 + 2 * RH + rN * H, g2 + 3 * RH + rN * H, N, H, nullptr, 0, nullptr, 0, nullptr, 0,
                      h1n, H, H, F + H};
            if ((rc = td_gru_fwd(g, s))) return rc;
        }
    }
    {   // x1 = TD + H2[:-1] W_td[:, h2]^T ; x2 = [att | H1[1:]] W_lang^T + b_lang -- all steps, one launch (operands of the backward)
        d3_gemm_prob p[2];
        p[0] = td_prob(R, E, x1, E);
        p[0].nseg = 1; p[0].seg[0] = td_seg(H2, H, a->W_td + E, ldtd, H);
        p[0].add = TD; p[0].ldadd = E;
        p[1] = td_prob(R, E, x2, E);
        p[1].nseg = 2;
        p[1].seg[0] = td_seg(att, F, a->W_lang, ldlang, F);
        p[1].seg[1] = td_seg(H1 + (size_t)N * H, H, a->W_lang + F, ldlang, H);
        p[1].bias = a->b_lang;
        if ((rc = hg_launch(p, 2, s))) return rc;
    }
    {   // classifier over all steps: c0 = relu(h2 Wc0^T + b), logits (batch-major rows) = c0 Wc2^T + b
        d3_gemm_prob p = td_prob(R, H, c0, H);
        p.nseg = 1; p.seg[0] = td_seg(H2 + (size_t)N * H, H, a->Wc0, H, H);
        p.bias = a->bc0; p.relu = 1;
        if ((rc = hg_launch(&p, 1, s))) return rc;
        d3_gemm_prob p2 = td_prob(R, V, a->logits, V);
        p2.nseg = 1; p2.seg[0] = td_seg(c0, H, a->Wc2, H, H);
        p2.bias = a->bc2; p2.perm_nb = N; p2.perm_s = S;
        if ((rc = hg_launch(&p2, 1, s))) return rc;
    }
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ backward through time
struct TdBwdLayout { size_t dlog, dc0, dH2, dgi1, dgh1, dgi2, dgh2, dx1, dx2, dq, tmpL, dh1q, dh1c, dh2c, dfp, dwp, dx1s, dattS, Wa, Wb, cs, cs_bytes, total; };
static TdBwdLayout td_bwd_layout(int N, int K, int S, int V, int H, int E, int F) {
    TdBwdLayout L;
    size_t o = 0;
    const size_t R = (size_t)S * N;
    auto take = [&](size_t bytes) { size_t at = o; o += d3_align(bytes); return at; };
    L.dlog = take(R * V * 4); L.dc0 = take(R * H * 4); L.dH2 = take(R * H * 4);
    L.dgi1 = take(R * 3 * H * 4); L.dgh1 = take(R * 3 * H * 4); L.dgi2 = take(R * 3 * H * 4); L.dgh2 = take(R * 3 * H * 4);
    L.dx1 = take(R * E * 4); L.dx2 = take(R * E * 4); L.dq = take(R * H * 4);
    L.tmpL = take((size_t)N * (F + H) * 4); L.dh1q = take((size_t)N * H * 4); L.dh1c = take((size_t)N * H * 4); L.dh2c = take((size_t)N * H * 4);
    L.dfp = take((size_t)N * K * H * 4); L.dwp = take(R * H * 4); L.dx1s = take((size_t)N * E * 4);
    L.dattS = take(R * F * 4);
    L.Wa = take((size_t)3 * H * (F + H) * 4); L.Wb = take((size_t)3 * H * H * 4);   // composed weights of the backward chain
    L.cs_bytes = hg_colsum_ws_bytes(8, V > 3 * H ? V : 3 * H); L.cs = take(L.cs_bytes);
    L.total = o;
    return L;
}

extern "C" size_t d3_topdown_bwd_ws_bytes(int N, int K, int S, int V, int H, int E, int F) {
    return td_bwd_layout(N, K, S, V, H, E, F).total;
}

// dlogits (N,S,V) -> every parameter gradient (written), dobj (N,K,F) and dtarget (N,F) (written).
// side (optional, round 5): a second stream for everything that only feeds PARAMETER gradients (the weight-gradient GEMMs batched over
// time, the bias column sums, the classifier's two weight gradients: ~0.3 ms of throughput-bound launches) -- the caller's stream
// then carries just the chain the rest of the backward waits for (dc0, dH2, the S-step recurrence, dobj / dtarget).  The function
// forks `side` off the caller's stream itself (events); JOINING is the caller's job: `side` must be waited for before anything reads
// a parameter gradient, and every buffer of `a` / `gd` must stay alive until then.
extern "C" int d3_topdown_xe_backward_ex(const d3_topdown_args *a, const d3_topdown_grads *gd, void *stream, void *side);
extern "C" int d3_topdown_xe_backward(const d3_topdown_args *a, const d3_topdown_grads *gd, void *stream) {
    return d3_topdown_xe_backward_ex(a, gd, stream, nullptr);
}
extern "C" int d3_topdown_xe_backward_ex(const d3_topdown_args *a, const d3_topdown_grads *gd, void *stream, void *side) {
    D3_CLEAR();
    int rc = td_check(a);
    if (rc) return rc;
    if (!gd || !gd->dlogits || !gd->ws) return D3_ERR_ARG;
    hipStream_t s = d3_stream(stream);
    hipStream_t sp = side ? (hipStream_t)side : s;          // the stream of the parameter-gradient work
    static thread_local hipEvent_t fork_ev[2] = {nullptr, nullptr};
    auto fork = [&](int k) -> int {                          // sp continues behind everything enqueued on s so far
        if (sp == s) return 0;
        if (!fork_ev[k]) D3_CHECK(hipEventCreateWithFlags(&fork_ev[k], hipEventDisableTiming));
        D3_CHECK(hipEventRecord(fork_ev[k], s));
        D3_CHECK(hipStreamWaitEvent(sp, fork_ev[k], 0));
        return 0;
    };
    const int N = a->N, K = a->K, S = a->S, V = a->V, H = a->H, E = a->E, F = a->F, R = S * N;
    const TdLayout L = td_layout(N, K, S, H, E, F);
    const TdBwdLayout B = td_bwd_layout(N, K, S, V, H, E, F);
    if (gd->ws_bytes < B.total) return D3_ERR_WORKSPACE;
    char *ws = (char *)a->ws, *bw = (char *)gd->ws;
    int *widx = (int *)(ws + L.widx), *bidx = (int *)(ws + L.bidx);
    float *fp = (float *)(ws + L.fp), *x1 = (float *)(ws + L.x1), *x2 = (float *)(ws + L.x2);
    float *H1 = (float *)(ws + L.H1), *H2 = (float *)(ws + L.H2), *g1 = (float *)(ws + L.g1), *g2 = (float *)(ws + L.g2);
    float *q = (float *)(ws + L.q), *av = (float *)(ws + L.a), *att = (float *)(ws + L.att), *c0 = (float *)(ws + L.c0);
    float *dlog = (float *)(bw + B.dlog), *dc0 = (float *)(bw + B.dc0), *dH2 = (float *)(bw + B.dH2);
    float *dgi1 = (float *)(bw + B.dgi1), *dgh1 = (float *)(bw + B.dgh1), *dgi2 = (float *)(bw + B.dgi2), *dgh2 = (float *)(bw + B.dgh2);
    float *dx1 = (float *)(bw + B.dx1), *dx2 = (float *)(bw + B.dx2), *dq = (float *)(bw + B.dq), *tmpL = (float *)(bw + B.tmpL);
    float *dh1q = (float *)(bw + B.dh1q), *dh1c = (float *)(bw + B.dh1c), *dh2c = (float *)(bw + B.dh2c);
    float *dfp = (float *)(bw + B.dfp), *dwp = (float *)(bw + B.dwp), *dx1s = (float *)(bw + B.dx1s), *dattS = (float *)(bw + B.dattS);
    float *Wa = (float *)(ws + L.Wa), *Wb = (float *)(ws + L.Wb);      // composed by the forward (same call's workspace)
    const int *act = (const int *)(ws + L.act), *nact = (const int *)(ws + L.nact);
    const long long ldtd = H + F + E, ldlang = F + H;
    const size_t RH = (size_t)R * H;
    // ---- classifier (batched over time).  dlog: time-major copy of dlogits (gathered rows), then
    //      dWc2 = dlog^T c0, dc0 = dlog Wc2 (relu-masked), dWc0 = dc0^T h2, dH2 = dc0 Wc0
    {
        d3_gemm_prob p = td_prob(R, H, dc0, H);
        p.nseg = 1; p.seg[0] = td_seg(gd->dlogits, V, a->Wc2, H, V, bidx, 0, 1);
        if ((rc = hg_launch(&p, 1, s))) return rc;
        td_relu_mask_kernel<<<(int)((RH + 255) / 256), 256, 0, s>>>(dc0, c0, (long long)RH);
    }
    if ((rc = fork(0))) return rc;
    {
        // time-major copy of dlogits for the k-major weight gradient (both operands must walk the rows in the same order)
        const long long tot = (long long)R * V;
        td_gather_rows_kernel<<<(int)((tot + 255) / 256), 256, 0, sp>>>(gd->dlogits, bidx, dlog, R, V);
        d3_gemm_prob p[3];
        p[0] = td_prob(V, H, gd->dWc2, H);
        p[0].nseg = 1; p[0].seg[0] = td_seg(dlog, V, c0, H, R, nullptr, 1, 1);
        p[1] = td_prob(H, H, gd->dWc0, H);
        p[1].nseg = 1; p[1].seg[0] = td_seg(dc0, H, H2 + (size_t)N * H, H, R, nullptr, 1, 1);
        p[2] = td_prob(R, H, dH2, H);
        p[2].nseg = 1; p[2].seg[0] = td_seg(dc0, H, a->Wc0, H, H, nullptr, 0, 1);
        if ((rc = hg_launch(&p[2], 1, s))) return rc;          // (the recurrence waits for dH2: first, on the caller's stream)
        if ((rc = hg_launch(&p[0], 1, sp))) return rc;
        if ((rc = hg_launch(&p[1], 1, sp))) return rc;
        const float *cx[2] = {dlog, dc0}; const long long cl[2] = {V, H}; const int cr[2] = {R, R}, cc[2] = {V, H};
        float *co[2] = {gd->dbc2, gd->dbc0};
        if ((rc = hg_colsum_multi(cx, cl, cr, cc, co, nullptr, 2, bw + B.cs, B.cs_bytes, sp))) return rc;
    }
    D3_CHECK(hipMemsetAsync(dh1c, 0, (size_t)N * H * 4, s));
    D3_CHECK(hipMemsetAsync(dh2c, 0, (size_t)N * H * 4, s));
    D3_CHECK(hipMemsetAsync(dfp, 0, (size_t)N * K * H * 4, s));
    // The backward recurrence is a chain of dependent ~8 us launches; two links are pure compositions of linear maps and are
    // taken out of it by composing the weights once per call (0.5 GFLOP):
    //   [datt | dh1 part] = (dgi2 Wih2) W_lang     = dgi2 Wa,  Wa = Wih2 W_lang        (3H x (F+H))
    //   dh2 carry        += (dgi1 Wih1) W_td[:, h2] = dgi1 Wb,  Wb = Wih1 W_td[:, E:E+H] (3H x H)
    // so dx2 / dx1 (still needed, for the weight gradients batched over time) leave the critical path: 8 -> 6 launches per step.
    const int nh = (N * H + 255) / 256;
    // Round 5: the two gate kernels of a step ride in the epilogues of the GEMMs that complete their input (d3_gemm_prob.gru): GRU-1's
    // gates behind dq W_hidd (the last of dh1's three contributions), GRU-2's gates of step t-1 behind dgi1 Wb (the last update of
    // the carried dh2) -- 6 -> 4 dependent launches per step; only the first step's GRU-2 gates keep a launch of their own.
    const bool fuse = d3_tune(D3T_TD_FUSE_GATES) != 0;
    auto gates_epi = [&](d3_gemm_prob &p, const float *d0, long long ld0, const float *d1, long long ld1, const float *g, size_t rN,
                         const float *hp, float *dgi, float *dgh, float *dhp) {
        p.gru = 1; p.gru_H = H;
        p.g_d0 = d0; p.g_ld0 = ld0; p.g_d1 = d1; p.g_ld1 = ld1;
        p.g_r = g + rN * H; p.g_z = g + RH + rN * H; p.g_n = g + 2 * RH + rN * H; p.g_ghn = g + 3 * RH + rN * H;
        p.g_hp = hp; p.g_ldh = H;
        p.g_dgi = dgi; p.g_lddgi = 3 * H; p.g_dgh = dgh; p.g_dhp = dhp;
    };
    for (int t = S - 1; t >= 0; t--) {
        const size_t rN = (size_t)t * N;
        float *h1p = H1 + rN * H, *h2p = H2 + rN * H;
        // GRU2 gates: dh2[t+1] = classifier part + carry from step t+1
        if (!fuse || t == S - 1)
            td_gru_bwd_gates_kernel<<<nh, 256, 0, s>>>(dH2 + rN * H, H, dh2c, H, nullptr, 0, g2 + rN * H, g2 + RH + rN * H, g2 + 2 * RH + rN * H,
                                                       g2 + 3 * RH + rN * H, h2p, H, dgi2 + rN * 3 * H, 3 * H, dgh2 + rN * 3 * H, dh2c, N, H, nullptr, 0);
        {   // dh2c += dgh2 Whh2 ; [datt | dh1 part] = dgi2 Wa ; dx2 = dgi2 Wih2 (off the chain)   (one launch)
            d3_gemm_prob p[3];
            p[0] = td_prob(N, H, dh2c, H);
            p[0].nseg = 1; p[0].seg[0] = td_seg(dgh2 + rN * 3 * H, 3 * H, a->Whh2, H, 3 * H, nullptr, 0, 1); p[0].accum = 1;
            p[1] = td_prob(N, F + H, tmpL, F + H);
            p[1].nseg = 1; p[1].seg[0] = td_seg(dgi2 + rN * 3 * H, 3 * H, Wa, F + H, 3 * H, nullptr, 0, 1);
            p[2] = td_prob(N, E, dx2 + rN * E, E);
            p[2].nseg = 1; p[2].seg[0] = td_seg(dgi2 + rN * 3 * H, 3 * H, a->Wih2, E, 3 * H, nullptr, 0, 1);
            if ((rc = hg_launch(p, 3, s))) return rc;
        }
        td_attn_bwd_kernel<TD_ATT_T><<<N, TD_ATT_T, (size_t)(F + K + 4 + td_attn_bwd_rsum_floats(H)) * 4, s>>>(tmpL, F + H, av + rN * K, att + rN * F, F, fp, q + rN * H, H, a->w_att,
                                                                 a->obj, act, nact, dq + rN * H, H, dfp, dwp + rN * H, dattS + rN * F, K, H, F);
        {   // dh1 (through map_hidd) = dq W_hidd  [+ GRU1 gates on dh1 = carry + map_lang part + this]
            d3_gemm_prob p = td_prob(N, H, dh1q, H);
            p.nseg = 1; p.seg[0] = td_seg(dq + rN * H, H, a->W_hidd, H, H, nullptr, 0, 1);
            if (fuse) gates_epi(p, dh1c, H, tmpL + F, F + H, g1, rN, h1p, dgi1 + rN * 3 * H, dgh1 + rN * 3 * H, dh1c);
            if ((rc = hg_launch(&p, 1, s))) return rc;
        }
        if (!fuse)
            td_gru_bwd_gates_kernel<<<nh, 256, 0, s>>>(dh1c, H, tmpL + F, F + H, dh1q, H, g1 + rN * H, g1 + RH + rN * H, g1 + 2 * RH + rN * H,
                                                       g1 + 3 * RH + rN * H, h1p, H, dgi1 + rN * 3 * H, 3 * H, dgh1 + rN * 3 * H, dh1c, N, H, nullptr, 0);
        {   // dh1c += dgh1 Whh1 ; dh2c += dgi1 Wb  [+ GRU2 gates of step t-1 on dh2 = classifier part + this carry] ; dx1 = dgi1 Wih1 (off the chain)
            d3_gemm_prob p[3];
            p[0] = td_prob(N, H, dh1c, H);
            p[0].nseg = 1; p[0].seg[0] = td_seg(dgh1 + rN * 3 * H, 3 * H, a->Whh1, H, 3 * H, nullptr, 0, 1); p[0].accum = 1;
            p[1] = td_prob(N, H, dh2c, H);
            p[1].nseg = 1; p[1].seg[0] = td_seg(dgi1 + rN * 3 * H, 3 * H, Wb, H, 3 * H, nullptr, 0, 1); p[1].accum = 1;
            if (fuse && t > 0) {
                const size_t rP = (size_t)(t - 1) * N;
                gates_epi(p[1], dH2 + rP * H, H, nullptr, 0, g2, rP, H2 + rP * H, dgi2 + rP * 3 * H, dgh2 + rP * 3 * H, dh2c);
            }
            p[2] = td_prob(N, E, dx1 + rN * E, E);
            p[2].nseg = 1; p[2].seg[0] = td_seg(dgi1 + rN * 3 * H, 3 * H, a->Wih1, E, 3 * H, nullptr, 0, 1);
            if ((rc = hg_launch(p, 3, s))) return rc;
        }
    }
    // ---- what the rest of the backward waits for, on the caller's stream: dtarget and dobj
    td_sum_time_kernel<<<(N * E + 255) / 256, 256, 0, s>>>(dx1, dx1s, S, N, E);      // (sum_t dx1[t]: the target feature is constant in t)
    if ((rc = fork(1))) return rc;
    {
        d3_gemm_prob p = td_prob(N, F, gd->dtarget, F);       // dtarget = dx1s W_td[:, E+H:]
        p.nseg = 1; p.seg[0] = td_seg(dx1s, E, a->W_td + E + H, ldtd, E, nullptr, 0, 1);
        if ((rc = hg_launch(&p, 1, s))) return rc;
        // dobj = sum_t a_t (x) datt_t (the attention's weighted sum), then += dfp W_feat (through map_feat)
        const int SC = S < 32 ? S : 32;
        td_dobj_kernel<<<dim3(N, K >= 64 ? 8 : 1), 256, (size_t)SC * (K + F) * 4, s>>>(av, dattS, gd->dobj, S, N, K, F, SC);
        d3_gemm_prob po = td_prob(N * K, F, gd->dobj, F);
        po.nseg = 1; po.seg[0] = td_seg(dfp, H, a->W_feat, F, H, nullptr, 0, 1); po.accum = 1;
        if ((rc = hg_launch(&po, 1, s))) return rc;
    }
    // ---- weight gradients, batched over time (k-major operands, K = R rows): the parameter-gradient stream
    {
        d3_gemm_prob p[4];
        // GRU cells
        p[0] = td_prob(3 * H, E, gd->dWih2, E); p[0].nseg = 1; p[0].seg[0] = td_seg(dgi2, 3 * H, x2, E, R, nullptr, 1, 1);
        p[1] = td_prob(3 * H, H, gd->dWhh2, H); p[1].nseg = 1; p[1].seg[0] = td_seg(dgh2, 3 * H, H2, H, R, nullptr, 1, 1);
        p[2] = td_prob(3 * H, E, gd->dWih1, E); p[2].nseg = 1; p[2].seg[0] = td_seg(dgi1, 3 * H, x1, E, R, nullptr, 1, 1);
        p[3] = td_prob(3 * H, H, gd->dWhh1, H); p[3].nseg = 1; p[3].seg[0] = td_seg(dgh1, 3 * H, H1, H, R, nullptr, 1, 1);
        if ((rc = hg_launch(p, 4, sp))) return rc;
        // all remaining bias gradients (and the attention vector's) in one two-stage column sum
        const float *cx[7] = {dgi2, dgh2, dgi1, dgh1, dx2, dx1, dwp};
        const long long cl[7] = {3 * H, 3 * H, 3 * H, 3 * H, E, E, H};
        const int cr[7] = {R, R, R, R, R, R, R}, cc[7] = {3 * H, 3 * H, 3 * H, 3 * H, E, E, H};
        float *co[7] = {gd->dbih2, gd->dbhh2, gd->dbih1, gd->dbhh1, gd->db_lang, gd->db_td, gd->dw_att};
        if ((rc = hg_colsum_multi(cx, cl, cr, cc, co, nullptr, 7, bw + B.cs, B.cs_bytes, sp))) return rc;
    }
    {
        // map_lang: dW (E, F+H) = dx2^T [att | h1[1:]] ; map_hidd: dW = dq^T h1[1:] ; map_topdown: dW (E, E+H+F) = dx1^T [emb[w] | h2[:-1] | target]
        d3_gemm_prob p[4];
        p[0] = td_prob(E, F, gd->dW_lang, ldlang); p[0].nseg = 1; p[0].seg[0] = td_seg(dx2, E, att, F, R, nullptr, 1, 1);
        p[1] = td_prob(E, H, gd->dW_lang + F, ldlang); p[1].nseg = 1; p[1].seg[0] = td_seg(dx2, E, H1 + (size_t)N * H, H, R, nullptr, 1, 1);
        p[2] = td_prob(H, H, gd->dW_hidd, H); p[2].nseg = 1; p[2].seg[0] = td_seg(dq, H, H1 + (size_t)N * H, H, R, nullptr, 1, 1);
        p[3] = td_prob(E, H, gd->dW_td + E, ldtd); p[3].nseg = 1; p[3].seg[0] = td_seg(dx1, E, H2, H, R, nullptr, 1, 1);
        if ((rc = hg_launch(p, 4, sp))) return rc;
    }
    {
        d3_gemm_prob p[2];
        // dW_td[:, E+H:] = dx1s^T target  (K = N rows)
        p[0] = td_prob(E, F, gd->dW_td + E + H, ldtd); p[0].nseg = 1; p[0].seg[0] = td_seg(dx1s, E, a->target, F, N, nullptr, 1, 1);
        // map_feat: dW_feat (H, F) = dfp^T obj (K = N*K rows)
        p[1] = td_prob(H, F, gd->dW_feat, F); p[1].nseg = 1; p[1].seg[0] = td_seg(dfp, H, a->obj, F, N * K, nullptr, 1, 1);
        if ((rc = hg_launch(p, 2, sp))) return rc;
    }
    {
        // dW_td[:, :E] = dx1^T emb[words]: k-major B with gathered rows is not a GEMM operand form; the embedding rows of the
        // R tokens are gathered into x-space first (R x E floats, reusing the dlog buffer which is dead by now: its readers ran on
        // this same stream)
        float *embg = dlog;
        const long long tot = (long long)R * E;
        td_gather_rows_kernel<<<(int)((tot + 255) / 256), 256, 0, sp>>>(a->emb, widx, embg, R, E);
        d3_gemm_prob p = td_prob(E, E, gd->dW_td, ldtd);
        p.nseg = 1; p.seg[0] = td_seg(dx1, E, embg, E, R, nullptr, 1, 1);
        if ((rc = hg_launch(&p, 1, sp))) return rc;
    }
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ one decode step (inference)
// The same step for the greedy / evaluation decodes (model/caption_module.py:350-383, 689-770), where the next word depends
// on the classifier output and nothing can be batched over time: 8 launches (x1, GRU1, map_hidd, attention, map_lang, GRU2,
// classifier.0, classifier.2).  fp = map_feat(obj) is computed once per decode by d3_topdown_feat_proj.  obj_div: consecutive
// samples sharing one (K,F) object block (the evaluation decode runs the K targets of a scene as K samples).
extern "C" size_t d3_topdown_step_ws_bytes(int N, int K, int H, int E, int F) {
    return 2 * d3_align((size_t)N * 4) + 2 * d3_align((size_t)N * E * 4) + 2 * d3_align((size_t)N * H * 4) + 2 * d3_align((size_t)N * F * 4) +
           d3_align((size_t)N * K * 4) + 256;
}

extern "C" int d3_topdown_feat_proj(const float *obj, const float *W_feat, float *fp, int rows, int H, int F, void *stream) {
    D3_CLEAR();
    d3_gemm_prob p = td_prob(rows, H, fp, H);
    p.nseg = 1; p.seg[0] = td_seg(obj, F, W_feat, F, F);
    return hg_launch(&p, 1, d3_stream(stream));
}

__global__ void td_word_idx_kernel(const long long *__restrict__ word, int *__restrict__ widx, int N, int V) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) { const long long w = word[n]; widx[n] = (int)(w < 0 ? 0 : w >= V ? V - 1 : w); }
}

extern "C" int d3_topdown_step(const d3_topdown_args *a, const long long *word, const float *fp, int obj_div, const float *h1_in,
                               const float *h2_in, float *h1_out, float *h2_out, float *logits, float *attn, void *ws_, size_t ws_bytes,
                               void *stream) {
    D3_CLEAR();
    if (!a || a->N < 1 || (a->H & 15) || (a->E & 3) || (a->F & 3) || a->F > 128 || obj_div < 1) return D3_ERR_ARG;
    const int N = a->N, K = a->K, V = a->V, H = a->H, E = a->E, F = a->F;
    if (ws_bytes < d3_topdown_step_ws_bytes(N, K, H, E, F)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    D3Carver cv(ws_, ws_bytes);
    int *widx = cv.take<int>(N);
    float *x1 = cv.take<float>((size_t)N * E), *x2 = cv.take<float>((size_t)N * E), *q = cv.take<float>((size_t)N * H);
    float *c0 = cv.take<float>((size_t)N * H), *att = cv.take<float>((size_t)N * F), *msum = cv.take<float>((size_t)N * F);
    int *act = cv.take<int>((size_t)N * K), *nact = cv.take<int>(N);
    const long long ldtd = H + F + E, ldlang = F + H;
    int rc;
    if (256 % F) return D3_ERR_ARG;
    td_word_idx_kernel<<<(N + 255) / 256, 256, 0, s>>>(word, widx, N, V);
    td_attn_prep_kernel<<<N, TD_PREP_T, (size_t)(TD_PREP_T / F) * F * 4, s>>>(a->mask, a->obj, act, nact, msum, K, F, obj_div);
    {
        d3_gemm_prob p = td_prob(N, E, x1, E);
        p.nseg = 3;
        p.seg[0] = td_seg(a->emb, E, a->W_td, ldtd, E, widx);
        p.seg[1] = td_seg(h2_in, H, a->W_td + E, ldtd, H);
        p.seg[2] = td_seg(a->target, F, a->W_td + E + H, ldtd, F);
        p.bias = a->b_td;
        if ((rc = hg_launch(&p, 1, s))) return rc;
    }
    {
        GruArgs g{x1, E, E, h1_in, H, a->Wih1, a->Whh1, a->bih1, a->bhh1, h1_out, H, nullptr, nullptr, nullptr, nullptr, N, H, nullptr, 0, nullptr, 0, nullptr, 0};
        if ((rc = td_gru_fwd(g, s))) return rc;
    }
    {
        d3_gemm_prob p = td_prob(N, H, q, H);
        p.nseg = 1; p.seg[0] = td_seg(h1_out, H, a->W_hidd, H, H);
        if ((rc = hg_launch(&p, 1, s))) return rc;
    }
    td_attn_fwd_kernel<TD_ATT_T><<<N, TD_ATT_T, (size_t)(2 * H + 2 * K + 2 * F) * 4, s>>>(fp, q, H, a->w_att, a->obj, act, nact, msum, attn, att, F, nullptr, 0, 1, K, H,
                                                                         F, obj_div);
    {
        d3_gemm_prob p = td_prob(N, E, x2, E);
        p.nseg = 2;
        p.seg[0] = td_seg(att, F, a->W_lang, ldlang, F);
        p.seg[1] = td_seg(h1_out, H, a->W_lang + F, ldlang, H);
        p.bias = a->b_lang;
        if ((rc = hg_launch(&p, 1, s))) return rc;
    }
    {
        GruArgs g{x2, E, E, h2_in, H, a->Wih2, a->Whh2, a->bih2, a->bhh2, h2_out, H, nullptr, nullptr, nullptr, nullptr, N, H, nullptr, 0, nullptr, 0, nullptr, 0};
        if ((rc = td_gru_fwd(g, s))) return rc;
    }
    {
        d3_gemm_prob p = td_prob(N, H, c0, H);
        p.nseg = 1; p.seg[0] = td_seg(h2_out, H, a->Wc0, H, H);
        p.bias = a->bc0; p.relu = 1;
        if ((rc = hg_launch(&p, 1, s))) return rc;
        d3_gemm_prob p2 = td_prob(N, V, logits, V);
        p2.nseg = 1; p2.seg[0] = td_seg(c0, H, a->Wc2, H, H);
        p2.bias = a->bc2;
        if ((rc = hg_launch(&p2, 1, s))) return rc;
    }
    D3_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ packed-sequence GRU (LangModule)
// nn.GRU(I -> H, batch_first) over pack_padded_sequence(x, lens) (model/lang_module.py:51-55, 146-150): per sample the
// recurrence runs for lens[n] steps; `hiddens` (N,T,H) is zero beyond a sample's length (pad_packed_sequence), `last` (N,H)
// its final state.  The input-side gates of ALL steps are one GEMM (N*T x I x 3H); a step is then ONE launch: h W_hh^T
// and the gate math fused (td_gru_fwd_kernel, precomputed-input form).  Backward through time: gate kernel + one
// k-major GEMM per step; dW_ih, dW_hh and the bias gradients batched over all steps afterwards.
struct GsLayout { size_t GI, Hs, g, total; };
static GsLayout gs_layout(int N, int T, int H) {
    GsLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += d3_align(b); return at; };
    L.GI = take((size_t)N * T * 3 * H * 4); L.Hs = take((size_t)(T + 1) * N * H * 4); L.g = take((size_t)4 * T * N * H * 4);
    L.total = o;
    return L;
}
extern "C" size_t d3_gru_seq_ws_bytes(int N, int T, int I, int H) { (void)I; return gs_layout(N, T, H).total; }
extern "C" size_t d3_gru_seq_bwd_ws_bytes(int N, int T, int I, int H) {
    (void)I;
    return d3_align((size_t)N * T * 3 * H * 4) * 2 + d3_align((size_t)N * H * 4) + d3_align(hg_colsum_ws_bytes(2, 3 * H)) + 256;
}

extern "C" int d3_gru_seq_forward(const float *x, const int *lens, const float *Wih, const float *Whh, const float *bih, const float *bhh,
                                  int N, int T, int I, int H, float *hiddens, float *last, void *ws_, size_t ws_bytes, void *stream) {
    D3_CLEAR();
    if (N < 1 || T < 1 || (H & 15) || (I & 3)) return D3_ERR_ARG;
    const GsLayout L = gs_layout(N, T, H);
    if (ws_bytes < L.total) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    char *ws = (char *)ws_;
    float *GI = (float *)(ws + L.GI), *Hs = (float *)(ws + L.Hs), *g = (float *)(ws + L.g);
    int rc;
    {
        d3_gemm_prob p = td_prob(N * T, 3 * H, GI, 3 * H);
        p.nseg = 1; p.seg[0] = td_seg(x, I, Wih, I, I);
        p.bias = bih;
        if ((rc = hg_launch(&p, 1, s))) return rc;
    }
    D3_CHECK(hipMemsetAsync(Hs, 0, (size_t)N * H * 4, s));
    const size_t NH = (size_t)N * H, TNH = (size_t)T * NH;
    for (int t = 0; t < T; t++) {
        GruArgs a{nullptr, 0, I, Hs + t * NH, H, nullptr, Whh, nullptr, bhh, Hs + (t + 1) * NH, H,
                  g + t * NH, g + TNH + t * NH, g + 2 * TNH + t * NH, g + 3 * TNH + t * NH, N, H,
                  GI + (size_t)t * 3 * H, (long long)T * 3 * H, lens, t, hiddens + (size_t)t * H, (long long)T * H};
        if ((rc = td_gru_fwd(a, s))) return rc;
    }
    D3_CHECK(hipMemcpyAsync(last, Hs + (size_t)T * NH, NH * 4, hipMemcpyDeviceToDevice, s));
    D3_LAUNCH_CHECK();
    return 0;
}

// d_hiddens (N,T,H) and d_last (N,H) (either may be NULL) -> dWih, dWhh, dbih, dbhh (written); dx (N,T,I) when non-NULL
extern "C" int d3_gru_seq_backward(const float *x, const int *lens, const float *Wih, const float *Whh, int N, int T, int I, int H,
                                   const float *d_hiddens, const float *d_last, const void *ws_, float *dWih, float *dWhh, float *dbih,
                                   float *dbhh, float *dx, void *ws2_, size_t ws2_bytes, void *stream) {
    D3_CLEAR();
    if (N < 1 || T < 1 || (H & 15) || (I & 3)) return D3_ERR_ARG;
    if (ws2_bytes < d3_gru_seq_bwd_ws_bytes(N, T, I, H)) return D3_ERR_WORKSPACE;
    hipStream_t s = d3_stream(stream);
    const GsLayout L = gs_layout(N, T, H);
    const char *ws = (const char *)ws_;
    const float *Hs = (const float *)(ws + L.Hs), *g = (const float *)(ws + L.g);
    char *w2 = (char *)ws2_;
    const size_t gsz = d3_align((size_t)N * T * 3 * H * 4);
    float *DGI = (float *)w2, *DGH = (float *)(w2 + gsz), *carry = (float *)(w2 + 2 * gsz);
    char *cs = w2 + 2 * gsz + d3_align((size_t)N * H * 4);
    const size_t NH = (size_t)N * H, TNH = (size_t)T * NH;
    int rc;
    if (d_last) D3_CHECK(hipMemcpyAsync(carry, d_last, NH * 4, hipMemcpyDeviceToDevice, s));
    else D3_CHECK(hipMemsetAsync(carry, 0, NH * 4, s));
    const int nh = (int)((NH + 255) / 256);
    for (int t = T - 1; t >= 0; t--) {
        // DGI rows batch-major (n*T + t) like x; DGH rows time-major (t*N + n) like the saved states
        td_gru_bwd_gates_kernel<<<nh, 256, 0, s>>>(carry, H, d_hiddens ? d_hiddens + (size_t)t * H : nullptr, (long long)T * H, nullptr, 0,
                                                   g + t * NH, g + TNH + t * NH, g + 2 * TNH + t * NH, g + 3 * TNH + t * NH, Hs + t * NH, H,
                                                   DGI + (size_t)t * 3 * H, (long long)T * 3 * H, DGH + (size_t)t * N * 3 * H, carry, N, H, lens, t);
        d3_gemm_prob p = td_prob(N, H, carry, H);
        p.nseg = 1; p.seg[0] = td_seg(DGH + (size_t)t * N * 3 * H, 3 * H, Whh, H, 3 * H, nullptr, 0, 1); p.accum = 1;
        if ((rc = hg_launch(&p, 1, s))) return rc;
    }
    {
        d3_gemm_prob p[2];
        p[0] = td_prob(3 * H, I, dWih, I); p[0].nseg = 1; p[0].seg[0] = td_seg(DGI, 3 * H, x, I, N * T, nullptr, 1, 1);
        p[1] = td_prob(3 * H, H, dWhh, H); p[1].nseg = 1; p[1].seg[0] = td_seg(DGH, 3 * H, Hs, H, N * T, nullptr, 1, 1);
        if ((rc = hg_launch(&p[0], 1, s))) return rc;
        if ((rc = hg_launch(&p[1], 1, s))) return rc;
        const float *cx[2] = {DGI, DGH}; const long long cl[2] = {3 * H, 3 * H}; const int cr[2] = {N * T, N * T}, cc[2] = {3 * H, 3 * H};
        float *co[2] = {dbih, dbhh};
        if ((rc = hg_colsum_multi(cx, cl, cr, cc, co, nullptr, 2, cs, hg_colsum_ws_bytes(2, 3 * H), s))) return rc;
        if (dx) {
            d3_gemm_prob q = td_prob(N * T, I, dx, I);
            q.nseg = 1; q.seg[0] = td_seg(DGI, 3 * H, Wih, I, 3 * H, nullptr, 0, 1);
            if ((rc = hg_launch(&q, 1, s))) return rc;
        }
    }
    D3_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------ decode-loop selection kernels (round 4)
// The sampling loops of the self-critical step (beam search + greedy baseline, model/caption_module.py:136-383) ran ~25 library
// launches per time step around the native decode step -- log_softmax, add, topk (sbtopk::gatherTopK: 45 us), div / mod, three
// gathers, cat, comparisons, two index_selects of the hidden states: ~7,000 element-wise launches per joint step.  One launch
// per step instead.
//
// d3_beam_select: one workgroup per sample.  logits (N*b, V): row n*b + j = live beam j of sample n (live = 1 at t = 0).
//   logp[j][v] = (x - max_j) - log(sum_v exp(x - max_j))             (torch's log_softmax expression)
//   cand[j*V + v] = sums_in[n][j] + logp[j][v]; the b best candidates best first (ties: the lower flat index) give
//   beam_ix = flat / V, tok = flat % V, chosen = logp, snap = sums_in[beam_ix] + chosen, ended = tok == eos (or `last`),
//   sums_out = snap - 1000 * ended (caption_module.py:300), seq_out[n][r][:t] = seq_prev[n][beam_ix][:t], seq_out[n][r][t] = tok,
//   and the two hidden states of row n*b + r are those of row n*b + beam_ix (the re-ordering of :305-307).
#define BS_T 256
__device__ __forceinline__ void td_beam_select_body(const int n, const int rs, const float *__restrict__ logits, const float *__restrict__ sums_in, int live, int b, int V,
                                                             int eos, int last, int t, int Tmax, const long long *__restrict__ seq_prev,
                                                             long long *__restrict__ seq_out, long long *__restrict__ tok_out,
                                                             float *__restrict__ snap_out, unsigned char *__restrict__ ended_out,
                                                             float *__restrict__ sums_out, const float *__restrict__ h1_in,
                                                             const float *__restrict__ h2_in, float *__restrict__ h1_out,
                                                             float *__restrict__ h2_out, int H) {
    __shared__ float s_max[8], s_lse[8], s_sum[8];
    __shared__ int s_pick[8], s_nan[8];
    __shared__ float s_pickv[8];
    __shared__ float w_v[BS_T / 64];
    __shared__ int w_i[BS_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (a) per live beam: max and log-sum-exp -- one wave per beam, shuffles only (round 4: block-wide trees of 8 barriers each before,
    // ~135 barriers per launch, 30 us on a 31-step dependent chain)
    for (int j = wv; j < live; j += BS_T / 64) {
        const float *x = logits + ((long long)n * rs + j) * V;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sm = 0.f;
        for (int v = lane; v < V; v += 64) sm += expf(x[v] - m);
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        if (lane == 0) { s_max[j] = m; s_lse[j] = logf(sm); s_sum[j] = sums_in[(long long)n * live + j]; }
    }
    __syncthreads();
    // (b) candidate scores staged in LDS once, then b block-wide arg-max passes over the LDS copy (a picked candidate is struck out)
    extern __shared__ float cs[];                     // live * V floats
    for (int j = 0; j < live; j++) {
        const float *x = logits + ((long long)n * rs + j) * V;
        const float mj = s_max[j], lj = s_lse[j], sj = s_sum[j];
        for (int v = tid; v < V; v += BS_T) cs[j * V + v] = sj + ((x[v] - mj) - lj);
    }
    __syncthreads();
    const int total = live * V;
    for (int r = 0; r < b; r++) {
        float bv = -INFINITY; int bi = 0x7FFFFFFF;
        for (int flat = tid; flat < total; flat += BS_T) {
            const float c = cs[flat];
            if (c > bv) { bv = c; bi = flat; }            // (ascending flat per thread: the first maximum wins)
        }
        for (int o = 32; o > 0; o >>= 1) {                // larger value, then lower flat index
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { w_v[wv] = bv; w_i[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = w_v[0]; int fi = w_i[0];
            for (int w = 1; w < BS_T / 64; w++) if (w_v[w] > fv || (w_v[w] == fv && w_i[w] < fi)) { fv = w_v[w]; fi = w_i[w]; }
            // NaN / all -inf scores: no comparison above is ever true and the sentinel index survives.  torch.topk returns NaN
            // scores there (a non-finite-loss guard can skip the step); an unguarded 0x7FFFFFFF would index far out of bounds.
            // Take the lowest candidate not picked yet (its score is NaN / -inf and propagates into snap / sums_out).
            if (fi == 0x7FFFFFFF) {
                fi = 0;
                for (int q = 0; q < r; q++) if (s_pick[q] == fi) { fi++; q = -1; }
                if (fi >= total) fi = 0;
                s_nan[r] = 1;
            } else s_nan[r] = 0;
            s_pick[r] = fi; s_pickv[r] = fv; cs[fi] = -INFINITY;
        }
        __syncthreads();
    }
    // (c) outputs
    if (tid < b) {
        const int flat = s_pick[tid], j = flat / V, v = flat - j * V;
        float chosen = (logits[((long long)n * rs + j) * V + v] - s_max[j]) - s_lse[j];
        if (s_nan[tid]) chosen = __builtin_nanf("");     // (degenerate scores: the pick is arbitrary, its score says so)
        const float snap = s_sum[j] + chosen;
        const bool ended = last || v == eos;
        const long long o = (long long)n * b + tid;
        tok_out[(long long)n * rs + tid] = v; snap_out[o] = snap; ended_out[o] = ended ? 1 : 0; sums_out[o] = snap - 1000.0f * (ended ? 1.f : 0.f);
        seq_out[o * Tmax + t] = v;
    }
    for (int e = tid; e < b * t; e += BS_T) {          // histories of the chosen beams
        const int r = e / t, c = e - r * t;
        const int j = s_pick[r] / V;
        seq_out[((long long)n * b + r) * Tmax + c] = seq_prev[((long long)n * b + j) * Tmax + c];
    }
    if (h1_in) {
        for (int e = tid; e < b * H; e += BS_T) {
            const int r = e / H, c = e - r * H;
            const int j = s_pick[r] / V;
            h1_out[((long long)n * rs + r) * H + c] = h1_in[((long long)n * rs + j) * H + c];
            h2_out[((long long)n * rs + r) * H + c] = h2_in[((long long)n * rs + j) * H + c];
        }
    }
}
__global__ __launch_bounds__(BS_T) void td_beam_select_kernel(const float *__restrict__ logits, const float *__restrict__ sums_in, int live, int b, int V,
                                                             int eos, int last, int t, int Tmax, const long long *__restrict__ seq_prev,
                                                             long long *__restrict__ seq_out, long long *__restrict__ tok_out,
                                                             float *__restrict__ snap_out, unsigned char *__restrict__ ended_out,
                                                             float *__restrict__ sums_out, const float *__restrict__ h1_in,
                                                             const float *__restrict__ h2_in, float *__restrict__ h1_out,
                                                             float *__restrict__ h2_out, int H) {
    td_beam_select_body(blockIdx.x, b, logits, sums_in, live, b, V, eos, last, t, Tmax, seq_prev, seq_out, tok_out, snap_out, ended_out, sums_out, h1_in,
                        h2_in, h1_out, h2_out, H);
}
extern "C" int d3_beam_select(const float *logits, const float *sums_in, int N, int live, int b, int V, int eos, int last, int t, int Tmax,
                              const long long *seq_prev, long long *seq_out, long long *tok_out, float *snap_out, unsigned char *ended_out,
                              float *sums_out, const float *h1_in, const float *h2_in, float *h1_out, float *h2_out, int H, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (live < 1 || live > b || b < 1 || b > 8 || V < 1 || t < 0 || t >= Tmax || (t > 0 && !seq_prev)) return D3_ERR_ARG;
    const size_t lds = (size_t)live * V * sizeof(float);
    if (lds > 60 * 1024) {        // (b = 8 beams over a 3004-word vocabulary: 94 KB)
        static bool attr_done[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !attr_done[dev]) {
            D3_CHECK(hipFuncSetAttribute((const void *)td_beam_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
        if (lds > 150 * 1024) return D3_ERR_ARG;
    }
    td_beam_select_kernel<<<N, BS_T, lds, d3_stream(stream)>>>(logits, sums_in, live, b, V, eos, last, t, Tmax, seq_prev, seq_out, tok_out, snap_out, ended_out,
                                                             sums_out, h1_in, h2_in, h1_out, h2_out, H);
    D3_LAUNCH_CHECK();
    return 0;
}
// greedy step: word = argmax_v logits[n][v] (first maximum), lp = its log-softmax value (caption_module.py:367-371)
__device__ __forceinline__ void td_greedy_select_body(const float *__restrict__ x, int V, long long *__restrict__ word, float *__restrict__ lp,
                                                      long long *__restrict__ word2) {
    __shared__ float w_v[BS_T / 64];
    __shared__ int w_i[BS_T / 64];
    __shared__ float w_s[BS_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float m = -INFINITY; int mi = 0x7FFFFFFF;
    for (int v = tid; v < V; v += BS_T) if (x[v] > m) { m = x[v]; mi = v; }
    for (int o = 32; o > 0; o >>= 1) {                    // larger value, then lower index (the first maximum)
        const float ov = __shfl_xor(m, o); const int oi = __shfl_xor(mi, o);
        if (ov > m || (ov == m && oi < mi)) { m = ov; mi = oi; }
    }
    if (lane == 0) { w_v[wv] = m; w_i[wv] = mi; }
    __syncthreads();
    m = w_v[0]; mi = w_i[0];
    for (int w = 1; w < BS_T / 64; w++) if (w_v[w] > m || (w_v[w] == m && w_i[w] < mi)) { m = w_v[w]; mi = w_i[w]; }
    // NaN / all -inf logits: no comparison is ever true; torch.max returns NaN there.  Index 0 with a NaN log-probability
    // instead of an out-of-bounds read through the sentinel.
    const bool degenerate = mi == 0x7FFFFFFF;
    if (degenerate) mi = 0;
    float sm = 0.f;
    for (int v = tid; v < V; v += BS_T) sm += expf(x[v] - m);
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    if (lane == 0) w_s[wv] = sm;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < BS_T / 64; w++) tot += w_s[w];
        *word = mi; *lp = degenerate ? __builtin_nanf("") : (x[mi] - m) - logf(tot); if (word2) *word2 = mi;
    }
}
// rows n * rs + off of logits (rs = 1, off = 0: dense); word2 (optional): the token also goes to row n * rs + off of the next step's input
__global__ __launch_bounds__(BS_T) void td_greedy_select_kernel(const float *__restrict__ logits, int V, long long *__restrict__ word, float *__restrict__ lp,
                                                               int rs, int off, long long *__restrict__ word2) {
    const long long row = (long long)blockIdx.x * rs + off;
    td_greedy_select_body(logits + row * V, V, word + blockIdx.x, lp + blockIdx.x, word2 ? word2 + row : nullptr);
}
// one selection launch for the joined decode (d3_topdown_beam_greedy): workgroups [0, N) select the beams of sample n (rows n * rs ..
// n * rs + b - 1), workgroups [N, 2N) the greedy row n * rs + b, whose hidden states are copied through (its row is not re-ordered)
__global__ __launch_bounds__(BS_T) void td_beam_greedy_select_kernel(const float *__restrict__ logits, const float *__restrict__ sums_in, int N, int live, int b,
                                                                    int V, int eos, int last, int t, int Tmax, const long long *__restrict__ seq_prev,
                                                                    long long *__restrict__ seq_out, long long *__restrict__ tok_out,
                                                                    float *__restrict__ snap_out, unsigned char *__restrict__ ended_out,
                                                                    float *__restrict__ sums_out, const float *__restrict__ h1_in,
                                                                    const float *__restrict__ h2_in, float *__restrict__ h1_out,
                                                                    float *__restrict__ h2_out, int H, long long *__restrict__ g_word,
                                                                    float *__restrict__ g_lp) {
    const int rs = b + 1;
    if ((int)blockIdx.x < N) {
        td_beam_select_body(blockIdx.x, rs, logits, sums_in, live, b, V, eos, last, t, Tmax, seq_prev, seq_out, tok_out, snap_out, ended_out, sums_out, h1_in,
                            h2_in, h1_out, h2_out, H);
        return;
    }
    const int n = blockIdx.x - N;
    const long long row = (long long)n * rs + b;
    td_greedy_select_body(logits + row * V, V, g_word + n, g_lp + n, tok_out + row);
    for (int c = threadIdx.x; c < H; c += BS_T) { h1_out[row * H + c] = h1_in[row * H + c]; h2_out[row * H + c] = h2_in[row * H + c]; }
}
extern "C" int d3_greedy_select(const float *logits, int N, int V, long long *word, float *lp, void *stream) {
    D3_CLEAR();
    if (N <= 0) return 0;
    if (V < 1) return D3_ERR_ARG;
    td_greedy_select_kernel<<<N, BS_T, 0, d3_stream(stream)>>>(logits, V, word, lp, 1, 0, nullptr);
    D3_LAUNCH_CHECK();
    return 0;
}

// ---- whole decodes in one call (round 4).  The loops around d3_topdown_step used to live on the host side of the boundary: two
// library calls, two tensor allocations and their argument marshalling per time step, ~60 us of interpreter time per step against
// ~30 us of launches -- and the self-critical step (model/caption_module.py:588-633) walks 61 such steps.  Same launches in the same
// order, issued from here.
// greedy (:350-383): h1_a / h2_a hold the initial (zero) states, words / lps are (max_len, N), first_word (N) is the sos row.
extern "C" int d3_topdown_greedy(const d3_topdown_args *a, const float *fp, int obj_div, float *h1_a, float *h2_a, float *h1_b, float *h2_b,
                                 float *logits, float *attn, void *ws, size_t ws_bytes, const long long *first_word, int max_len,
                                 long long *words, float *lps, void *stream) {
    if (!a || max_len < 1 || !first_word || !words || !lps) return D3_ERR_ARG;
    const long long *w = first_word;
    float *i1 = h1_a, *i2 = h2_a, *o1 = h1_b, *o2 = h2_b;
    for (int t = 0; t < max_len; t++) {
        int rc = d3_topdown_step(a, w, fp, obj_div, i1, i2, o1, o2, logits, attn, ws, ws_bytes, stream);
        if (rc) return rc;
        rc = d3_greedy_select(logits, a->N, a->V, words + (size_t)t * a->N, lps + (size_t)t * a->N, stream);
        if (rc) return rc;
        w = words + (size_t)t * a->N;
        float *x = i1; i1 = o1; o1 = x;
        x = i2; i2 = o2; o2 = x;
    }
    return 0;
}
// The two decodes of one self-critical step as ONE chain (model/caption_module.py:588-633: beam search = "sampled", greedy = baseline,
// same samples, same parameters): a->N = samples * (b + 1) rows, row n * (b + 1) + j = beam j of sample n for j < b and the greedy row
// for j = b; all rows of a sample share its object block (obj_div = b + 1).  One decode step + one selection launch per time step
// instead of two of each: the recurrence is a chain of dependent ~7 us launches, so the step costs its length, not its rows.
// Every row's arithmetic is that of the separate decodes (a row of the step never reads another row).  Beam outputs as
// d3_topdown_beam; g_words / g_lps (glen, samples), glen >= max_len (the reference decodes max_spk_len + 1 greedy steps).
static int td_beam_select_lds(size_t lds, const void *kernel) {
    if (lds > 150 * 1024) return D3_ERR_ARG;
    if (lds > 60 * 1024) D3_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
    return 0;
}
extern "C" int d3_topdown_beam_greedy(const d3_topdown_args *a, const float *fp, int b, float *const *h1, float *const *h2, float *logits,
                                      float *attn, void *ws, size_t ws_bytes, const long long *first_word, int eos, int max_len,
                                      long long *allseq, float *snap_all, unsigned char *ended_all, float *sums0, float *sums1,
                                      long long *tok, int glen, long long *g_words, float *g_lps, void *stream) {
    if (!a || b < 1 || b > 8 || a->N % (b + 1) || max_len < 1 || glen < max_len || !h1 || !h2 || !g_words || !g_lps) return D3_ERR_ARG;
    const int rs = b + 1, N = a->N / rs, H = a->H, V = a->V;
    const size_t per = (size_t)N * b;
    hipStream_t s = d3_stream(stream);
    int rc = td_beam_select_lds((size_t)b * V * sizeof(float), (const void *)td_beam_greedy_select_kernel);
    if (rc) return rc;
    float *P1 = h1[1], *Q1 = h1[2], *R1 = h1[0], *P2 = h2[1], *Q2 = h2[2], *R2 = h2[0];
    rc = d3_topdown_step(a, first_word, fp, rs, R1, R2, P1, P2, logits, attn, ws, ws_bytes, stream);
    if (rc) return rc;
    float *s_in = sums0, *s_out = sums1;
    int live = 1;
    for (int t = 0; t < max_len; t++) {
        const int last = t == max_len - 1;
        td_beam_greedy_select_kernel<<<2 * N, BS_T, (size_t)live * V * sizeof(float), s>>>(
            logits, s_in, N, live, b, V, eos, last, t, max_len, t > 0 ? allseq + (size_t)(t - 1) * per * max_len : nullptr,
            allseq + (size_t)t * per * max_len, tok, snap_all + (size_t)t * per, ended_all + (size_t)t * per, s_out, P1, P2, Q1, Q2, H,
            g_words + (size_t)t * N, g_lps + (size_t)t * N);
        D3_LAUNCH_CHECK();
        float *x = s_in; s_in = s_out; s_out = x;
        live = b;
        if (t + 1 >= glen) break;
        rc = d3_topdown_step(a, tok, fp, rs, Q1, Q2, R1, R2, logits, attn, ws, ws_bytes, stream);
        if (rc) return rc;
        x = P1; P1 = R1; R1 = Q1; Q1 = x;
        x = P2; P2 = R2; R2 = Q2; Q2 = x;
    }
    for (int t = max_len; t < glen; t++) {          // the greedy rows' remaining steps (the beam rows ride along, unread)
        td_greedy_select_kernel<<<N, BS_T, 0, s>>>(logits, V, g_words + (size_t)t * N, g_lps + (size_t)t * N, rs, b, tok);
        D3_LAUNCH_CHECK();
        if (t + 1 >= glen) break;
        rc = d3_topdown_step(a, tok, fp, rs, P1, P2, R1, R2, logits, attn, ws, ws_bytes, stream);
        if (rc) return rc;
        float *x = P1; P1 = R1; R1 = x;
        x = P2; P2 = R2; R2 = x;
    }
    return 0;
}
// beam search (:136-349) over a->N = samples * b rows (row n * b + j = beam j of sample n; the b rows of a sample share its object
// block: obj_div = b).  Three buffers per hidden state rotate: latest step output -> (select: re-ordered) -> next step's output;
// h1[0] / h2[0] hold the initial (zero) states.  allseq (max_len, samples, b, max_len) zero-filled by the caller, snap_all /
// ended_all (max_len, samples, b), sums0 (samples, b) zero-filled, sums1 / tok scratch: every step's beams are kept, the caller
// ranks the finished ones (d3net_amd/speaker.py).
extern "C" int d3_topdown_beam(const d3_topdown_args *a, const float *fp, int b, float *const *h1, float *const *h2, float *logits,
                               float *attn, void *ws, size_t ws_bytes, const long long *first_word, int eos, int max_len,
                               long long *allseq, float *snap_all, unsigned char *ended_all, float *sums0, float *sums1,
                               long long *tok, void *stream) {
    if (!a || b < 1 || a->N % b || max_len < 1 || !h1 || !h2) return D3_ERR_ARG;
    const int N = a->N / b, H = a->H;
    const size_t per = (size_t)N * b;
    float *P1 = h1[1], *Q1 = h1[2], *R1 = h1[0], *P2 = h2[1], *Q2 = h2[2], *R2 = h2[0];
    int rc = d3_topdown_step(a, first_word, fp, b, R1, R2, P1, P2, logits, attn, ws, ws_bytes, stream);
    if (rc) return rc;
    float *s_in = sums0, *s_out = sums1;
    int live = 1;                                   // t = 0: a single live beam per sample (:176-179)
    for (int t = 0; t < max_len; t++) {
        const int last = t == max_len - 1;
        rc = d3_beam_select(logits, s_in, N, live, b, a->V, eos, last, t, max_len, t > 0 ? allseq + (size_t)(t - 1) * per * max_len : nullptr,
                            allseq + (size_t)t * per * max_len, tok, snap_all + (size_t)t * per, ended_all + (size_t)t * per, s_out, P1, P2, Q1, Q2,
                            H, stream);
        if (rc) return rc;
        float *x = s_in; s_in = s_out; s_out = x;
        live = b;
        if (last) break;
        rc = d3_topdown_step(a, tok, fp, b, Q1, Q2, R1, R2, logits, attn, ws, ws_bytes, stream);
        if (rc) return rc;
        x = P1; P1 = R1; R1 = Q1; Q1 = x;
        x = P2; P2 = R2; R2 = Q2; Q2 = x;
    }
    return 0;
}
