// CTC prefix scorer for joint CTC/attention decoding on gfx950.
//
// Replaces the per-token Python loop of the reference's `CTCPrefixScoreTH.__call__` (src/decoding/ctc_scorer.py:58-178:
// `for t in range(start, end)` over ALL encoder frames on (T, 2, B*W, V) tensors) and `index_select_state` (:180-207).
// Every (hypothesis i, token c) pair is an independent chain over time
//     r^n_t = lse(r^n_{t-1}, phi_{t-1}) + x_t[c] ,   r^b_t = lse(r^n_{t-1}, r^b_{t-1}) + x_t[blank] ,
//     log psi = lse_t(phi_{t-1} + x_t[c])  (+ r^n_{start-1}),      phi_t = (c == last_i) ? r'^b_t : lse(r'^n_t, r'^b_t)
// so one thread runs one chain with its forward variables in registers: the reference's (T,2,B*W,V) state tensor
// (50 MB per emitted token at T'=250, W=5, V=5001) is never materialised.  The only state kept between tokens is the
// previous prefix's (T,2,B*W) forward variables; the chains of the tokens that beam search actually selected are re-run
// (B*W chains) to advance it — including the reference's quirk that every beam inherits beam 0's variables.
// HBM traffic per token: the log-posteriors x (B,T,V) read once, coalesced over c (HBM/L2 bound, fp32 math).
#include "common.hpp"

namespace {

constexpr float LOGZERO = -10000000000.0f;

// log(e^a + e^b).  Hardware exp2 / log2 (v_exp_f32 / v_log_f32, ~1 ulp) instead of libm's expf / log1pf: the 250-step scan is one
// dependent chain of these per frame on a wave that is alone on its SIMD, so the instruction count of the chain IS the latency.
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m + __logf(1.f + __expf(-fabsf(a - b)));
}

// x = log_softmax(logits) with the padding rule of ctc_scorer.py:39-42 (frames >= len: logzero, blank = 0)
template <typename T>
__global__ __launch_bounds__(256) void prefix_prepare_kernel(const T* __restrict__ logits, long ld_b, long ld_t, const float* __restrict__ lse,
                                                              const int* __restrict__ lens, int Tn, int O, int blank, float* __restrict__ x) {
    const int row = blockIdx.x;                 // b*T + t
    const int b = row / Tn, t = row - b * Tn;
    const bool pad = t >= lens[b];
    const T* src = logits + (long)b * ld_b + (long)t * ld_t;
    const float l = lse[row];
    for (int c = threadIdx.x; c < O; c += 256)
        x[(long)row * O + c] = pad ? (c == blank ? 0.f : LOGZERO) : (float)src[c] - l;
}

// initial forward variables (:74-82): r^n = logzero, r^b_t = cumsum_t x_t[blank] (sequential fp32, as torch.cumsum)
__global__ void prefix_init_kernel(const float* __restrict__ x, int B, int Tn, int O, int blank, int W, float* __restrict__ r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // hypothesis
    if (i >= B * W) return;
    const int b = i / W;
    float acc = 0.f;
    for (int t = 0; t < Tn; ++t) {
        acc += x[((long)b * Tn + t) * O + blank];
        r[((long)t * 2 + 0) * B * W + i] = LOGZERO;
        r[((long)t * 2 + 1) * B * W + i] = acc;
    }
}

struct ChainArgs {
    const float* x; int B, T, O, blank, W;
    const float* r_prev;            // (T, 2, n_bh)
    const long* last_ids; long ld_last;   // last token of every hypothesis' prefix (input_ids[:, -1])
    int out_len;                    // len(prefix) - 1
    // chain selection: null -> all (i, c); else K chains (hyp[k], tok[k])
    const int* hyp; const long* tok; long ld_tok; int K;
    float* r_out;                   // (T, 2, K) or null
    float* psi_out;                 // (K) / (n_bh, O) or null
    const float* s_prev;            // (n_bh) or null (= 0)
    float* scores_out;              // (n_bh, O) or null: psi - s_prev with blank / exact-zero -> logzero
    int beam0;                      // chain selection without an index tensor: K chains, chain k = (hypothesis (k / W) W — beam 0 of k's utterance, the reference's quirk —, tok[k])
    const float* psi_prev;          // s_prev gathered here instead: psi_prev[((i / W) W) O + last_i]   ((n_bh, O) of the previous call)
    int rp_full;                    // r_prev is the previous call's (T, 2, n_bh, O) chains of EVERY (hypothesis, token): hypothesis i continues from [.., (i / W) W, last_i]
};

__global__ __launch_bounds__(256) void prefix_chain_kernel(ChainArgs p) {
    const int n_bh = p.B * p.W;
    long k = (long)blockIdx.x * 256 + threadIdx.x;
    int i, c;
    if (p.hyp || p.beam0) {
        if (k >= p.K) return;
        i = p.hyp ? p.hyp[k] : (int)(k / p.W) * p.W;
        c = (int)p.tok[k * p.ld_tok];
    } else {
        const int cpb = (p.O + 255) / 256;                    // blocks per hypothesis
        i = blockIdx.x / cpb;
        c = (blockIdx.x % cpb) * 256 + threadIdx.x;
        if (i >= n_bh || c >= p.O) return;
        k = (long)i * p.O + c;
    }
    const int b = i / p.W;
    const float* xb = p.x + (long)b * p.T * p.O;
    const bool same = (long)c == p.last_ids[(long)i * p.ld_last];
    const int start = p.out_len > 1 ? p.out_len : 1;
    const long rp_plane = p.rp_full ? (long)n_bh * p.O : n_bh;                 // distance between the (t, n / b) planes of r_prev, and this chain's column in them
    const long rp_col = p.rp_full ? (long)(i / p.W) * p.W * p.O + p.last_ids[(long)i * p.ld_last] : i;
    const bool picked = p.hyp || p.beam0;
    const long Kt = picked ? p.K : (long)n_bh * p.O;
    float r0 = LOGZERO, r1 = LOGZERO;
    if (p.out_len == 0) r0 = xb[c];
    if (p.r_out) {
        for (int t = 0; t < start && t < p.T; ++t) {
            p.r_out[((long)t * 2 + 0) * Kt + k] = (t == 0 && p.out_len == 0) ? r0 : LOGZERO;
            p.r_out[((long)t * 2 + 1) * Kt + k] = LOGZERO;
        }
    }
    // r[start-1] : only r[0] can be non-logzero (out_len == 0)
    if (start - 1 != 0) r0 = LOGZERO;
    float pm = r0, ps = 1.f;                     // online logsumexp of psi, seeded with r^n_{start-1}
    // the loads of a step do not depend on the recursion: fetch UNROLL steps ahead of the dependent lse chain (a lone wave per SIMD has
    // nobody to hide the ~0.5 us load latency behind; per-step loads made the scan latency-bound at ~1 us per frame)
    constexpr int UNROLL = 8;
    for (int t0 = start; t0 < p.T; t0 += UNROLL) {
        float rp0[UNROLL], rp1[UNROLL], xc[UNROLL], xbl[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int t = t0 + u;
            const bool ok = t < p.T;
            rp0[u] = ok ? p.r_prev[((long)(t - 1) * 2 + 0) * rp_plane + rp_col] : 0.f;
            rp1[u] = ok ? p.r_prev[((long)(t - 1) * 2 + 1) * rp_plane + rp_col] : 0.f;
            xc[u] = ok ? xb[(long)t * p.O + c] : 0.f;
            xbl[u] = ok ? xb[(long)t * p.O + p.blank] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int t = t0 + u;
            if (t >= p.T) break;
            const float phi = same ? rp1[u] : lse2(rp0[u], rp1[u]);
            const float n0 = lse2(r0, phi) + xc[u];
            const float n1 = lse2(r0, r1) + xbl[u];
            r0 = n0; r1 = n1;
            if (p.r_out) {
                p.r_out[((long)t * 2 + 0) * Kt + k] = r0;
                p.r_out[((long)t * 2 + 1) * Kt + k] = r1;
            }
            const float term = phi + xc[u];
            if (term > pm) { ps = ps * __expf(pm - term) + 1.f; pm = term; }
            else ps += __expf(term - pm);
        }
    }
    float psi = pm + logf(ps);
    if (p.scores_out || !picked) { if (c == p.blank) psi = LOGZERO; }           // :173
    if (p.psi_out) p.psi_out[k] = psi;
    if (p.scores_out) {
        const float sp = p.psi_prev ? p.psi_prev[(long)(i / p.W) * p.W * p.O + p.last_ids[(long)i * p.ld_last]] : (p.s_prev ? p.s_prev[i] : 0.f);
        float s = psi - sp;
        if (s == 0.f) s = LOGZERO;                                              // :176
        p.scores_out[k] = s;
    }
}

}  // namespace

// x_out (B,T,O) fp32 = padded log-posteriors; r0_out (T,2,B*W) initial forward variables. lse: scratch (B*T) fp32.
extern "C" int mi_row_lse(const void* x, long ld, int dtype, int V, float* lse, int M, hipStream_t stream);
extern "C" int mi_ctc_prefix_prepare(const void* logits, long ld_b, long ld_t, int dtype, const int* lens, int B, int T, int O,
                                     int blank, int W, float* lse_scratch, float* x_out, float* r0_out, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || O <= 0 || W <= 0 || blank < 0 || blank >= O) return MI_ERR_ARG;
    // row LSE needs contiguous (B*T) rows with stride ld_t; per-batch stride handled by looping batches when ld_b != T*ld_t
    for (int b = 0; b < B; ++b) {
        const char* base = (const char*)logits + (size_t)b * ld_b * (dtype == 0 ? 4 : 2);
        int rc = mi_row_lse(base, ld_t, dtype, O, lse_scratch + (size_t)b * T, T, stream);
        if (rc != MI_OK) return rc;
    }
    if (dtype == 0)
        hipLaunchKernelGGL(prefix_prepare_kernel<float>, dim3(B * T), dim3(256), 0, stream, (const float*)logits, ld_b, ld_t, lse_scratch, lens, T, O, blank, x_out);
    else
        hipLaunchKernelGGL(prefix_prepare_kernel<bf16_t>, dim3(B * T), dim3(256), 0, stream, (const bf16_t*)logits, ld_b, ld_t, lse_scratch, lens, T, O, blank, x_out);
    hipLaunchKernelGGL(prefix_init_kernel, dim3(cdiv(B * W, 64)), dim3(64), 0, stream, x_out, B, T, O, blank, W, r0_out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// score all (hyp, token) pairs: scores_out (n_bh, O), psi_out (n_bh, O)
extern "C" int mi_ctc_prefix_score(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, const long* last_ids,
                                   long ld_last, int out_len, const float* s_prev, float* psi_out, float* scores_out, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || O <= 0 || W <= 0) return MI_ERR_ARG;
    ChainArgs a{x, B, T, O, blank, W, r_prev, last_ids, ld_last, out_len, nullptr, nullptr, 0, 0, nullptr, psi_out, s_prev, scores_out, 0, nullptr, 0};
    hipLaunchKernelGGL(prefix_chain_kernel, dim3(B * W * cdiv(O, 256)), dim3(256), 0, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// advance the state along the selected chains (hyp[k], tok[k]), k < K: r_out (T,2,K)
extern "C" int mi_ctc_prefix_select(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, const long* last_ids,
                                    long ld_last, int out_len, const int* hyp, const long* tok, long ld_tok, int K, float* r_out,
                                    hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || O <= 0 || W <= 0 || K <= 0) return MI_ERR_ARG;
    ChainArgs a{x, B, T, O, blank, W, r_prev, last_ids, ld_last, out_len, hyp, tok, ld_tok, K, r_out, nullptr, nullptr, nullptr, 0, nullptr, 0};
    hipLaunchKernelGGL(prefix_chain_kernel, dim3(cdiv(K, 256)), dim3(256), 0, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// One decoding step of the processor after the first (= mi_ctc_prefix_select along (beam 0 of every utterance, the token each hypothesis ended on) followed by
// mi_ctc_prefix_score with s_prev gathered from the previous call's psi), without the index tensors in between: two launches from one call.
//   previous call: r_old (T,2,n_bh), last_old (its prefixes' last tokens), out_len_old, psi_old (n_bh, O);  this call: last, out_len
//   out: r_prev (T,2,n_bh) = the state this call's chains start from (kept for the next call), psi (n_bh, O), scores (n_bh, O)
extern "C" int mi_ctc_prefix_advance(const float* x, int B, int T, int O, int blank, int W, const float* r_old, const long* last_old, long ld_last_old, int out_len_old,
                                     const float* psi_old, const long* last, long ld_last, int out_len, float* r_prev, float* psi, float* scores, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || O <= 0 || W <= 0 || !r_old || !last_old || !psi_old || !last || !r_prev || !psi || !scores) return MI_ERR_ARG;
    const int n_bh = B * W;
    ChainArgs sel{x, B, T, O, blank, W, r_old, last_old, ld_last_old, out_len_old, nullptr, last, ld_last, n_bh, r_prev, nullptr, nullptr, nullptr, 1, nullptr, 0};
    hipLaunchKernelGGL(prefix_chain_kernel, dim3(cdiv(n_bh, 256)), dim3(256), 0, stream, sel);
    ChainArgs sc{x, B, T, O, blank, W, r_prev, last, ld_last, out_len, nullptr, nullptr, 0, 0, nullptr, psi, nullptr, scores, 0, psi_old, 0};
    hipLaunchKernelGGL(prefix_chain_kernel, dim3(n_bh * cdiv(O, 256)), dim3(256), 0, stream, sc);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// The same step in ONE launch when the chains of every (hypothesis, token) may be kept (T x 2 x n_bh x O floats: 50 MB at T' = 250, W = 5, V = 5001 — what the reference
// materialises per token, ctc_scorer.py:58-178; here it is written once, coalesced over the tokens, by the scan that computes it anyway): the state a hypothesis continues from
// is then a column of the previous call's `r_all`, and the re-run of the selected chains (mi_ctc_prefix_select: a second dependent 250-frame scan per token) disappears.
//   r_prev: rp_full = 0 -> (T, 2, n_bh) of mi_ctc_prefix_prepare (first token, psi_old null);  rp_full = 1 -> the previous call's r_all, psi_old its psi.
extern "C" int mi_ctc_prefix_score_full(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, int rp_full, const float* psi_old, const long* last,
                                        long ld_last, int out_len, float* r_all, float* psi, float* scores, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || O <= 0 || W <= 0 || !r_prev || !last || !r_all || !psi || !scores || (rp_full && !psi_old) || (rp_full != 0 && rp_full != 1)) return MI_ERR_ARG;
    ChainArgs a{x, B, T, O, blank, W, r_prev, last, ld_last, out_len, nullptr, nullptr, 0, 0, r_all, psi, nullptr, scores, 0, rp_full ? psi_old : nullptr, rp_full};
    hipLaunchKernelGGL(prefix_chain_kernel, dim3(B * W * cdiv(O, 256)), dim3(256), 0, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
