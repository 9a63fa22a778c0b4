// One beam-search step of joint CTC / attention decoding, entirely on the device (BASELINE config 5: streaming bs = 1 latency).
//
// Replaces, per emitted token, what the reference runs on the host through transformers' GenerationMixin beam loop
// (reference call site: src/models/ctc_encoder_plus_autoregressive_decoder.py:360-482, processors src/decoding/ctc_scorer.py:259-365; the loop is third-party code:
// transformers/generation/utils.py `_beam_search` of the installed 5.x — the rules the fixtures tests/golden/gen_*.npz pin, restated on the CPU in oracle/generate_ref.py):
//     scores = log_softmax(decoder logits);  [CTC processor: scores[:, pad] = logzero;  scores = (1 - w) scores + w ctc];  cand = scores + running beam score
//     top 2W of the W * V candidates of an utterance, best first;  a candidate STOPS when its token is EOS or its length reaches max_length
//     the first W candidates that did not stop run on as the next beams (at max_length: stopped ones, lowered by 1e9 — the loop ends there)
//     stopped candidates among the first W ranks join the kept hypotheses with score / (generated tokens) ** length_penalty; the best W are kept, best first
//     early-stop rule: the utterance is finished when W hypotheses are kept and best running score / (hypothetical length) ** length_penalty cannot beat the worst of them
//     (early_stopping False / True: the current length; "never" with a positive penalty: max_length), or — early_stopping True — as soon as W hypotheses are kept
//     input ids follow their beams and get the new token appended
// The host loop did this with two device -> host copies, a Python walk and three host -> device copies per token (~0.3 ms of a 0.8 ms token); here it is one
// launch, nothing leaves the device until decoding ends, and the host only enqueues.  Arithmetic is the host loop's, operation for operation (fp32 subtract,
// two multiplies and an add without contraction, fp32 beam add, fp32 division for the kept hypotheses), so both produce the same hypotheses bit for bit.
#include "common.hpp"
#include "../../include/hfasr_hip.h"

namespace {

// BEAM_STOP = 1 / 2 / 3 (tools/beam_step_phases.sh only; never in the product build): the kernel returns after its candidate pass / its 2W selection rounds / the one-thread
// walk — where its time goes (results are garbage in every mode but 0)
#ifndef BEAM_STOP
#define BEAM_STOP 0
#endif
constexpr float LOGZERO = -10000000000.0f;
constexpr int BS_THREADS = 1024, BS_WAVES = BS_THREADS / 64, BS_MAXW = 16;

struct BeamArgs {
    const float* logits; long ldl;        // (B * W, V) decoder logits of the last position
    const float* lse;                     // (B * W) their row log-sum-exp (mi_row_lse)
    const float* ctc;                     // (B * W, V) CTC prefix scores or null
    float w_att, w_ctc;
    int mask_pad;                         // the CTC processor masks the pad token (ctc_scorer.py:325); without it nothing does
    int pad, eos, B, W, V, cur_len, max_length, Lmax;
    float denom, heur_denom;              // (generated tokens of a hypothesis closed in this step) ** length_penalty; (hypothetical length of the early-stop rule) ** length_penalty
    int early_stopping;                   // 0 False, 1 True, 2 "never"
    long* ids;                            // (B * W, Lmax), updated in place
    float* beam_scores;                   // (B * W), updated in place
    long* new_tok;                        // (B * W)
    long* beam_idx;                       // (B * W)
    int* done; int* nfin; float* fin_score; int* fin_len; long* fin_tok;      // (B), (B), (B, W), (B, W), (B, W, Lmax): the kept hypotheses, best first
    float* top_s; int* top_i;             // optional (B, 2W): the candidates the step walked
    int* done_out;                        // optional (B): copy of the done flags after the step (host-mapped pinned memory: the host polls it without a copy on the stream)
};

// the host loop's arithmetic, one rounding per operation: no multiply-add contraction here (HIP's __fmul_rn / __fadd_rn are plain operators, so the pragma is what holds it)
__device__ __forceinline__ float cand_value(const BeamArgs& p, int b, int e) {
#pragma clang fp contract(off)
    const int beam = e / p.V, tok = e - beam * p.V;
    const long row = (long)b * p.W + beam;
    float s = p.logits[row * p.ldl + tok] - p.lse[row];
    if (p.mask_pad && tok == p.pad) s = LOGZERO;
    if (p.ctc) {
        const float a = p.w_att * s, c = p.w_ctc * p.ctc[row * p.V + tok];
        s = a + c;
    }
    return s + p.beam_scores[row];
}

// candidates are ordered by (value descending, index ascending); `after` = the last one taken
__device__ __forceinline__ bool comes_after(float v, int e, float pv, int pe) { return v < pv || (v == pv && e > pe); }
__device__ __forceinline__ bool better(float v, int e, float bv, int be) { return v > bv || (v == bv && e < be); }

// logits / CTC scores of the candidates tid, tid + 1024, ... of the first NG groups of eight (clamped addresses past the last candidate; the rest zero)
template <int NG>
__device__ __forceinline__ void load_pass(const BeamArgs& p, int b, int W, int tid, int qstep, int rstep, float (&lg)[32], float (&ct)[32]) {
    int beam = tid / p.V, tok = tid - beam * p.V;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < NG * 8) {
            const int bc = beam < W ? beam : W - 1;
            const long row = (long)b * W + bc;
            lg[i] = p.logits[row * p.ldl + tok];
            ct[i] = p.ctc ? p.ctc[row * p.V + tok] : 0.f;
            beam += qstep; tok += rstep;
            if (tok >= p.V) { tok -= p.V; ++beam; }
        } else { lg[i] = 0.f; ct[i] = 0.f; }
    }
}

__global__ __launch_bounds__(BS_THREADS) void beam_step_kernel(BeamArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long* stage = reinterpret_cast<long*>(smem);                       // [W][cur_len] the utterance's input ids before the step
    __shared__ float wv[2 * BS_WAVES];
    __shared__ int we[2 * BS_WAVES];
    __shared__ float tops[2 * BS_MAXW];
    __shared__ int topi[2 * BS_MAXW];
    __shared__ float nbs[BS_MAXW];
    __shared__ long nbt[BS_MAXW];
    __shared__ int nbb[BS_MAXW];                                       // source beam (within the utterance) of every next beam
    __shared__ int fsrc[BS_MAXW], nf_new;                              // new rank -> old rank (>= 0) or ~(candidate rank) of every kept hypothesis
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = p.W, N = W * p.V, R = 2 * W;
    const bool was_done = p.done[b] != 0;
    // the kept hypotheses' scores / lengths / count: requested here by W threads, read by the one-thread walk below from LDS (it used to fetch them itself, one dependent
    // global round trip after another at the end of the kernel)
    __shared__ float pf_fs[BS_MAXW];
    __shared__ int pf_fl[BS_MAXW], pf_nf;
    if (tid < W) { pf_fs[tid] = p.fin_score[(long)b * W + tid]; pf_fl[tid] = p.fin_len[(long)b * W + tid]; }
    if (tid == 0) pf_nf = p.nfin[b];

    // ---- top 2W: every thread owns the candidates tid, tid + 1024, ... and offers its best one not yet taken; the owner of a round's winner re-scans
    // (a thread keeps its candidates' values in registers when they fit — W * V <= 32 Ki: the owner of a round's winner then re-scans 25 registers instead of
    // re-reading 25 x 3 global values; at W = 5, V = 5001 the ten rounds of re-reads were most of this kernel's 139 us per token)
    constexpr int CPT = 32;
    const bool cached = N <= CPT * BS_THREADS;
    float cv[CPT];
    float mv = -INFINITY; int me = 0x7fffffff;
    float mv2 = -INFINITY; int me2 = 0x7fffffff;                       // cached form: the thread's second best (valid while have2), so that a taken best needs no rescan
    bool have2 = true;
    if (!was_done) {
        if (cached) {
            // Round 5: every load of the pass is requested before the first is used.  Written as `e < N ? cand_value(...) : -inf` each candidate's four loads sat under
            // a condition — a basic block of their own, a full wait at every join: 25 dependent round trips to logits another kernel had just written (27 of the
            // kernel's 40 us at W = 5).  Now the indices are clamped, (beam, token) advance by (1024 / V, 1024 % V) without a division per candidate, the per-row
            // scalars come from LDS, and the arithmetic — cand_value's, operation for operation — runs on registers.
            __shared__ float row_lse[BS_MAXW], row_bs[BS_MAXW];
            if (tid < W) { row_lse[tid] = p.lse[(long)b * W + tid]; row_bs[tid] = p.beam_scores[(long)b * W + tid]; }
            const int qstep = BS_THREADS / p.V, rstep = BS_THREADS - qstep * p.V;
            float lg[CPT], ct[CPT];
            const int ng = (N + 8 * BS_THREADS - 1) / (8 * BS_THREADS);      // groups of eight candidates per thread that hold any at all (block-uniform)
            if (ng >= 4) load_pass<4>(p, b, W, tid, qstep, rstep, lg, ct);     // one straight-line block per count: every load of the pass is in flight together
            else if (ng == 3) load_pass<3>(p, b, W, tid, qstep, rstep, lg, ct);
            else if (ng == 2) load_pass<2>(p, b, W, tid, qstep, rstep, lg, ct);
            else load_pass<1>(p, b, W, tid, qstep, rstep, lg, ct);
            __syncthreads();
            {
#pragma clang fp contract(off)
                int beam = tid / p.V, tok = tid - beam * p.V;
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int e = tid + i * BS_THREADS;
                    float sc = lg[i] - row_lse[beam < W ? beam : 0];
                    if (p.mask_pad && tok == p.pad) sc = LOGZERO;
                    if (p.ctc) {
                        const float a = p.w_att * sc, c = p.w_ctc * ct[i];
                        sc = a + c;
                    }
                    sc = sc + row_bs[beam < W ? beam : 0];
                    cv[i] = e < N ? sc : -INFINITY;
                    if (cv[i] > mv) { mv2 = mv; me2 = me; mv = cv[i]; me = e; }           // ascending e: the first of equal values stays (= better())
                    else if (cv[i] > mv2) { mv2 = cv[i]; me2 = e; }
                    beam += qstep; tok += rstep;
                    if (tok >= p.V) { tok -= p.V; ++beam; }
                }
            }
        } else
            for (int e = tid; e < N; e += BS_THREADS) {
                const float v = cand_value(p, b, e);
                if (better(v, e, mv, me)) { mv = v; me = e; }
            }
    }
    if (BEAM_STOP == 1) { if (mv == 12345.f) p.new_tok[0] = me; return; }
    unsigned alive = 0xffffffffu;                                      // cached form: bit i = candidate tid + 1024 i not yet taken
    for (int r = 0; r < R && !was_done; ++r) {
        const float wm = wave_max(mv);
        const float ecand = (mv == wm && me != 0x7fffffff) ? -(float)me : -INFINITY;      // indices < 2^24: exact in fp32
        const float em = wave_max(ecand);
        float* wvr = wv + (r & 1) * BS_WAVES;                          // winners of odd and even rounds in separate slots: ONE barrier per round (a wave cannot run two rounds ahead)
        int* wer = we + (r & 1) * BS_WAVES;
        if (lane == 0) { wvr[wave] = wm; wer[wave] = em == -INFINITY ? 0x7fffffff : (int)(-em); }
        __syncthreads();
        // the 16 wave winners -> the block's: one DPP reduction per wave (lane k holds wave k's), not a 16-entry scan per thread
        const float xv = lane < BS_WAVES ? wvr[lane] : -INFINITY;
        const int xe = lane < BS_WAVES ? wer[lane] : 0x7fffffff;
        const float bv = wave_max(xv);
        const float bem = wave_max((xv == bv && xe != 0x7fffffff) ? -(float)xe : -INFINITY);
        const int be = bem == -INFINITY ? 0x7fffffff : (int)(-bem);
        if (tid == 0) { tops[r] = bv; topi[r] = be; }
        if (be != 0x7fffffff && (be % BS_THREADS) == tid) {            // my candidate was taken: next best of my subset
            if (cached) {
                alive &= ~(1u << (be / BS_THREADS));
                if (have2) { mv = mv2; me = me2; have2 = false; }      // the second best of the pass (no rescan: a thread is rarely taken from twice)
                else {
                    mv = -INFINITY; me = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < CPT; ++i) {
                        const float v = ((alive >> i) & 1u) ? cv[i] : -INFINITY;
                        if (v > mv) { mv = v; me = tid + i * BS_THREADS; }
                    }
                }
            } else {
                mv = -INFINITY; me = 0x7fffffff;
                for (int e = tid; e < N; e += BS_THREADS) {
                    const float v = cand_value(p, b, e);
                    if (comes_after(v, e, bv, be) && better(v, e, mv, me)) { mv = v; me = e; }
                }
            }
        }
    }
    __syncthreads();

    if (BEAM_STOP == 2) { if (tid == 0 && tops[0] == 12345.f) p.new_tok[0] = topi[0]; return; }
    // ---- the utterance's ids before the step, and its kept hypotheses
    long* fstage = stage + W * p.cur_len;                              // [W][Lmax]
    for (int i = tid; i < W * p.cur_len; i += BS_THREADS) {
        const int k = i / p.cur_len, j = i - k * p.cur_len;
        stage[i] = p.ids[((long)b * W + k) * p.Lmax + j];
    }
    if (!was_done)
        for (int i = tid; i < W * p.Lmax; i += BS_THREADS) fstage[i] = p.fin_tok[(long)b * W * p.Lmax + i];

    // ---- walk the candidates (one thread: at most 2W steps)
    if (tid == 0) {
        for (int k = 0; k < W; ++k) { nbs[k] = 0.f; nbt[k] = was_done ? p.pad : 0; nbb[k] = 0; fsrc[k] = k; }
        nf_new = pf_nf;
        if (!was_done) {
            const bool at_max = p.cur_len + 1 >= p.max_length;
            bool hit[2 * BS_MAXW];
            int nr = 0;
            for (int r = 0; r < R; ++r) {
                if (topi[r] == 0x7fffffff) break;
                const int tok = topi[r] % p.V;
                hit[r] = tok == p.eos || at_max;
                ++nr;
            }
            // the W best candidates that did not stop run on; when fewer are left (max_length) the stopped ones follow, lowered by 1e9
            int k = 0;
            for (int r = 0; r < nr && k < W; ++r)
                if (!hit[r]) { nbs[k] = tops[r]; nbt[k] = topi[r] % p.V; nbb[k] = topi[r] / p.V; ++k; }
            for (int r = 0; r < nr && k < W; ++r)
                if (hit[r]) { nbs[k] = tops[r] + -1.0e9f; nbt[k] = topi[r] % p.V; nbb[k] = topi[r] / p.V; ++k; }
            // stopped candidates among the first W ranks compete with the kept hypotheses: best W, best first, an equal score behind the older one
            int nf = nf_new;
            float fs[BS_MAXW];
            int fl[BS_MAXW];
            for (int i = 0; i < nf; ++i) { fs[i] = pf_fs[i]; fl[i] = pf_fl[i]; }
            for (int r = 0; r < nr && r < W; ++r) {
                if (!hit[r]) continue;
                const float sc = tops[r] / p.denom;
                int pos = nf;
                while (pos > 0 && sc > fs[pos - 1]) --pos;
                if (pos >= W) continue;
                const int last = nf < W ? nf : W - 1;
                for (int i = last; i > pos; --i) { fs[i] = fs[i - 1]; fl[i] = fl[i - 1]; fsrc[i] = fsrc[i - 1]; }
                fs[pos] = sc; fl[pos] = p.cur_len + 1; fsrc[pos] = ~r;
                if (nf < W) ++nf;
            }
            for (int i = 0; i < nf; ++i) { p.fin_score[(long)b * W + i] = fs[i]; p.fin_len[(long)b * W + i] = fl[i]; }
            nf_new = nf;
            p.nfin[b] = nf;
            // early-stop rule on the state after the step
            const float best = nbs[0] / p.heur_denom;
            const bool unsat = best > (nf == W ? fs[W - 1] : -1.0e9f);
            if (!unsat || (p.early_stopping == 1 && nf == W) || at_max) p.done[b] = 1;
        }
        if (p.done_out) p.done_out[b] = p.done[b];
        if (p.top_s)
            for (int r = 0; r < R; ++r) { p.top_s[(long)b * R + r] = was_done ? 0.f : tops[r]; p.top_i[(long)b * R + r] = was_done ? 0 : topi[r]; }
    }
    __syncthreads();
    if (BEAM_STOP == 3) return;

    // ---- the kept hypotheses move to their new ranks (old rows from the LDS copy, new ones = their beam's ids + the closing token)
    if (!was_done)
        for (int i = 0; i < nf_new; ++i) {
            const int src = fsrc[i];
            if (src == i) continue;
            long* dst = p.fin_tok + ((long)b * W + i) * p.Lmax;
            if (src >= 0) {
                for (int j = tid; j < p.Lmax; j += BS_THREADS) dst[j] = fstage[src * p.Lmax + j];
            } else {
                const int r = ~src, beam = topi[r] / p.V;
                for (int j = tid; j < p.Lmax; j += BS_THREADS) dst[j] = j < p.cur_len ? stage[beam * p.cur_len + j] : (j == p.cur_len ? (long)(topi[r] % p.V) : (long)p.pad);
            }
        }
    for (int i = tid; i < W * p.cur_len; i += BS_THREADS) {
        const int k = i / p.cur_len, j = i - k * p.cur_len;
        p.ids[((long)b * W + k) * p.Lmax + j] = stage[nbb[k] * p.cur_len + j];
    }
    if (tid < W) {
        const long row = (long)b * W + tid;
        p.ids[row * p.Lmax + p.cur_len] = nbt[tid];
        p.new_tok[row] = nbt[tid];
        p.beam_idx[row] = (long)b * W + nbb[tid];
        p.beam_scores[row] = nbs[tid];
    }
}

}  // namespace

extern "C" int mi_beam_step(const float* logits, long ldl, const float* lse, const float* ctc, float w_att, float w_ctc, int mask_pad, int pad, int eos, int B, int W,
                            int V, int cur_len, int max_length, int Lmax, float denom, float heur_denom, int early_stopping, long* ids, float* beam_scores, long* new_tok,
                            long* beam_idx, int* done, int* nfin, float* fin_score, int* fin_len, long* fin_tok, float* top_s, int* top_i, int* done_out, hipStream_t stream) {
    MI_ENTER();
    if (!logits || !lse || !ids || !beam_scores || !new_tok || !beam_idx || !done || !nfin || !fin_score || !fin_len || !fin_tok) return MI_ERR_ARG;
    if (B <= 0 || W <= 0 || W > BS_MAXW || V <= 1 || (long)W * V >= (1l << 24) || cur_len <= 0 || cur_len >= Lmax || cur_len >= max_length || max_length > Lmax || pad < 0 ||
        pad >= V || !(denom > 0.f) || !(heur_denom > 0.f) || early_stopping < 0 || early_stopping > 2)
        return MI_ERR_ARG;
    const size_t lds = (size_t)W * (cur_len + Lmax) * sizeof(long);
    if (lds > 96 * 1024) return MI_ERR_UNSUPPORTED;
    BeamArgs a{logits, ldl, lse, ctc, w_att, w_ctc, mask_pad, pad, eos, B, W, V, cur_len, max_length, Lmax, denom, heur_denom, early_stopping, ids, beam_scores, new_tok, beam_idx,
               done, nfin, fin_score, fin_len, fin_tok, top_s, top_i, done_out};
    if (!ensure_dynamic_lds<0>(reinterpret_cast<const void*>(beam_step_kernel), 96 * 1024)) return MI_ERR_LAUNCH;        // per device (a function-local static configured only the first one)
    hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(BS_THREADS), lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
