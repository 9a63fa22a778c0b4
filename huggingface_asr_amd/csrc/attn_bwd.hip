// Softmax stage of the attention backward pass on gfx950 (the matrix products around it are mi_bgemm_bf16).
//
// Reference forward being differentiated: e_branchformer.py:100-135 + tf wav2vec2_conformer:528-565 (relative shift):
//   s[i,j] = ((q_i+u)·k_j + (q_i+v)·p[T-1-i+j]) / sqrt(hd) + mask,  P = softmax_j(s),  ctx = P V
// and the GPT-2 decoder's causal self-attention / length-masked cross-attention (tf gpt2 eager attention).
// Head-major layouts: AC, dP, dS, P are (H, B, Tq, Tk); BD, dBD are (H, B, Tq, 2T-1) — for one head the (b, t) rows are
// contiguous, so the batch-reduced products of the position branch are plain strided GEMMs.
#include "common.hpp"

namespace {

struct SmArgs {
    const float* ac; const float* bd;     // bd may be null
    const float* dp;                      // backward only
    bf16_t* prob;                         // fwd: out, bwd: in
    bf16_t* ds; bf16_t* dbd;              // backward outputs (dbd may be null)
    const int* lengths;                   // valid keys per utterance or null
    int H, B, Tq, Tk, causal;
    long ld_s, ld_p;                      // row strides (elements) of the (.., Tk) and (.., 2T-1) matrices
    float scale;
    float drop_p; unsigned long long drop_key;   // attention-probability dropout (dropout.hip's counter-based mask); 0 = off
    bf16_t* prob_drop;                    // fwd: dropped probabilities for the PV product (null when drop_p == 0)
};

// logical index of P[h][b][i][j] = ((h*B + b)*Tq + i)*Tk + j  (independent of the padded row stride)
__device__ __forceinline__ float sm_keep(const SmArgs& p, long row, int j) {
    const unsigned long long idx = (unsigned long long)row * (unsigned long long)p.Tk + (unsigned long long)j;
    return mask_u01_at(p.drop_key, idx) >= p.drop_p ? 1.f / (1.f - p.drop_p) : 0.f;
}

__device__ __forceinline__ bool key_masked(const SmArgs& p, int b, int i, int j) {
    if (p.lengths && j >= p.lengths[b]) return true;
    if (p.causal && j > i + (p.Tk - p.Tq)) return true;
    return false;
}

// one wave per (h, b, i) row
__global__ __launch_bounds__(256) void softmax_fwd_kernel(SmArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.H * p.B * p.Tq;
    if (row >= rows) return;
    const int i = (int)(row % p.Tq), b = (int)((row / p.Tq) % p.B);
    const float* ac = p.ac + row * p.ld_s;
    const float* bd = p.bd ? p.bd + row * p.ld_p + (p.Tq - 1 - i) : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < p.Tk; j += 64) {
        if (key_masked(p, b, i, j)) continue;
        mx = fmaxf(mx, (ac[j] + (bd ? bd[j] : 0.f)) * p.scale);
    }
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < p.Tk; j += 64) {
        if (key_masked(p, b, i, j)) continue;
        s += __expf((ac[j] + (bd ? bd[j] : 0.f)) * p.scale - mx);
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    bf16_t* out = p.prob + row * p.ld_s;
    for (int j = lane; j < p.Tk; j += 64) {
        float v = 0.f;
        if (!key_masked(p, b, i, j)) v = __expf((ac[j] + (bd ? bd[j] : 0.f)) * p.scale - mx) * inv;
        out[j] = f2bf(v);
        if (p.prob_drop) p.prob_drop[row * p.ld_s + j] = f2bf(v * sm_keep(p, row, j));
    }
}

// dS = P * (dP - sum_j P dP) * scale;  dBD[i, T-1-i+j] = dS[i,j], zero elsewhere
__global__ __launch_bounds__(256) void softmax_bwd_kernel(SmArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.H * p.B * p.Tq;
    if (row >= rows) return;
    const int i = (int)(row % p.Tq);
    const bf16_t* pr = p.prob + row * p.ld_s;
    const float* dp = p.dp + row * p.ld_s;
    // with dropout, dp is the gradient w.r.t. the DROPPED probabilities: dP = dp * keep / (1 - p) (mask regenerated)
    const bool dr = p.drop_p > 0.f;
    float dot = 0.f;
    for (int j = lane; j < p.Tk; j += 64) dot += bf2f(pr[j]) * dp[j] * (dr ? sm_keep(p, row, j) : 1.f);
    dot = wave_sum(dot);
    bf16_t* ds = p.ds + row * p.ld_s;
    for (int j = lane; j < p.Tk; j += 64) ds[j] = f2bf(bf2f(pr[j]) * (dp[j] * (dr ? sm_keep(p, row, j) : 1.f) - dot) * p.scale);
    if (p.dbd) {
        const int P = 2 * p.Tq - 1, off = p.Tq - 1 - i;
        bf16_t* dbd = p.dbd + row * p.ld_p;
        for (int q = lane; q < P; q += 64) {
            const int j = q - off;
            dbd[q] = (j >= 0 && j < p.Tk) ? f2bf(bf2f(pr[j]) * (dp[j] * (dr ? sm_keep(p, row, j) : 1.f) - dot) * p.scale) : (bf16_t)0.f;
        }
    }
}

// Register forms for Tk <= 64 NR (the encoder's 250 frames: NR = 4): a row lives in the wave's registers, so scores are read once, exponentiated
// once and a dropout keep factor is hashed once per element (the loops above read three times, exponentiate twice and — backward, with dropout —
// hash three times).  Same arithmetic in the same order: results are bit-identical to the loop forms.
template <int NR>
__global__ __launch_bounds__(256) void softmax_fwd_reg_kernel(SmArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.H * p.B * p.Tq;
    if (row >= rows) return;
    const int i = (int)(row % p.Tq), b = (int)((row / p.Tq) % p.B);
    const float* ac = p.ac + row * p.ld_s;
    const float* bd = p.bd ? p.bd + row * p.ld_p + (p.Tq - 1 - i) : nullptr;
    float v[NR];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        const bool ok = j < p.Tk && !key_masked(p, b, i, j);
        v[r] = ok ? (ac[j] + (bd ? bd[j] : 0.f)) * p.scale : -INFINITY;
        mx = fmaxf(mx, v[r]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        v[r] = v[r] > -INFINITY ? __expf(v[r] - mx) : 0.f;
        s += v[r];
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    bf16_t* out = p.prob + row * p.ld_s;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j >= p.Tk) continue;
        const float pv = v[r] * inv;
        out[j] = f2bf(pv);
        if (p.prob_drop) p.prob_drop[row * p.ld_s + j] = f2bf(pv * sm_keep(p, row, j));
    }
}
template <int NR>
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(SmArgs p) {
    __shared__ bf16_t srow[4][64 * NR];                   // a wave's dS row, re-read at the relative-position shift for dBD
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    const long rows = (long)p.H * p.B * p.Tq;
    if (row >= rows) return;
    const int i = (int)(row % p.Tq);
    const bf16_t* pr = p.prob + row * p.ld_s;
    const float* dp = p.dp + row * p.ld_s;
    const bool dr = p.drop_p > 0.f;
    float pv[NR], gv[NR];
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        pv[r] = 0.f; gv[r] = 0.f;
        if (j < p.Tk) {
            pv[r] = bf2f(pr[j]);
            gv[r] = dp[j] * (dr ? sm_keep(p, row, j) : 1.f);
            dot += pv[r] * gv[r];
        }
    }
    dot = wave_sum(dot);
    bf16_t* ds = p.ds + row * p.ld_s;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        const bf16_t o = f2bf(pv[r] * (gv[r] - dot) * p.scale);
        srow[wave][j] = o;
        if (j < p.Tk) ds[j] = o;
    }
    if (p.dbd) {
        __builtin_amdgcn_wave_barrier();                  // LDS accesses of one wave complete in order: the row above is visible to the reads below
        const int P = 2 * p.Tq - 1, off = p.Tq - 1 - i;
        bf16_t* dbd = p.dbd + row * p.ld_p;
        for (int q = lane; q < P; q += 64) {
            const int j = q - off;
            dbd[q] = (j >= 0 && j < p.Tk) ? srow[wave][j] : (bf16_t)0.f;
        }
    }
}

}  // namespace

// ld_s: row stride of ac / prob / dp / ds (>= Tk); ld_p: row stride of bd / dbd (>= 2 Tq - 1).  Padding columns are never read.
// prob_drop / drop_p / seed / stream_id: attention-probability dropout (prob stays un-dropped for the backward pass); pass NULL / 0 for none
extern "C" int mi_attn_softmax_fwd(const float* ac, const float* bd, const int* lengths, void* prob, void* prob_drop, int H, int B, int Tq, int Tk,
                                   long ld_s, long ld_p, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t st) {
    MI_ENTER();
    if (H <= 0 || B <= 0 || Tq <= 0 || Tk <= 0 || (bd && Tq != Tk) || ld_s < Tk || (bd && ld_p < 2 * Tq - 1)) return MI_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !prob_drop)) return MI_ERR_ARG;
    SmArgs p{ac, bd, nullptr, (bf16_t*)prob, nullptr, nullptr, lengths, H, B, Tq, Tk, causal, ld_s, ld_p, scale, drop_p,
             ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed, drop_p > 0.f ? (bf16_t*)prob_drop : nullptr};
    if (Tk <= 256) hipLaunchKernelGGL(softmax_fwd_reg_kernel<4>, dim3(cdiv((long)H * B * Tq, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv((long)H * B * Tq, 4)), dim3(256), 0, st, p);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_attn_softmax_bwd(const void* prob, const float* dp, void* ds, void* dbd, int H, int B, int Tq, int Tk, long ld_s, long ld_p,
                                   float scale, float drop_p, unsigned seed, unsigned stream_id, hipStream_t st) {
    MI_ENTER();
    if (H <= 0 || B <= 0 || Tq <= 0 || Tk <= 0 || (dbd && Tq != Tk) || ld_s < Tk || (dbd && ld_p < 2 * Tq - 1)) return MI_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return MI_ERR_ARG;
    SmArgs p{nullptr, nullptr, dp, (bf16_t*)prob, (bf16_t*)ds, (bf16_t*)dbd, nullptr, H, B, Tq, Tk, 0, ld_s, ld_p, scale, drop_p,
             ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed, nullptr};
    if (Tk <= 256) hipLaunchKernelGGL(softmax_bwd_reg_kernel<4>, dim3(cdiv((long)H * B * Tq, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv((long)H * B * Tq, 4)), dim3(256), 0, st, p);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
