// Whole GPT-2 decoder token step as ONE C call (BASELINE config 5): embeddings -> L x (ln_1, fused QKV, KV-cache
// append, causal attention over the cache, c_proj + residual, ln_cross_attn, q projection, cross-attention over the cached encoder
// K/V, c_proj + residual, ln_2, gelu_new MLP + residual) -> ln_f on the last new position -> lm_head logits.
//
// Replaces, per emitted token, what the reference runs through transformers' GPT2Model.forward with a KV cache
// (multi_head_gpt2.py:80-170, tf gpt2 modeling :262-310) — about a hundred small launches; driving them from C++ instead of one
// ctypes call each removes the host time between launches, which dominates bs=1 latency.  Also: one kernel that re-orders every
// layer's KV cache after a beam-search step (transformers' `_reorder_cache`).
#include "common.hpp"
#include "../../include/hfasr_hip.h"

namespace {

// K / V rows of the new tokens -> cache[(b, past + u), :]
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* __restrict__ qkv, long ldq, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                         int B, int U, int past, int Lmax, int d) {
    const int d8 = d >> 3;
    const long total = (long)B * U * d8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % d8);
        const long m = i / d8;
        const int b = (int)(m / U), u = (int)(m % U);
        const long dst = ((long)b * Lmax + past + u) * d + c * 8;
        *reinterpret_cast<bf16x8*>(kc + dst) = *reinterpret_cast<const bf16x8*>(qkv + m * ldq + d + c * 8);
        *reinterpret_cast<bf16x8*>(vc + dst) = *reinterpret_cast<const bf16x8*>(qkv + m * ldq + 2 * d + c * 8);
    }
}

// ---- skinny fused linear for the token step (M <= 8 rows, e.g. 1 utterance x 5 beams): y = act(LN?(x) · W^T + b) (+ resid)
// A 128x128 MFMA tile is 96 % padding at M = 5 and every separate LayerNorm / residual launch costs more than its work, so this
// kernel does it GEMV-style: the block normalises the M rows into LDS (bf16-rounded, like the LN kernel's output), then every wave
// streams 4 weight rows (16-B loads, K split over the lanes) and reduces the M dot products with wave shuffles.
constexpr int SK_MAXM = 8, SK_COLS = 4;

struct SkArgs {
    const float* x32; long ldx;            // fp32 input (residual stream) when ln_g != null or x16 == null
    const bf16_t* x16; long ldx16;         // bf16 input (attention context / MLP hidden)
    const float* ln_g; const float* ln_b; float eps;
    const bf16_t* W; long ldw; const float* bias;
    float* out32; long ldo32;              // fp32 output: out32 = resid32 + acc (in-place residual add) or plain logits
    const float* resid32;
    bf16_t* out16; long ldo16;             // bf16 output
    int M, N, K, act;                      // act: 0 none, 2 gelu_new
    bf16_t* kc; bf16_t* vc; int U, past, Lmax, dkv;      // optional (kc != null): columns [dkv, 2 dkv) / [2 dkv, 3 dkv) are ALSO appended to the K / V caches at row past + u
};

template <int NCH>      // 64-wide chunks of a LayerNorm row a lane holds: 8 for K = 512 (exact), 32 for any K <= 2048
__global__ __launch_bounds__(256) void skinny_linear_kernel(SkArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                     // [M][K] bf16 (the operand precision of every linear on this path)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * SK_COLS;
    // the weight vectors of this wave's columns do not depend on the input: request them first (SK_COLS x K/512 independent 16-B loads in
    // flight under the LayerNorm prologue), consume them after the barrier
    constexpr int MAXP = 4;                                               // K <= 2048
    bf16x8 wv[SK_COLS][MAXP];
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c)
#pragma unroll
        for (int ps = 0; ps < MAXP; ++ps) {
            const int n = n0 + c, k = ps * 512 + lane * 8;
            wv[c][ps] = (n < p.N && k < p.K) ? *reinterpret_cast<const bf16x8*>(p.W + (long)n * p.ldw + k) : z8;
        }
    for (int m = wave; m < p.M; m += 4) {
        if (p.x16) {
            for (int k = lane; k < p.K; k += 64) xs[m * p.K + k] = p.x16[(long)m * p.ldx16 + k];
        } else if (p.ln_g) {
            const float* xr = p.x32 + (long)m * p.ldx;
            float xv[NCH], gv[NCH], bv[NCH];                              // row + affine in registers, requested together: ONE global round trip
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int k = lane + 64 * i;
                const bool ok = k < p.K;
                xv[i] = ok ? xr[k] : 0.f; gv[i] = ok ? p.ln_g[k] : 0.f; bv[i] = ok ? p.ln_b[k] : 0.f;
                s += xv[i];
            }
            const float mean = wave_sum(s) / p.K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) { const int k = lane + 64 * i; const float a = k < p.K ? xv[i] - mean : 0.f; q += a * a; }
            const float rstd = rsqrtf(wave_sum(q) / p.K + p.eps);
#pragma unroll
            for (int i = 0; i < NCH; ++i) { const int k = lane + 64 * i; if (k < p.K) xs[m * p.K + k] = f2bf((xv[i] - mean) * rstd * gv[i] + bv[i]); }
        } else {
            for (int k = lane; k < p.K; k += 64) xs[m * p.K + k] = f2bf(p.x32[(long)m * p.ldx + k]);
        }
    }
    __syncthreads();
    // bf16 x bf16 products accumulated in fp32 two at a time (v_dot2c_f32_bf16): a quarter of the instructions of convert + fma per element
    float acc[SK_COLS][SK_MAXM];
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c)
#pragma unroll
        for (int m = 0; m < SK_MAXM; ++m) acc[c][m] = 0.f;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
        const int k = ps * 512 + lane * 8;
        if (k >= p.K) continue;
#pragma unroll
        for (int m = 0; m < SK_MAXM; ++m) {
            if (m >= p.M) continue;
            const bf16x8 x8 = *reinterpret_cast<const bf16x8*>(xs + m * p.K + k);
#pragma unroll
            for (int c = 0; c < SK_COLS; ++c) {
                const bf16x8 w8 = wv[c][ps];
                float a = acc[c][m];
                a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 0, 1), __builtin_shufflevector(x8, x8, 0, 1), a, false);
                a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 2, 3), __builtin_shufflevector(x8, x8, 2, 3), a, false);
                a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 4, 5), __builtin_shufflevector(x8, x8, 4, 5), a, false);
                a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 6, 7), __builtin_shufflevector(x8, x8, 6, 7), a, false);
                acc[c][m] = a;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c) {
        const int n = n0 + c;
        if (n >= p.N) break;                                          // wave-uniform
#pragma unroll
        for (int m = 0; m < SK_MAXM; ++m)
            if (m < p.M) acc[c][m] = wave_sum(acc[c][m]);
        if (lane < p.M) {
            float v = 0.f;
#pragma unroll
            for (int m = 0; m < SK_MAXM; ++m) if (m == lane) v = acc[c][m];
            v += p.bias ? p.bias[n] : 0.f;
            if (p.act == 2) v = gelu_tanh(v);
            if (p.out32) p.out32[(long)lane * p.ldo32 + n] = (p.resid32 ? p.resid32[(long)lane * p.ldo32 + n] : 0.f) + v;
            else {
                const bf16_t o = f2bf(v);
                p.out16[(long)lane * p.ldo16 + n] = o;
                if (p.kc && n >= p.dkv) {                              // fused KV-cache append (was a separate launch per layer)
                    const int b = lane / p.U, u = lane - b * p.U;
                    const long row = ((long)b * p.Lmax + p.past + u) * p.dkv;
                    if (n < 2 * p.dkv) p.kc[row + n - p.dkv] = o; else p.vc[row + n - 2 * p.dkv] = o;
                }
            }
        }
    }
}

int skinny(const SkArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.M > SK_MAXM || (a.K % 8) || (a.ldw % 8) || a.K > 2048) return MI_ERR_ARG;
    const size_t lds = (size_t)a.M * a.K * sizeof(bf16_t);
    if (lds > 150 * 1024) return MI_ERR_UNSUPPORTED;
    if (a.K <= 512) hipLaunchKernelGGL(skinny_linear_kernel<8>, dim3(cdiv(a.N, 4 * SK_COLS)), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(skinny_linear_kernel<32>, dim3(cdiv(a.N, 4 * SK_COLS)), dim3(256), lds, st, a);
    return MI_OK;
}

// ---- attention of at most 8 new rows over cached keys (token step): one block per (row, head).  The MFMA attention kernel pads these rows to a 128-query tile and walks the
// keys in dependent steps (9-12 us over 250 encoder frames); here LPK = HD / 8 lanes share a key (16-B pieces of its K and V rows), the block's four waves split the keys, a wave
// requests all 2 NB pieces of a batch before it uses the first, soft-max in fp32 with a running max across batches, key slots and waves merged through LDS.
constexpr int DA_WAVES = 4;
struct DecAttnArgs {
    const bf16_t* q; long ldq; const bf16_t* K; const bf16_t* V; long ldkv, bstride;
    const int* enc_len; bf16_t* ctx; long ldo;
    int M, U, H, past, self, T_enc; float scale;
};
template <int HD>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecAttnArgs p) {
    constexpr int LPK = HD / 8, KPI = 64 / LPK, NB = 8, EPL = HD / 64;       // lanes per key, keys per wave and load round, rounds per batch, output dims per lane
    __shared__ float red[DA_WAVES * 64 * 8];                                  // [wave][key slot][dim]: KPI HD = 512 floats per wave
    __shared__ float rs[DA_WAVES * 8];
    __shared__ float rm[DA_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPK, c = lane % LPK;
    const int t = blockIdx.x;
    const int m = t / p.H, h = t - m * p.H, b = m / p.U, u = m - b * p.U;
    const int nkeys = p.self ? p.past + u + 1 : (p.enc_len ? (p.enc_len[b] < p.T_enc ? p.enc_len[b] : p.T_enc) : p.T_enc);
    const bf16_t* kb = p.K + (long)b * p.bstride + h * HD + c * 8;
    const bf16_t* vb = p.V + (long)b * p.bstride + h * HD + c * 8;
    const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(p.q + (long)m * p.ldq + h * HD + c * 8);
    float mrun = -INFINITY, lsum = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int key0 = wave * KPI; key0 < nkeys; key0 += DA_WAVES * KPI * NB) {           // wave-uniform
        bf16x8 k8[NB], v8[NB];
        float sv[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {                                 // unconditional, clamped: a load under a condition is its own basic block and costs a full wait at the join
            const int key = key0 + i * DA_WAVES * KPI + g;
            const long off = (long)(key < nkeys ? key : 0) * p.ldkv;
            k8[i] = *reinterpret_cast<const bf16x8*>(kb + off);
            v8[i] = *reinterpret_cast<const bf16x8*>(vb + off);
        }
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float sd = 0.f;
            sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 0, 1), __builtin_shufflevector(q8, q8, 0, 1), sd, false);
            sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 2, 3), __builtin_shufflevector(q8, q8, 2, 3), sd, false);
            sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 4, 5), __builtin_shufflevector(q8, q8, 4, 5), sd, false);
            sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 6, 7), __builtin_shufflevector(q8, q8, 6, 7), sd, false);
            sd += dpp_f32<0xB1, 0xF>(0.f, sd);                         // the key's LPK lanes: quad, half row (8 lanes) [, row (16 lanes)]
            sd += dpp_f32<0x4E, 0xF>(0.f, sd);
            sd += dpp_f32<0x141, 0xF>(0.f, sd);
            if (LPK == 16) sd += dpp_f32<0x140, 0xF>(0.f, sd);
            sv[i] = (key0 + i * DA_WAVES * KPI + g) < nkeys ? sd * p.scale : -INFINITY;
            bm = fmaxf(bm, sv[i]);
        }
        bm = wave_max(bm);                                             // finite: key slot 0 of the batch's first round is < nkeys
        const float mnew = fmaxf(mrun, bm);
        const float f = mrun == -INFINITY ? 0.f : __expf(mrun - mnew);
        lsum *= f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= f;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float pr = sv[i] == -INFINITY ? 0.f : __expf(sv[i] - mnew);
            lsum += pr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(pr, bf2f(v8[i][j]), acc[j]);
        }
    }
    // key slots and waves -> LDS (a wave without keys has mrun = -inf and zeros)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(wave * KPI + g) * HD + c * 8 + j] = acc[j];
    if (c == 0) rs[wave * KPI + g] = lsum;
    if (lane == 0) rm[wave] = mrun;
    __syncthreads();
    if (wave == 0) {
        float gm = -INFINITY;
#pragma unroll
        for (int w = 0; w < DA_WAVES; ++w) gm = fmaxf(gm, rm[w]);
        float tot = 0.f, o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
        for (int w = 0; w < DA_WAVES; ++w) {
            const float f = rm[w] == -INFINITY ? 0.f : __expf(rm[w] - gm);
#pragma unroll
            for (int k = 0; k < KPI; ++k) {
                tot = fmaf(rs[w * KPI + k], f, tot);
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = fmaf(red[(w * KPI + k) * HD + lane * EPL + e], f, o[e]);
            }
        }
        const float inv = tot > 0.f ? 1.f / tot : 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p.ctx[(long)m * p.ldo + h * HD + lane * EPL + e] = f2bf(o[e] * inv);
    }
}

int decode_attn(const DecAttnArgs& a, int hd, hipStream_t st) {
    if (hd == 64) hipLaunchKernelGGL(decode_attn_kernel<64>, dim3(a.M * a.H), dim3(256), 0, st, a);
    else if (hd == 128) hipLaunchKernelGGL(decode_attn_kernel<128>, dim3(a.M * a.H), dim3(256), 0, st, a);
    else return MI_ERR_UNSUPPORTED;
    return MI_OK;
}

constexpr int MAX_LAYERS = 48;
struct ReorderArgs { const bf16_t* src[2 * MAX_LAYERS]; bf16_t* dst[2 * MAX_LAYERS]; };

// dst[t][b, :rows, :] = src[t][beam_idx[b], :rows, :] for every cache tensor t (K and V of every layer)
__global__ __launch_bounds__(256) void kv_reorder_kernel(ReorderArgs a, const long* __restrict__ beam_idx, int BW, int rows, int Lmax, int d) {
    const int t = blockIdx.y;
    const int d8 = d >> 3;
    const long total = (long)BW * rows * d8;
    const bf16_t* src = a.src[t];
    bf16_t* dst = a.dst[t];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % d8);
        const long m = i / d8;
        const int b = (int)(m / rows), r = (int)(m % rows);
        const long sb = beam_idx[b];
        *reinterpret_cast<bf16x8*>(dst + ((long)b * Lmax + r) * d + c * 8) = *reinterpret_cast<const bf16x8*>(src + (sb * Lmax + r) * d + c * 8);
    }
}

struct Carver {
    char* base; size_t off;
    void* take(size_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; }
};
}  // namespace
// decoder_fused.hip: the three-launches-per-layer form of the token step
bool gpt2_step_fused_ok(const mi_gpt2_config& c, int B, int U);
size_t gpt2_step_fused_floats(const mi_gpt2_config& c, int M);
int gpt2_step_fused(const mi_gpt2_config& c, const void* const* weights, const long* ids, float emb_scale, int M, int past, int Lmax, void* const* kcache, void* const* vcache,
                    const void* const* cross_kv, int T_enc, const int* enc_len, float* fws, bf16_t* hid, hipStream_t st);
namespace {
struct StepWs { float* x; bf16_t *a, *qkv, *ctx, *qq, *m, *hid; float* fws; size_t bytes; };
StepWs carve(const mi_gpt2_config& c, int M, int B, int U, void* base) {
    Carver k{(char*)base, 0};
    StepWs w;
    w.x = (float*)k.take((size_t)M * c.d * 4);
    w.a = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.qkv = (bf16_t*)k.take((size_t)M * 3 * c.d * 2);
    w.ctx = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.qq = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.m = (bf16_t*)k.take((size_t)M * 4 * c.d * 2);
    w.hid = (bf16_t*)k.take((size_t)B * c.d * 2);
    w.fws = gpt2_step_fused_ok(c, B, U) ? (float*)k.take(gpt2_step_fused_floats(c, M) * sizeof(float)) : nullptr;
    w.bytes = k.off;
    return w;
}

#define RUN(expr) do { int rc__ = (expr); if (rc__ != MI_OK) return rc__; } while (0)

}  // namespace

extern "C" size_t mi_gpt2_step_workspace_bytes(const mi_gpt2_config* cfg, int B, int U) { return carve(*cfg, B * U, B, U, nullptr).bytes; }

// weights: [wte f32 (V,d), pos f32 (n_pos,d), lnf_g, lnf_b, lm_head bf16 (V,d)] then per layer 18 pointers in the order
//   ln1_g, ln1_b, wqkv, bqkv, wo, bo, lnc_g, lnc_b, wq, bq, wco, bco, ln2_g, ln2_b, wfc, bfc, wpr, bpr     (matrices bf16 (out,in), vectors f32)
// ids_new (B,U) int64; kcache / vcache: L pointers to (B, Lmax, d) bf16; cross_kv: L pointers to (B*T_enc, 2d) bf16 [K | V];
// logits (B, ld_logits) f32 of the LAST new position.  Appends the new tokens' K/V at rows [past, past + U).
extern "C" int mi_gpt2_step(const mi_gpt2_config* cfg, const void* const* weights, const long* ids_new, int B, int U, int past, int Lmax,
                            void* const* kcache, void* const* vcache, const void* const* cross_kv, int T_enc, const int* enc_len,
                            float emb_scale, void* workspace, size_t workspace_bytes, float* logits, long ld_logits, hipStream_t st) {
    MI_ENTER();
    const mi_gpt2_config& c = *cfg;
    if (B <= 0 || U <= 0 || past < 0 || past + U > Lmax || c.L <= 0 || c.L > MAX_LAYERS || c.d % c.H || (c.d % 8)) return MI_ERR_ARG;
    const int hd = c.d / c.H;
    if (hd != 64 && hd != 128) return MI_ERR_UNSUPPORTED;
    const int M = B * U, d = c.d;
    StepWs w = carve(c, M, B, U, workspace);
    if (w.bytes > workspace_bytes) return MI_ERR_ARG;
    const float scale = 1.0f / sqrtf((float)hd);
    auto Gf = [&](int i) { return (const float*)weights[i]; };
    auto Lw = [&](int l, int i) { return weights[5 + l * 18 + i]; };
    auto Lf = [&](int l, int i) { return (const float*)weights[5 + l * 18 + i]; };
    auto ln = [&](const float* x, long ldx, const float* g, const float* b, bf16_t* out, int rows) {
        return mi_layernorm_chain(x, ldx, nullptr, 1, nullptr, nullptr, 0.f, nullptr, 0, g, b, c.eps, out, d, nullptr, 0, nullptr, nullptr, nullptr, 0, rows, d, st);
    };
    if (c.step_form != 1 && w.fws) {
        // ---- fused token step (decoder_fused.hip): three launches per layer, every cross-workgroup reduction folded into the next launch's prologue (the embedding too)
        RUN(gpt2_step_fused(c, weights, ids_new, emb_scale, M, past, Lmax, kcache, vcache, cross_kv, T_enc, enc_len, w.fws, w.hid, st));
        SkArgs a{}; a.x16 = w.hid; a.ldx16 = d; a.W = (const bf16_t*)weights[4]; a.ldw = d; a.out32 = logits; a.ldo32 = ld_logits; a.M = B; a.N = c.V; a.K = d; a.act = 0;
        RUN(skinny(a, st));
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    RUN(mi_embed_tokens(ids_new, Gf(0), emb_scale, Gf(1), past, U, d, M, c.V, w.x, st));
    if (M <= SK_MAXM && (d % 8) == 0 && 4 * d <= 2048) {
        // ---- skinny token step: LayerNorms, biases, activations and residual adds fused into GEMV-style linears (8 launches per layer)
        auto lin_ln = [&](const float* g, const float* b, const void* W, const float* bias, int N, bf16_t* out, int act) {
            SkArgs a{}; a.x32 = w.x; a.ldx = d; a.ln_g = g; a.ln_b = b; a.eps = c.eps; a.W = (const bf16_t*)W; a.ldw = d; a.bias = bias;
            a.out16 = out; a.ldo16 = N; a.M = M; a.N = N; a.K = d; a.act = act;
            return skinny(a, st);
        };
        auto lin_res = [&](const bf16_t* in, int K, const void* W, const float* bias) {      // x += in · W^T + b
            SkArgs a{}; a.x16 = in; a.ldx16 = K; a.W = (const bf16_t*)W; a.ldw = K; a.bias = bias; a.out32 = w.x; a.ldo32 = d; a.resid32 = w.x;
            a.M = M; a.N = d; a.K = K; a.act = 0;
            return skinny(a, st);
        };
        for (int l = 0; l < c.L; ++l) {
            bf16_t* kc = (bf16_t*)kcache[l];
            bf16_t* vc = (bf16_t*)vcache[l];
            const bf16_t* ckv = (const bf16_t*)cross_kv[l];
            {   // LN1 + fused QKV projection; its K / V columns go straight into the caches as well
                SkArgs a{}; a.x32 = w.x; a.ldx = d; a.ln_g = Lf(l, 0); a.ln_b = Lf(l, 1); a.eps = c.eps; a.W = (const bf16_t*)Lw(l, 2); a.ldw = d; a.bias = Lf(l, 3);
                a.out16 = w.qkv; a.ldo16 = 3 * d; a.M = M; a.N = 3 * d; a.K = d; a.act = 0;
                a.kc = kc; a.vc = vc; a.U = U; a.past = past; a.Lmax = Lmax; a.dkv = d;
                RUN(skinny(a, st));
            }
            RUN(decode_attn(DecAttnArgs{w.qkv, 3 * d, kc, vc, d, (long)Lmax * d, nullptr, w.ctx, d, M, U, c.H, past, 1, 0, scale}, hd, st));
            RUN(lin_res(w.ctx, d, Lw(l, 4), Lf(l, 5)));
            RUN(lin_ln(Lf(l, 6), Lf(l, 7), Lw(l, 8), Lf(l, 9), d, w.qq, 0));
            RUN(decode_attn(DecAttnArgs{w.qq, d, ckv, ckv + d, 2 * d, (long)T_enc * 2 * d, enc_len, w.ctx, d, M, U, c.H, 0, 0, T_enc, scale}, hd, st));
            RUN(lin_res(w.ctx, d, Lw(l, 10), Lf(l, 11)));
            RUN(lin_ln(Lf(l, 12), Lf(l, 13), Lw(l, 14), Lf(l, 15), 4 * d, w.m, 2));
            RUN(lin_res(w.m, 4 * d, Lw(l, 16), Lf(l, 17)));
        }
        // ln_f on the last new position of every sequence + lm head -> fp32 logits
        SkArgs a{}; a.x32 = w.x + (size_t)(U - 1) * d; a.ldx = (long)U * d; a.ln_g = Gf(2); a.ln_b = Gf(3); a.eps = c.eps; a.W = (const bf16_t*)weights[4];
        a.ldw = d; a.out32 = logits; a.ldo32 = ld_logits; a.M = B; a.N = c.V; a.K = d; a.act = 0;
        RUN(skinny(a, st));
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    for (int l = 0; l < c.L; ++l) {
        bf16_t* kc = (bf16_t*)kcache[l];
        bf16_t* vc = (bf16_t*)vcache[l];
        const bf16_t* ckv = (const bf16_t*)cross_kv[l];
        RUN(ln(w.x, d, Lf(l, 0), Lf(l, 1), w.a, M));
        RUN(mi_gemm_bf16(w.a, d, Lw(l, 2), d, Lf(l, 3), 1, w.qkv, 3 * d, 0, nullptr, 0, 1.f, 0, M, 3 * d, d, 0, 0, st));
        {
            const long total = (long)M * (d / 8);
            hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w.qkv, (long)3 * d, kc, vc, B, U, past, Lmax, d);
        }
        RUN(mi_attention_qkv_bf16(w.qkv, 3 * d, kc, d, vc, d, nullptr, 0, nullptr, nullptr, nullptr, w.ctx, d, B, U, past + U, (long)Lmax * d, c.H, hd,
                                  scale, 1, st));
        RUN(mi_gemm_bf16(w.ctx, d, Lw(l, 4), d, Lf(l, 5), 1, w.x, d, 1, w.x, d, 1.f, 0, M, d, d, 0, 0, st));
        RUN(ln(w.x, d, Lf(l, 6), Lf(l, 7), w.a, M));
        RUN(mi_gemm_bf16(w.a, d, Lw(l, 8), d, Lf(l, 9), 1, w.qq, d, 0, nullptr, 0, 1.f, 0, M, d, d, 0, 0, st));
        RUN(mi_attention_qkv_bf16(w.qq, d, ckv, 2 * d, ckv + d, 2 * d, nullptr, 0, nullptr, nullptr, enc_len, w.ctx, d, B, U, T_enc, 0, c.H, hd, scale, 0, st));
        RUN(mi_gemm_bf16(w.ctx, d, Lw(l, 10), d, Lf(l, 11), 1, w.x, d, 1, w.x, d, 1.f, 0, M, d, d, 0, 0, st));
        RUN(ln(w.x, d, Lf(l, 12), Lf(l, 13), w.a, M));
        RUN(mi_gemm_bf16(w.a, d, Lw(l, 14), d, Lf(l, 15), 1, w.m, 4 * d, 0, nullptr, 0, 1.f, 2, M, 4 * d, d, 0, 0, st));
        RUN(mi_gemm_bf16(w.m, 4 * d, Lw(l, 16), 4 * d, Lf(l, 17), 1, w.x, d, 1, w.x, d, 1.f, 0, M, d, 4 * d, 0, 0, st));
    }
    // ln_f on the last new position of every sequence (rows b*U + U-1: a strided view), then the lm head
    RUN(ln(w.x + (size_t)(U - 1) * d, (long)U * d, Gf(2), Gf(3), w.hid, B));
    RUN(mi_gemm_bf16(w.hid, d, weights[4], d, nullptr, 0, logits, ld_logits, 1, nullptr, 0, 1.f, 0, B, c.V, d, 0, 0, st));
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// beam-search re-ordering of every layer's KV cache in one launch: dst_x[l][b] = src_x[l][beam_idx[b]] for the first `rows` positions
extern "C" int mi_kv_cache_reorder(const void* const* src_k, const void* const* src_v, void* const* dst_k, void* const* dst_v, const long* beam_idx,
                                   int L, int BW, int rows, int Lmax, int d, hipStream_t st) {
    MI_ENTER();
    if (L <= 0 || L > MAX_LAYERS || BW <= 0 || rows < 0 || rows > Lmax || (d % 8)) return MI_ERR_ARG;
    if (rows == 0) return MI_OK;
    ReorderArgs a;
    for (int l = 0; l < L; ++l) {
        a.src[2 * l] = (const bf16_t*)src_k[l]; a.src[2 * l + 1] = (const bf16_t*)src_v[l];
        a.dst[2 * l] = (bf16_t*)dst_k[l]; a.dst[2 * l + 1] = (bf16_t*)dst_v[l];
    }
    const long total = (long)BW * rows * (d / 8);
    const long g = (total + 255) / 256;
    hipLaunchKernelGGL(kv_reorder_kernel, dim3((unsigned)(g > 1024 ? 1024 : g), 2 * L), dim3(256), 0, st, a, beam_idx, BW, rows, Lmax, d);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
