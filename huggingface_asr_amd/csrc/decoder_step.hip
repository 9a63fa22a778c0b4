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
struct StepWs { float* x; bf16_t *a, *qkv, *ctx, *qq, *m, *hid; size_t bytes; };
StepWs carve(const mi_gpt2_config& c, int M, int B, void* base) {
    Carver k{(char*)base, 0};
    StepWs w;
    w.x = (float*)k.take((size_t)M * c.d * 4);
    w.a = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.qkv = (bf16_t*)k.take((size_t)M * 3 * c.d * 2);
    w.ctx = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.qq = (bf16_t*)k.take((size_t)M * c.d * 2);
    w.m = (bf16_t*)k.take((size_t)M * 4 * c.d * 2);
    w.hid = (bf16_t*)k.take((size_t)B * c.d * 2);
    w.bytes = k.off;
    return w;
}

#define RUN(expr) do { int rc__ = (expr); if (rc__ != MI_OK) return rc__; } while (0)

}  // namespace

extern "C" size_t mi_gpt2_step_workspace_bytes(const mi_gpt2_config* cfg, int B, int U) { return carve(*cfg, B * U, B, nullptr).bytes; }

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
    StepWs w = carve(c, M, B, workspace);
    if (w.bytes > workspace_bytes) return MI_ERR_ARG;
    const float scale = 1.0f / sqrtf((float)hd);
    auto Gf = [&](int i) { return (const float*)weights[i]; };
    auto Lw = [&](int l, int i) { return weights[5 + l * 18 + i]; };
    auto Lf = [&](int l, int i) { return (const float*)weights[5 + l * 18 + i]; };
    auto ln = [&](const float* x, long ldx, const float* g, const float* b, bf16_t* out, int rows) {
        return mi_layernorm_chain(x, ldx, nullptr, 1, nullptr, nullptr, 0.f, nullptr, 0, g, b, c.eps, out, d, nullptr, 0, nullptr, nullptr, nullptr, 0, rows, d, st);
    };
    RUN(mi_embed_tokens(ids_new, Gf(0), emb_scale, Gf(1), past, U, d, M, c.V, w.x, st));
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
