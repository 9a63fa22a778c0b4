// Dropout for the training step on gfx950 — counter-based masks (no RNG state, no stored masks).
//
// Reference: torch.nn.Dropout at the points the reference applies it in train() mode (e_branchformer.py:132,203,288,301,451;
// tf wav2vec2_conformer FFN :353,356, encoder input :674; GPT-2 embd / attn / resid dropouts).  torch's Philox stream cannot be
// reproduced, so the mask is DEFINED here as a pure function of (seed, stream, logical element index):
//     keep(idx) = u16(idx) * 2^-16 >= p,   u16 = 16-bit field (idx & 3) — from the top — of splitmix64((idx >> 2) ^ key),   key = (stream << 32) ^ seed
// the same hash as huggingface_asr_amd/synth.py (`dropout_keep`), so the CPU oracle is run with the identical masks in the parity
// tests and the backward pass regenerates the mask instead of storing it.   out = x * keep / (1 - p).
#include "common.hpp"

namespace {

__device__ __forceinline__ float keep_scale(unsigned long long key, unsigned long long idx, float p, float inv_keep) {
    return mask_u01_at(key, idx) >= p ? inv_keep : 0.f;
}

// element (m, n) of an (M, N) matrix has logical index m * N + n whatever the leading dimensions are
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dropout_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ out, long ldo, int M, int N,
                                                       float alpha, float p, unsigned long long key) {
    const long total = (long)M * N;
    const float inv_keep = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float v = (float)x[(long)m * ldx + n] * alpha * keep_scale(key, (unsigned long long)i, p, inv_keep);
        out[(long)m * ldo + n] = (TO)v;
    }
}

// Fast form (rows of a multiple of 8 elements, 16-B aligned, fewer than 2^31 elements): a thread owns 8 consecutive elements — one 16-B access per
// bf16 operand, one 32-bit division per 8 elements instead of a 64-bit division per element; the eight hashes are what is left (the mask stays
// the same pure function of the logical element index, so the oracle's masks and the backward's regenerated masks are unchanged).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dropout_vec8_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ out, long ldo, int M, int N,
                                                            float alpha, float p, unsigned long long key) {
    const unsigned n8 = (unsigned)N >> 3;
    const unsigned total8 = (unsigned)M * n8;
    const float inv_keep = 1.f / (1.f - p);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total8; i += gridDim.x * 256u) {
        const unsigned m = i / n8, c = (i - m * n8) << 3;
        float v[8];
        if constexpr (sizeof(TI) == 2) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (long)m * ldx + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + (long)m * ldx + c), b = *reinterpret_cast<const f32x4*>(x + (long)m * ldx + c + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        float ks[8];                                                              // 8 elements = 2 hashes
        mask_keep4(key, (unsigned long long)i << 1, p, inv_keep, ks);
        mask_keep4(key, ((unsigned long long)i << 1) + 1, p, inv_keep, ks + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha * ks[e];
        if constexpr (sizeof(TO) == 2) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x8*>(out + (long)m * ldo + c) = o;
        } else {
            *reinterpret_cast<f32x4*>(out + (long)m * ldo + c) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(out + (long)m * ldo + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
}
__global__ __launch_bounds__(256) void dropout_add_vec4_kernel(float* __restrict__ y, long ldy, const float* __restrict__ resid, long ldr,
                                                                const float* __restrict__ t, long ldt, int M, int N, float alpha, float p,
                                                                unsigned long long key) {
    const unsigned n4 = (unsigned)N >> 2;
    const unsigned total4 = (unsigned)M * n4;
    const float inv_keep = 1.f / (1.f - p);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned m = i / n4, c = (i - m * n4) << 2;
        const f32x4 r = *reinterpret_cast<const f32x4*>(resid + (long)m * ldr + c), tv = *reinterpret_cast<const f32x4*>(t + (long)m * ldt + c);
        float ks[4];
        mask_keep4(key, (unsigned long long)i, p, inv_keep, ks);
        f32x4 o;
        o.x = r.x + alpha * tv.x * ks[0];
        o.y = r.y + alpha * tv.y * ks[1];
        o.z = r.z + alpha * tv.z * ks[2];
        o.w = r.w + alpha * tv.w * ks[3];
        *reinterpret_cast<f32x4*>(y + (long)m * ldy + c) = o;
    }
}

// y = resid + alpha * dropout(t)
__global__ __launch_bounds__(256) void dropout_add_kernel(float* __restrict__ y, long ldy, const float* __restrict__ resid, long ldr,
                                                           const float* __restrict__ t, long ldt, int M, int N, float alpha, float p,
                                                           unsigned long long key) {
    const long total = (long)M * N;
    const float inv_keep = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        y[(long)m * ldy + n] = resid[(long)m * ldr + n] + alpha * t[(long)m * ldt + n] * keep_scale(key, (unsigned long long)i, p, inv_keep);
    }
}

int grid_for(long n) { const long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

}  // namespace

// out = alpha * dropout(x);  in_dtype / out_dtype: 0 fp32, 1 bf16;  out may alias x when the dtypes match
extern "C" int mi_dropout(const void* x, long ldx, int in_dtype, void* out, long ldo, int out_dtype, int M, int N, float alpha, float p,
                          unsigned seed, unsigned stream_id, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || p < 0.f || p >= 1.f) return MI_ERR_ARG;
    const unsigned long long key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    const bool vec = (N % 8) == 0 && (ldx % 8) == 0 && (ldo % 8) == 0 && (long)M * N < (1L << 31) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec) {
        const int g8 = grid_for((long)M * N / 8);
        if (in_dtype == 0 && out_dtype == 0) hipLaunchKernelGGL((dropout_vec8_kernel<float, float>), dim3(g8), dim3(256), 0, st, (const float*)x, ldx, (float*)out, ldo, M, N, alpha, p, key);
        else if (in_dtype == 0 && out_dtype == 1) hipLaunchKernelGGL((dropout_vec8_kernel<float, bf16_t>), dim3(g8), dim3(256), 0, st, (const float*)x, ldx, (bf16_t*)out, ldo, M, N, alpha, p, key);
        else if (in_dtype == 1 && out_dtype == 1) hipLaunchKernelGGL((dropout_vec8_kernel<bf16_t, bf16_t>), dim3(g8), dim3(256), 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, N, alpha, p, key);
        else if (in_dtype == 1 && out_dtype == 0) hipLaunchKernelGGL((dropout_vec8_kernel<bf16_t, float>), dim3(g8), dim3(256), 0, st, (const bf16_t*)x, ldx, (float*)out, ldo, M, N, alpha, p, key);
        else return MI_ERR_ARG;
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const int g = grid_for((long)M * N);
    if (in_dtype == 0 && out_dtype == 0) hipLaunchKernelGGL((dropout_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)x, ldx, (float*)out, ldo, M, N, alpha, p, key);
    else if (in_dtype == 0 && out_dtype == 1) hipLaunchKernelGGL((dropout_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)x, ldx, (bf16_t*)out, ldo, M, N, alpha, p, key);
    else if (in_dtype == 1 && out_dtype == 1) hipLaunchKernelGGL((dropout_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, N, alpha, p, key);
    else if (in_dtype == 1 && out_dtype == 0) hipLaunchKernelGGL((dropout_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx, (float*)out, ldo, M, N, alpha, p, key);
    else return MI_ERR_ARG;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_dropout_add_f32(float* y, long ldy, const float* resid, long ldr, const float* t, long ldt, int M, int N, float alpha, float p,
                                  unsigned seed, unsigned stream_id, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || p < 0.f || p >= 1.f) return MI_ERR_ARG;
    const unsigned long long key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    const bool vec = (N % 4) == 0 && (ldy % 4) == 0 && (ldr % 4) == 0 && (ldt % 4) == 0 && (long)M * N < (1L << 31) &&
                     ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(t)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(dropout_add_vec4_kernel, dim3(grid_for((long)M * N / 4)), dim3(256), 0, st, y, ldy, resid, ldr, t, ldt, M, N, alpha, p, key);
    else hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, y, ldy, resid, ldr, t, ldt, M, N, alpha, p, key);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
