// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libhfasr_hip.so.
// Wave = 64 lanes.  bf16 storage, fp32 accumulation everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <unordered_map>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MI_OK 0
#define MI_ERR_ARG (-1)
#define MI_ERR_LAUNCH (-2)
#define MI_ERR_UNSUPPORTED (-3)

// hipGetLastError() is per-thread and sticky across unrelated runtime calls (PyTorch leaves benign errors such as a
// failed hipPointerGetAttributes behind): clear it on entry, so that MI_CHECK_LAUNCH reports only our own launches.
#define MI_ENTER() (void)hipGetLastError()

extern "C" void mi_record_hip_error(int code, const char* file, int line);   // encoder.hip (diagnostics only)

#define MI_CHECK_LAUNCH()                                            \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) {                                     \
            mi_record_hip_error((int)e__, __FILE__, __LINE__);       \
            return MI_ERR_LAUNCH;                                    \
        }                                                            \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// erf-GELU, the reference's "gelu" activation (ACT2FN["gelu"] / nn.GELU()):  0.5 x (1 + erf(x / sqrt 2)).
// Every forward GELU of the path is rounded to bf16 (8 significant bits) right after it, and in the GEMM epilogues it is the critical path: 128 values
// per lane, VALU-bound (17 instructions per value with an Abramowitz-Stegun erf, two of them quarter-rate).  So the forward uses
//     x * Phi(x),  Phi(x) ~= 1 / (1 + 2^(xc * (c1 + c3 xc^2 + c5 xc^4))),  xc = clamp(x, -10, 10)      (c* = -log2(e) * minimax fit of logit(Phi), odd in x)
// 9 instructions (med3, mul, 2 fma, mul, exp2, add, rcp, mul); |error| <= 2.6e-5 absolute for |x| <= 10 (the usual tanh form: 4.7e-4), relative error <= 5e-4 for
// x > -2 — below a quarter of a bf16 half-ulp wherever |GELU| > 0.02 — and values in the negative tail (|GELU| < 0.016) within 2.6e-5 of exact.
// The clamp is what makes it total: the quintic's x^5 coefficient has the opposite sign of the others, so the un-clamped exponent turns around at |x| ~ 11.1
// (GELU(12) came out as 9e-12, GELU(-12) as -12).  At |xc| = 10 the exponent is -+28.5, i.e. Phi = 1 - 3e-9 / 3e-9: beyond it the result is x * Phi(+-10)
// = x resp. -0 to fp32 precision, +-inf included (x = -inf: -inf * 2.6e-9 = -inf is avoided by the select below).  The backward (gelu_erf_grad below) uses the same fit for Phi.
__device__ __forceinline__ float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -10.0f, 10.0f);
    const float x2 = xc * xc;
    float p = fmaf(x2, 1.01426783e-3f, -1.06775756e-1f);          // -log2(e) * (-7.03036837e-4, 7.40113137e-2)
    p = fmaf(p, x2, -2.30112135f);                                //  -log2(e) * 1.59501574
    const float e = __builtin_amdgcn_exp2f(p * xc);
    const float r = x * __builtin_amdgcn_rcpf(1.0f + e);
    return x < -10.0f ? -0.0f : r;                                // erf-GELU(x <= -10) = -0 to fp32 precision (|x Phi(x)| < 8e-23); also keeps -inf from producing -inf * tiny
}

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }      // x -> -inf: rcp(inf) = 0;  x -> +inf: rcp(1) = 1

// GPT-2's "gelu_new" (tanh form; transformers activations.NewGELUActivation), used by the decoder MLP
__device__ __forceinline__ float gelu_tanh(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float e = __expf(2.f * u);                       // tanh(u) = 1 - 2/(e^{2u}+1), safe at both ends
    return 0.5f * x * (2.f - 2.f / (e + 1.f));
}

// derivatives of the two GELUs (training: element-wise kernels, the conv front end's backward, the dX GEMM's training epilogue)
// GELU'(x) = Phi(x) + x phi(x), with the forward's logistic fit for Phi and phi from one exp2 (|error| <= 5e-5 absolute over the whole line): 12 instructions, three of them quarter-rate — the
// erf form (fast_erf + a separate exp: 22) made every epilogue and element-wise kernel that applies the derivative VALU-bound (the dX GEMM's training epilogue, conv1's
// backward, mi_act_bwd); every consumer rounds the product to bf16 (relative 2e-3).  The clamp keeps +-inf finite: Phi(+-10) = 1 / 0 to fp32, x phi(x) -> 0.
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -10.0f, 10.0f);
    const float x2 = xc * xc;
    float p = fmaf(x2, 1.01426783e-3f, -1.06775756e-1f);
    p = fmaf(p, x2, -2.30112135f);
    const float cdf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xc));
    const float pdf = __builtin_amdgcn_exp2f(-0.72134752044f * x2);                    // exp(-x^2 / 2)
    return fmaf(xc * 0.3989422804014327f, pdf, cdf);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float e = __expf(2.f * u);
    const float th = 1.f - 2.f / (e + 1.f);
    return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x * x);
}

// Wave-wide reductions by DPP (no LDS round trips: `__shfl_xor` is a ds_bpermute_b32 per step on gfx9, ~100 cycles of latency each, and the
// small latency-bound kernels — token-step GEMVs, CTC, row statistics — are chains of them).  Within a row of 16 lanes: quad_perm, row_half_mirror,
// row_mirror leave the row total in every lane; row_bcast:15 / row_bcast:31 carry it across the four rows into lane 63; v_readlane broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1, 0xF>(0.f, v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(0.f, v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xF>(0.f, v);       // row_half_mirror
    v += dpp_f32<0x140, 0xF>(0.f, v);       // row_mirror: every lane of a row holds the row total
    v += dpp_f32<0x142, 0xA>(0.f, v);       // row_bcast:15 into rows 1 and 3
    v += dpp_f32<0x143, 0xC>(0.f, v);       // row_bcast:31 into rows 2 and 3: lane 63 holds the wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));        // rows outside the mask keep `old` = their own value
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// max / sum of a value with its partner in the other half of the wave (lane ^ 32): one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float half_swap_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device) — a function-local `static bool` configures only
// the device that happened to be current at the first call.  One table per translation unit (kernel -> bit mask of configured devices).
static inline bool ensure_dynamic_lds_ptr(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    unsigned long long& m = done[kernel];
    if ((m >> dev) & 1ull) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    m |= 1ull << dev;
    return true;
}
template <int TAG>
static inline bool ensure_dynamic_lds(const void* kernel, size_t bytes) { return ensure_dynamic_lds_ptr(kernel, bytes); }

// Fixed-order reduction of `rows` partial rows of n floats (the second stage of every parameter-gradient reduction of the training step: no float atomics anywhere, so the
// backward is bit-reproducible run to run): a block owns 16 columns over ALL rows — thread (column, row group g) adds rows g, g + 16, ... with eight loads in flight, the 16
// groups are combined in order through LDS — and hands (column, total) to `emit`.  A few dozen to a few thousand rows: all launch and memory latency, no bandwidth.
struct EmitAdd { float* out; __device__ void operator()(int i, float v) const { out[i] += v; } };
template <typename Emit>
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ partial, int rows, int n, Emit emit) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx;
    float s = 0.f;
    if (i < n) {
        const float* src = partial + i;
        int r = ty;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(r + 16 * u) * n];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < rows; r += 16) s += src[(long)r * n];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][tx];
        emit(i, t);
    }
}
template <typename Emit>
static inline void rows_reduce_launch(const float* partial, int rows, int n, Emit emit, hipStream_t st) {
    hipLaunchKernelGGL(rows_reduce_kernel<Emit>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, partial, rows, n, emit);
}

// Counter-based uniform draws of the dropout masks (dropout.hip, attn_bwd.hip; host twin: huggingface_asr_amd/synth.py `dropout_keep`):
// one splitmix64 round over (quad index ^ key) yields FOUR 16-bit draws — element idx takes bits 63..48, 47..32, 31..16, 15..0 for idx & 3 = 0, 1, 2, 3 —
// so a mask costs one 64-bit hash per four elements.  The hash (two 64-bit multiplies = eight quarter-rate 32-bit ones, ~200 issue cycles) is what bounds the mask
// kernels and was, at one hash per PAIR with 24-bit draws (rounds 2-3), the largest VALU item of the training GEMM epilogues — more than the GELU next to it;
// 16 bits resolve p to 1.5e-5 (keep probability at p = 0.1: 0.899994).
__device__ __forceinline__ unsigned long long mask_hash(unsigned long long key, unsigned long long quad) {
    unsigned long long z = (quad ^ key) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float mask_u01(unsigned long long h, int j) {        // j = idx & 3
    const unsigned w = j < 2 ? (unsigned)(h >> 32) : (unsigned)h;
    return (float)((j & 1) ? (w & 0xFFFFu) : (w >> 16)) * (1.0f / 65536.0f);
}
__device__ __forceinline__ float mask_u01_at(unsigned long long key, unsigned long long idx) { return mask_u01(mask_hash(key, idx >> 2), (int)(idx & 3)); }
// keep factors of the four elements of one (aligned) quad
__device__ __forceinline__ void mask_keep4(unsigned long long key, unsigned long long quad, float p, float inv_keep, float* ks) {
    const unsigned long long h = mask_hash(key, quad);
#pragma unroll
    for (int j = 0; j < 4; ++j) ks[j] = mask_u01(h, j) >= p ? inv_keep : 0.f;
}

