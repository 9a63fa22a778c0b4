// BEST-RQ pre-training pieces on gfx950 (reference src/models/bestrq.py):
//   * RandomProjectionQuantizer.forward (:66-82): targets[m] = argmin_c || CB[c] - normalize(x[m] · P) ||  over the codebook,
//     x = 4 stacked log-mel frames (in_dim = 320), P (in_dim, cd) and CB (C, cd) frozen random buffers.  fp32 throughout (a bf16
//     projection would flip near-ties), one wave per row: lanes split in_dim for the projection, then split the codebook.
//   * BestRQMask._mask_hidden_states (:84-97): masked frames are REPLACED by N(0, std) noise.  torch's `normal_` stream cannot be
//     reproduced; the noise is counter-based (Box-Muller on the splitmix64 hash of dropout.hip / synth.py), so the oracle runs with
//     the identical noise in the parity tests.
#include "common.hpp"

namespace {

constexpr int RPQ_MAX_H = 64;        // books * codebook_dim held in registers

__device__ __forceinline__ unsigned long long bq_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float bq_uniform(unsigned long long key, unsigned long long idx) {
    const unsigned long long h = bq_splitmix64(bq_splitmix64(idx ^ key) + key);
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

// one wave per row.  QUIRK kept from the reference (bestrq.py:81): `F.normalize(x[:, None] @ P)` uses F.normalize's DEFAULT dim = 1,
// i.e. the projection is normalised ACROSS THE CODEBOOKS axis (for one book: every coordinate becomes its own sign), not over codebook_dim.
__global__ __launch_bounds__(256) void rpq_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ P, const float* __restrict__ CB,
                                                   long* __restrict__ out, int M, int in_dim, int cd, int C, int nb) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nh = nb * cd;
    float h[RPQ_MAX_H];
#pragma unroll
    for (int j = 0; j < RPQ_MAX_H; ++j) h[j] = 0.f;
    const float* xr = x + (long)row * ldx;
    for (int i = lane; i < in_dim; i += 64) {
        const float v = xr[i];
#pragma unroll
        for (int j = 0; j < RPQ_MAX_H; ++j)
            if (j < nh) h[j] = fmaf(v, P[((long)(j / cd) * in_dim + i) * cd + (j % cd)], h[j]);
    }
#pragma unroll
    for (int j = 0; j < RPQ_MAX_H; ++j)
        if (j < nh) h[j] = wave_sum(h[j]);
    // normalise over the books axis: h[k][j] / max(sqrt(sum_k' h[k'][j]^2), 1e-12)
    float den[RPQ_MAX_H];
#pragma unroll
    for (int j = 0; j < RPQ_MAX_H; ++j) den[j] = 0.f;
#pragma unroll
    for (int j = 0; j < RPQ_MAX_H; ++j)
        if (j < nh) {
#pragma unroll
            for (int jj = 0; jj < RPQ_MAX_H; ++jj)
                if (jj < nh && (jj % cd) == (j % cd)) den[j] = fmaf(h[jj], h[jj], den[j]);
        }
#pragma unroll
    for (int j = 0; j < RPQ_MAX_H; ++j)
        if (j < nh) h[j] = h[j] / fmaxf(sqrtf(den[j]), 1e-12f);
    for (int k = 0; k < nb; ++k) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float* cb = CB + ((long)k * C + c) * cd;
            float d2 = 0.f;
#pragma unroll
            for (int j = 0; j < RPQ_MAX_H; ++j)
                if (j < nh && (j / cd) == k) { const float t = cb[j - k * cd] - h[j]; d2 = fmaf(t, t, d2); }
            if (d2 < best) { best = d2; bi = c; }                        // ascending c per lane: first minimum wins
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) out[(long)k * M + row] = bi;
    }
}

// x[m, :] = std * N(0,1) where time_mask[m] != 0;  element (m, c): Box-Muller on two uniforms of logical index m * N + c
__global__ __launch_bounds__(256) void mask_noise_kernel(float* __restrict__ x, long ld, const unsigned char* __restrict__ tmask, int M, int N, float std,
                                                          unsigned long long key) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        if (!tmask[m]) continue;
        const float u1 = fmaxf(bq_uniform(key, 2ull * (unsigned long long)i), 2.98023224e-08f);
        const float u2 = bq_uniform(key, 2ull * (unsigned long long)i + 1ull);
        x[(long)m * ld + c] = std * sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
}

}  // namespace

// x (M, in_dim) f32 (row stride ldx), P (books, in_dim, cd), CB (books, C, cd) -> targets (books, M) int64
extern "C" int mi_rpq_targets(const float* x, long ldx, const float* P, const float* CB, long* targets, int M, int in_dim, int cd, int C, int books,
                              hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || in_dim <= 0 || cd <= 0 || books <= 0 || books * cd > RPQ_MAX_H || C <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(rpq_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, ldx, P, CB, targets, M, in_dim, cd, C, books);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_mask_noise_f32(float* x, long ld, const unsigned char* time_mask, int M, int N, float std, unsigned seed, unsigned stream_id,
                                 hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || !time_mask) return MI_ERR_ARG;
    const unsigned long long key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    const long g = ((long)M * N + 255) / 256;
    hipLaunchKernelGGL(mask_noise_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, x, ld, time_mask, M, N, std, key);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
