// Whisper front end + small glue kernels for the Whisper-style encoder (BASELINE.json config 4) on gfx950.
//
// Replaces transformers' WhisperFeatureExtractor numpy path (feature_extraction_whisper.py `_np_extract_fbank_features`,
// selected by the reference's configs/default_data_preprocessing_whisper.json:20-29): reflect-padded 400/160 STFT with a
// periodic Hann window, 400-point DFT (stored as complex64), power, 80 Slaney mel filters, log10(max(.,1e-10)), last frame
// dropped, clamp to (clip max - 8), (x+4)/4.  n_fft = 400 is not a power of two, so the transform is a direct float64 DFT
// against a 400-entry twiddle table in LDS (1 GFLOP per 30 s clip — noise next to the encoder's 0.35 TFLOP).
// The encoder itself (Conv1d x2 as implicit GEMMs, pre-LN MHA + FFN layers) runs on the shared kernels; this file only adds
// the (B,mel,T) -> channels-last bf16 transpose and the "+ embed_positions" step.
#include "common.hpp"

namespace {

constexpr int WN = 400, WHOP = 160, WBINS = 201;

struct WhArgs {
    const float* wave; long ldw; const int* num_samples; int n_samples;    // clips are zero-padded / cut to n_samples
    const double* window; const double* twiddle;                           // (400), (400,2) cos/sin(2 pi k/400)
    const double* mel_t; int nmel;                                          // (nmel, 201)
    float* out; int frames;                                                 // (B, frames, nmel) log10 mel (before clamp)
    int B;
};

__device__ __forceinline__ float sample_reflect(const float* w, int ns, int n_samples, int i) {
    // index into the reflect-padded, zero-extended clip: position i of pad(w_ext, 200, 'reflect'), w_ext has n_samples
    int j = i - WN / 2;
    if (j < 0) j = -j;
    if (j >= n_samples) j = 2 * (n_samples - 1) - j;
    return j < ns ? w[j] : 0.f;
}

__global__ __launch_bounds__(256) void whisper_logmel_kernel(WhArgs p) {
    __shared__ double xs[WN];
    __shared__ double2 tw[WN];
    __shared__ double pw[WBINS];
    const int tid = threadIdx.x;
    for (int i = tid; i < WN; i += 256) tw[i] = double2{p.twiddle[2 * i], p.twiddle[2 * i + 1]};
    const long total = (long)p.B * p.frames;
    for (long f = blockIdx.x; f < total; f += gridDim.x) {
        const int b = (int)(f / p.frames), t = (int)(f % p.frames);
        const float* w = p.wave + (long)b * p.ldw;
        const int ns = p.num_samples ? min(p.num_samples[b], p.n_samples) : p.n_samples;
        __syncthreads();
        for (int i = tid; i < WN; i += 256) xs[i] = (double)sample_reflect(w, ns, p.n_samples, t * WHOP + i) * p.window[i];
        __syncthreads();
        if (tid < WBINS) {
            double re = 0.0, im = 0.0;
            int idx = 0;
            for (int n = 0; n < WN; ++n) {
                const double2 c = tw[idx];
                re += xs[n] * c.x;
                im -= xs[n] * c.y;
                idx += tid;
                if (idx >= WN) idx -= WN;
            }
            const double r32 = (double)(float)re, i32 = (double)(float)im;
            const double mag = sqrt(r32 * r32 + i32 * i32);
            pw[tid] = mag * mag;
        }
        __syncthreads();
        if (tid < p.nmel) {
            const double* mt = p.mel_t + (long)tid * WBINS;
            double acc = 0.0;
            for (int k = 0; k < WBINS; ++k) acc += pw[k] * mt[k];
            p.out[((long)b * p.frames + t) * p.nmel + tid] = (float)log10(fmax(acc, 1e-10));
        }
    }
}

// per clip: m = max over (frames, nmel); out = (max(x, m - 8) + 4) / 4, written both as (B, nmel, frames) fp32 (the HF
// `input_features` layout) and as channels-last bf16 (B, frames, nmel) for the first conv
__global__ __launch_bounds__(256) void whisper_norm_kernel(const float* __restrict__ x, int frames, int nmel,
                                                            float* __restrict__ out_ft, bf16_t* __restrict__ out_cl) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (long)b * frames * nmel;
    float m = -INFINITY;
    for (int i = tid; i < frames * nmel; i += 256) m = fmaxf(m, xb[i]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int i = tid; i < frames * nmel; i += 256) {
        const int t = i / nmel, f = i % nmel;
        const float v = (fmaxf(xb[i], m - 8.0f) + 4.0f) / 4.0f;
        if (out_ft) out_ft[((long)b * nmel + f) * frames + t] = v;
        if (out_cl) out_cl[(long)b * frames * nmel + i] = f2bf(v);
    }
}

// (B, C, T) fp32 -> (B, T, C) bf16
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int Cn, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < Cn && t < T) ? x[((long)b * Cn + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (t < T && c < Cn) out[((long)b * T + t) * Cn + c] = f2bf(tile[tx][r]);
    }
}

// x[m, :] = float(a[m, :]) + pos[m % T, :]
__global__ __launch_bounds__(256) void add_pos_kernel(const bf16_t* __restrict__ a, const float* __restrict__ pos, float* __restrict__ x,
                                                       int M, int T, int d) {
    const int d4 = d >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)M * d4; i += (long)gridDim.x * 256) {
        const long m = i / d4; const int c = (int)(i % d4);
        const bf16x4 v = reinterpret_cast<const bf16x4*>(a + m * d)[c];
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos + (m % T) * d)[c];
        reinterpret_cast<f32x4*>(x + m * d)[c] = f32x4{bf2f(v[0]) + pe.x, bf2f(v[1]) + pe.y, bf2f(v[2]) + pe.z, bf2f(v[3]) + pe.w};
    }
}

}  // namespace

// wave (B, ldw) fp32; scratch (B, frames, nmel) fp32; outputs: features (B, nmel, frames) fp32 (nullable) and channels-last bf16 (nullable)
extern "C" int mi_whisper_logmel(const float* wave, long ldw, const int* num_samples, int n_samples, const double* window,
                                 const double* twiddle, const double* mel_t, int nmel, int B, float* scratch,
                                 float* out_features, void* out_cl_bf16, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || n_samples < WN || nmel <= 0 || nmel > 256) return MI_ERR_ARG;
    const int frames = n_samples / WHOP;                       // 1 + n_samples/160 frames, last one dropped
    WhArgs a{wave, ldw, num_samples, n_samples, window, twiddle, mel_t, nmel, scratch, frames, B};
    const long total = (long)B * frames;
    hipLaunchKernelGGL(whisper_logmel_kernel, dim3((unsigned)(total < 8192 ? total : 8192)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(whisper_norm_kernel, dim3(B), dim3(256), 0, stream, scratch, frames, nmel, out_features, (bf16_t*)out_cl_bf16);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_transpose_cast_bct_btc(const float* x, void* out_bf16, int B, int C, int T, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || C <= 0 || T <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, stream, x, (bf16_t*)out_bf16, B, C, T);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_add_positions(const void* a_bf16, const float* pos, float* x, int M, int T, int d, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || T <= 0 || d <= 0 || (d % 4)) return MI_ERR_ARG;
    const long total = (long)M * (d / 4);
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, stream,
                       (const bf16_t*)a_bf16, pos, x, M, T, d);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
