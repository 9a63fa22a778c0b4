// Whisper front end + small glue kernels for the Whisper-style encoder (BASELINE.json config 4) on gfx950.
//
// Replaces transformers' WhisperFeatureExtractor numpy path (feature_extraction_whisper.py `_np_extract_fbank_features`,
// selected by the reference's configs/default_data_preprocessing_whisper.json:20-29): reflect-padded 400/160 STFT with a
// periodic Hann window, 400-point DFT (stored as complex64), power, 80 Slaney mel filters, log10(max(.,1e-10)), last frame
// dropped, clamp to (clip max - 8), (x+4)/4.  n_fft = 400 is not a power of two, so the transform is a direct float64 DFT
// against a 400-entry twiddle table in LDS, four bins per lane through the factor-4 symmetry of 400 (whisper_logmel_kernel).
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

// float max through integer atomics (values of either sign; the target starts at -inf)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// A wave per frame, four frames per block.  400 = 4 x 100, so the bins k, 100-k, 100+k and 200-k (k = 0..50) share their twiddles up to
// factors (-i)^n and a conjugation: lane k accumulates S_j = sum over n = j (mod 4) of x[n] e^{-2 pi i nk/400} (two FMAs per sample) and
//   X[k] = S0 + S1 + S2 + S3,  X[200-k] = S0* - S1* + S2* - S3*,  X[100+k] = S0 - i S1 - S2 + i S3,  X[100-k] = S0* - i S1* - S2* + i S3*
// — a quarter of the multiply-adds and of the twiddle gathers of one-lane-per-bin.  The mel filters are applied over their non-zero span only,
// and the per-clip maximum the normalisation needs is taken here (one global atomic per frame).
__global__ __launch_bounds__(256) void whisper_logmel_kernel(WhArgs p, float* __restrict__ clipmax) {
    __shared__ double xs[4][WN];
    __shared__ double2 tw[WN];
    __shared__ double pw[4][WBINS + 3];
    __shared__ int mlo[256], mhi[256];
    __shared__ float smax[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < WN; i += 256) tw[i] = double2{p.twiddle[2 * i], p.twiddle[2 * i + 1]};
    // non-zero span of every mel filter (triangles: contiguous).  Thread = bin, loop over the filters: one coalesced load per filter and thread, the span's ends from
    // wave ballots.  (Rounds 1-3: thread = filter walking its 201 bins — 201 DEPENDENT L2 round trips at the start of every block, ~150 us x 3 block rounds: most of
    // the kernel's 704 us.)
    for (int i = tid; i < 256; i += 256) { mlo[i] = WBINS; mhi[i] = 0; }
    __syncthreads();
    for (int m = 0; m < p.nmel; ++m) {
        const bool nz = tid < WBINS && p.mel_t[(long)m * WBINS + tid] != 0.0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(nz);
        if (lane == 0 && bal != 0ull) {
            atomicMin(&mlo[m], wave * 64 + (int)__builtin_ctzll(bal));
            atomicMax(&mhi[m], wave * 64 + 64 - (int)__builtin_clzll(bal));
        }
    }
    const long total = (long)p.B * p.frames;
    for (long f0 = (long)blockIdx.x * 4; f0 < total; f0 += (long)gridDim.x * 4) {
        const long f = f0 + wave;
        const bool valid = f < total;
        const int b = valid ? (int)(f / p.frames) : 0, t = valid ? (int)(f % p.frames) : 0;
        __syncthreads();
        if (tid < 4) smax[tid] = -INFINITY;
        if (valid) {
            const float* w = p.wave + (long)b * p.ldw;
            const int ns = p.num_samples ? min(p.num_samples[b], p.n_samples) : p.n_samples;
            for (int i = lane; i < WN; i += 64) xs[wave][i] = (double)sample_reflect(w, ns, p.n_samples, t * WHOP + i) * p.window[i];
        }
        __syncthreads();
        if (valid && lane <= 50) {
            const int k = lane;
            double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};      // S_j = re[j] + i im[j]
            // The twiddle of sample n = 4 m + j is e^{2 pi i (4 m + j) k / 400}: four independent rotators c_j, started from the table at j k and advanced by the table's
            // entry at 4 k (float64 complex multiply: 100 steps grow the error to ~1e-14, the test's tolerance on log10 is 3e-5).  Gathering tw[n k mod 400] per lane
            // and step, as rounds 1-3 did, is a 51-lane LDS access whose bank is (n k) mod 16: up to 51-way conflicts — the kernel took 704 us per 16 x 30 s batch,
            // 7.5 % of config 4's step, most of it LDS replays.  The samples stay a broadcast read.
            double2 c[4];
            {
                int idx = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { c[j] = tw[idx]; idx += k; if (idx >= WN) idx -= WN; }
            }
            const double2 w4 = tw[(4 * k) % WN];
            for (int n = 0; n < WN; n += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double x = xs[wave][n + j];
                    re[j] = fma(x, c[j].x, re[j]);
                    im[j] = fma(-x, c[j].y, im[j]);
                    const double nx = fma(c[j].x, w4.x, -(c[j].y * w4.y)), ny = fma(c[j].x, w4.y, c[j].y * w4.x);
                    c[j] = double2{nx, ny};
                }
            }
            auto put = [&](int bin, double r, double i_) {
                const double r32 = (double)(float)r, i32 = (double)(float)i_;          // the reference stores the STFT as complex64
                pw[wave][bin] = r32 * r32 + i32 * i32;           // |X|^2 directly (the reference's abs-then-square differs by one float64 rounding)
            };
            put(k, (re[0] + re[2]) + (re[1] + re[3]), (im[0] + im[2]) + (im[1] + im[3]));
            put(200 - k, (re[0] + re[2]) - (re[1] + re[3]), -((im[0] + im[2]) - (im[1] + im[3])));
            // -i S1 = (im1, -re1),  +i S3 = (-im3, re3)
            put(100 + k, (re[0] - re[2]) + (im[1] - im[3]), (im[0] - im[2]) - (re[1] - re[3]));
            // S* terms: S0* - i S1* - S2* + i S3*;  -i (re1 - i im1) = (-im1, -re1),  +i (re3 - i im3) = (im3, re3)
            put(100 - k, (re[0] - re[2]) - (im[1] - im[3]), -(im[0] - im[2]) - (re[1] - re[3]));
        }
        __syncthreads();
        for (int q = tid; q < 4 * p.nmel; q += 256) {
            const int fr = q / p.nmel, m = q - fr * p.nmel;
            const long ff = f0 + fr;
            if (ff >= total) continue;
            const double* mt = p.mel_t + (long)m * WBINS;
            double acc = 0.0;
            for (int kk = mlo[m]; kk < mhi[m]; ++kk) acc = fma(pw[fr][kk], mt[kk], acc);
            const float v = log10f((float)fmax(acc, 1e-10));           // float32 log10 of the float64 energy: 1 ulp (< 1e-6) from float32(log10_64(acc))
            p.out[ff * p.nmel + m] = v;
            atomic_max_f32(&smax[fr], v);
        }
        __syncthreads();
        if (tid < 4 && f0 + tid < total) atomic_max_f32(clipmax + (int)((f0 + tid) / p.frames), smax[tid]);
    }
}

// out = (max(x, m - 8) + 4) / 4 with m = the clip's maximum (taken by the log-mel kernel), written both as (B, nmel, frames) fp32 (the HF
// `input_features` layout, optional) and as channels-last bf16 (B, frames, nmel) for the first conv
__global__ __launch_bounds__(256) void whisper_norm_kernel(const float* __restrict__ x, const float* __restrict__ clipmax, int frames, int nmel, int B,
                                                            float* __restrict__ out_ft, bf16_t* __restrict__ out_cl) {
    const long per = (long)frames * nmel, total = per * B;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per);
        const int r = (int)(i - (long)b * per);
        const int t = r / nmel, f = r - t * nmel;
        const float v = (fmaxf(x[i], clipmax[b] - 8.0f) + 4.0f) / 4.0f;
        if (out_ft) out_ft[((long)b * nmel + f) * frames + t] = v;
        if (out_cl) out_cl[i] = f2bf(v);
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

// wave (B, ldw) fp32; scratch (B * frames * nmel + B) fp32; outputs: features (B, nmel, frames) fp32 (nullable) and channels-last bf16 (nullable)
extern "C" int mi_whisper_logmel(const float* wave, long ldw, const int* num_samples, int n_samples, const double* window,
                                 const double* twiddle, const double* mel_t, int nmel, int B, float* scratch,
                                 float* out_features, void* out_cl_bf16, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || n_samples < WN || nmel <= 0 || nmel > 256) return MI_ERR_ARG;
    const int frames = n_samples / WHOP;                       // 1 + n_samples/160 frames, last one dropped
    WhArgs a{wave, ldw, num_samples, n_samples, window, twiddle, mel_t, nmel, scratch, frames, B};
    const long total = (long)B * frames;
    float* clipmax = scratch + total * nmel;                   // B floats behind the (B, frames, nmel) scratch
    if (hipMemsetD32Async((hipDeviceptr_t)clipmax, 0xFF800000u /* -inf */, B, stream) != hipSuccess) return MI_ERR_LAUNCH;
    const long nblk = (total + 3) / 4;
    hipLaunchKernelGGL(whisper_logmel_kernel, dim3((unsigned)(nblk < 4096 ? nblk : 4096)), dim3(256), 0, stream, a, clipmax);
    const long nel = total * nmel;
    hipLaunchKernelGGL(whisper_norm_kernel, dim3((unsigned)((nel + 255) / 256 < 8192 ? (nel + 255) / 256 : 8192)), dim3(256), 0, stream, scratch, clipmax, frames, nmel, B,
                       out_features, (bf16_t*)out_cl_bf16);
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
