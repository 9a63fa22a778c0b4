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
constexpr int MELW = 640;                                  // LDS slots for the packed mel weights (>= 2 * WBINS + a triangle's slack)

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

// Two frames per wave, eight per block.  400 = 8 x 50: with S_j(k) = sum over n = j (mod 8) of x[n] e^{-2 pi i n k / 400} (k = 0..25, one lane each; lanes 0-25 the wave's
// first frame, 32-57 its second) the bins are 8-point transforms of the eight partial sums,
//     X[k + 50 m] = sum_j S_j(k) w8^{j m},      X[50 m - k] = sum_j conj(S_j(k)) w8^{j m},      w8 = e^{-2 pi i / 8}
// (x is real: S_j(-k) = conj S_j(k)), m = 0..3 of the first and 1..4 of the second cover 0..200.  A sample costs two FMAs and one rotation of its residue class's twiddle
// (eight rotators per lane, each advanced by e^{-2 pi i 8 k / 400}; float64: 50 steps grow the error to ~1e-14).  The four-fold form of rounds 1-3 ran ONE frame per
// wave on 51 lanes: the same 2 400 float64 instructions per wave for half the frames (PMC: the kernel is bound by their issue: 146 M per 16 x 30 s batch).
// The mel filters are applied over their non-zero span only, and the per-clip maximum the normalisation needs is taken here (one global atomic per frame).
__global__ __launch_bounds__(256) void whisper_logmel_kernel(WhArgs p, float* __restrict__ clipmax) {
    constexpr int FB = 8;                                   // frames per block
    __shared__ double xs[FB][WN];
    __shared__ double2 tw[WN];
    __shared__ double pw[FB][WBINS + 3];
    __shared__ int mlo[256], mhi[256];
    __shared__ float smax[FB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < WN; i += 256) tw[i] = double2{p.twiddle[2 * i], p.twiddle[2 * i + 1]};
    // non-zero span of every mel filter (triangles: contiguous).  Thread = bin, loop over the filters: one coalesced load per filter and thread, the span's ends from
    // wave ballots (a thread per filter walking its 201 bins was 201 dependent L2 round trips at the start of every block).
    mlo[tid] = WBINS; mhi[tid] = 0;
    __syncthreads();
    for (int m0 = 0; m0 < p.nmel; m0 += 8) {                 // eight filters' loads in flight at a time (one at a time the loop is 80 L2 round trips per block)
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (tid < WBINS && m0 + j < p.nmel) ? p.mel_t[(long)(m0 + j) * WBINS + tid] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(v[j] != 0.0);
            if (lane == 0 && bal != 0ull) {
                atomicMin(&mlo[m0 + j], wave * 64 + (int)__builtin_ctzll(bal));
                atomicMax(&mhi[m0 + j], wave * 64 + 64 - (int)__builtin_clzll(bal));
            }
        }
    }
    // the filters' non-zero weights packed back to back in LDS (every bin lies under at most two triangles): the filter loop below read them from global memory — a chain
    // of L2 round trips per (frame, filter) task, three tasks per thread and block iteration — which, not the transform, was most of the kernel's time
    __shared__ double mw[MELW];
    __shared__ int moff[257];
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < p.nmel; ++m) { moff[m] = o; o += max(mhi[m] - mlo[m], 0); }
        moff[p.nmel] = o;
    }
    __syncthreads();
    const int mtot = moff[p.nmel];
    const bool mel_lds = mtot <= MELW;
    if (mel_lds)
        for (int i = tid; i < mtot; i += 256) {
            int lo = 0, hi = p.nmel - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (moff[mid] <= i) lo = mid; else hi = mid - 1; }
            mw[i] = p.mel_t[(long)lo * WBINS + mlo[lo] + (i - moff[lo])];
        }
    const long total = (long)p.B * p.frames;
    // A block owns a CONTIGUOUS run of frames — one clip, rarely two — and keeps that clip's maximum to itself until the clip changes or the block ends: ~800 global
    // atomics per launch.  One `atomic max` per frame on the clip's single word (rounds 1-3; 48 000 per batch on 16 addresses, 3 000 serialised L2 round trips each) was
    // what the kernel's 650-700 us actually were: halving its arithmetic, its set-up or its mel loads changed nothing.
    const long per = ((total + gridDim.x - 1) / gridDim.x + FB - 1) / FB * FB;
    const long fbeg = (long)blockIdx.x * per, fend = fbeg + per < total ? fbeg + per : total;
    int cur_clip = -1;                                      // (thread 0)
    float cur_max = -INFINITY;
    for (long f0 = fbeg; f0 < fend; f0 += FB) {
        __syncthreads();
        if (tid < FB) smax[tid] = -INFINITY;
        // the eight windowed frames: 32 threads per frame
        {
            const int fr = tid >> 5, l32 = tid & 31;
            const long f = f0 + fr;
            if (f < fend) {
                const int b = (int)(f / p.frames), t = (int)(f % p.frames);
                const float* w = p.wave + (long)b * p.ldw;
                const int ns = p.num_samples ? min(p.num_samples[b], p.n_samples) : p.n_samples;
                for (int i = l32; i < WN; i += 32) xs[fr][i] = (double)sample_reflect(w, ns, p.n_samples, t * WHOP + i) * p.window[i];
            }
        }
        __syncthreads();
        const int half = lane >> 5, k = lane & 31, fr = 2 * wave + half;
        if (f0 + fr < fend && k <= 25) {
            double sr[8], si[8];                              // S_j = sr[j] + i si[j]
            double2 c[8];
            {
                int idx = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) { sr[j] = 0.0; si[j] = 0.0; c[j] = tw[idx]; idx += k; }       // e^{+2 pi i j k / 400}: j k <= 175
            }
            const double2 w8k = tw[8 * k];                    // 8 k <= 200
            const double* xf = xs[fr];
            for (int n = 0; n < WN; n += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double x = xf[n + j];
                    sr[j] = fma(x, c[j].x, sr[j]);
                    si[j] = fma(-x, c[j].y, si[j]);
                    const double nx = fma(c[j].x, w8k.x, -(c[j].y * w8k.y)), ny = fma(c[j].x, w8k.y, c[j].y * w8k.x);
                    c[j] = double2{nx, ny};
                }
            }
            // 8-point transforms of (S_j) and (conj S_j):  w8^q = (cr[q], ci[q]), q = j m mod 8
            const double h = 0.70710678118654752440;
            const double cr[8] = {1.0, h, 0.0, -h, -1.0, -h, 0.0, h}, ci[8] = {0.0, -h, -1.0, -h, 0.0, h, 1.0, h};
            auto put = [&](int bin, double r, double i_) {
                const double r32 = (double)(float)r, i32 = (double)(float)i_;          // the reference stores the STFT as complex64
                pw[fr][bin] = r32 * r32 + i32 * i32;           // |X|^2 directly (the reference's abs-then-square differs by one float64 rounding)
            };
#pragma unroll
            for (int m = 0; m <= 4; ++m) {
                double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;  // a = sum S_j w^{jm}, b = sum conj(S_j) w^{jm}
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = (j * m) & 7;
                    ar += sr[j] * cr[q] - si[j] * ci[q];  ai += sr[j] * ci[q] + si[j] * cr[q];
                    br += sr[j] * cr[q] + si[j] * ci[q];  bi += sr[j] * ci[q] - si[j] * cr[q];
                }
                if (m <= 3) put(k + 50 * m, ar, ai);
                if (m >= 1 && k >= 1) put(50 * m - k, br, bi);
                if (m == 4 && k == 0) put(200, ar, ai);
            }
        }
        __syncthreads();
        for (int q = tid; q < FB * p.nmel; q += 256) {
            const int fq = q / p.nmel, m = q - fq * p.nmel;
            const long ff = f0 + fq;
            if (ff >= fend) continue;
            const double* mt = mel_lds ? mw + moff[m] - mlo[m] : p.mel_t + (long)m * WBINS;
            double acc = 0.0;
            for (int kk = mlo[m]; kk < mhi[m]; ++kk) acc = fma(pw[fq][kk], mt[kk], acc);
            const float v = log10f((float)fmax(acc, 1e-10));           // float32 log10 of the float64 energy: 1 ulp (< 1e-6) from float32(log10_64(acc))
            p.out[ff * p.nmel + m] = v;
            atomic_max_f32(&smax[fq], v);
        }
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < FB && f0 + i < fend; ++i) {
                const int b = (int)((f0 + i) / p.frames);
                if (b != cur_clip) {
                    if (cur_clip >= 0) atomic_max_f32(clipmax + cur_clip, cur_max);
                    cur_clip = b; cur_max = -INFINITY;
                }
                cur_max = fmaxf(cur_max, smax[i]);
            }
    }
    if (tid == 0 && cur_clip >= 0) atomic_max_f32(clipmax + cur_clip, cur_max);
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
    const long nblk = (total + 7) / 8;
    hipLaunchKernelGGL(whisper_logmel_kernel, dim3((unsigned)(nblk < 768 ? nblk : 768)), dim3(256), 0, stream, a, clipmax);      // three blocks per CU, resident for the whole launch: a block's set-up (twiddles, mel spans, packed weights) runs once per CU slot, not once per 8 frames
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
