// Kaldi-style log-mel filterbank + CMVN on gfx950, float64 inside like the reference's numpy branch.
//
// Restates transformers' Speech2TextFeatureExtractor._extract_fbank_features (numpy branch,
// feature_extraction_speech_to_text.py:122-138 -> audio_utils.spectrogram :955-1016) as used by the
// reference's CustomFeatureExtractor (src/utilities/feature_extractors.py:51-61):
//   wave*2^15 -> frames 400/hop 160 (snip edges) -> per-frame DC removal -> pre-emphasis 0.97
//   -> povey window -> 512-pt FFT (result rounded to complex64) -> |.|^2 -> 80 Kaldi mel triangles
//   -> max(., 1.19e-7) -> log -> float32,
// and utterance_cmvn (:141-163) / global_normalize (feature_extractors.py:47-49).
// One wave per frame: the real 512-point transform as a complex 256-point radix-4 Stockham FFT (4 stages) + unpack, in wave-private LDS, twiddles/window/filters from
// host-built float64 tables (identical numbers to numpy's).  52 MFLOP per 10 s clip -> latency/LDS bound,
// not HBM bound (0.64 MB in, 0.32 MB out per clip).
#include "common.hpp"

namespace {

constexpr int FRAME = 400, HOP = 160, NFFT = 512, NBINS = 257;
constexpr int MEL_W = 640;                                              // LDS slots for the packed mel weights (>= 2 * NBINS + a triangle's slack)

struct FbankArgs {
    const float* wave; long ldw;         // (B, ldw) samples
    const int* num_samples;              // (B) valid samples per clip, or null -> N
    int N;
    const double* window;                // (400)
    const double* twiddle;               // (256, 2)  exp(-2*pi*i*k/512)
    const double* mel_t;                 // (nmel, 257) dense transposed filterbank
    const int* mel_lo; const int* mel_hi;   // (nmel) non-zero bin range [lo, hi)
    float* out; long ld_out_b; int T_out;   // (B, T_out, nmel): frames >= T_b are left untouched
    int B, nmel; double mel_floor; double preemph;
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NZ = NFFT / 2;                                          // the complex transform has 256 points: two 4-KiB buffers per wave, 36 KiB per block -> 4 blocks per CU
    double2* buf0 = reinterpret_cast<double2*>(smem) + wave * 2 * NZ;
    double2* buf1 = buf0 + NZ;
    double2* twl = reinterpret_cast<double2*>(smem) + 4 * 2 * NZ;         // [NFFT/2] twiddles, shared by the four waves (an LDS read instead of a global load per butterfly)
    for (int i = threadIdx.x; i < NFFT / 2; i += 256) twl[i] = double2{p.twiddle[2 * i], p.twiddle[2 * i + 1]};
    // the mel triangles' non-zero spans, packed back to back in LDS (every bin lies under at most two triangles: <= 2 * 257 weights): the filter loop of step 5 read its
    // weights from global memory, one dependent L2 round trip per bin of the longest span (~40) and frame — most of the kernel's time
    double* mwt = reinterpret_cast<double*>(twl + NFFT / 2);             // [MEL_W]
    int* moff = reinterpret_cast<int*>(mwt + MEL_W);                     // [nmel + 1] (nmel <= 127)
    int* mlo_s = moff + 128;                                            // [nmel]: the spans' first bins
    if ((int)threadIdx.x < p.nmel) { mlo_s[threadIdx.x] = p.mel_lo[threadIdx.x]; moff[threadIdx.x + 1] = max(p.mel_hi[threadIdx.x] - p.mel_lo[threadIdx.x], 0); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int f = 0; f < p.nmel; ++f) { const int n = moff[f + 1]; moff[f] = o; o += n; }
        moff[p.nmel] = o;
    }
    __syncthreads();
    const int mtot = moff[p.nmel];
    const bool mel_lds = mtot <= MEL_W;
    if (mel_lds)
        for (int i = threadIdx.x; i < mtot; i += 256) {                  // one load per thread: the filter of slot i by bisection of the offsets
            int lo = 0, hi = p.nmel - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (moff[mid] <= i) lo = mid; else hi = mid - 1; }
            mwt[i] = p.mel_t[(long)lo * NBINS + mlo_s[lo] + (i - moff[lo])];
        }
    __syncthreads();
    const long total = (long)p.B * p.T_out;
    for (long fidx = (long)blockIdx.x * 4 + wave; fidx < total; fidx += (long)gridDim.x * 4) {
        const int b = (int)(fidx / p.T_out), t = (int)(fidx % p.T_out);
        const int ns = p.num_samples ? p.num_samples[b] : p.N;
        const int nframes = ns >= FRAME ? 1 + (ns - FRAME) / HOP : 0;
        if (t >= nframes) continue;                         // wave-uniform
        const float* w = p.wave + (long)b * p.ldw + (long)t * HOP;
        // 1. load (x * 2^15 in float32, then float64), DC offset.  Lane owns the sample PAIRS (2m, 2m+1), m = lane + 64 i: the real 512-point
        //    transform is done as a complex 256-point FFT of z[m] = x[2m] + i x[2m+1] and unpacked afterwards (half the butterflies)
        double xa[4], xb[4], xp[4];
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n0 = 2 * (lane + 64 * i), n1 = n0 + 1;
            xa[i] = (n0 < FRAME) ? (double)(w[n0] * 32768.0f) : 0.0;
            xb[i] = (n1 < FRAME) ? (double)(w[n1] * 32768.0f) : 0.0;
            xp[i] = (n0 >= 1 && n0 < FRAME) ? (double)(w[n0 - 1] * 32768.0f) : 0.0;
            s += xa[i] + xb[i];
        }
        const double mean = wave_sum_d(s) / (double)FRAME;
        // 2. pre-emphasis on the DC-removed frame, window, into the FFT buffer (zero padded to 512 samples = 256 pairs)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = lane + 64 * i, n0 = 2 * m, n1 = n0 + 1;
            double v0 = 0.0, v1 = 0.0;
            if (n0 < FRAME) {
                const double c0 = xa[i] - mean;
                v0 = ((n0 == 0) ? c0 * (1.0 - p.preemph) : c0 - p.preemph * (xp[i] - mean)) * p.window[n0];
                if (n1 < FRAME) v1 = ((xb[i] - mean) - p.preemph * c0) * p.window[n1];
            }
            buf0[m] = double2{v0, v1};
        }
        wave_lds_sync();
        // 3. Stockham radix-4 DIF, 256 points = 4 stages, one butterfly per lane and stage.  w(j) = exp(-2 pi i j / 256) from the 512th-root table.
        auto w256 = [&](int j) { const double2 t = twl[2 * (j & 127)]; return (j & 128) ? double2{-t.x, -t.y} : t; };
        auto cmul = [](double2 a, double2 b) { return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; };
        double2* src = buf0;
        double2* dst = buf1;
        int nn = 256, st = 1;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
            const int n4 = nn >> 2;
            const int q = lane & (st - 1), pp = lane / st;         // butterfly `lane`: 0..63 = (n/4) * s
            const double2 a = src[q + st * pp], b = src[q + st * (pp + n4)], c = src[q + st * (pp + 2 * n4)], d = src[q + st * (pp + 3 * n4)];
            const double2 apc = {a.x + c.x, a.y + c.y}, amc = {a.x - c.x, a.y - c.y};
            const double2 bpd = {b.x + d.x, b.y + d.y}, bmd = {b.x - d.x, b.y - d.y};
            const int j = pp * st;                                 // exp(-2 pi i pp / nn) = w256(pp * st)
            dst[q + st * (4 * pp + 0)] = double2{apc.x + bpd.x, apc.y + bpd.y};
            dst[q + st * (4 * pp + 1)] = cmul(double2{amc.x + bmd.y, amc.y - bmd.x}, w256(j));          // a - i b - c + i d
            dst[q + st * (4 * pp + 2)] = cmul(double2{apc.x - bpd.x, apc.y - bpd.y}, w256(2 * j));
            dst[q + st * (4 * pp + 3)] = cmul(double2{amc.x - bmd.y, amc.y + bmd.x}, w256(3 * j));      // a + i b - c - i d
            wave_lds_sync();
            double2* tmp = src; src = dst; dst = tmp;
            nn = n4; st <<= 2;
        }
        // 4. unpack the real transform, X[k] = E[k] + W512^k O[k] with E = (Z[k] + conj Z[256-k]) / 2, O = (Z[k] - conj Z[256-k]) / (2i), and take the
        //    power of the complex64-rounded spectrum (kept in LDS: dst reused as 257 doubles)
        double* pw = reinterpret_cast<double*>(dst);
        double pk[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            pk[i] = 0.0;
            if (k < NBINS) {
                const double2 zk = src[k & 255], zc = src[(256 - k) & 255];
                const double2 e = {0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y)};
                const double2 o = {0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x)};              // (zk - conj zc) / (2i)
                const double2 wk = (k < 256) ? twl[k] : double2{-1.0, 0.0};
                const double2 ow = cmul(o, wk);
                const double re = (double)(float)(e.x + ow.x), im = (double)(float)(e.y + ow.y);
                pk[i] = re * re + im * im;                  // (|X|^2 directly: sqrt-then-square of the reference differs from it by one float64 rounding, 1e-16 relative; a float64
                                                            // square root is ~20 instructions of the ~850 per frame that bound this kernel — PMC: f64 VALU issue = its duration)
            }
        }
        wave_lds_sync();                                            // every lane has read Z before dst/src are reused for the powers
#pragma unroll
        for (int i = 0; i < 5; ++i) { const int k = lane + 64 * i; if (k < NBINS) pw[k] = pk[i]; }
        wave_lds_sync();
        // 5. mel filters, floor, log
        float* o = p.out + (long)b * p.ld_out_b + (long)t * p.nmel;
        for (int f = lane; f < p.nmel; f += 64) {
            const int lo = mlo_s[f], hi = lo + (moff[f + 1] - moff[f]);
            double acc = 0.0;
            if (mel_lds) {
                const double* mw = mwt + moff[f] - lo;
                for (int k = lo; k < hi; ++k) acc += pw[k] * mw[k];       // same products in the same order as the global-memory form
            } else {
                const double* mt = p.mel_t + (long)f * NBINS;
                for (int k = lo; k < hi; ++k) acc += pw[k] * mt[k];
            }
            o[f] = logf((float)fmax(acc, p.mel_floor));     // float32 log of the float64 energy: within 1 ulp (1e-6 absolute) of float32(log64(acc)); the float64 log was ~60 instructions per value
        }
        wave_lds_sync();
    }
}

// utterance CMVN in numpy's float32 evaluation order (axis-0 reductions accumulate row by row):
//   mean = sum_t x / n ; y = x - mean ; std = sqrt(sum_t (y - sum_t y / n)^2 / n) ; out = y / std ; padded frames -> pad
__global__ __launch_bounds__(128) void cmvn_kernel(float* x, long ld_b, const int* frames, int T, int nmel,
                                                    int norm_means, int norm_vars, float pad) {
    const int b = blockIdx.x, f = threadIdx.x;
    if (f >= nmel) return;
    float* xb = x + (long)b * ld_b;
    const int n = frames ? min(frames[b], T) : T;
    if (n > 0) {
        const float fn = (float)n;
        if (norm_means) {
            float s = 0.f;
            for (int t = 0; t < n; ++t) s += xb[(long)t * nmel + f];
            const float mean = s / fn;
            for (int t = 0; t < T; ++t) xb[(long)t * nmel + f] -= mean;      // np.subtract(x, mean): all rows
        }
        if (norm_vars) {
            float s = 0.f;
            for (int t = 0; t < n; ++t) s += xb[(long)t * nmel + f];
            const float m2 = s / fn;
            float q = 0.f;
            for (int t = 0; t < n; ++t) { const float d = xb[(long)t * nmel + f] - m2; q += d * d; }
            const float sd = sqrtf(q / fn);
            for (int t = 0; t < T; ++t) xb[(long)t * nmel + f] /= sd;
        }
    }
    for (int t = n; t < T; ++t) xb[(long)t * nmel + f] = pad;
}

// Fast form of cmvn_kernel: one block per (utterance, group of G mel bins); the (T x G) slab is staged in LDS, G
// threads run numpy's sequential float32 reductions out of LDS (bit-identical order), everybody writes the result.
template <int G>
__global__ __launch_bounds__(256) void cmvn_lds_kernel(float* x, long ld_b, const int* frames, int T, int nmel,
                                                        int norm_means, int norm_vars, float pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);          // [T][G]
    float* red = slab + (size_t)T * G;                     // [2][G] : mean, sd
    const int b = blockIdx.x, f0 = blockIdx.y * G, tid = threadIdx.x;
    float* xb = x + (long)b * ld_b;
    const int n = frames ? min(frames[b], T) : T;
    const bool vec4 = (G % 4) == 0 && (nmel % 4) == 0 && f0 + G <= nmel && (reinterpret_cast<uintptr_t>(xb) & 15) == 0;       // 16-B accesses: a quarter of the load / store instructions
    if (vec4) {
        for (int i = tid; i < T * (G / 4); i += 256) {
            const int t = i / (G / 4), g4 = i % (G / 4);
            reinterpret_cast<f32x4*>(slab)[i] = *reinterpret_cast<const f32x4*>(xb + (long)t * nmel + f0 + 4 * g4);
        }
    } else {
        for (int i = tid; i < T * G; i += 256) {
            const int t = i / G, g = i % G;
            slab[i] = (f0 + g < nmel) ? xb[(long)t * nmel + f0 + g] : 0.f;
        }
    }
    __syncthreads();
    if (tid < G) {
        // numpy's float32 reductions along the time axis are sequential per column, and so are these — but sixteen values are fetched from LDS ahead of the
        // sixteen dependent adds that consume them (one LDS round trip per 16 steps of the chain instead of one per step: a single partial wave has no other
        // way to hide it).  Same operations in the same order: bit-identical.
        constexpr int UN = 16;
        float mean = 0.f, sd = 1.f;
        const float fn = (float)max(n, 1);
        const float* col = slab + tid;
        if (n > 0 && norm_means) {
            float s = 0.f;
            int t = 0;
            for (; t + UN <= n; t += UN) {
                float v[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) v[j] = col[(t + j) * G];
#pragma unroll
                for (int j = 0; j < UN; ++j) s += v[j];
            }
            for (; t < n; ++t) s += col[t * G];
            mean = s / fn;
        }
        if (n > 0 && norm_vars) {
            float s = 0.f;
            int t = 0;
            for (; t + UN <= n; t += UN) {
                float v[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) v[j] = col[(t + j) * G];
#pragma unroll
                for (int j = 0; j < UN; ++j) s += v[j] - mean;
            }
            for (; t < n; ++t) s += col[t * G] - mean;
            const float m2 = s / fn;
            float q = 0.f;
            for (t = 0; t + UN <= n; t += UN) {
                float v[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) v[j] = col[(t + j) * G];
#pragma unroll
                for (int j = 0; j < UN; ++j) { const float d = (v[j] - mean) - m2; q += d * d; }
            }
            for (; t < n; ++t) { const float d = (col[t * G] - mean) - m2; q += d * d; }
            sd = sqrtf(q / fn);
        }
        red[tid] = mean; red[G + tid] = sd;
    }
    __syncthreads();
    if (vec4) {
        for (int i = tid; i < T * (G / 4); i += 256) {
            const int t = i / (G / 4), g4 = i % (G / 4);
            f32x4 v = {pad, pad, pad, pad};
            if (t < n) {
                v = reinterpret_cast<const f32x4*>(slab)[i];
                if (norm_means) v = f32x4{v.x - red[4 * g4], v.y - red[4 * g4 + 1], v.z - red[4 * g4 + 2], v.w - red[4 * g4 + 3]};
                if (norm_vars) v = f32x4{v.x / red[G + 4 * g4], v.y / red[G + 4 * g4 + 1], v.z / red[G + 4 * g4 + 2], v.w / red[G + 4 * g4 + 3]};
            }
            *reinterpret_cast<f32x4*>(xb + (long)t * nmel + f0 + 4 * g4) = v;
        }
        return;
    }
    for (int i = tid; i < T * G; i += 256) {
        const int t = i / G, g = i % G;
        if (f0 + g >= nmel) continue;
        float v = pad;
        if (t < n) {
            v = slab[i];
            if (norm_means) v -= red[g];
            if (norm_vars) v /= red[G + g];
        }
        xb[(long)t * nmel + f0 + g] = v;
    }
}

__global__ void global_norm_kernel(float* x, long total, int nmel, const float* means, const float* stds) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % nmel);
        x[i] = (x[i] - means[f]) / stds[f];
    }
}

// ---- the dataloader-side `default_transform` of the reference on the device (src/utilities/callbacks.py:108-118): `audio_object_stripper` = np.trim_zeros
// (leading AND trailing samples equal to zero, src/utilities/data_utils.py:173-177), then zero-padding up to `min_len` (8000) samples.
// trim_bounds: one block per clip finds the first / last non-zero sample; trim_copy: the stripped clip, left-aligned, zeros behind it.
__global__ __launch_bounds__(256) void trim_bounds_kernel(const float* wave, long ldw, const int* num_samples, int N, int min_len, int* first, int* valid, int* eff) {
    __shared__ int slo[4], shi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ns = num_samples ? min(num_samples[b], N) : N;
    const float* w = wave + (long)b * ldw;
    int lo = 0x7fffffff, hi = -1;
    for (int i = tid; i < ns; i += 256)
        if (w[i] != 0.f) { lo = min(lo, i); hi = max(hi, i); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((tid & 63) == 0) { slo[tid >> 6] = lo; shi[tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lo = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
        hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
        const int v = hi < 0 ? 0 : hi - lo + 1;             // an all-zero clip strips to nothing (and is then padded to min_len zeros)
        first[b] = hi < 0 ? 0 : lo;
        valid[b] = v;
        eff[b] = max(v, min_len);
    }
}
__global__ __launch_bounds__(256) void trim_copy_kernel(const float* wave, long ldw, const int* first, const int* valid, float* out, long ldo, int N_out) {
    const int b = blockIdx.y;
    const int f = first[b], v = valid[b];
    const float* w = wave + (long)b * ldw + f;
    float* o = out + (long)b * ldo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N_out; i += gridDim.x * 256) o[i] = i < v ? w[i] : 0.f;
}

}  // namespace

// wave (B, ldw) f32 in [-1,1]; out (B, T_out, nmel) f32 (frames beyond a clip's own count are not written).
extern "C" int mi_fbank_f64(const float* wave, long ldw, const int* num_samples, int N, const double* window,
                            const double* twiddle, const double* mel_t, const int* mel_lo, const int* mel_hi,
                            float* out, int T_out, int B, int nmel, double mel_floor, double preemph, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || N < FRAME || T_out <= 0 || nmel <= 0) return MI_ERR_ARG;
    FbankArgs a{wave, ldw, num_samples, N, window, twiddle, mel_t, mel_lo, mel_hi, out, (long)T_out * nmel, T_out,
                B, nmel, mel_floor, preemph};
    const long total = (long)B * T_out;
    const int grid = (int)((total + 3) / 4 < 768 ? (total + 3) / 4 : 768);       // three blocks per CU: a block's set-up (twiddles, packed mel weights) serves ~10 frames per wave
    if (nmel > 127) return MI_ERR_ARG;
    const size_t lds = (4 * 2 * (NFFT / 2) + NFFT / 2) * sizeof(double2) + MEL_W * sizeof(double) + 256 * sizeof(int);
    hipLaunchKernelGGL(fbank_kernel, dim3(grid), dim3(256), lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_cmvn_utterance(float* x, const int* frames, int B, int T, int nmel, int norm_means, int norm_vars,
                                 float pad, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || nmel <= 0 || nmel > 128) return MI_ERR_ARG;
    if ((size_t)T * 16 * 4 + 128 <= 140 * 1024) {
        hipLaunchKernelGGL(cmvn_lds_kernel<16>, dim3(B, cdiv(nmel, 16)), dim3(256), (size_t)T * 16 * 4 + 128, stream, x, (long)T * nmel,
                           frames, T, nmel, norm_means, norm_vars, pad);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    if ((size_t)T * 4 * 4 + 32 <= 140 * 1024) {
        hipLaunchKernelGGL(cmvn_lds_kernel<4>, dim3(B, cdiv(nmel, 4)), dim3(256), (size_t)T * 4 * 4 + 32, stream, x, (long)T * nmel,
                           frames, T, nmel, norm_means, norm_vars, pad);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    hipLaunchKernelGGL(cmvn_kernel, dim3(B), dim3(128), 0, stream, x, (long)T * nmel, frames, T, nmel, norm_means, norm_vars, pad);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_cmvn_global(float* x, long total, int nmel, const float* means, const float* stds, hipStream_t stream) {
    MI_ENTER();
    if (total <= 0 || nmel <= 0) return MI_ERR_ARG;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(global_norm_kernel, dim3(grid), dim3(256), 0, stream, x, total, nmel, means, stds);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// out (B, ldo >= N_out) <- clip b stripped of leading / trailing zero samples, zeros behind; eff_len[b] = max(stripped length, min_len) = the sample count the
// feature extractor sees; first / valid (B) = offset and length of the stripped clip.  N_out >= max(N, min_len).
extern "C" int mi_trim_zeros_pad_f32(const float* wave, long ldw, const int* num_samples, int N, int B, int min_len, float* out, long ldo, int N_out,
                                     int* first, int* valid, int* eff_len, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || N <= 0 || min_len < 0 || N_out < N || N_out < min_len || ldo < N_out) return MI_ERR_ARG;
    hipLaunchKernelGGL(trim_bounds_kernel, dim3(B), dim3(256), 0, stream, wave, ldw, num_samples, N, min_len, first, valid, eff_len);
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(trim_copy_kernel, dim3(cdiv(N_out, 256 * 8) < 256 ? cdiv(N_out, 256 * 8) : 256, B), dim3(256), 0, stream, wave, ldw, first, valid, out, ldo, N_out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
