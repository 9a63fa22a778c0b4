// Speed perturbation on the device: the polyphase sinc resampler behind torchaudio.transforms.SpeedPerturbation, the first
// step of the reference's training pre-processing (configs/default_data_preprocessing2d.json:3-19), for a whole (B, N) waveform batch.
//
//   out[b][j*nw + i] = sum_k kernel[i][k] * x[b][j*orig + k - width]        (x = 0 outside [0, N)),  kernel (nw, 2*width + orig) fp32
//
// i.e. torch.nn.functional.conv1d(pad(x, (width, width + orig)), kernel, stride = orig) with the nw phases interleaved
// (torchaudio functional.py `_apply_sinc_resample_kernel`); the kernel table is built on the host in float64 as torchaudio does.
// HBM-bound: every input sample is read (kw / orig ~ 2.5 times, from L1/L2) and every output written once; a thread owns one output sample.
#include "common.hpp"

namespace {

constexpr int SP_MAX_TAPS = 96, SP_MAX_PHASES = 32;

__global__ __launch_bounds__(256) void speed_resample_kernel(const float* __restrict__ x, long ld, int N, int orig, int nw, int width, int kw,
                                                              const float* __restrict__ kernel, float* __restrict__ out, long ld_out, int N_out, int B) {
    __shared__ float sk[SP_MAX_PHASES * SP_MAX_TAPS];
    for (int i = threadIdx.x; i < nw * kw; i += 256) sk[i] = kernel[i];
    __syncthreads();
    const long total = (long)B * N_out;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / N_out), n = (int)(idx - (long)b * N_out);
        const int j = n / nw, i = n - j * nw;
        const float* xb = x + (long)b * ld;
        const float* kr = sk + i * kw;
        const int s0 = j * orig - width;
        float acc = 0.f;
        for (int k = 0; k < kw; ++k) {
            const int s = s0 + k;
            if (s >= 0 && s < N) acc = fmaf(kr[k], xb[s], acc);
        }
        out[(long)b * ld_out + n] = acc;
    }
}

__global__ void speed_lengths_kernel(const int* __restrict__ len, int B, int src, int tgt, int N_out, int* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long v = ((long)len[b] * tgt + src - 1) / src;                      // ceil(len * target / source)
    out[b] = (int)(v < N_out ? v : N_out);
}

}  // namespace

// wave (B, N) fp32 rows `ld` apart -> out (B, N_out) rows `ld_out` apart, N_out = ceil(nw * N / orig); lengths / out_lengths (B) int32 or null.
extern "C" int mi_speed_resample_f32(const float* wave, long ld, const int* lengths, int B, int N, int orig, int nw, const float* kernel, int width,
                                     float* out, long ld_out, int N_out, int* out_lengths, hipStream_t stream) {
    MI_ENTER();
    const int kw = 2 * width + orig;
    if (B <= 0 || N <= 0 || orig <= 0 || nw <= 0 || width < 0 || nw > SP_MAX_PHASES || kw > SP_MAX_TAPS || nw * kw > SP_MAX_PHASES * SP_MAX_TAPS) return MI_ERR_ARG;
    if (N_out != (int)(((long)nw * N + orig - 1) / orig) || ld < N || ld_out < N_out) return MI_ERR_ARG;
    const long total = (long)B * N_out;
    const long nb = (total + 255) / 256;
    hipLaunchKernelGGL(speed_resample_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, wave, ld, N, orig, nw, width, kw, kernel, out, ld_out, N_out, B);
    MI_CHECK_LAUNCH();
    if (lengths && out_lengths) {
        hipLaunchKernelGGL(speed_lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, lengths, B, orig, nw, N_out, out_lengths);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}
