// Feature-level SpecAugment on the device (reference src/augmentations/spec_aug.py, the ESPnet implementation the
// recipes run inside dataloader workers: configs/default_data_preprocessing2d.json:36-58): bicubic time warp (:104-137) + frequency
// masks + time masks (:40-98) in ONE pass over the (B, T, F) batch.  The random parameters are drawn on the host in the reference's own
// order (huggingface_asr_amd/augment.py), so a seeded run reproduces the reference's augmentation.
//
// Time warp = torch.nn.functional.interpolate(mode="bicubic", align_corners=False) of the segments [0, center) -> [0, warped) and
// [center, len) -> [warped, len): along time a 4-tap cubic convolution (A = -0.75), src = (dst + 0.5) * in/out - 0.5, taps clamped to the
// segment; along frequency the scale is 1, where the cubic kernel degenerates to the identity.
#include "common.hpp"

namespace {

__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, u = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// params (B, stride) int32: [len, has_warp, center, warped, nf pairs (pos, width), nt pairs (pos, width)]
__global__ __launch_bounds__(256) void specaug_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int F,
                                                       const int* __restrict__ params, int stride, int nf, int nt, float pad_value) {
    const long total = (long)B * T * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i % F);
        const int t = (int)((i / F) % T);
        const int b = (int)(i / ((long)F * T));
        const int* p = params + (long)b * stride;
        const int len = p[0];
        float v;
        if (t >= len) {
            v = pad_value;
        } else if (!p[1]) {
            v = x[i];
        } else {
            const int center = p[2], warped = p[3];
            int s0, in_n, d, out_n;
            if (t < warped) { s0 = 0; in_n = center; d = t; out_n = warped; }
            else { s0 = center; in_n = len - center; d = t - warped; out_n = len - warped; }
            const float scale = (float)in_n / (float)out_n;
            const float src = scale * ((float)d + 0.5f) - 0.5f;
            const float fl = floorf(src);
            const int ix = (int)fl;
            float w[4];
            cubic_w(src - fl, w);
            const float* xb = x + ((long)b * T + s0) * F + f;
            v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int j = ix - 1 + k;
                j = j < 0 ? 0 : (j > in_n - 1 ? in_n - 1 : j);
                v = fmaf(w[k], xb[(long)j * F], v);
            }
        }
        bool masked = false;
        for (int m = 0; m < nf; ++m) { const int pos = p[4 + 2 * m], wd = p[5 + 2 * m]; masked |= (f >= pos && f < pos + wd); }
        for (int m = 0; m < nt; ++m) { const int pos = p[4 + 2 * nf + 2 * m], wd = p[5 + 2 * nf + 2 * m]; masked |= (t >= pos && t < pos + wd); }
        out[i] = masked ? 0.f : v;
    }
}

}  // namespace

// x, out (B, T, F) f32 (out != x); params (B, 4 + 2 nf + 2 nt) int32 on the device (layout above); pad_value fills t >= len
extern "C" int mi_specaug_f32(const float* x, float* out, int B, int T, int F, const int* params, int nf, int nt, float pad_value, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || F <= 0 || nf < 0 || nt < 0 || x == out) return MI_ERR_ARG;
    const long g = ((long)B * T * F + 255) / 256;
    hipLaunchKernelGGL(specaug_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, st, x, out, B, T, F, params, 4 + 2 * nf + 2 * nt, nf, nt, pad_value);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
