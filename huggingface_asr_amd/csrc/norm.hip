// LayerNorm chain kernel (wave-per-row, fp32 statistics) for gfx950.
//
// The E-Branchformer layer has LayerNorms back to back around the fp32 residual stream
// (reference e_branchformer.py:233,236,242,257,261; tf wav2vec2_conformer :328-333,:707): this kernel
// runs, per row and in one pass over HBM,
//   stage 0  x = (t >= len[b]) ? 0 : x              (tf:662-665 "make sure padded tokens output 0")
//   stage 1  y = g1 ? LN(x; g1, b1) : x ; optionally stored fp32 (in place allowed)
//   stage 2  a = LN(y; ga, ba) -> bf16 and/or fp32 ;  b = LN(y; gb, bb) -> bf16   (same statistics)
// so "final_layer_norm of layer i + ff1 norm of layer i+1" or "self_attn_layer_norm + cgMLP_layer_norm"
// cost one read of the row.  One wave per row, row held in registers (d <= 2048, d % 4 == 0).
#include "common.hpp"

namespace {

constexpr int MAXV = 8;   // float4 per lane -> d <= 64*4*8 = 2048

struct LnArgs {
    const float* x; long ldx;
    const int* lengths; int T;          // optional row mask (row = b*T + t)
    const float* g1; const float* b1; float* y32; long ldy;
    const float* ga; const float* ba; bf16_t* outa; long lda; float* outa32; long lda32;
    const float* gb; const float* bb; bf16_t* outb; long ldb;
    int M, d; float eps1, eps2;
    bf16_t* yb; long ldyb; float* ystats;       // LayerNorm-fold producer (mi_layernorm_fold): bf16 copy of y and its (sum, sumsq) in slot 0 of the row's 32-float statistics record
};

template <int NV>
__global__ __launch_bounds__(256) void ln_chain_kernel(LnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int d4 = p.d >> 2;
    f32x4 v[NV];
    bool masked = false;
    if (p.lengths) {
        const int b = row / p.T, t = row - b * p.T;
        masked = t >= p.lengths[b];
    }
    const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (long)row * p.ldx);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < d4 && !masked) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // padding lanes (c >= d4) hold zeros: they must not enter the variance -> handle by masking below
    auto nvalid = [&](int i) { return (lane + 64 * i) < d4; };

    float mean, rstd;
    if (p.g1) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        mean = wave_sum(s) / p.d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (nvalid(i)) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += a * a + b * b + c * c + e * e;
            }
        rstd = rsqrtf(wave_sum(q) / p.d + p.eps1);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (nvalid(i)) {
                const int c = lane + 64 * i;
                const f32x4 g = reinterpret_cast<const f32x4*>(p.g1)[c], b = reinterpret_cast<const f32x4*>(p.b1)[c];
                v[i] = (v[i] - mean) * rstd * g + b;
            }
    }
    if (p.y32) {
        f32x4* yr = reinterpret_cast<f32x4*>(p.y32 + (long)row * p.ldy);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (nvalid(i)) yr[lane + 64 * i] = v[i];
    }
    if (p.yb) {
        float s = 0.f, qq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (nvalid(i)) {
                reinterpret_cast<bf16x4*>(p.yb + (long)row * p.ldyb)[lane + 64 * i] = bf16x4{f2bf(v[i].x), f2bf(v[i].y), f2bf(v[i].z), f2bf(v[i].w)};
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                qq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        s = wave_sum(s); qq = wave_sum(qq);
        if (lane == 0) *reinterpret_cast<f32x2*>(p.ystats + (long)row * 32) = f32x2{s, qq};
    }
    if (!p.ga) return;
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        mean = wave_sum(s) / p.d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (nvalid(i)) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += a * a + b * b + c * c + e * e;
            }
        rstd = rsqrtf(wave_sum(q) / p.d + p.eps2);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (nvalid(i)) {
            const int c = lane + 64 * i;
            const f32x4 n = (v[i] - mean) * rstd;
            const f32x4 a = n * reinterpret_cast<const f32x4*>(p.ga)[c] + reinterpret_cast<const f32x4*>(p.ba)[c];
            if (p.outa) {
                bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
                reinterpret_cast<bf16x4*>(p.outa + (long)row * p.lda)[c] = o;
            }
            if (p.outa32) reinterpret_cast<f32x4*>(p.outa32 + (long)row * p.lda32)[c] = a;
            if (p.gb) {
                const f32x4 b = n * reinterpret_cast<const f32x4*>(p.gb)[c] + reinterpret_cast<const f32x4*>(p.bb)[c];
                bf16x4 o = {f2bf(b.x), f2bf(b.y), f2bf(b.z), f2bf(b.w)};
                reinterpret_cast<bf16x4*>(p.outb + (long)row * p.ldb)[c] = o;
            }
        }
}

// rotary embedding on the INPUT of the Q/K projections (tf wav2vec2_conformer :509-526, applied at
// e_branchformer.py:88-93): out[.., h, c] = x*cos[t][c] + rot(x)*sin[t][c], rot = cat(-x[hd/2:], x[:hd/2]).
__global__ __launch_bounds__(256) void rotary_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo,
                                                      const float* __restrict__ cs, const float* __restrict__ sn,
                                                      int M, int T, int H, int hd) {
    const int half = hd >> 1;
    const long total = (long)M * H * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % half);
        const int h = (int)((i / half) % H);
        const long row = i / ((long)half * H);
        const int t = (int)(row % T);
        const float x1 = bf2f(x[row * ldx + h * hd + c]), x2 = bf2f(x[row * ldx + h * hd + half + c]);
        const float c1 = cs[t * hd + c], s1 = sn[t * hd + c], c2 = cs[t * hd + half + c], s2 = sn[t * hd + half + c];
        out[row * ldo + h * hd + c] = f2bf(x1 * c1 - x2 * s1);
        out[row * ldo + h * hd + half + c] = f2bf(x2 * c2 + x1 * s2);
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo, int M, int d) {
    const int d4 = d >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)M * d4; i += (long)gridDim.x * 256) {
        const long m = i / d4; const int c = (int)(i % d4);
        const f32x4 v = reinterpret_cast<const f32x4*>(x + m * ldx)[c];
        reinterpret_cast<bf16x4*>(out + m * ldo)[c] = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
    }
}

}  // namespace

// fp32 (M,d) -> bf16 (M,d) round-to-nearest-even (GEMM A operands made from fp32 states: encoder output -> decoder)
extern "C" int mi_cast_f32_bf16(const float* x, long ldx, void* out, long ldo, int M, int d, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || (d % 4) || (ldx % 4) || (ldo % 4)) return MI_ERR_ARG;
    const long total = (long)M * (d / 4);
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, stream,
                       x, ldx, (bf16_t*)out, ldo, M, d);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_rotary_bf16(const void* x, long ldx, void* out, long ldo, const float* cos_t, const float* sin_t,
                              int M, int T, int H, int hd, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || T <= 0 || H <= 0 || hd <= 0 || (hd & 1)) return MI_ERR_ARG;
    const long total = (long)M * H * (hd / 2);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rotary_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, cos_t, sin_t, M, T, H, hd);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_layernorm_chain(const float* x, long ldx, const int* lengths, int T,
                                  const float* g1, const float* b1, float eps1, float* y32, long ldy,
                                  const float* ga, const float* ba, float eps2, void* outa_bf16, long lda,
                                  float* outa_f32, long lda32,
                                  const float* gb, const float* bb, void* outb_bf16, long ldb,
                                  int M, int d, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || (d % 4) != 0 || d > 64 * 4 * MAXV) return MI_ERR_ARG;
    if ((ldx % 4) || (y32 && (ldy % 4)) || (outa_bf16 && (lda % 4)) || (outa_f32 && (lda32 % 4)) || (outb_bf16 && (ldb % 4)))
        return MI_ERR_ARG;
    LnArgs p{x, ldx, lengths, T, g1, b1, y32, ldy, ga, ba, (bf16_t*)outa_bf16, lda, outa_f32, lda32,
             gb, bb, (bf16_t*)outb_bf16, ldb, M, d, eps1, eps2, nullptr, 0, nullptr};
    const int nv = cdiv(d / 4, 64);
    dim3 grid(cdiv(M, 4)), block(256);
    if (nv <= 1) hipLaunchKernelGGL(ln_chain_kernel<1>, grid, block, 0, stream, p);
    else if (nv <= 2) hipLaunchKernelGGL(ln_chain_kernel<2>, grid, block, 0, stream, p);
    else if (nv <= 4) hipLaunchKernelGGL(ln_chain_kernel<4>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(ln_chain_kernel<8>, grid, block, 0, stream, p);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// LayerNorm-fold producer (encoder.hip `ln_fold` path): y = g1 ? LN(mask(x); g1, b1) : mask(x) stored three ways in one pass — fp32 (the residual stream, in place allowed),
// bf16 (the A operand of the GEMMs that fold the NEXT LayerNorm into their epilogue) and the row's (sum, sum of squares) in pair 0 of its 32-float statistics record
// (`ln_npart` = 1 for mi_gemm_lnfold_bf16).
extern "C" int mi_layernorm_fold(const float* x, long ldx, const int* lengths, int T, const float* g1, const float* b1, float eps1, float* y32, long ldy,
                                 void* yb_bf16, long ldyb, float* stats, int M, int d, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || (d % 4) != 0 || d > 64 * 4 * MAXV || !yb_bf16 || !stats) return MI_ERR_ARG;
    if ((ldx % 4) || (y32 && (ldy % 4)) || (ldyb % 4) || (reinterpret_cast<uintptr_t>(stats) & 7)) return MI_ERR_ARG;
    LnArgs p{x, ldx, lengths, T, g1, b1, y32, ldy, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, eps1, eps1, (bf16_t*)yb_bf16, ldyb, stats};
    const int nv = cdiv(d / 4, 64);
    dim3 grid(cdiv(M, 4)), block(256);
    if (nv <= 1) hipLaunchKernelGGL(ln_chain_kernel<1>, grid, block, 0, stream, p);
    else if (nv <= 2) hipLaunchKernelGGL(ln_chain_kernel<2>, grid, block, 0, stream, p);
    else if (nv <= 4) hipLaunchKernelGGL(ln_chain_kernel<4>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(ln_chain_kernel<8>, grid, block, 0, stream, p);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
