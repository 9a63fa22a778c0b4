#ifndef VAR
#define VAR 0
#endif
// debug copy of dwconv31_kernel<true> (huggingface_asr_amd/csrc/conv.hip) with LDS-tile / register-window dumps
#include "common.hpp"
namespace {
struct DwArgs {
    const bf16_t* in; long ld_in;        // conv input rows (gate half for CSGU, m for MERGE)
    const bf16_t* mul; long ld_mul;      // CSGU: x_r
    const float* stats;                  // CSGU: (M,2) mean/rstd of the gate rows
    const float* gamma; const float* beta;   // CSGU LayerNorm affine (C)
    const float* w; const float* bias;   // (C, K) taps, (C)
    bf16_t* out; long ld_out;
    int B, T, C, K, pad_left, dilation, act; float* dbg_tile; float* dbg_tile2;    // act: 0 identity, 1 gelu, 2 relu, 3 silu
};
// Fast form for the reference's kernel size 31, dilation 1.  A block owns 64 channels x 128 time steps: the (128+30) x 64
// input tile (16-B global loads, LayerNorm applied on the way in), the gate operand x_r and the result tile all live in
// LDS, every thread keeps a 62-sample window + the 31 taps of its channel in registers and produces 32 consecutive
// outputs (93 LDS reads per 992 FMAs), and the result leaves as 16-B-per-lane rows.
constexpr int DWF_K = 31, DWF_TT = 128, DWF_CT = 64, DWF_ROWS = DWF_TT + DWF_K - 1, DWF_PER = DWF_TT / 4;

template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv31_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                          // [DWF_ROWS][64] fp32
    float* sw = tile + DWF_ROWS * DWF_CT;                                  // [31][64]
    bf16_t* io = reinterpret_cast<bf16_t*>(sw + DWF_K * DWF_CT);           // [128][64] bf16: x_r in, result out
    const int c0 = blockIdx.x * DWF_CT, t0 = blockIdx.y * DWF_TT, b = blockIdx.z;
    const int tid = threadIdx.x;
    for (int i = tid; i < DWF_K * DWF_CT; i += 256) {
        const int k = i / DWF_CT, cc = i % DWF_CT;
        sw[i] = p.w[(long)(c0 + cc) * DWF_K + k];
    }
#if VAR == 2
    const f32x4 hg0 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + (tid & 7) * 8), hg1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + (tid & 7) * 8 + 4);
    const f32x4 hb0 = *reinterpret_cast<const f32x4*>(p.beta + c0 + (tid & 7) * 8), hb1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + (tid & 7) * 8 + 4);
#endif
    for (int id = tid; id < DWF_ROWS * (DWF_CT / 8); id += 256) {
        const int r = id >> 3, ch = id & 7;
        const int t = t0 - p.pad_left + r;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < p.T) {
            const long row = (long)b * p.T + t;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.in + row * p.ld_in + c0 + ch * 8);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bf2f(v[j]);
            if (CSGU) {
                const float mu = p.stats[2 * row], rs = p.stats[2 * row + 1];
#if VAR == 2
                const f32x4 g0 = hg0, g1 = hg1, b0 = hb0, b1 = hb1;
#else
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch * 8 + 4);
#endif
#if VAR == 1
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_s_sleep(4);
                __builtin_amdgcn_sched_barrier(0);
#endif
#if VAR == 4
                // round 2: is the failure a LOST WRITE-BACK of the packed FMA?  The destination pair is a register pair of its own, pre-set to a sentinel
                // (777 / 778); the FMA is written by hand so that the destination is neither a source nor the accumulator input.  A corrupted tile word
                // equal to the sentinel = the low-half result of v_pk_fma_f32 never reached the register file.
                f32x2 xn[4], gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}}, bb[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}}, dd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xn[q] = f32x2{(f[2 * q] - mu) * rs, (f[2 * q + 1] - mu) * rs};
                    dd[q] = f32x2{777.0f, 778.0f};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(dd[q]) : "v"(xn[q]), "v"(gg[q]), "v"(bb[q]));
                lo = f32x4{dd[0].x, dd[0].y, dd[1].x, dd[1].y};
                hi = f32x4{dd[2].x, dd[2].y, dd[3].x, dd[3].y};
#elif VAR == 6
                // packed normalisation (compiler: v_pk_mul_f32), SCALAR affine step written by hand (v_fma_f32 x 8): which packed instruction is the victim?
                float xn6[8], o6[8];
                const float gg6[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb6[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) xn6[q] = (f[q] - mu) * rs;
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(o6[q]) : "v"(xn6[q]), "v"(gg6[q]), "v"(bb6[q]));
                lo = f32x4{o6[0], o6[1], o6[2], o6[3]};
                hi = f32x4{o6[4], o6[5], o6[6], o6[7]};
#elif VAR == 7
                // SCALAR normalisation written by hand (v_sub_f32 / v_mul_f32), packed affine step left to the compiler (v_pk_fma_f32)
                float xn7[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float d;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(f[q]), "v"(mu));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(xn7[q]) : "v"(d), "v"(rs));
                }
                lo = f32x4{xn7[0] * g0.x + b0.x, xn7[1] * g0.y + b0.y, xn7[2] * g0.z + b0.z, xn7[3] * g0.w + b0.w};
                hi = f32x4{xn7[4] * g1.x + b1.x, xn7[5] * g1.y + b1.y, xn7[6] * g1.z + b1.z, xn7[7] * g1.w + b1.w};
#else
                lo = f32x4{(f[0] - mu) * rs * g0.x + b0.x, (f[1] - mu) * rs * g0.y + b0.y, (f[2] - mu) * rs * g0.z + b0.z, (f[3] - mu) * rs * g0.w + b0.w};
                hi = f32x4{(f[4] - mu) * rs * g1.x + b1.x, (f[5] - mu) * rs * g1.y + b1.y, (f[6] - mu) * rs * g1.z + b1.z, (f[7] - mu) * rs * g1.w + b1.w};
#endif
            } else {
                lo = f32x4{f[0], f[1], f[2], f[3]};
                hi = f32x4{f[4], f[5], f[6], f[7]};
            }
        }
        *reinterpret_cast<f32x4*>(tile + r * DWF_CT + ch * 8) = lo;
        *reinterpret_cast<f32x4*>(tile + r * DWF_CT + ch * 8 + 4) = hi;
    }
    if (CSGU) {
        for (int id = tid; id < DWF_TT * (DWF_CT / 8); id += 256) {
            const int r = id >> 3, ch = id & 7;
            const int t = t0 + r;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t < p.T) v = *reinterpret_cast<const bf16x8*>(p.mul + ((long)b * p.T + t) * p.ld_mul + c0 + ch * 8);
            *reinterpret_cast<bf16x8*>(io + r * DWF_CT + ch * 8) = v;
        }
    }
    __syncthreads();
    if (p.dbg_tile) {
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = tid; i < DWF_ROWS * DWF_CT; i += 256) p.dbg_tile[blk * DWF_ROWS * DWF_CT + i] = tile[i];
    }
    const int tx = tid & 63, ty = tid >> 6;
    const int c = c0 + tx;
    float wk[DWF_K];
#pragma unroll
    for (int k = 0; k < DWF_K; ++k) wk[k] = sw[k * DWF_CT + tx];
    float win[DWF_PER + DWF_K - 1];
#pragma unroll
    for (int i = 0; i < DWF_PER + DWF_K - 1; ++i) win[i] = tile[(ty * DWF_PER + i) * DWF_CT + tx];

    if (p.dbg_tile2) {
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = 0; i < DWF_PER + DWF_K - 1; ++i) p.dbg_tile2[(blk * 4 + ty) * 62 * 64 + i * 64 + tx] = win[i];
    }
    const float bias = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int j = 0; j < DWF_PER; ++j) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < DWF_K; ++k) acc = fmaf(wk[k], win[j + k], acc);
        const int rl = ty * DWF_PER + j;
        float o;
        if (CSGU) {
            if (p.act == 1) acc = gelu_erf(acc);
            else if (p.act == 2) acc = fmaxf(acc, 0.f);
            else if (p.act == 3) acc = acc / (1.f + __expf(-acc));
            o = bf2f(io[rl * DWF_CT + tx]) * acc;
        } else {
            o = win[j + (DWF_K - 1) / 2] + acc;
        }
        io[rl * DWF_CT + tx] = f2bf(o);
    }
    __syncthreads();
    for (int id = tid; id < DWF_TT * (DWF_CT / 8); id += 256) {
        const int r = id >> 3, ch = id & 7;
        const int t = t0 + r;
        if (t < p.T)
            *reinterpret_cast<bf16x8*>(p.out + ((long)b * p.T + t) * p.ld_out + c0 + ch * 8) = *reinterpret_cast<const bf16x8*>(io + r * DWF_CT + ch * 8);
    }
}
}
extern "C" int probe3_launch(const void* in, long ld_in, const float* stats, const float* gamma, const float* beta, const void* mul, long ld_mul, const float* w, const float* bias,
                             void* out, long ld_out, int B, int T, int C, float* dbg_tile, float* dbg_tile2, hipStream_t stream) {
    DwArgs a{(const bf16_t*)in, ld_in, (const bf16_t*)mul, ld_mul, stats, gamma, beta, w, bias, (bf16_t*)out, ld_out, B, T, C, 31, 15, 1, 0, dbg_tile, dbg_tile2};
    dim3 gridf(C / DWF_CT, (T + DWF_TT - 1) / DWF_TT, B);
    const size_t ldsf = (size_t)(DWF_ROWS * DWF_CT + DWF_K * DWF_CT) * sizeof(float) + (size_t)DWF_TT * DWF_CT * sizeof(bf16_t);
    hipLaunchKernelGGL(dwconv31_kernel<true>, gridf, dim3(256), ldsf, stream, a);
    return (int)hipGetLastError();
}
