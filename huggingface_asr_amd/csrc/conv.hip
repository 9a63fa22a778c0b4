// HBM-bound convolution kernels of the E-Branchformer path (gfx950).
//
//  * conv2d_first: Conv2d(1 -> C, KxK, stride s) + GELU over the padded (B,T,F) log-mel layout
//    (reference extractors.py:71-96 first layer, :111 `input_values[:, None]`), output channels-last
//    (B, T1, F1, C) bf16 so that the second conv is a K-contiguous implicit GEMM (gemm_bf16.hip).
//  * dwconv_time: depthwise Conv1d(k<=31) along time on (B,T,C) bf16, two fused forms:
//      CSGU  (e_branchformer.py:184-204): out = x_r * act(conv(LN(x_g)) + b)   [LN applied on the fly
//             from per-row mean/rstd, so the normalised gate never round-trips HBM]
//      MERGE (e_branchformer.py:296-299): out = m + conv(m) + b
//  * row_stats: per-row mean / rstd of the gate half for the CSGU LayerNorm (fp32).
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ conv2d_first
// one thread = one output position x 8 consecutive channels; 32 threads cover C=256 (generic C%8==0).  General geometry: (KH, KW) taps,
// strides (st, sf), pads (pad_t, pad_f) — the gate conv of GatedConv2dShared is (12,3) / (8,2) / (4,1) (extractors.py:41-47).
// ACT: 1 = GELU (the layer's activation), 0 = raw pre-activation (the operands of mi_gated_act_bf16).
template <int ACT>
__global__ __launch_bounds__(256) void conv2d_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                            int B, int T, int F, int C, int KH, int KW, int st, int sf, int pad_t, int pad_f,
                                                            int T1, int F1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sw = reinterpret_cast<float*>(smem);          // [KH*KW][C]  (tap-major so 8 channels are contiguous)
    float* sb = sw + KH * KW * C;                        // [C]
    for (int i = threadIdx.x; i < KH * KW * C; i += blockDim.x) {
        const int c = i % C, tap = i / C;
        sw[i] = w[c * KH * KW + tap];
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) sb[i] = bias[i];
    __syncthreads();
    const int cg = C >> 3;                               // channel groups of 8
    const long total = (long)B * T1 * F1 * cg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        const long pos = idx / cg;
        const int f1 = (int)(pos % F1);
        const int t1 = (int)((pos / F1) % T1);
        const int b = (int)(pos / ((long)F1 * T1));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = sb[g * 8 + j];
        for (int kh = 0; kh < KH; ++kh) {
            const int t = t1 * st - pad_t + kh;
            if (t < 0 || t >= T) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int f = f1 * sf - pad_f + kw;
                if (f < 0 || f >= F) continue;
                const float xv = x[((long)b * T + t) * F + f];
                const float* wp = sw + (kh * KW + kw) * C + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wp[j], acc[j]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(ACT ? gelu_erf(acc[j]) : acc[j]);
        *reinterpret_cast<bf16x8*>(out + pos * C + g * 8) = o;
    }
}

// 3x3 fast form: a thread owns 8 output channels (72 taps + 8 biases in registers) and walks output positions.
__global__ __launch_bounds__(256) void conv2d_first3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             int B, int T, int F, int C, int stride, int pad_t, int pad_f,
                                                             int T1, int F1) {
    const int cg = C >> 3, ppb = 256 / cg;
    const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
    float wr[9][8], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias[g * 8 + j];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wr[tap][j] = w[(g * 8 + j) * 9 + tap];
    }
    const int total = B * T1 * F1;                       // < 2^31 (checked by the launcher): 32-bit index arithmetic, the 64-bit divisions cost more than the conv
    for (int pos = blockIdx.x * ppb + pl; pos < total; pos += gridDim.x * ppb) {
        const int bt = pos / F1;
        const int f1 = pos - bt * F1;
        const int b = bt / T1;
        const int t1 = bt - b * T1;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = br[j];
        const float* xb = x + (long)b * T * F;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = t1 * stride - pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = f1 * stride - pad_f + kw;
                const float xv = (t >= 0 && t < T && f >= 0 && f < F) ? xb[(long)t * F + f] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[kh * 3 + kw][j], acc[j]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(gelu_erf(acc[j]));
        *reinterpret_cast<bf16x8*>(out + (long)pos * C + g * 8) = o;
    }
}

// The same layer on the matrix cores (round 4; C = 256).  pre[ch][pos] = sum_tap w[ch][tap] x[pos][tap] + b[ch] is a GEMM with K = 9: one 32-position tile x 32 channels is
// ONE v_mfma_f32_32x32x16_bf16 — but x, w and b are fp32 here (the un-fused kernel above multiplies them in fp32), so each factor is split into two bf16 parts
// (hi = bf16(v), lo = bf16(v - hi)) and the K = 32 slots of TWO MFMAs carry   x_hi w_hi (9) | x_lo w_hi (9) | x_hi w_lo (9) | 1 b_hi | 1 b_lo | 0 0 0:
// the result differs from the fp32 FMA chain by the dropped x_lo w_lo terms (2^-16 relative) — far inside the bf16 rounding of the output.
// A wave owns 32 consecutive output positions x all 256 channels (16 MFMAs, 128 accumulators: rows = channels, so four consecutive registers are four consecutive
// channels), applies GELU, packs to bf16 and passes the 16-KiB tile through a wave-private, XOR-swizzled LDS region so that it leaves as whole 512-B rows (1 KiB per
// store instruction).  Measured and not kept: the next tile's nine features requested one tile ahead (114.5 vs 100.7 us: the loads are L2 hits; the extra live registers
// and the waits the compiler places cost more than the round trip two waves per SIMD leave exposed).  Why the matrix cores: the VALU form spends 38 lane-instructions per output (72 FMAs + 72 for the GELU + address arithmetic per 8 channels; PMC: 97 M wave
// instructions per launch, 181 us at BASELINE config 2 for a 328-MB write that takes 70 us); here the GELU and the pack are all that is left on the VALU (13 per output).
__device__ __forceinline__ bf16_t c1_slot(const bf16_t (&hi)[9], const bf16_t (&lo_)[9], bf16_t one_hi, bf16_t one_lo, int s, bool lo_in_middle) {
    // slots 0-8: hi | 9-17: (x operand: lo, w operand: hi) | 18-26: (x: hi, w: lo) | 27, 28: (x: 1, 1; w: b_hi, b_lo) | 29-31: 0
    if (s < 9) return hi[s];
    if (s < 18) return lo_in_middle ? lo_[s - 9] : hi[s - 9];
    if (s < 27) return lo_in_middle ? hi[s - 18] : lo_[s - 18];
    if (s == 27) return one_hi;
    if (s == 28) return one_lo;
    return (bf16_t)0.f;
}
__global__ __launch_bounds__(256) void conv2d_first3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                  int B, int T, int F, int stride, int pad_t, int pad_f, int T1, int F1) {
    constexpr int C = 256, NG = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    char* stage = smem + wave * (32 * C * 2);              // [32 positions][512 B], 16-B chunk c of row p at chunk c ^ p
    // ---- the weight operand: row = channel 32 g + l31, k = slots 16 j + 8 h + i
    bf16x8 wa[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ch = 32 * g + l31;
        bf16_t whi[9], wlo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { const float v = w[ch * 9 + t]; whi[t] = f2bf(v); wlo[t] = f2bf(v - bf2f(whi[t])); }
        const float bv = bias[ch];
        const bf16_t bhi = f2bf(bv), blo = f2bf(bv - bf2f(bhi));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16_t s0 = c1_slot(whi, wlo, bhi, blo, 16 * j + i, false), s1 = c1_slot(whi, wlo, bhi, blo, 16 * j + 8 + i, false);
                wa[g][j][i] = h ? s1 : s0;
            }
    }
    const int total = B * T1 * F1;                          // < 2^31 - 2^20 (launcher)
    const int ntiles = (total + 31) >> 5;
    const bf16_t one = f2bf(1.f), zero = f2bf(0.f);
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int m0 = tile << 5;
        const int pos = min(m0 + l31, total - 1);
        const int bt = pos / F1, f1 = pos - bt * F1, b = bt / T1, t1 = bt - b * T1;
        const float* xb = x + (long)b * T * F;
        bf16_t xhi[9], xlo[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = t1 * stride - pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = f1 * stride - pad_f + kw;
                const float v = (t >= 0 && t < T && f >= 0 && f < F) ? xb[(long)t * F + f] : 0.f;
                xhi[kh * 3 + kw] = f2bf(v); xlo[kh * 3 + kw] = f2bf(v - bf2f(xhi[kh * 3 + kw]));
            }
        }
        bf16x8 xb_[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16_t s0 = c1_slot(xhi, xlo, one, one, 16 * j + i, true), s1 = c1_slot(xhi, xlo, one, one, 16 * j + 8 + i, true);
                xb_[j][i] = h ? s1 : s0;
            }
        // two halves of four channel groups: 64 accumulators live at a time (all eight at once put the kernel at 300 registers = one wave per SIMD)
#pragma unroll
        for (int gh = 0; gh < NG; gh += 4) {
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[gh + g][0], xb_[0], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[gh + g][1], xb_[1], acc[g], 0, 0, 0);
            }
            // lane = position l31; registers 4 q .. 4 q + 3 of group g = channels 32 g + 8 q + 4 h .. + 3  ->  8 B of chunk (4 g + q) of row l31
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x4 o = {f2bf(gelu_erf(acc[g][4 * q])), f2bf(gelu_erf(acc[g][4 * q + 1])), f2bf(gelu_erf(acc[g][4 * q + 2])), f2bf(gelu_erf(acc[g][4 * q + 3]))};
                    *reinterpret_cast<bf16x4*>(stage + l31 * 512 + (((4 * (gh + g) + q) ^ l31) << 4) + h * 8) = o;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = 2 * i + h;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + p * 512 + ((l31 ^ p) << 4));
            if (m0 + p < total) *reinterpret_cast<uint4*>(out + (long)(m0 + p) * C + l31 * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the next tile overwrites the region
    }
}

// GatedConv2d (extractors.py:23-32) as the first layer, 3x3: out = GELU((conv(x) + b) * sigmoid(gate(x) + bg)) — two filter banks over the same nine
// input samples.  A thread owns 4 output channels (2 x 36 taps + 8 biases in registers) and walks output positions.
__global__ __launch_bounds__(256) void conv2d_first3_gated_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                   const float* __restrict__ gw, const float* __restrict__ gbias, bf16_t* __restrict__ out,
                                                                   int B, int T, int F, int C, int stride, int pad_t, int pad_f, int T1, int F1) {
    const int cg = C >> 2, ppb = 256 / cg;
    const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
    float wr[9][4], gr[9][4], br[4], gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        br[j] = bias[g * 4 + j]; gb[j] = gbias[g * 4 + j];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) { wr[tap][j] = w[(g * 4 + j) * 9 + tap]; gr[tap][j] = gw[(g * 4 + j) * 9 + tap]; }
    }
    const int total = B * T1 * F1;                       // < 2^31 (checked by the launcher)
    if (pl < ppb)
    for (int pos = blockIdx.x * ppb + pl; pos < total; pos += gridDim.x * ppb) {
        const int bt = pos / F1;
        const int f1 = pos - bt * F1;
        const int b = bt / T1;
        const int t1 = bt - b * T1;
        float acc[4], gacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = br[j]; gacc[j] = gb[j]; }
        const float* xb = x + (long)b * T * F;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = t1 * stride - pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = f1 * stride - pad_f + kw;
                const float xv = (t >= 0 && t < T && f >= 0 && f < F) ? xb[(long)t * F + f] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] = fmaf(xv, wr[kh * 3 + kw][j], acc[j]); gacc[j] = fmaf(xv, gr[kh * 3 + kw][j], gacc[j]); }
            }
        }
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(gelu_erf(acc[j] * sigmoid_f(gacc[j])));
        *reinterpret_cast<bf16x4*>(out + (long)pos * C + g * 4) = o;
    }
}

// out[(b,t,f), c] = GELU(z[(b,t,f), zc(c)] * sigmoid(g[(b, t / share, f), gc(c)])): the product + activation of the context-aware Conv2d layers
// (extractors.py:31-32 share = 1; :49-54 share = 4: one gate row per four output time steps) applied to raw conv outputs.
// `blk`: the conv / gate columns of one row are interleaved in blocks of `blk` channels — zc(c) = (c / blk) * 2 blk + c % blk, gc(c) = zc(c) + blk —
// when both come out of ONE stacked GEMM (z == g, the packing of the fused implicit-GEMM epilogue); blk = 0: two separate tensors, plain columns.
__global__ __launch_bounds__(256) void gated_act_kernel(const bf16_t* __restrict__ z, long ldz, const bf16_t* __restrict__ g, long ldg, bf16_t* __restrict__ out, long ldo,
                                                         int B, int T, int Fq, int C, int share, int blk) {
    const int c8 = C >> 3;
    const long total = (long)B * T * Fq * c8;
    const int Tg = T / share;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c8) * 8;
        const long row = i / c8;
        const int f = (int)(row % Fq);
        const long bt = row / Fq;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const long grow = ((long)b * Tg + t / share) * Fq + f;
        const int zc = blk ? (cc / blk) * 2 * blk + cc % blk : cc;
        const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + row * ldz + zc);
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + grow * ldg + (blk ? zc + blk : cc));
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(gelu_erf(bf2f(zv[j]) * sigmoid_f(bf2f(gv[j]))));
        *reinterpret_cast<bf16x8*>(out + row * ldo + cc) = o;
    }
}

// ------------------------------------------------------------------------------------------------ row_stats
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, long ldx, int d, float eps,
                                                         float* __restrict__ stats, int M) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + (long)row * ldx);
    const int d8 = d >> 3;
    float s = 0.f, q = 0.f;
    for (int c = lane; c < d8; c += 64) {
        const bf16x8 v = xr[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f(v[j]);
    }
    const float mean = wave_sum(s) / d;
    for (int c = lane; c < d8; c += 64) {
        const bf16x8 v = xr[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float a = bf2f(v[j]) - mean; q += a * a; }
    }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// ------------------------------------------------------------------------------------------------ dwconv_time
constexpr int DW_TT = 64;     // output time steps per block
constexpr int DW_CT = 64;     // channels per block
constexpr int DW_KMAX = 31;

struct DwArgs {
    const bf16_t* in; long ld_in;        // conv input rows (gate half for CSGU, m for MERGE)
    const bf16_t* mul; long ld_mul;      // CSGU: x_r
    const float* stats;                  // CSGU: (M,2) mean/rstd of the gate rows
    const float* gamma; const float* beta;   // CSGU LayerNorm affine (C)
    const float* w; const float* bias;   // (C, K) taps, (C)
    bf16_t* out; long ld_out;
    int B, T, C, K, pad_left, dilation, act;    // act: 0 identity, 1 gelu, 2 relu, 3 silu
};

template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv_time_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (p.K - 1) * p.dilation;
    const int rows = DW_TT + halo;
    float* tile = reinterpret_cast<float*>(smem);        // [rows][DW_CT]
    float* sw = tile + rows * DW_CT;                     // [K][DW_CT]
    const int c0 = blockIdx.x * DW_CT, t0 = blockIdx.y * DW_TT, b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = c0 + tx;
    const bool cok = c < p.C;
    for (int i = threadIdx.x; i < p.K * DW_CT; i += 256) {
        const int k = i / DW_CT, cc = i % DW_CT;
        sw[i] = (c0 + cc < p.C) ? p.w[(long)(c0 + cc) * p.K + k] : 0.f;
    }
    float g = 1.f, be = 0.f;
    if (CSGU && cok) { g = p.gamma[c]; be = p.beta[c]; }
    for (int r = ty; r < rows; r += 4) {
        const int t = t0 - p.pad_left + r;
        float v = 0.f;                                   // zero padding is applied to the conv INPUT (post-LN)
        if (cok && t >= 0 && t < p.T) {
            const long row = (long)b * p.T + t;
            v = bf2f(p.in[row * p.ld_in + c]);
            if (CSGU) v = (v - p.stats[2 * row]) * p.stats[2 * row + 1] * g + be;
        }
        tile[r * DW_CT + tx] = v;
    }
    __syncthreads();
    if (!cok) return;
    const float bias = p.bias ? p.bias[c] : 0.f;
    // each thread: 16 consecutive outputs of one channel
#pragma unroll 4
    for (int j = 0; j < DW_TT / 4; ++j) {
        const int tl = ty * (DW_TT / 4) + j;
        const int t = t0 + tl;
        if (t >= p.T) break;
        float acc = bias;
        for (int k = 0; k < p.K; ++k) acc = fmaf(sw[k * DW_CT + tx], tile[(tl + k * p.dilation) * DW_CT + tx], acc);
        const long row = (long)b * p.T + t;
        float o;
        if (CSGU) {
            if (p.act == 1) acc = gelu_erf(acc);
            else if (p.act == 2) acc = fmaxf(acc, 0.f);
            else if (p.act == 3) acc = acc / (1.f + __expf(-acc));
            o = p.mul ? bf2f(p.mul[row * p.ld_mul + c]) * acc : acc;       // mul == NULL: the conv alone (linear-after-conv / split-gate form)
        } else {
            o = tile[(tl + p.pad_left) * DW_CT + tx] + acc;
        }
        p.out[row * p.ld_out + c] = f2bf(o);
    }
}

// Fast form for the reference's kernel size 31, dilation 1.  A block owns 64 channels x 64 time steps (4 blocks per CU; 128 steps at 2 per CU was 1 % slower end to end): the (64+30) x 64
// input tile (16-B global loads, LayerNorm applied on the way in), the gate operand x_r and the result tile all live in
// LDS, every thread keeps a 46-sample window + the 31 taps of its channel in registers and produces 16 consecutive
// outputs (77 LDS reads per 496 FMAs), and the result leaves as 16-B-per-lane rows.
constexpr int DWF_K = 31, DWF_TT = 64, DWF_CT = 64, DWF_ROWS = DWF_TT + DWF_K - 1, DWF_PER = DWF_TT / 4;
constexpr int DWF_WS = DWF_CT + 1, DWF_WBUF = (DWF_K * DWF_WS + 3) / 4 * 4;     // weight rows padded to 65 floats; buffer rounded so `io` stays 16-B aligned

template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv31_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                          // [DWF_ROWS][64] fp32
    float* sw = tile + DWF_ROWS * DWF_CT;                                  // [31][DWF_WS]: tap-major, row stride 65 (conflict-free transposed fill)
    bf16_t* io = reinterpret_cast<bf16_t*>(sw + DWF_WBUF);                 // [128][64] bf16: x_r in, result out
    const int c0 = blockIdx.x * DWF_CT, t0 = blockIdx.y * DWF_TT, b = blockIdx.z;
    const int tid = threadIdx.x;
    // the block's 64 x 31 taps are one contiguous run of the (C, 31) weight: read it linearly (coalesced), all passes in flight together
    constexpr int NPW = (DWF_K * DWF_CT + 255) / 256;
    float wv[NPW];
#pragma unroll
    for (int q = 0; q < NPW; ++q) wv[q] = p.w[(long)c0 * DWF_K + min(tid + q * 256, DWF_K * DWF_CT - 1)];
    // Fill: every global load of the block (5 passes of the conv input, their row statistics, 4 passes of the gate operand, the LayerNorm
    // affine) is issued before the first one is consumed — one memory round trip per block instead of one per pass; the block has only
    // one or two companions on its CU, so a pass-by-pass load -> wait -> ds_write loop leaves the memory pipe idle most of the time.
    constexpr int NPI = (DWF_ROWS * (DWF_CT / 8) + 255) / 256, NPO = DWF_TT * (DWF_CT / 8) / 256;
    const int ch = tid & 7;                                                // 256 % 8 == 0: a thread keeps its 16-B channel chunk in every pass
    bf16x8 vin[NPI], vio[NPO];
    float mu[NPI], rs[NPI];
    bool ok[NPI];
#pragma unroll
    for (int q = 0; q < NPI; ++q) {                                        // branch-free: out-of-range rows read a clamped (valid) row and are zeroed below
        const int id = tid + q * 256, r = id >> 3;
        const int t = t0 - p.pad_left + r;
        ok[q] = id < DWF_ROWS * (DWF_CT / 8) && t >= 0 && t < p.T;
        const long row = (long)b * p.T + min(max(t, 0), p.T - 1);
        vin[q] = *reinterpret_cast<const bf16x8*>(p.in + row * p.ld_in + c0 + ch * 8);
        mu[q] = 0.f; rs[q] = 0.f;
        if (CSGU) { mu[q] = p.stats[2 * row]; rs[q] = p.stats[2 * row + 1]; }
    }
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (CSGU) {
        g0 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch * 8); g1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch * 8 + 4);
        b0 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch * 8);  b1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch * 8 + 4);
#pragma unroll
        for (int q = 0; q < NPO; ++q) {
            const int r = (tid + q * 256) >> 3, t = t0 + r;
            vio[q] = *reinterpret_cast<const bf16x8*>(p.mul + ((long)b * p.T + min(t, p.T - 1)) * p.ld_mul + c0 + ch * 8);   // rows past T: never stored
        }
    }
    __builtin_amdgcn_sched_barrier(0);                                     // keep the loads above in front of their uses below
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
        const int i = tid + q * 256;
        if (i < DWF_K * DWF_CT) sw[(i % DWF_K) * DWF_WS + i / DWF_K] = wv[q];
    }
#pragma unroll
    for (int q = 0; q < NPI; ++q) {
        const int id = tid + q * 256, r = id >> 3;
        if (id < DWF_ROWS * (DWF_CT / 8)) {
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            if (ok[q]) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = bf2f(vin[q][j]);
                if (CSGU) {
                    lo = f32x4{(f[0] - mu[q]) * rs[q] * g0.x + b0.x, (f[1] - mu[q]) * rs[q] * g0.y + b0.y, (f[2] - mu[q]) * rs[q] * g0.z + b0.z, (f[3] - mu[q]) * rs[q] * g0.w + b0.w};
                    hi = f32x4{(f[4] - mu[q]) * rs[q] * g1.x + b1.x, (f[5] - mu[q]) * rs[q] * g1.y + b1.y, (f[6] - mu[q]) * rs[q] * g1.z + b1.z, (f[7] - mu[q]) * rs[q] * g1.w + b1.w};
                } else {
                    lo = f32x4{f[0], f[1], f[2], f[3]};
                    hi = f32x4{f[4], f[5], f[6], f[7]};
                }
            }
            *reinterpret_cast<f32x4*>(tile + r * DWF_CT + ch * 8) = lo;
            *reinterpret_cast<f32x4*>(tile + r * DWF_CT + ch * 8 + 4) = hi;
        }
    }
    if (CSGU) {
#pragma unroll
        for (int q = 0; q < NPO; ++q) {
            const int r = (tid + q * 256) >> 3;
            *reinterpret_cast<bf16x8*>(io + r * DWF_CT + ch * 8) = vio[q];
        }
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    const int c = c0 + tx;
    float wk[DWF_K];
#pragma unroll
    for (int k = 0; k < DWF_K; ++k) wk[k] = sw[k * DWF_WS + tx];
    float win[DWF_PER + DWF_K - 1];
#pragma unroll
    for (int i = 0; i < DWF_PER + DWF_K - 1; ++i) win[i] = tile[(ty * DWF_PER + i) * DWF_CT + tx];
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

}  // namespace

extern "C" int mi_conv2d_first_gelu(const float* x, const float* w, const float* bias, void* out_cl_bf16,
                                    int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f,
                                    int T1, int F1, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || F <= 0 || C <= 0 || (C % 8) != 0 || K <= 0 || K > 7) return MI_ERR_ARG;
    const int cgs = C / 8;
    if (K == 3 && C == 256 && (long)B * T1 * F1 < (1L << 31) - (1L << 20) && (reinterpret_cast<uintptr_t>(out_cl_bf16) & 15) == 0) {      // the matrix-core form
        const long ntiles = ((long)B * T1 * F1 + 31) / 32;
        const long nb = (ntiles + 3) / 4;
        hipLaunchKernelGGL(conv2d_first3_mfma_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 4 * 32 * 256 * 2, stream, x, w, bias, (bf16_t*)out_cl_bf16,
                           B, T, F, stride, pad_t, pad_f, T1, F1);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    if (K == 3 && cgs <= 256 && (256 % cgs) == 0 && (long)B * T1 * F1 < (1L << 31) - 256L * 8192) {
        const long npos = (long)B * T1 * F1;
        const int ppb = 256 / cgs;
        const long nb = (npos + ppb - 1) / ppb;
        hipLaunchKernelGGL(conv2d_first3_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, x, w, bias,
                           (bf16_t*)out_cl_bf16, B, T, F, C, stride, pad_t, pad_f, T1, F1);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const long total = (long)B * T1 * F1 * (C / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const size_t lds = (size_t)(K * K * C + C) * sizeof(float);
    hipLaunchKernelGGL(conv2d_first_kernel<1>, dim3(grid), dim3(256), lds, stream, x, w, bias, (bf16_t*)out_cl_bf16,
                       B, T, F, C, K, K, stride, stride, pad_t, pad_f, T1, F1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// Conv2d(1 -> C) of general geometry over the (B,T,F) features -> channels-last (B,T1,F1,C) bf16; act: 1 GELU, 0 raw (pre-activation).
// The raw form feeds mi_gated_act_bf16 (conv and gate operands of the context-aware front ends, extractors.py:23-54).
extern "C" int mi_conv2d_first_geo(const float* x, const float* w, const float* bias, void* out_cl_bf16, int B, int T, int F, int C,
                                   int KH, int KW, int stride_t, int stride_f, int pad_t, int pad_f, int T1, int F1, int act, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || F <= 0 || C <= 0 || (C % 8) != 0 || KH <= 0 || KW <= 0 || KH * KW > 64 || stride_t <= 0 || stride_f <= 0 || T1 <= 0 || F1 <= 0) return MI_ERR_ARG;
    if ((T1 - 1) * stride_t - pad_t >= T || (F1 - 1) * stride_f - pad_f >= F) return MI_ERR_ARG;      // every window must start inside the input
    const size_t lds = (size_t)(KH * KW * C + C) * sizeof(float);
    if (lds > 64 * 1024) return MI_ERR_UNSUPPORTED;
    const long total = (long)B * T1 * F1 * (C / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (act) hipLaunchKernelGGL(conv2d_first_kernel<1>, dim3(grid), dim3(256), lds, stream, x, w, bias, (bf16_t*)out_cl_bf16, B, T, F, C, KH, KW, stride_t, stride_f, pad_t, pad_f, T1, F1);
    else hipLaunchKernelGGL(conv2d_first_kernel<0>, dim3(grid), dim3(256), lds, stream, x, w, bias, (bf16_t*)out_cl_bf16, B, T, F, C, KH, KW, stride_t, stride_f, pad_t, pad_f, T1, F1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// GatedConv2d first layer, fused (3x3 only): out = GELU((conv + b) * sigmoid(gate + bg)) -> channels-last bf16.  MI_ERR_UNSUPPORTED for other shapes:
// the caller then runs the two raw convs + mi_gated_act_bf16.
extern "C" int mi_conv2d_first_gated_gelu(const float* x, const float* w, const float* bias, const float* gw, const float* gbias, void* out_cl_bf16,
                                          int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f, int T1, int F1, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || F <= 0 || C <= 0 || T1 <= 0 || F1 <= 0) return MI_ERR_ARG;
    const int cgs = C / 4;
    if (K != 3 || (C % 4) != 0 || cgs > 256 || (long)B * T1 * F1 >= (1L << 31) - 256L * 8192) return MI_ERR_UNSUPPORTED;
    const long npos = (long)B * T1 * F1;
    const int ppb = 256 / cgs;
    const long nb = (npos + ppb - 1) / ppb;
    hipLaunchKernelGGL(conv2d_first3_gated_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, x, w, bias, gw, gbias,
                       (bf16_t*)out_cl_bf16, B, T, F, C, stride, pad_t, pad_f, T1, F1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// out (B*T*Fq, C) = GELU(z * sigmoid(gate)), gate row = (b, t / share, f); see gated_act_kernel for `blk`.  T % share == 0 (the reference's view, extractors.py:52).
extern "C" int mi_gated_act_bf16(const void* z, long ldz, const void* g, long ldg, void* out, long ldo, int B, int T, int Fq, int C, int share, int blk,
                                 hipStream_t stream) {
    MI_ENTER();
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (B <= 0 || T <= 0 || Fq <= 0 || C <= 0 || (C % 8) != 0 || share <= 0 || (T % share) != 0 || blk < 0 || (blk && ((blk % 8) != 0 || (C % blk) != 0))) return MI_ERR_ARG;
    if ((ldz % 8) || (ldg % 8) || (ldo % 8) || !al16(z) || !al16(g) || !al16(out)) return MI_ERR_ARG;
    const long blocks = ((long)B * T * Fq * (C / 8) + 255) / 256;
    hipLaunchKernelGGL(gated_act_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, (const bf16_t*)z, ldz, (const bf16_t*)g, ldg,
                       (bf16_t*)out, ldo, B, T, Fq, C, share, blk);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_row_stats_bf16(const void* x, long ldx, int d, float eps, float* stats, int M, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || (d % 8) != 0 || (ldx % 8) != 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, d, eps, stats, M);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

static int dw_launch(const DwArgs& a, bool csgu, hipStream_t stream) {
    if (a.B <= 0 || a.T <= 0 || a.C <= 0 || a.K <= 0 || a.K > DW_KMAX || a.dilation < 1) return MI_ERR_ARG;
    const int halo = (a.K - 1) * a.dilation;
    const size_t lds = (size_t)((DW_TT + halo) * DW_CT + a.K * DW_CT) * sizeof(float);
    if (lds > 160 * 1024) return MI_ERR_UNSUPPORTED;
    const bool fast = !(csgu && !a.mul) && a.K == DWF_K && a.dilation == 1 && a.pad_left == (DWF_K - 1) / 2 && (a.C % DWF_CT) == 0 &&
                      (a.ld_in % 8) == 0 && (((uintptr_t)a.in) & 15) == 0 && (a.ld_out % 8) == 0 && (((uintptr_t)a.out) & 15) == 0 &&
                      (!csgu || ((a.ld_mul % 8) == 0 && (((uintptr_t)a.mul) & 15) == 0)) && (!csgu || (((uintptr_t)a.gamma | (uintptr_t)a.beta) & 15) == 0);
    if (fast) {
        dim3 gridf(a.C / DWF_CT, cdiv(a.T, DWF_TT), a.B);
        const size_t ldsf = (size_t)(DWF_ROWS * DWF_CT + DWF_WBUF) * sizeof(float) + (size_t)DWF_TT * DWF_CT * sizeof(bf16_t);
        if (csgu) hipLaunchKernelGGL(dwconv31_kernel<true>, gridf, dim3(256), ldsf, stream, a);
        else hipLaunchKernelGGL(dwconv31_kernel<false>, gridf, dim3(256), ldsf, stream, a);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    dim3 grid(cdiv(a.C, DW_CT), cdiv(a.T, DW_TT), a.B);
    if (csgu) hipLaunchKernelGGL(dwconv_time_kernel<true>, grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(dwconv_time_kernel<false>, grid, dim3(256), lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// CSGU: u (B*T, 2C) bf16 = [x_r | x_g];  out (B*T, C) = x_r * act(dwconv(LN(x_g)) + bias)
extern "C" int mi_csgu_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta,
                            const float* w, const float* bias, void* out, long ldo,
                            int B, int T, int C, int K, int pad_left, int dilation, int act, hipStream_t stream) {
    MI_ENTER();
    DwArgs a{};
    a.in = (const bf16_t*)u + C; a.ld_in = ldu; a.mul = (const bf16_t*)u; a.ld_mul = ldu; a.stats = stats;
    a.gamma = gamma; a.beta = beta; a.w = w; a.bias = bias; a.out = (bf16_t*)out; a.ld_out = ldo;
    a.B = B; a.T = T; a.C = C; a.K = K; a.pad_left = pad_left; a.dilation = dilation; a.act = act;
    return dw_launch(a, true, stream);
}

// The CSGU conv alone: out (B*T, C) bf16 = dwconv(LN(x_g)) + bias, for `csgu_use_linear_after_conv` (e_branchformer.py:172-175,196-201: conv -> Linear -> act -> gate),
// whose Linear runs as a GEMM between this and mi_gate_act_mul_bf16.
extern "C" int mi_csgu_conv_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta,
                                 const float* w, const float* bias, void* out, long ldo,
                                 int B, int T, int C, int K, int pad_left, int dilation, hipStream_t stream) {
    MI_ENTER();
    DwArgs a{};
    a.in = (const bf16_t*)u + C; a.ld_in = ldu; a.mul = nullptr; a.ld_mul = 0; a.stats = stats;
    a.gamma = gamma; a.beta = beta; a.w = w; a.bias = bias; a.out = (bf16_t*)out; a.ld_out = ldo;
    a.B = B; a.T = T; a.C = C; a.K = K; a.pad_left = pad_left; a.dilation = dilation; a.act = 0;
    return dw_launch(a, true, stream);
}

__global__ __launch_bounds__(256) void gate_act_mul_kernel(const bf16_t* r, long ldr, const bf16_t* g, long ldg, bf16_t* out, long ldo, long M, int C, int act) {
    const long n = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        float v = bf2f(g[row * ldg + c]);
        if (act == 1) v = gelu_erf(v);
        else if (act == 2) v = fmaxf(v, 0.f);
        else if (act == 3) v = v / (1.f + __expf(-v));
        out[row * ldo + c] = f2bf(bf2f(r[row * ldr + c]) * v);
    }
}

// out = x_r * act(g): the gate of the split CSGU form (act: 0 identity, 1 gelu, 2 relu, 3 silu)
extern "C" int mi_gate_act_mul_bf16(const void* r, long ldr, const void* g, long ldg, void* out, long ldo, long M, int C, int act, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || C <= 0 || act < 0 || act > 3) return MI_ERR_ARG;
    const long blocks = (M * C + 255) / 256;
    hipLaunchKernelGGL(gate_act_mul_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                       (const bf16_t*)r, ldr, (const bf16_t*)g, ldg, (bf16_t*)out, ldo, M, C, act);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// MERGE: out = m + dwconv(m) + bias on (B*T, C) bf16
extern "C" int mi_dwconv_residual_bf16(const void* m, long ldm, const float* w, const float* bias, void* out, long ldo,
                                       int B, int T, int C, int K, int pad_left, hipStream_t stream) {
    MI_ENTER();
    DwArgs a{};
    a.in = (const bf16_t*)m; a.ld_in = ldm; a.w = w; a.bias = bias; a.out = (bf16_t*)out; a.ld_out = ldo;
    a.B = B; a.T = T; a.C = C; a.K = K; a.pad_left = pad_left; a.dilation = 1; a.act = 0;
    return dw_launch(a, false, stream);
}
