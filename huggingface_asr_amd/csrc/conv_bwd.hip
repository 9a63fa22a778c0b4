// Backward kernels of the convolutional pieces of the path on gfx950 (training step):
//   * depthwise Conv1d over time (CSGU e_branchformer.py:184-204 with its LayerNorm + gating, merge block :296-304),
//   * the Conv2d sub-sampling front end (extractors.py:71-113): im2col for the weight gradient of conv2 and a fused
//     col2im + GELU' + conv1 weight-gradient kernel (conv1's activation gradient is never materialised).
// Gradients of the forward kernels in conv.hip; checked against torch autograd of the CPU oracle in tests/.
#include "common.hpp"

namespace {


// ------------------------------------------------------------------------------------------------ depthwise conv backward
constexpr int DB_TT = 64, DB_CT = 64, DB_KMAX = 31;

struct DwBwdArgs {
    const bf16_t* x; long ldx;            // conv input rows (CSGU: the gate half x_g, LayerNorm applied on load)
    const bf16_t* r; long ldr;            // CSGU: x_r
    const float* stats; const float* gamma; const float* beta;   // CSGU LayerNorm
    const bf16_t* dy; long lddy;          // CSGU: ds (gradient of x_r * conv);  MERGE: gradient of m + conv(m)
    const float* w; const float* bias;    // (C,K), (C)
    bf16_t* dx; long lddx;                // CSGU: gradient w.r.t. LN(x_g);  MERGE: dy + conv^T(dy)
    bf16_t* dr; long lddr;                // CSGU: ds * conv
    float* dw; float* db;                 // accumulated
    int B, T, C, K, pad_left, dilation;
};

// block = (64 channels, utterance b); walks the time tiles, keeps the K tap gradients of its channel in registers
template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwBwdArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = p.K - 1;
    const int rows = DB_TT + halo;
    float* tx_ = reinterpret_cast<float*>(smem);          // [rows][64] conv input, t from t0 - pad
    float* ty_ = tx_ + rows * DB_CT;                      // [rows][64] conv output gradient, t from t0 + pad - halo
    float* sw = ty_ + rows * DB_CT;                       // [K][64]
    float* red = sw + DB_KMAX * DB_CT;                    // [4][K+1][64] tap-gradient partials
    const int c0 = blockIdx.x * DB_CT, b = blockIdx.y;
    const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int c = c0 + tx;
    const bool cok = c < p.C;
    for (int i = threadIdx.x; i < p.K * DB_CT; i += 256) {
        const int k = i / DB_CT, cc = i % DB_CT;
        sw[i] = (c0 + cc < p.C) ? p.w[(long)(c0 + cc) * p.K + k] : 0.f;
    }
    float g = 1.f, be = 0.f, bias = 0.f;
    if (CSGU && cok) { g = p.gamma[c]; be = p.beta[c]; }
    if (cok && p.bias) bias = p.bias[c];
    float gw[DB_KMAX];
#pragma unroll
    for (int k = 0; k < DB_KMAX; ++k) gw[k] = 0.f;
    float gb = 0.f;
    const int per = DB_TT / 4;
    for (int t0 = 0; t0 < p.T; t0 += DB_TT) {
        __syncthreads();
        for (int rr = tq; rr < rows; rr += 4) {
            const int t = t0 - p.pad_left + rr;
            float v = 0.f;
            if (cok && t >= 0 && t < p.T) {
                const long row = (long)b * p.T + t;
                v = bf2f(p.x[row * p.ldx + c]);
                if (CSGU) v = (v - p.stats[2 * row]) * p.stats[2 * row + 1] * g + be;
            }
            tx_[rr * DB_CT + tx] = v;
            const int t2 = t0 + p.pad_left - halo + rr;
            float d = 0.f;
            if (cok && t2 >= 0 && t2 < p.T) {
                const long row = (long)b * p.T + t2;
                d = bf2f(p.dy[row * p.lddy + c]);
                if (CSGU && p.r) d *= bf2f(p.r[row * p.ldr + c]);      // r == NULL: dy already is the conv-output gradient (split-gate form)
            }
            ty_[rr * DB_CT + tx] = d;
        }
        __syncthreads();
        if (!cok) continue;
        for (int j = 0; j < per; ++j) {
            const int tl = tq * per + j, t = t0 + tl;
            if (t >= p.T) break;
            // dx[t] = sum_k w[k] * dyc[t + pad - k]  -> ty_ row (tl + halo - k)
            float acc = 0.f;
            for (int k = 0; k < p.K; ++k) acc = fmaf(sw[k * DB_CT + tx], ty_[(tl + halo - k) * DB_CT + tx], acc);
            const float dyc = ty_[(tl + halo - p.pad_left) * DB_CT + tx];      // conv-output gradient at t
            const long row = (long)b * p.T + t;
            if (CSGU) {
                if (p.r) {
                    float cv = bias;                                          // conv output at t (recomputed)
                    for (int k = 0; k < p.K; ++k) cv = fmaf(sw[k * DB_CT + tx], tx_[(tl + k) * DB_CT + tx], cv);
                    p.dr[row * p.lddr + c] = f2bf(bf2f(p.dy[row * p.lddy + c]) * cv);
                }
            } else {
                acc += dyc;                                                   // residual path of m + conv(m)
            }
            p.dx[row * p.lddx + c] = f2bf(acc);
            gb += dyc;
#pragma unroll
            for (int k = 0; k < DB_KMAX; ++k)
                if (k < p.K) gw[k] = fmaf(dyc, tx_[(tl + k) * DB_CT + tx], gw[k]);
        }
    }
    __syncthreads();
    float* mine = red + (size_t)tq * (DB_KMAX + 1) * DB_CT;
#pragma unroll
    for (int k = 0; k < DB_KMAX; ++k) mine[k * DB_CT + tx] = gw[k];
    mine[DB_KMAX * DB_CT + tx] = gb;
    __syncthreads();
    for (int i = threadIdx.x; i < (p.K + 1) * DB_CT; i += 256) {
        const int k = i / DB_CT, cc = i % DB_CT;
        if (c0 + cc >= p.C) continue;
        const int kk = (k == p.K) ? DB_KMAX : k;
        float s = 0.f;
        for (int q = 0; q < 4; ++q) s += red[(size_t)q * (DB_KMAX + 1) * DB_CT + kk * DB_CT + cc];
        partial[((long)b * p.C + c0 + cc) * 32 + kk] = s;                  // this utterance's partial (slot 31 = the bias); dw_partial_reduce_kernel adds the utterances in order
    }
}


// ---- dilated form (the reference's CAUSAL CSGU: e_branchformer.py:153-160 passes (K-1)/2 in CausalConv1d's dilation slot, so the conv is dilated by 15 with a left
// pad of 450; also any causal / dilated merge conv).  A streaming model's training path, not a hot one: no LDS tiling (the halo would be 450 rows), every tap is read
// from L2.  block = 64 channels x 64 time steps of utterance b; thread = (channel, 16 consecutive steps); tap gradients in registers, reduced like the kernel above.
template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv_bwd_dilated_kernel(DwBwdArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);          // [4][K+1][64]
    const int c0 = blockIdx.x * DB_CT, t0 = blockIdx.y * DB_TT, b = blockIdx.z;
    const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int c = c0 + tx;
    const bool cok = c < p.C;
    float wk[DB_KMAX], gw[DB_KMAX];
#pragma unroll
    for (int k = 0; k < DB_KMAX; ++k) { wk[k] = (cok && k < p.K) ? p.w[(long)c * p.K + k] : 0.f; gw[k] = 0.f; }
    float g = 1.f, be = 0.f, bias = 0.f, gb = 0.f;
    if (CSGU && cok) { g = p.gamma[c]; be = p.beta[c]; }
    if (cok && p.bias) bias = p.bias[c];
    auto xin = [&](int t) -> float {                       // conv input at time t (LayerNorm applied for CSGU), zero outside the utterance
        if (t < 0 || t >= p.T) return 0.f;
        const long row = (long)b * p.T + t;
        float v = bf2f(p.x[row * p.ldx + c]);
        if (CSGU) v = (v - p.stats[2 * row]) * p.stats[2 * row + 1] * g + be;
        return v;
    };
    auto dyc = [&](int t) -> float {                       // gradient w.r.t. the conv output at time t
        if (t < 0 || t >= p.T) return 0.f;
        const long row = (long)b * p.T + t;
        float d = bf2f(p.dy[row * p.lddy + c]);
        if (CSGU && p.r) d *= bf2f(p.r[row * p.ldr + c]);
        return d;
    };
    if (cok) {
        for (int j = 0; j < DB_TT / 4; ++j) {
            const int t = t0 + tq * (DB_TT / 4) + j;
            if (t >= p.T) break;
            const long row = (long)b * p.T + t;
            // y[t] = bias + sum_k w[k] x[t - pad + k dil];   dx[u] = sum_k w[k] dyc[u + pad - k dil]
            float acc = 0.f, cv = bias;
            const float dt = dyc(t);
#pragma unroll
            for (int k = 0; k < DB_KMAX; ++k)
                if (k < p.K) {
                    const float xv = xin(t - p.pad_left + k * p.dilation);
                    acc = fmaf(wk[k], dyc(t + p.pad_left - k * p.dilation), acc);
                    cv = fmaf(wk[k], xv, cv);
                    gw[k] = fmaf(dt, xv, gw[k]);
                }
            if (CSGU) { if (p.r) p.dr[row * p.lddr + c] = f2bf(bf2f(p.dy[row * p.lddy + c]) * cv); }
            else acc += dt;                                    // residual path of m + conv(m)
            p.dx[row * p.lddx + c] = f2bf(acc);
            gb += dt;
        }
    }
    float* mine = red + (size_t)tq * (DB_KMAX + 1) * DB_CT;
#pragma unroll
    for (int k = 0; k < DB_KMAX; ++k) mine[k * DB_CT + tx] = gw[k];
    mine[DB_KMAX * DB_CT + tx] = gb;
    __syncthreads();
    for (int i = threadIdx.x; i < (p.K + 1) * DB_CT; i += 256) {
        const int k = i / DB_CT, cc = i % DB_CT;
        if (c0 + cc >= p.C) continue;
        const int kk = (k == p.K) ? DB_KMAX : k;
        float s2 = 0.f;
        for (int q = 0; q < 4; ++q) s2 += red[(size_t)q * (DB_KMAX + 1) * DB_CT + kk * DB_CT + cc];
        partial[(((long)b * gridDim.y + blockIdx.y) * p.C + c0 + cc) * 32 + kk] = s2;     // one partial per (utterance, time tile)
    }
}

// ---- fast form for the reference's kernel size 31 (pad 15, dilation 1), mirror of dwconv31_kernel (conv.hip):
// block = (64 channels, utterance), 128 time steps per tile; the (128+30) x 64 conv-input tile (LayerNorm applied on the way in)
// and the conv-output-gradient tile live in LDS as fp32, every thread keeps both 62-sample windows, the 31 taps and the 31
// tap gradients of its channel in registers and produces 32 consecutive steps.  Tap / bias gradients leave as per-utterance
// partials (workspace (B, C, 32)), summed over B by dw_partial_reduce_kernel — no float atomics.
constexpr int FB_K = 31, FB_TT = 64, FB_CT = 64, FB_ROWS = FB_TT + FB_K - 1, FB_PER = FB_TT / 4;

template <bool CSGU>
__global__ __launch_bounds__(256) void dwconv31_bwd_kernel(DwBwdArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile_x = reinterpret_cast<float*>(smem);                     // [FB_ROWS][64]
    float* tile_d = tile_x + FB_ROWS * FB_CT;                           // [FB_ROWS][64]
    bf16_t* io = reinterpret_cast<bf16_t*>(tile_d + FB_ROWS * FB_CT);   // [128][64]: raw ds in (CSGU), result out
    const int c0 = blockIdx.x * FB_CT, b = blockIdx.y;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int c = c0 + tx;
    float wk[FB_K], gw[FB_K];
#pragma unroll
    for (int k = 0; k < FB_K; ++k) { wk[k] = p.w[(long)c * FB_K + k]; gw[k] = 0.f; }
    float gb = 0.f;
    const float bias = (CSGU && p.bias) ? p.bias[c] : 0.f;
    const int ch8 = tid & 7;                               // 256 % 8 == 0: a thread keeps its 16-B channel chunk in every fill pass
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (CSGU) {
        g0 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch8 * 8); g1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + ch8 * 8 + 4);
        b0 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch8 * 8);  b1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + ch8 * 8 + 4);
    }
    // The global loads of a tile (3 passes x {conv input, output gradient, gate operand, row statistics}) are all issued before the first is consumed, and — round 3 —
    // those of tile n + 1 are issued as soon as tile n's have gone to LDS, so they fly during tile n's arithmetic and stores: one block per CU (97 KiB of LDS) walks
    // an utterance's 4-8 tiles, and each tile used to start with a full memory round trip and nothing else to run.
    constexpr int NPI = (FB_ROWS * (FB_CT / 8) + 255) / 256;
    struct Fill { bf16x8 vx[NPI], vg[NPI], vr[NPI]; float mu[NPI], rs[NPI]; };
    auto fetch = [&](int t0, Fill& f) {
#pragma unroll
        for (int q = 0; q < NPI; ++q) {                                // branch-free: out-of-range rows read a clamped (valid) row and are zeroed when they go to LDS
            const int id = tid + q * 256, r = id >> 3;
            const int t = t0 - 15 + r;
            const long row = (long)b * p.T + min(max(t, 0), p.T - 1);
            f.vx[q] = *reinterpret_cast<const bf16x8*>(p.x + row * p.ldx + c0 + ch8 * 8);
            f.vg[q] = *reinterpret_cast<const bf16x8*>(p.dy + row * p.lddy + c0 + ch8 * 8);
            f.mu[q] = 0.f; f.rs[q] = 0.f;
            f.vr[q] = f.vg[q];
            if (CSGU) {
                f.mu[q] = p.stats[2 * row]; f.rs[q] = p.stats[2 * row + 1];
                f.vr[q] = *reinterpret_cast<const bf16x8*>(p.r + row * p.ldr + c0 + ch8 * 8);
            }
        }
    };
    Fill cur, nxt;
    fetch(0, cur);
    for (int t0 = 0; t0 < p.T; t0 += FB_TT) {
        __syncthreads();
        {
#pragma unroll
            for (int q = 0; q < NPI; ++q) {
                const int id = tid + q * 256, r = id >> 3;
                if (id >= FB_ROWS * (FB_CT / 8)) continue;
                const int t = t0 - 15 + r;
                const bool ok = t >= 0 && t < p.T;
                f32x4 xl = {0.f, 0.f, 0.f, 0.f}, xh = xl, dl = xl, dh = xl;
                if (ok) {
                    float f[8], e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { f[j] = bf2f(cur.vx[q][j]); e[j] = bf2f(cur.vg[q][j]); }
                    if (CSGU) {
                        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f[j] = (f[j] - cur.mu[q]) * cur.rs[q] * gm[j] + bt[j];
                            e[j] *= bf2f(cur.vr[q][j]);
                        }
                        if (r >= 15 && r < 15 + FB_TT) *reinterpret_cast<bf16x8*>(io + (r - 15) * FB_CT + ch8 * 8) = cur.vg[q];
                    }
                    xl = f32x4{f[0], f[1], f[2], f[3]}; xh = f32x4{f[4], f[5], f[6], f[7]};
                    dl = f32x4{e[0], e[1], e[2], e[3]}; dh = f32x4{e[4], e[5], e[6], e[7]};
                }
                *reinterpret_cast<f32x4*>(tile_x + r * FB_CT + ch8 * 8) = xl;
                *reinterpret_cast<f32x4*>(tile_x + r * FB_CT + ch8 * 8 + 4) = xh;
                *reinterpret_cast<f32x4*>(tile_d + r * FB_CT + ch8 * 8) = dl;
                *reinterpret_cast<f32x4*>(tile_d + r * FB_CT + ch8 * 8 + 4) = dh;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(min(t0 + FB_TT, (p.T - 1) / FB_TT * FB_TT), nxt);           // unconditional: the last tile re-reads itself (no branch for the compiler to wait at)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        float xw[FB_PER + FB_K - 1], dwn[FB_PER + FB_K - 1];
#pragma unroll
        for (int i = 0; i < FB_PER + FB_K - 1; ++i) {
            xw[i] = tile_x[(ty * FB_PER + i) * FB_CT + tx];
            dwn[i] = tile_d[(ty * FB_PER + i) * FB_CT + tx];
        }
        __syncthreads();                                   // windows are in registers: io / tiles may be overwritten below
#pragma unroll
        for (int j = 0; j < FB_PER; ++j) {
            const int rl = ty * FB_PER + j;
            const float dyc = dwn[j + 15];                 // rows beyond T hold zeros: no contribution
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < FB_K; ++k) acc = fmaf(wk[k], dwn[j + 30 - k], acc);
            gb += dyc;
#pragma unroll
            for (int k = 0; k < FB_K; ++k) gw[k] = fmaf(dyc, xw[j + k], gw[k]);
            if (CSGU) {
                float cv = bias;
#pragma unroll
                for (int k = 0; k < FB_K; ++k) cv = fmaf(wk[k], xw[j + k], cv);
                const float dsv = bf2f(io[rl * FB_CT + tx]);
                io[rl * FB_CT + tx] = f2bf(dsv * cv);                                       // dr
                reinterpret_cast<bf16_t*>(tile_x)[rl * FB_CT + tx] = f2bf(acc);            // dgn (tile_x reused as a bf16 [128][64] buffer)
            } else {
                io[rl * FB_CT + tx] = f2bf(acc + dyc);                                      // dm = dy + conv^T(dy)
            }
        }
        __syncthreads();
        for (int id = tid; id < FB_TT * (FB_CT / 8); id += 256) {
            const int r = id >> 3, ch = id & 7;
            const int t = t0 + r;
            if (t >= p.T) continue;
            const long row = (long)b * p.T + t;
            if (CSGU) {
                *reinterpret_cast<bf16x8*>(p.dr + row * p.lddr + c0 + ch * 8) = *reinterpret_cast<const bf16x8*>(io + r * FB_CT + ch * 8);
                *reinterpret_cast<bf16x8*>(p.dx + row * p.lddx + c0 + ch * 8) = *reinterpret_cast<const bf16x8*>(reinterpret_cast<bf16_t*>(tile_x) + r * FB_CT + ch * 8);
            } else {
                *reinterpret_cast<bf16x8*>(p.dx + row * p.lddx + c0 + ch * 8) = *reinterpret_cast<const bf16x8*>(io + r * FB_CT + ch * 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    // reduce the 4 time groups' tap gradients through LDS, store this utterance's partial
    __syncthreads();
    float* red = tile_x;                                   // [4][32][64]
#pragma unroll
    for (int k = 0; k < FB_K; ++k) red[(ty * 32 + k) * FB_CT + tx] = gw[k];
    red[(ty * 32 + 31) * FB_CT + tx] = gb;
    __syncthreads();
    for (int i = tid; i < 32 * FB_CT; i += 256) {
        const int k = i / FB_CT, cc = i % FB_CT;
        const float sum = red[k * FB_CT + cc] + red[(32 + k) * FB_CT + cc] + red[(64 + k) * FB_CT + cc] + red[(96 + k) * FB_CT + cc];
        partial[((long)b * p.C + c0 + cc) * 32 + k] = sum;
    }
}

// dw[c][k] += sum_b partial[b][c][k] (k < K <= 31),  db[c] += sum_b partial[b][c][31]; the B partials (utterances, or (utterance, time tile) pairs) are added in a fixed
// order by rows_reduce_kernel (16 columns x 16 row groups per block; a thread per (channel, tap) walking all B rows alone took 27 us per call at B = 96)
struct EmitDw { float* dw; float* db; int K; __device__ void operator()(int i, float v) const { const int c = i >> 5, k = i & 31; if (k < K) dw[(long)c * K + k] += v; else if (k == 31 && db) db[c] += v; } };
static inline void dw_partial_reduce_launch(const float* partial, int B, int C, int K, float* dw, float* db, hipStream_t st) {
    rows_reduce_launch(partial, B, C * 32, EmitDw{dw, db, K}, st);
}

// ------------------------------------------------------------------------------------------------ im2col (channels-last)
// in (B,Tin,Fin,Cin) bf16 -> col (B*Tout*Fout, KH*KW*Cin) bf16, k = (kh*KW + kw)*Cin + c  (the A operand of conv2 made explicit)
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ col, int B, int Tin, int Fin,
                                                      int Cin, int KH, int KW, int stride, int stride_f, int pad_t, int pad_f, int Tout, int Fout) {
    const int c8 = Cin / 8;
    const long total = (long)B * Tout * Fout * KH * KW * c8;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        long r = i / c8;
        const int tap = (int)(r % (KH * KW)); r /= (KH * KW);
        const int fo = (int)(r % Fout); r /= Fout;
        const int to = (int)(r % Tout);
        const int b = (int)(r / Tout);
        const int kh = tap / KW, kw = tap % KW;
        const int ti = to * stride - pad_t + kh, fi = fo * stride_f - pad_f + kw;
        bf16x8 v = zero8;
        if (ti >= 0 && ti < Tin && fi >= 0 && fi < Fin)
            v = *reinterpret_cast<const bf16x8*>(in + (((long)b * Tin + ti) * Fin + fi) * Cin + cc * 8);
        *reinterpret_cast<bf16x8*>(col + i * 8) = v;
    }
}

// Fixed-order sum over the threads of a block that share a channel group: thread (g, pl) of cg x ppb contributes NV values; emit(g, v, sum over pl ascending).  Through an
// LDS staging area of 256 x NV floats (in place of LDS float atomics, whose order is not reproducible).  Every thread of the block calls it.
template <int NV, typename Emit>
__device__ __forceinline__ void group_sum_ordered(float* stage, const float (&vals)[NV], int g, int pl, int cg, int ppb, Emit emit) {
    __syncthreads();
    if (pl < ppb) {
#pragma unroll
        for (int v = 0; v < NV; ++v) stage[((long)pl * cg + g) * NV + v] = vals[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cg * NV; i += 256) {
        float sum = 0.f;
        for (int q = 0; q < ppb; ++q) sum += stage[(long)q * cg * NV + i];
        emit(i / NV, i % NV, sum);
    }
}
struct EmitConv1 { float* dw; float* db; int NT; __device__ void operator()(int i, float v) const { const int ch = i / (NT + 1), k = i % (NT + 1); if (k == NT) db[ch] += v; else dw[ch * NT + k] += v; } };

// ------------------------------------------------------------------------------------------------ conv1 backward (3x3)
// Fused: dact1[b,t1,f1,c] = sum over the conv2 taps that read this position of dcol[(b,t2,f2), (kh,kw,c)]   (col2im gather)
//        dpre1 = dact1 * gelu'(pre1), pre1 recomputed from the features;  dW1[c][tap] += dpre1 * x[tap];  db1[c] += dpre1.
// Thread layout of conv2d_first3_kernel: a thread owns 8 channels and walks positions.
// FIX32: conv2 is the usual 3x3 / stride 2 and there are fewer than 2^31 positions: tap loops unrolled over constants, shifts instead of
// divisions by the stride, 32-bit position arithmetic.
// PH: the activation gradient arrives as the four PHASE buffers of mi_conv2d_s2k3_dgrad_bf16 (conv2's data gradient as four stride-1 convolutions, one per parity
// of (t1 + pad2_t, f1 + pad2_f)) instead of the (B*T2*F2, 9 C) gradient of conv2's im2col operand: ONE 16-B load per position and channel group instead of up to four
struct Conv1Ph { const bf16_t* buf[4]; int U[2], V[2], ulo[2], vlo[2]; };
template <bool FIX32, bool PH = false>
__global__ __launch_bounds__(256) void conv1_bwd3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const bf16_t* __restrict__ dcol, float* __restrict__ partial,
                                                          int B, int T, int F, int C, int stride, int pad_t, int pad_f, int T1, int F1,
                                                          int K2r, int stride2r, int pad2_t, int pad2_f, int T2, int F2, Conv1Ph ph) {
    const int K2 = FIX32 ? 3 : K2r, stride2 = FIX32 ? 2 : stride2r;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stage = reinterpret_cast<float*>(smem);       // [256][40]: the block's ordered reduction (below)
    const int cg = C >> 3, ppb = 256 / cg;
    const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
    float wr[9][8], br[8], gwr[9][8], gbr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias[g * 8 + j]; gbr[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) { wr[tap][j] = w[(g * 8 + j) * 9 + tap]; gwr[tap][j] = 0.f; }
    }
    const long total = (long)B * T1 * F1;
    const long ldcol = (long)K2 * K2 * C;
    if (FIX32) {
        // Software-pipelined form (round 3).  The gather of a position is at most 2 x 2 taps of conv2 (stride 2: along each axis either the tap of matching parity, or
        // taps 0 and 2): ALWAYS four 16-B loads — a tap that does not exist re-reads the first one and is multiplied by zero — plus the nine features, all issued
        // for position n + 1 before position n's ~500 VALU instructions run.  Before, every position began with its own loads and nothing in flight: at two waves
        // per SIMD (200 VGPRs) the kernel ran at 0.9 TB/s on a 700-MB read.
        struct Pos { bf16x8 v[PH ? 1 : 4]; float x[9]; unsigned mask; };          // mask: bits 0-3 the taps that exist, bits 4-12 the features inside the input
        auto fetch = [&](long pos, Pos& q) {
            const unsigned up = (unsigned)pos, qq = up / (unsigned)F1;
            const int f1 = (int)(up - qq * (unsigned)F1), b = (int)(qq / (unsigned)T1), t1 = (int)(qq - (unsigned)b * (unsigned)T1);
            const int at = t1 + pad2_t, af = f1 + pad2_f;
            const int kh0 = at & 1, kw0 = af & 1;
            const int t2a = (at - kh0) >> 1, f2a = (af - kw0) >> 1;
            const bool vt[2] = {t2a < T2, kh0 == 0 && t2a >= 1 && t2a - 1 < T2};
            const bool vf[2] = {f2a < F2, kw0 == 0 && f2a >= 1 && f2a - 1 < F2};
            const bf16_t* base = dcol + g * 8;
            unsigned mask = 0;
            if constexpr (PH) {
                const int u = (at >> 1) - ph.ulo[kh0], v = (af >> 1) - ph.vlo[kw0];
                const bf16_t* pb = kh0 ? (kw0 ? ph.buf[3] : ph.buf[2]) : (kw0 ? ph.buf[1] : ph.buf[0]);
                q.v[0] = *reinterpret_cast<const bf16x8*>(pb + (((long)b * ph.U[kh0] + u) * ph.V[kw0] + v) * C + g * 8);
                mask = 1u;
            } else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = vt[i] && vf[j];
                    const int t2 = ok ? t2a - i : 0, f2 = ok ? f2a - j : 0, tap = ok ? (kh0 + 2 * i) * 3 + (kw0 + 2 * j) : 0;
                    const long off = (((long)b * T2 + t2) * F2 + f2) * ldcol + (long)tap * C;
                    q.v[i * 2 + j] = *reinterpret_cast<const bf16x8*>(base + (ok ? off : 0));
                    mask |= (ok ? 1u : 0u) << (i * 2 + j);
                }
            const float* xb = x + (long)b * T * F;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int t = t1 * stride - pad_t + kh, tc = min(max(t, 0), T - 1);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int f = f1 * stride - pad_f + kw, fc = min(max(f, 0), F - 1);
                    q.x[kh * 3 + kw] = xb[tc * F + fc];                              // clamped address: the load is unconditional; the mask bit says whether it counts
                    mask |= ((t == tc && f == fc) ? 1u : 0u) << (4 + kh * 3 + kw);
                }
            }
            q.mask = mask;
        };
        const long step = (long)gridDim.x * ppb;
        long pos = (long)blockIdx.x * ppb + pl;
        if (pl < ppb && pos < total) {
            Pos cur, nxt;
            fetch(pos, cur);
            for (;;) {
                const long nx = pos + step;
                const bool more = nx < total;
                fetch(more ? nx : pos, nxt);                 // unconditional (the last one re-reads this position): nothing the compiler has to wait for early
                __builtin_amdgcn_sched_barrier(0);           // the loads above are issued BEFORE this position's arithmetic, and nothing below may be hoisted over them
                float da[8], xv[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) xv[tap] = (cur.mask >> (4 + tap)) & 1u ? cur.x[tap] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    da[j] = 0.f;
#pragma unroll
                    for (int i = 0; i < (PH ? 1 : 4); ++i) da[j] += (PH || ((cur.mask >> i) & 1u)) ? bf2f(cur.v[i][j]) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float pre = br[j];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) pre = fmaf(xv[tap], wr[tap][j], pre);
                    const float dpre = da[j] * gelu_erf_grad(pre);
                    gbr[j] += dpre;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) gwr[tap][j] = fmaf(dpre, xv[tap], gwr[tap][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!more) break;
                cur = nxt;
                pos = nx;
            }
        }
    } else if (pl < ppb)
    for (long pos = (long)blockIdx.x * ppb + pl; pos < total; pos += (long)gridDim.x * ppb) {
        int f1, t1, b;
        if (FIX32) {
            const unsigned up = (unsigned)pos, q = up / (unsigned)F1;
            f1 = (int)(up - q * (unsigned)F1); b = (int)(q / (unsigned)T1); t1 = (int)(q - (unsigned)b * (unsigned)T1);
        } else {
            f1 = (int)(pos % F1);
            t1 = (int)((pos / F1) % T1);
            b = (int)(pos / ((long)F1 * T1));
        }
        // gather the activation gradient
        float da[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) da[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < (FIX32 ? 3 : K2); ++kh) {
            const int nt = t1 + pad2_t - kh;
            if (nt < 0 || (FIX32 ? (nt & 1) : (nt % stride2)) != 0) continue;
            const int t2 = FIX32 ? nt >> 1 : nt / stride2;
            if (t2 >= T2) continue;
#pragma unroll
            for (int kw = 0; kw < (FIX32 ? 3 : K2); ++kw) {
                const int nf = f1 + pad2_f - kw;
                if (nf < 0 || (FIX32 ? (nf & 1) : (nf % stride2)) != 0) continue;
                const int f2 = FIX32 ? nf >> 1 : nf / stride2;
                if (f2 >= F2) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcol + (((long)b * T2 + t2) * F2 + f2) * ldcol + (long)(kh * K2 + kw) * C + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) da[j] += bf2f(v[j]);
            }
        }
        float xv[9];
        const float* xb = x + (long)b * T * F;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = t1 * stride - pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = f1 * stride - pad_f + kw;
                xv[kh * 3 + kw] = (t >= 0 && t < T && f >= 0 && f < F) ? xb[(long)t * F + f] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float pre = br[j];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) pre = fmaf(xv[tap], wr[tap][j], pre);
            const float dpre = da[j] * gelu_erf_grad(pre);
            gbr[j] += dpre;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) gwr[tap][j] = fmaf(dpre, xv[tap], gwr[tap][j]);
        }
    }
    // the block's sums over its ppb position lanes, in lane order, as row blockIdx.x of `partial` ([channel][9 taps | bias]); a second kernel adds the rows in order
    float* prow = partial + (long)blockIdx.x * C * 10;
    float va[40], vb[40];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int tap = 0; tap < 5; ++tap) va[tap * 8 + j] = gwr[tap][j];
#pragma unroll
        for (int tap = 5; tap < 9; ++tap) vb[(tap - 5) * 8 + j] = gwr[tap][j];
        vb[32 + j] = gbr[j];
    }
    group_sum_ordered<40>(stage, va, g, pl, cg, ppb, [&](int gg, int v, float sum) { prow[(gg * 8 + (v & 7)) * 10 + (v >> 3)] = sum; });
    group_sum_ordered<40>(stage, vb, g, pl, cg, ppb, [&](int gg, int v, float sum) { prow[(gg * 8 + (v & 7)) * 10 + 5 + (v >> 3)] = sum; });
}

// ------------------------------------------------------------------------------------------------ conv1 backward on the matrix cores (round 4; C = 256, phase buffers)
// Per 32-position tile and 128 channels (a wave; the two waves of a pair split the channels):
//   pre[ch][pos]  = two K = 16 MFMAs on hi / lo split operands, exactly the forward (conv.hip conv2d_first3_mfma_kernel: slots x_hi w_hi | x_lo w_hi | x_hi w_lo | b_hi | b_lo)
//   dpre          = dact1 * GELU'(pre)                      (dact1: the tile's rows come in as 16-B chunks, pass through LDS and are read back in the accumulator layout)
//   dW[ch][tap]  += sum_pos dpre[pos][ch] x[pos][tap]       = a GEMM over the tile's 32 positions: dpre (hi / lo bf16 images [pos][ch]) and the tile's features
//                                                             ([pos][x_hi 0-8 | 1 | x_lo 0-8 | 0]) both go through LDS and come back by `ds_read_b64_tr_b16` as the A / B
//                                                             operands (same k-permutation on both sides, as in gemm_tn.hip); column 9 (the ones) is db.
// The accumulators (128 channels x 32 columns per wave) live in registers for the whole kernel; per block they are summed in wave order into one partial row.
// The VALU form above spends ~300 instructions per position and 8 channels (72 FMAs for pre, 72 for dW, GELU', gather): 1.51 ms at BASELINE config 3 against 0.45 ms of reads.
__device__ __forceinline__ int c1b_swz16(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }       // 256-B rows: the transposed read's four rows x 64 B land on disjoint banks
typedef short c1b_s16x4 __attribute__((ext_vector_type(4)));
typedef short c1b_s16x8 __attribute__((ext_vector_type(8)));
template <int OFF>
__device__ __forceinline__ void c1b_tr_issue(unsigned a0, unsigned a1, c1b_s16x4& lo, c1b_s16x4& hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16x8 c1b_tr_join(c1b_s16x4& lo, c1b_s16x4& hi) {
    asm volatile("" : "+v"(lo), "+v"(hi));
    const c1b_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// lane addresses of the two transposed reads of a 32-column fragment starting at column cb of a [row][ROWB bytes] image (k-block 0); SWZ16: the 256-B rows' chunk swizzle, else row & 3
template <int ROWB, bool SWZ16>
__device__ __forceinline__ void c1b_tr_addr(unsigned base, int cb, int lane, unsigned& a0, unsigned& a1) {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int col = cb + (g & 1) * 16 + 4 * p4;
    const int lc = col >> 3, within = (col & 7) * 2;
    const int k0 = 4 * (g >> 1) + q4, k1 = k0 + 8;
    a0 = base + k0 * ROWB + ((lc ^ (SWZ16 ? c1b_swz16(k0) : (k0 & 3))) << 4) + within;
    a1 = base + k1 * ROWB + ((lc ^ (SWZ16 ? c1b_swz16(k1) : (k1 & 3))) << 4) + within;
}
__device__ __forceinline__ bf16_t c1b_slot(const bf16_t (&hi)[9], const bf16_t (&lo_)[9], bf16_t one_hi, bf16_t one_lo, int s, bool lo_in_middle) {
    if (s < 9) return hi[s];
    if (s < 18) return lo_in_middle ? lo_[s - 9] : hi[s - 9];
    if (s < 27) return lo_in_middle ? hi[s - 18] : lo_[s - 18];
    if (s == 27) return one_hi;
    if (s == 28) return one_lo;
    return (bf16_t)0.f;
}
constexpr int C1B_WAVE_LDS = 2 * 8192 + 2048;          // per wave: dact staging / dpre hi image | dpre lo image | feature image   (the final 128 x 32 fp32 accumulators reuse the first 16 KiB)
__global__ __launch_bounds__(256, 2) void conv1_bwd3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ partial,
                                                               int B, int T, int F, int stride, int pad_t, int pad_f, int T1, int F1, int pad2_t, int pad2_f, Conv1Ph ph) {
    constexpr int C = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pair = wave >> 1, cb0 = (wave & 1) * 128;
    char* reg = smem + wave * C1B_WAVE_LDS;
    char* DH = reg;                                     // [32][256 B]: first the dact tile, then — in place — dpre's hi image (both in the transposed read's layout: chunk c of row p at c ^ swz16(p))
    char* DL = reg + 8192;
    char* XI = reg + 16384;                             // [32][64 B]
    // weight operand of the forward product: rows = channels cb0 + 32 g + l31
    bf16x8 wa[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = cb0 + 32 * g + l31;
        bf16_t whi[9], wlo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { const float v = w[ch * 9 + t]; whi[t] = f2bf(v); wlo[t] = f2bf(v - bf2f(whi[t])); }
        const float bv = bias[ch];
        const bf16_t bhi = f2bf(bv), blo = f2bf(bv - bf2f(bhi));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16_t s0 = c1b_slot(whi, wlo, bhi, blo, 16 * j + i, false), s1 = c1b_slot(whi, wlo, bhi, blo, 16 * j + 8 + i, false);
                wa[g][j][i] = h ? s1 : s0;
            }
    }
    unsigned xa0, xa1, da0[4], da1[4];
    c1b_tr_addr<64, false>((unsigned)(size_t)XI, 0, lane, xa0, xa1);
#pragma unroll
    for (int g = 0; g < 4; ++g) c1b_tr_addr<256, true>((unsigned)(size_t)DH, 32 * g, lane, da0[g], da1[g]);
    f32x16 dacc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[g][r] = 0.f;
    const int total = B * T1 * F1;
    const int ntiles = (total + 31) >> 5;
    const bf16_t one = f2bf(1.f), zero = f2bf(0.f);
    for (int tile = blockIdx.x * 2 + pair; tile < ntiles; tile += gridDim.x * 2) {
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
        // this lane's position in its phase buffer (conv1_bwd3_kernel<.., PH>'s fetch)
        const int at = t1 + pad2_t, af = f1 + pad2_f, kh0 = at & 1, kw0 = af & 1;
        const bf16_t* pbuf = kh0 ? (kw0 ? ph.buf[3] : ph.buf[2]) : (kw0 ? ph.buf[1] : ph.buf[0]);
        const bf16_t* rowp = pbuf + (((long)b * ph.U[kh0] + ((at >> 1) - ph.ulo[kh0])) * ph.V[kw0] + ((af >> 1) - ph.vlo[kw0])) * C + cb0;
        const unsigned long long rp = reinterpret_cast<unsigned long long>(rowp);
        // (1) the tile's dact rows, 16-B chunks: row p = 4 i + (lane >> 4), chunk c = lane & 15
        uint4 dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 4 * i + (lane >> 4), c = lane & 15;
            const unsigned lo32 = (unsigned)__shfl((int)(unsigned)rp, p, 64), hi32 = (unsigned)__shfl((int)(unsigned)(rp >> 32), p, 64);
            const bf16_t* src = reinterpret_cast<const bf16_t*>(((unsigned long long)hi32 << 32) | lo32);
            dv[i] = (m0 + p < total) ? *reinterpret_cast<const uint4*>(src + c * 8) : uint4{0u, 0u, 0u, 0u};
        }
        // (2) pre = W x (+ b), rows = channels, lane = position
        bf16x8 xb_[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16_t s0 = c1b_slot(xhi, xlo, one, one, 16 * j + i, true), s1 = c1b_slot(xhi, xlo, one, one, 16 * j + 8 + i, true);
                xb_[j][i] = h ? s1 : s0;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 4 * i + (lane >> 4), c = lane & 15;
            *reinterpret_cast<uint4*>(DH + p * 256 + ((c ^ c1b_swz16(p)) << 4)) = dv[i];
        }
        // the feature image: row = position, columns x_hi 0-8 | 1 | x_lo 0-8 | 0 (32 bf16 = four 16-B chunks, chunk c at c ^ (row & 3)); the two lanes of a position write two chunks each
        {
            bf16x8 c0, c1;
            if (h == 0) { c0 = bf16x8{xhi[0], xhi[1], xhi[2], xhi[3], xhi[4], xhi[5], xhi[6], xhi[7]}; c1 = bf16x8{xhi[8], one, xlo[0], xlo[1], xlo[2], xlo[3], xlo[4], xlo[5]}; }
            else { c0 = bf16x8{xlo[6], xlo[7], xlo[8], zero, zero, zero, zero, zero}; c1 = bf16x8{zero, zero, zero, zero, zero, zero, zero, zero}; }
            *reinterpret_cast<bf16x8*>(XI + l31 * 64 + (((2 * h) ^ (l31 & 3)) << 4)) = c0;
            *reinterpret_cast<bf16x8*>(XI + l31 * 64 + (((2 * h + 1) ^ (l31 & 3)) << 4)) = c1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // (2) + (4), two channel groups at a time (32 accumulators live): pre = W x (+ b) with rows = channels, lane = position; dpre = dact * GELU'(pre), split into
        // bf16 hi / lo and written as [position][channel] images in the transposed read's layout
        // (dact comes back from the staging rows in the accumulator layout — registers 4 q .. 4 q + 3 of group g = channels 32 g + 8 q + 4 h .. + 3 of position l31 — and
        // the staging rows use the transposed read's chunk swizzle, so a lane's 8-B piece of dpre's hi image lands exactly where its dact piece was: in place, group by group)
        const int sw = c1b_swz16(l31);
#pragma unroll
        for (int gh = 0; gh < 4; gh += 2) {
            bf16x4 da[2][4];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) da[g][q] = *reinterpret_cast<const bf16x4*>(DH + l31 * 256 + (((4 * (gh + g) + q) ^ sw) << 4) + h * 8);
            f32x16 acc[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[gh + g][0], xb_[0], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[gh + g][1], xb_[1], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = bf2f(da[g][q][e]) * gelu_erf_grad(acc[g][4 * q + e]);
                        vh[e] = f2bf(d); vl[e] = f2bf(d - bf2f(vh[e]));
                    }
                    const int off = l31 * 256 + (((4 * (gh + g) + q) ^ sw) << 4) + h * 8;
                    *reinterpret_cast<bf16x4*>(DH + off) = vh;
                    *reinterpret_cast<bf16x4*>(DL + off) = vl;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (5) dW += dpre^T x over the tile's 32 positions: two k-blocks of 16, hi and lo images
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            c1b_s16x4 xl, xh, dl_[4][2], dh_[4][2];                      // [group][hi / lo image]
            if (kb == 0) {
                c1b_tr_issue<0>(xa0, xa1, xl, xh);
#pragma unroll
                for (int g = 0; g < 4; ++g) { c1b_tr_issue<0>(da0[g], da1[g], dl_[g][0], dh_[g][0]); c1b_tr_issue<8192>(da0[g], da1[g], dl_[g][1], dh_[g][1]); }
            } else {
                c1b_tr_issue<16 * 64>(xa0, xa1, xl, xh);
#pragma unroll
                for (int g = 0; g < 4; ++g) { c1b_tr_issue<16 * 256>(da0[g], da1[g], dl_[g][0], dh_[g][0]); c1b_tr_issue<8192 + 16 * 256>(da0[g], da1[g], dl_[g][1], dh_[g][1]); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bf16x8 xf = c1b_tr_join(xl, xh);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1b_tr_join(dl_[g][0], dh_[g][0]), xf, dacc[g], 0, 0, 0);
                dacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1b_tr_join(dl_[g][1], dh_[g][1]), xf, dacc[g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the block's partial row [channel][9 taps | bias]: every wave's 128 x 32 accumulators through LDS, pairs added in order
    float* fin = reinterpret_cast<float*>(reg);          // [128 channels][32 columns]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[(32 * g + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = dacc[g][r];
    __syncthreads();
    float* prow = partial + (long)blockIdx.x * C * 10;
    for (int i = threadIdx.x; i < C * 10; i += 256) {
        const int ch = i / 10, k = i - ch * 10;
        float sum = 0.f;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const float* fw = reinterpret_cast<const float*>(smem + (pr * 2 + (ch >> 7)) * C1B_WAVE_LDS) + (ch & 127) * 32;
            sum += k < 9 ? fw[k] + fw[10 + k] : fw[9];
        }
        prow[i] = sum;
    }
}

int grid_for(long n, int cap) { const long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

int dw_bwd_launch(const DwBwdArgs& a, bool csgu, float* workspace, hipStream_t st) {
    if (a.B <= 0 || a.T <= 0 || a.C <= 0 || a.K <= 0 || a.K > DB_KMAX || a.dilation < 1 || a.pad_left < 0 || a.pad_left > (a.K - 1) * a.dilation || !workspace) return MI_ERR_ARG;
    if (a.dilation > 1) {                                   // the causal (dilated) CSGU / merge conv
        const size_t ldsd = (size_t)4 * (DB_KMAX + 1) * DB_CT * sizeof(float);
        dim3 gridd(cdiv(a.C, DB_CT), cdiv(a.T, DB_TT), a.B);
        if (csgu) hipLaunchKernelGGL(dwconv_bwd_dilated_kernel<true>, gridd, dim3(256), ldsd, st, a, workspace);
        else hipLaunchKernelGGL(dwconv_bwd_dilated_kernel<false>, gridd, dim3(256), ldsd, st, a, workspace);
        MI_CHECK_LAUNCH();
        if (a.dw) dw_partial_reduce_launch(workspace, a.B * (int)gridd.y, a.C, a.K, a.dw, a.db, st);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool fast = workspace && !(csgu && !a.r) && a.K == FB_K && a.pad_left == 15 && (a.C % FB_CT) == 0 && (a.ldx % 8) == 0 && (a.lddy % 8) == 0 && (a.lddx % 8) == 0 &&
                      al16(a.x) && al16(a.dy) && al16(a.dx) && (!csgu || ((a.ldr % 8) == 0 && (a.lddr % 8) == 0 && al16(a.r) && al16(a.dr) && al16(a.gamma) && al16(a.beta)));
    if (fast) {
        const size_t ldsf = (size_t)2 * FB_ROWS * FB_CT * sizeof(float) + (size_t)FB_TT * FB_CT * sizeof(bf16_t);
        dim3 gridf(a.C / FB_CT, a.B);
        if (csgu) hipLaunchKernelGGL(dwconv31_bwd_kernel<true>, gridf, dim3(256), ldsf, st, a, workspace);
        else hipLaunchKernelGGL(dwconv31_bwd_kernel<false>, gridf, dim3(256), ldsf, st, a, workspace);
        MI_CHECK_LAUNCH();
        if (a.dw) dw_partial_reduce_launch(workspace, a.B, a.C, a.K, a.dw, a.db, st);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const int rows = DB_TT + a.K - 1;
    const size_t lds = (size_t)(2 * rows * DB_CT + DB_KMAX * DB_CT + 4 * (DB_KMAX + 1) * DB_CT) * sizeof(float);
    dim3 grid(cdiv(a.C, DB_CT), a.B);
    if (csgu) hipLaunchKernelGGL(dwconv_bwd_kernel<true>, grid, dim3(256), lds, st, a, workspace);
    else hipLaunchKernelGGL(dwconv_bwd_kernel<false>, grid, dim3(256), lds, st, a, workspace);
    MI_CHECK_LAUNCH();
    if (a.dw) dw_partial_reduce_launch(workspace, a.B, a.C, a.K, a.dw, a.db, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace

// `workspace`: B*C*32 floats (per-utterance tap-gradient partials), B*ceil(T/64)*C*32 for the dilated form (a partial per time tile); the partials are added in a fixed
// order by a second kernel: no float atomics, the gradients are bit-reproducible.
// CSGU backward (identity activation; dilation > 1 = the causal form): u (B*T, 2C) = [x_r | x_g], ds = gradient of x_r * (dwconv(LN(x_g)) + b)
//   -> dr (B*T, C) = ds * conv,  dgn (B*T, C) = gradient w.r.t. LN(x_g),  dw (C,K) +=, db (C) +=
// dr == NULL: the split-gate form (mi_csgu_conv_bf16 forward) — `ds` is the gradient of the conv output itself, no gate operand is applied or produced.
extern "C" int mi_csgu_bwd_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta, const float* w,
                                const float* bias, const void* ds, long ldds, void* dr, long lddr, void* dgn, long lddgn,
                                float* dw, float* db, int B, int T, int C, int K, int pad_left, int dilation, float* workspace, hipStream_t st) {
    MI_ENTER();
    DwBwdArgs a{};
    a.dilation = dilation;
    a.x = (const bf16_t*)u + C; a.ldx = ldu; a.r = dr ? (const bf16_t*)u : nullptr; a.ldr = ldu; a.stats = stats; a.gamma = gamma; a.beta = beta;
    a.dy = (const bf16_t*)ds; a.lddy = ldds; a.w = w; a.bias = bias; a.dx = (bf16_t*)dgn; a.lddx = lddgn; a.dr = (bf16_t*)dr; a.lddr = lddr;
    a.dw = dw; a.db = db; a.B = B; a.T = T; a.C = C; a.K = K; a.pad_left = pad_left;
    return dw_bwd_launch(a, true, workspace, st);
}

__device__ __forceinline__ float gate_act(float v, int act) {
    if (act == 1) return gelu_erf(v);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return v / (1.f + __expf(-v));
    return v;
}
__device__ __forceinline__ float gate_act_grad(float v, int act) {
    if (act == 1) return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
    if (act == 2) return v > 0.f ? 1.f : 0.f;
    if (act == 3) { const float sg = 1.f / (1.f + __expf(-v)); return sg * (1.f + v * (1.f - sg)); }
    return 1.f;
}

__global__ __launch_bounds__(256) void gate_act_mul_bwd_kernel(const bf16_t* r, long ldr, const bf16_t* g, long ldg, const bf16_t* ds, long ldds,
                                                               bf16_t* dr, long lddr, bf16_t* dg, long lddg, long M, int C, int act) {
    const long n = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        const float gv = bf2f(g[row * ldg + c]), d = bf2f(ds[row * ldds + c]);
        dr[row * lddr + c] = f2bf(d * gate_act(gv, act));
        dg[row * lddg + c] = f2bf(d * bf2f(r[row * ldr + c]) * gate_act_grad(gv, act));
    }
}

// backward of mi_gate_act_mul_bf16 (s = x_r * act(g)):  dr = ds * act(g),  dg = ds * x_r * act'(g)
extern "C" int mi_gate_act_mul_bwd_bf16(const void* r, long ldr, const void* g, long ldg, const void* ds, long ldds, void* dr, long lddr,
                                        void* dg, long lddg, long M, int C, int act, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || C <= 0 || act < 0 || act > 3) return MI_ERR_ARG;
    const long blocks = (M * C + 255) / 256;
    hipLaunchKernelGGL(gate_act_mul_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, (const bf16_t*)r, ldr, (const bf16_t*)g, ldg,
                       (const bf16_t*)ds, ldds, (bf16_t*)dr, lddr, (bf16_t*)dg, lddg, M, C, act);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// merge-block backward: y = m + dwconv(m) + b  ->  dm = dy + conv^T(dy),  dw +=, db +=
extern "C" int mi_dwconv_residual_bwd_bf16(const void* m, long ldm, const float* w, const void* dy, long lddy, void* dm, long lddm,
                                           float* dw, float* db, int B, int T, int C, int K, int pad_left, int dilation, float* workspace, hipStream_t st) {
    MI_ENTER();
    DwBwdArgs a{};
    a.dilation = dilation;
    a.x = (const bf16_t*)m; a.ldx = ldm; a.dy = (const bf16_t*)dy; a.lddy = lddy; a.w = w; a.dx = (bf16_t*)dm; a.lddx = lddm;
    a.dw = dw; a.db = db; a.B = B; a.T = T; a.C = C; a.K = K; a.pad_left = pad_left;
    return dw_bwd_launch(a, false, workspace, st);
}

extern "C" int mi_im2col_cl_bf16(const void* in, void* col, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride, int pad_t,
                                 int pad_f, int Tout, int Fout, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || Cin <= 0 || (Cin % 8) != 0 || Tout <= 0 || Fout <= 0) return MI_ERR_ARG;
    const long total = (long)B * Tout * Fout * KH * KW * (Cin / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)col, B, Tin, Fin, Cin, KH, KW,
                       stride, stride, pad_t, pad_f, Tout, Fout);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// im2col with separate time / frequency strides (the gate conv of GatedConv2dShared: (12,3) / (8,2) / (4,1), extractors.py:41-47)
extern "C" int mi_im2col_cl_geo_bf16(const void* in, void* col, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride_t, int stride_f, int pad_t,
                                     int pad_f, int Tout, int Fout, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || Cin <= 0 || (Cin % 8) != 0 || Tout <= 0 || Fout <= 0 || stride_t <= 0 || stride_f <= 0) return MI_ERR_ARG;
    const long total = (long)B * Tout * Fout * KH * KW * (Cin / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)col, B, Tin, Fin, Cin, KH, KW,
                       stride_t, stride_f, pad_t, pad_f, Tout, Fout);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------ context-aware front ends (extractors.py:23-65): backward pieces
// col2im gather: din[(b,ti,fi), c] (+)= sum over the taps (kh,kw) whose window (to,fo) covers (ti,fi) of dcol[(b,to,fo), (kh*KW + kw)*Cin + c]
namespace {
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ din, int B, int Tin, int Fin, int Cin, int KH, int KW,
                                                      int st_t, int st_f, int pad_t, int pad_f, int Tout, int Fout, int accumulate) {
    const int c8 = Cin / 8;
    const long total = (long)B * Tin * Fin * c8;
    const long ldcol = (long)KH * KW * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c8) * 8;
        long r = i / c8;
        const int fi = (int)(r % Fin); r /= Fin;
        const int ti = (int)(r % Tin);
        const int b = (int)(r / Tin);
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            const int nt = ti + pad_t - kh;
            if (nt < 0 || (nt % st_t) != 0) continue;
            const int to = nt / st_t;
            if (to >= Tout) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int nf = fi + pad_f - kw;
                if (nf < 0 || (nf % st_f) != 0) continue;
                const int fo = nf / st_f;
                if (fo >= Fout) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcol + (((long)b * Tout + to) * Fout + fo) * ldcol + (long)(kh * KW + kw) * Cin + cc);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += bf2f(v[j]);
            }
        }
        bf16x8* dst = reinterpret_cast<bf16x8*>(din + i * 8);
        bf16x8 o;
        if (accumulate) {
            const bf16x8 old = *dst;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += bf2f(old[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(a[j]);
        *dst = o;
    }
}

// backward of mi_gated_act_bf16: y = z * sigmoid(g), out = GELU(y).  One thread per (gate row, 8 channels); it walks the `share` conv rows of its gate row:
//   dy = dout * GELU'(y);  dz = dy * sigmoid(g);  dg = sum over the shared rows of dy * z * sigmoid(g) (1 - sigmoid(g))
__global__ __launch_bounds__(256) void gated_act_bwd_kernel(const bf16_t* __restrict__ dout, long lddo, const bf16_t* __restrict__ z, long ldz, const bf16_t* __restrict__ g, long ldg,
                                                             bf16_t* __restrict__ dz, long lddz, bf16_t* __restrict__ dg, long lddg, int B, int T, int Fq, int C, int share) {
    const int c8 = C >> 3, Tg = T / share;
    const long total = (long)B * Tg * Fq * c8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c8) * 8;
        const long grow = i / c8;
        const int f = (int)(grow % Fq);
        const long bt = grow / Fq;
        const int tg = (int)(bt % Tg), b = (int)(bt / Tg);
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + grow * ldg + cc);
        float sg[8], dga[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sg[j] = sigmoid_f(bf2f(gv[j])); dga[j] = 0.f; }
        for (int s2 = 0; s2 < share; ++s2) {
            const long row = ((long)b * T + tg * share + s2) * Fq + f;
            const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + row * ldz + cc);
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dout + row * lddo + cc);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float zz = bf2f(zv[j]);
                const float dy = bf2f(dv[j]) * gelu_erf_grad(zz * sg[j]);
                o[j] = f2bf(dy * sg[j]);
                dga[j] = fmaf(dy * zz, sg[j] * (1.f - sg[j]), dga[j]);
            }
            *reinterpret_cast<bf16x8*>(dz + row * lddz + cc) = o;
        }
        bf16x8 og;
#pragma unroll
        for (int j = 0; j < 8; ++j) og[j] = f2bf(dga[j]);
        *reinterpret_cast<bf16x8*>(dg + grow * lddg + cc) = og;
    }
}

// weight / bias gradient of a Conv2d(1 -> C) of general geometry from the gradient of its raw output: dw[c][tap] += sum_pos dy[pos][c] * x[window(pos)][tap], db[c] += sum_pos dy[pos][c].
// A thread owns two channels and all NT taps (2 NT + 2 accumulators) and walks output positions; the block's sums (position lanes added in order) are row blockIdx.x of
// `partial` ([channel][NT taps | bias]), added over the blocks in order by rows_reduce_kernel: no atomics.
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ partial,
                                                           int B, int T, int F, int C, int st_t, int st_f, int pad_t, int pad_f, int T1, int F1) {
    constexpr int NT = KH * KW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stage = reinterpret_cast<float*>(smem);       // [256][NT + 1]
    const int cp = C >> 1, ppb = 256 / cp;
    const int g = threadIdx.x % cp, pl = threadIdx.x / cp;
    float a0[NT], a1[NT], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
    const long total = (long)B * T1 * F1;
    if (pl < ppb)
    for (long pos = (long)blockIdx.x * ppb + pl; pos < total; pos += (long)gridDim.x * ppb) {
        const int f1 = (int)(pos % F1);
        const int t1 = (int)((pos / F1) % T1);
        const int b = (int)(pos / ((long)F1 * T1));
        const bf16x2 d = *reinterpret_cast<const bf16x2*>(dy + pos * C + g * 2);
        const float d0 = bf2f(d[0]), d1 = bf2f(d[1]);
        s0 += d0; s1 += d1;
        const float* xb = x + (long)b * T * F;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int t = t1 * st_t - pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int f = f1 * st_f - pad_f + kw;
                const float xv = (t >= 0 && t < T && f >= 0 && f < F) ? xb[(long)t * F + f] : 0.f;
                a0[kh * KW + kw] = fmaf(d0, xv, a0[kh * KW + kw]);
                a1[kh * KW + kw] = fmaf(d1, xv, a1[kh * KW + kw]);
            }
        }
    }
    float* prow = partial + (long)blockIdx.x * C * (NT + 1);
    float va[NT + 1], vb[NT + 1];
#pragma unroll
    for (int k = 0; k < NT; ++k) { va[k] = a0[k]; vb[k] = a1[k]; }
    va[NT] = s0; vb[NT] = s1;
    group_sum_ordered<NT + 1>(stage, va, g, pl, cp, ppb, [&](int gg, int v, float sum) { prow[(gg * 2) * (NT + 1) + v] = sum; });
    group_sum_ordered<NT + 1>(stage, vb, g, pl, cp, ppb, [&](int gg, int v, float sum) { prow[(gg * 2 + 1) * (NT + 1) + v] = sum; });
}
}  // namespace

// din (B,Tin,Fin,Cin) bf16 (+)= col2im(dcol (B*Tout*Fout, KH*KW*Cin) bf16): the input gradient of an implicit-GEMM conv from the gradient of its im2col operand
extern "C" int mi_col2im_cl_bf16(const void* dcol, void* din, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride_t, int stride_f, int pad_t, int pad_f,
                                 int Tout, int Fout, int accumulate, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || Cin <= 0 || (Cin % 8) != 0 || Tin <= 0 || Fin <= 0 || Tout <= 0 || Fout <= 0 || stride_t <= 0 || stride_f <= 0 || KH <= 0 || KW <= 0) return MI_ERR_ARG;
    const long total = (long)B * Tin * Fin * (Cin / 8);
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)din, B, Tin, Fin, Cin, KH, KW, stride_t, stride_f,
                       pad_t, pad_f, Tout, Fout, accumulate);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// backward of mi_gated_act_bf16 (plain column layout: z / dz (B*T*Fq, C), g / dg (B*(T/share)*Fq, C))
extern "C" int mi_gated_act_bwd_bf16(const void* dout, long lddo, const void* z, long ldz, const void* g, long ldg, void* dz, long lddz, void* dg, long lddg,
                                     int B, int T, int Fq, int C, int share, hipStream_t st) {
    MI_ENTER();
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (B <= 0 || T <= 0 || Fq <= 0 || C <= 0 || (C % 8) != 0 || share <= 0 || (T % share) != 0) return MI_ERR_ARG;
    if ((lddo % 8) || (ldz % 8) || (ldg % 8) || (lddz % 8) || (lddg % 8) || !al16(dout) || !al16(z) || !al16(g) || !al16(dz) || !al16(dg)) return MI_ERR_ARG;
    const long total = (long)B * (T / share) * Fq * (C / 8);
    hipLaunchKernelGGL(gated_act_bwd_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, st, (const bf16_t*)dout, lddo, (const bf16_t*)z, ldz, (const bf16_t*)g, ldg,
                       (bf16_t*)dz, lddz, (bf16_t*)dg, lddg, B, T, Fq, C, share);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// x (B,T,F) f32, dy (B,T1,F1,C) bf16 = gradient of the RAW output of a Conv2d(1 -> C, (KH,KW), strides, pads) -> dw (C, KH*KW) +=, db (C) +=.  (KH,KW) = (3,3) or (12,3).
// workspace: mi_conv2d_first_wgrad_workspace_floats(...) floats (one partial row per block; the rows are added in a fixed order)
static unsigned conv1_wgrad_grid(int B, int C, int T1, int F1) {
    const int cp = C / 2, ppb = 256 / (cp > 0 ? cp : 1);
    const long npos = (long)B * T1 * F1;
    const long nb = (npos + (ppb > 0 ? ppb : 1) - 1) / (ppb > 0 ? ppb : 1);
    return (unsigned)(nb < 1024 ? nb : 1024);
}
extern "C" size_t mi_conv2d_first_wgrad_workspace_floats(int B, int C, int KH, int KW, int T1, int F1) {
    if (B <= 0 || C <= 0 || T1 <= 0 || F1 <= 0) return 0;
    return (size_t)conv1_wgrad_grid(B, C, T1, F1) * (size_t)C * (size_t)(KH * KW + 1);
}
extern "C" int mi_conv2d_first_wgrad(const float* x, const void* dy, float* dw, float* db, int B, int T, int F, int C, int KH, int KW, int stride_t, int stride_f,
                                     int pad_t, int pad_f, int T1, int F1, float* workspace, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || C <= 0 || (C % 2) != 0 || C > 512 || T1 <= 0 || F1 <= 0 || stride_t <= 0 || stride_f <= 0 || !workspace) return MI_ERR_ARG;
    const int cp = C / 2, ppb = 256 / cp;
    if (ppb < 1) return MI_ERR_UNSUPPORTED;
    const unsigned grid = conv1_wgrad_grid(B, C, T1, F1);
    const size_t lds = (size_t)256 * (KH * KW + 1) * sizeof(float);
    if (KH == 3 && KW == 3)
        hipLaunchKernelGGL((conv1_wgrad_kernel<3, 3>), dim3(grid), dim3(256), lds, st, x, (const bf16_t*)dy, workspace, B, T, F, C, stride_t, stride_f, pad_t, pad_f, T1, F1);
    else if (KH == 12 && KW == 3)
        hipLaunchKernelGGL((conv1_wgrad_kernel<12, 3>), dim3(grid), dim3(256), lds, st, x, (const bf16_t*)dy, workspace, B, T, F, C, stride_t, stride_f, pad_t, pad_f, T1, F1);
    else return MI_ERR_UNSUPPORTED;
    MI_CHECK_LAUNCH();
    rows_reduce_launch(workspace, (int)grid, C * (KH * KW + 1), EmitConv1{dw, db, KH * KW}, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// x (B,T,F) f32, w (C,9), bias (C); dcol (B*T2*F2, K2*K2*C) bf16 = gradient of conv2's im2col operand -> dw (C,9) +=, db (C) +=
// workspace: mi_conv2d_first_bwd_workspace_floats(B, C, T1, F1) floats (one partial row of 10 C floats per block; the rows are added in a fixed order)
static unsigned conv1_bwd_grid(int B, int C, int T1, int F1) {
    const int cgs = C / 8, ppb = 256 / (cgs > 0 ? cgs : 1);
    const long npos = (long)B * T1 * F1;
    const long nb = (npos + (ppb > 0 ? ppb : 1) - 1) / (ppb > 0 ? ppb : 1);
    return (unsigned)(nb < 2048 ? nb : 2048);
}
extern "C" size_t mi_conv2d_first_bwd_workspace_floats(int B, int C, int T1, int F1) {
    if (B <= 0 || C <= 0 || T1 <= 0 || F1 <= 0) return 0;
    return (size_t)conv1_bwd_grid(B, C, T1, F1) * (size_t)C * 10;
}
extern "C" int mi_conv2d_first_bwd(const float* x, const float* w, const float* bias, const void* dcol, float* dw, float* db,
                                   int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f, int T1, int F1,
                                   int K2, int stride2, int pad2_t, int pad2_f, int T2, int F2, float* workspace, hipStream_t st) {
    MI_ENTER();
    const int cgs = C / 8;
    if (B <= 0 || K != 3 || (C % 8) != 0 || cgs > 256 || stride2 <= 0) return MI_ERR_UNSUPPORTED;
    if (!workspace) return MI_ERR_ARG;
    const long npos = (long)B * T1 * F1;
    const unsigned grid = conv1_bwd_grid(B, C, T1, F1);
    const size_t lds = (size_t)256 * 40 * sizeof(float);
    if (K2 == 3 && stride2 == 2 && npos < (1L << 31))
        hipLaunchKernelGGL(conv1_bwd3_kernel<true>, dim3(grid), dim3(256), lds, st, x, w, bias, (const bf16_t*)dcol, workspace,
                           B, T, F, C, stride, pad_t, pad_f, T1, F1, K2, stride2, pad2_t, pad2_f, T2, F2, Conv1Ph{});
    else
        hipLaunchKernelGGL(conv1_bwd3_kernel<false>, dim3(grid), dim3(256), lds, st, x, w, bias, (const bf16_t*)dcol, workspace,
                           B, T, F, C, stride, pad_t, pad_f, T1, F1, K2, stride2, pad2_t, pad2_f, T2, F2, Conv1Ph{});
    MI_CHECK_LAUNCH();
    rows_reduce_launch(workspace, (int)grid, C * 10, EmitConv1{dw, db, 9}, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------ conv2's data gradient as four stride-1 convolutions
// A 3x3 / stride-2 Conv2d reads input position t1 with tap kh iff (t1 + pad - kh) is even: along each axis the parity pt = (t1 + pad) & 1 decides WHICH taps (pt = 0:
// kh in {0, 2}; pt = 1: kh = 1), and with t1 + pad = 2u + pt the contributing outputs are t2 = u - i for kh = pt + 2i.  So
//     dact1[b, t1, f1, c] = sum_{i, j, co} dY2[b, u - i, v - j, co] * W2[co][pt + 2i][pf + 2j][c]
// is, per parity class (pt, pf), a dense stride-1 convolution of dY2 (B, T2, F2, C2) with a (2 - pt) x (2 - pf) kernel: four implicit GEMMs on the forward conv kernel
// (gemm_8p.hip, the operand gathered by the LDS-DMA addresses), M = B * U * V rows each, K = taps * C2 — the same 9 C1 C2 MACs per output position of conv2 in total.
// What it replaces: dcol = dY2 · W2 as ONE GEMM into the (B*T2*F2, 9 C1) gradient of the im2col operand (4.4 GB at BASELINE config 3) which conv1's backward then
// gathered back (col2im) — now 4 phase buffers of together B*T1*F1*C1 elements (2.0 GB), each element written once and read once.
extern "C" int mi_conv2d_cl_geo_bf16(const void* in, const void* weight, const float* bias, void* out, int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW,
                                     int stride_t, int stride_f, int pad_t, int pad_f, int Tout, int Fout, int act, int gated, hipStream_t stream);      // gemm_bf16.hip
namespace {
struct S2Axis { int lo[2], n[2], pad[2]; bool ok; };
S2Axis s2_axis(int L1, int p, int L2) {
    S2Axis a{};
    a.ok = L1 > 0 && L2 > 0 && p >= 0;
    for (int pt = 0; pt < 2 && a.ok; ++pt) {
        const int nt = pt == 0 ? 2 : 1;
        const int lo = p - pt > 0 ? (p - pt + 1) / 2 : 0;             // first u with t1 = 2u + pt - p >= 0
        const int hi = (L1 - 1 + p - pt) / 2;                          // last u with t1 <= L1 - 1   (L1 - 1 + p - pt >= 0 whenever lo <= hi)
        a.lo[pt] = lo; a.n[pt] = hi - lo + 1; a.pad[pt] = nt - 1 - lo;
        // the stride-1 conv reads input row u' - pad + i', i' in [0, nt): a non-negative leading pad, and the first tap of the last row inside the input
        a.ok = a.n[pt] > 0 && a.pad[pt] >= 0 && (L1 - 1 + p - pt) >= 0 && (a.n[pt] - 1 - a.pad[pt]) < L2;
    }
    return a;
}
// packed[phase] (C1, taps * C2): row c, column (i' * ntf + j') * C2 + co = wT[((kh * 3 + kw) * C1 + c)][co], kh = pt + 2 (ntt - 1 - i'), kw = pf + 2 (ntf - 1 - j')
__global__ __launch_bounds__(256) void s2k3_pack_kernel(const bf16_t* __restrict__ wT, long ldwt, bf16_t* __restrict__ packed, int C1, int C2) {
    const int c8 = C2 >> 3;
    const long total = (long)9 * C1 * c8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int cc = (int)(e % c8);
        long r = e / c8;
        const int c = (int)(r % C1);
        const int tap = (int)(r / C1), kh = tap / 3, kw = tap % 3;
        const int pt = kh & 1, pf = kw & 1, ntt = 2 - pt, ntf = 2 - pf;
        const int ip = ntt - 1 - (kh >> 1), jp = ntf - 1 - (kw >> 1);
        const int phase = pt * 2 + pf;
        const long base = phase == 0 ? 0 : phase == 1 ? (long)4 * C1 * C2 : phase == 2 ? (long)6 * C1 * C2 : (long)8 * C1 * C2;
        const long K = (long)ntt * ntf * C2;
        *reinterpret_cast<bf16x8*>(packed + base + c * K + (long)(ip * ntf + jp) * C2 + cc * 8) =
            *reinterpret_cast<const bf16x8*>(wT + ((long)tap * C1 + c) * ldwt + cc * 8);
    }
}
}  // namespace

// elements of the phase-buffer allocation (bf16) for a (B, T1, F1, C1) activation gradient; 0 = this geometry is not supported (the caller keeps the im2col-gradient path)
extern "C" size_t mi_conv2d_s2k3_dgrad_elems(int B, int T1, int F1, int C1, int T2, int F2, int pad_t, int pad_f) {
    const S2Axis at = s2_axis(T1, pad_t, T2), af = s2_axis(F1, pad_f, F2);
    if (!at.ok || !af.ok || B <= 0 || C1 <= 0) return 0;
    size_t n = 0;
    for (int pt = 0; pt < 2; ++pt)
        for (int pf = 0; pf < 2; ++pf) n += (size_t)B * at.n[pt] * af.n[pf] * C1;
    return n;
}
// wT (9 * C1 rows (kh, kw, c), >= C2 columns co; row stride ldwt) bf16 — the transposed copy of conv2's weight the trainer keeps for the data gradient — into the four
// phase weight matrices, back to back: 9 * C1 * C2 elements
extern "C" int mi_conv2d_s2k3_dgrad_pack_bf16(const void* wT, long ldwt, void* packed, int C1, int C2, hipStream_t st) {
    MI_ENTER();
    if (C1 <= 0 || C2 <= 0 || (C2 % 8) || (ldwt % 8) || ldwt < C2 || !wT || !packed) return MI_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(wT) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15)) return MI_ERR_ARG;
    hipLaunchKernelGGL(s2k3_pack_kernel, dim3(grid_for((long)9 * C1 * (C2 / 8), 1024)), dim3(256), 0, st, (const bf16_t*)wT, ldwt, (bf16_t*)packed, C1, C2);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// dY2 (B, T2, F2, C2) bf16 contiguous, packed = mi_conv2d_s2k3_dgrad_pack_bf16's output -> phases (mi_conv2d_s2k3_dgrad_elems elements): the buffers of
// (pt, pf) = (0,0), (0,1), (1,0), (1,1) back to back, each (B, U_pt, V_pf, C1) with U / V the number of positions of that parity.  replaces: the input gradient autograd
// derives for the second Conv2d of extractors.py:82-89.
extern "C" int mi_conv2d_s2k3_dgrad_bf16(const void* dY2, const void* packed, void* phases, int B, int T1, int F1, int C1, int T2, int F2, int C2,
                                         int pad_t, int pad_f, hipStream_t st) {
    MI_ENTER();
    const S2Axis at = s2_axis(T1, pad_t, T2), af = s2_axis(F1, pad_f, F2);
    if (!at.ok || !af.ok) return MI_ERR_UNSUPPORTED;
    if (B <= 0 || C1 <= 0 || C2 <= 0 || !dY2 || !packed || !phases) return MI_ERR_ARG;
    const bf16_t* wp = (const bf16_t*)packed;
    bf16_t* op = (bf16_t*)phases;
    for (int pt = 0; pt < 2; ++pt)
        for (int pf = 0; pf < 2; ++pf) {
            const int ntt = 2 - pt, ntf = 2 - pf;
            const int rc = mi_conv2d_cl_geo_bf16(dY2, wp, nullptr, op, B, T2, F2, C2, C1, ntt, ntf, 1, 1, at.pad[pt], af.pad[pf], at.n[pt], af.n[pf], 0, 0, st);
            if (rc != MI_OK) return rc;
            wp += (long)C1 * ntt * ntf * C2;
            op += (long)B * at.n[pt] * af.n[pf] * C1;
        }
    return MI_OK;
}
// mi_conv2d_first_bwd with the activation gradient in phase buffers (conv2: 3x3, stride 2, leading pads pad2_t / pad2_f)
extern "C" int mi_conv2d_first_bwd_phases(const float* x, const float* w, const float* bias, const void* phases, float* dw, float* db,
                                          int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f, int T1, int F1,
                                          int pad2_t, int pad2_f, int T2, int F2, float* workspace, hipStream_t st) {
    MI_ENTER();
    const int cgs = C / 8;
    const long npos = (long)B * T1 * F1;
    if (B <= 0 || K != 3 || (C % 8) != 0 || cgs > 256 || npos >= (1L << 31)) return MI_ERR_UNSUPPORTED;
    const S2Axis at = s2_axis(T1, pad2_t, T2), af = s2_axis(F1, pad2_f, F2);
    if (!at.ok || !af.ok) return MI_ERR_UNSUPPORTED;
    if (!workspace || !phases || (reinterpret_cast<uintptr_t>(phases) & 15)) return MI_ERR_ARG;
    Conv1Ph ph{};
    const bf16_t* op = (const bf16_t*)phases;
    for (int pt = 0; pt < 2; ++pt) {
        ph.U[pt] = at.n[pt]; ph.ulo[pt] = at.lo[pt]; ph.V[pt] = af.n[pt]; ph.vlo[pt] = af.lo[pt];
        for (int pf = 0; pf < 2; ++pf) { ph.buf[pt * 2 + pf] = op; op += (long)B * at.n[pt] * af.n[pf] * C; }
    }
    unsigned grid = conv1_bwd_grid(B, C, T1, F1);
    if (C == 256) {                                        // the matrix-core form (two position tiles per block at a time, 128 channels per wave)
        const long ntiles = (npos + 31) / 32;
        const unsigned g2 = (unsigned)((ntiles + 1) / 2 < 512 ? (ntiles + 1) / 2 : 512);
        if (g2 < grid) grid = g2;                          // never more partial rows than the workspace holds
        hipLaunchKernelGGL(conv1_bwd3_mfma_kernel, dim3(grid), dim3(256), (size_t)4 * C1B_WAVE_LDS, st, x, w, bias, workspace,
                           B, T, F, stride, pad_t, pad_f, T1, F1, pad2_t, pad2_f, ph);
        MI_CHECK_LAUNCH();
        rows_reduce_launch(workspace, (int)grid, C * 10, EmitConv1{dw, db, 9}, st);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const size_t lds = (size_t)256 * 40 * sizeof(float);
    hipLaunchKernelGGL((conv1_bwd3_kernel<true, true>), dim3(grid), dim3(256), lds, st, x, w, bias, (const bf16_t*)nullptr, workspace,
                       B, T, F, C, stride, pad_t, pad_f, T1, F1, 3, 2, pad2_t, pad2_f, T2, F2, ph);
    MI_CHECK_LAUNCH();
    rows_reduce_launch(workspace, (int)grid, C * 10, EmitConv1{dw, db, 9}, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
