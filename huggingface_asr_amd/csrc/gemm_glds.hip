// LDS-DMA pipelined bf16 MFMA GEMM for gfx950 (fast path of mi_gemm_bf16 / mi_conv2d_cl_bf16, K % 64 == 0).
//
//   C[M,N] = epi(A[M,K] · W[N,K]^T)        block tile 128x128x64, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA 32x32x16
//
// Staging: `global_load_lds_dwordx4` (HBM/L2 -> LDS without VGPRs) into a ring of STAGES K-tiles; each wave issues
// 8 x 1 KiB pieces per K-tile (4 for A, 4 for W).  The LDS image is lane-linear per piece (8 rows x 128 B), so the
// bank-conflict swizzle (16-B chunk ^ ((row>>1)&7), conflict-free for ds_read_b128) is applied to the per-lane SOURCE
// address and again on the read (guide §5.4 rule 21).  Tiles stay in flight across the single raw s_barrier per K-tile:
// counted `s_waitcnt vmcnt(8)` (3-stage ring: tile t+1 keeps flying while tile t is consumed) — never vmcnt(0) in the
// steady state.  Rows beyond M / N are clamped to a valid row (their results are never stored); conv zero padding reads
// from a 16-B zero page.  Epilogue: accumulators (+bias, GELU) are staged through the now-free LDS ring as fp32 and leave
// as full 16-B-per-lane rows (bf16: 8 elements per store), with the residual read the same way.
#include "gemm_args.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;       // 32 KiB
constexpr int A_BYTES = BM * BK * 2;

__device__ __attribute__((aligned(16))) uint4 g_zero_page = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int fsw(int row) { return (row >> 1) & 7; }


// Epilogue out-phase shared by both kernels: `ct` holds ROWS x TBN raw fp32 accumulators; every thread of the block
// applies bias / GELU / residual to whole 8-column chunks and stores 16 B (bf16) or 2 x 16 B (fp32) per chunk.  Residual
// values are all loaded before the first store (resid may alias C, which would otherwise serialise load->store chains).
template <int ROWS, int TBN, int NTHR>
__device__ __forceinline__ void epilogue_out(const GemmArgs& p, const float* ct, int m_base, int n0, int tid) {
    const bool vec_ok = (p.col_T == 0) && (n0 + TBN <= p.N) && ((p.bias_mode != 1) || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
                        (p.out_f32 ? ((p.ldc & 3) == 0) : ((p.ldc & 7) == 0)) && (!p.resid || (p.ldr & 3) == 0);
    if (vec_ok) {
        constexpr int CH = TBN / 8;
        constexpr int ITER = (ROWS * CH + NTHR - 1) / NTHR;
        f32x4 r0[ITER], r1[ITER];
        if (p.resid) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int id = tid + it * NTHR;
                const int rl = id / CH, ch = id % CH;
                const int m = m_base + rl, n = n0 + ch * 8;
                if (id < ROWS * CH && m < p.M) {
                    r0[it] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n);
                    r1[it] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n + 4);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int id = tid + it * NTHR;
            const int rl = id / CH, ch = id % CH;
            const int m = m_base + rl, n = n0 + ch * 8;
            if (id >= ROWS * CH || m >= p.M) continue;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ct + rl * TBN + ch * 8);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ct + rl * TBN + ch * 8 + 4);
            if (p.bias_mode == 1) {
                v0 += *reinterpret_cast<const f32x4*>(p.bias + n);
                v1 += *reinterpret_cast<const f32x4*>(p.bias + n + 4);
            } else if (p.bias_mode == 2) {
                const float bm = p.bias[m];
                v0 += bm; v1 += bm;
            }
            if (p.act == 1) {
                v0 = f32x4{gelu_erf(v0.x), gelu_erf(v0.y), gelu_erf(v0.z), gelu_erf(v0.w)};
                v1 = f32x4{gelu_erf(v1.x), gelu_erf(v1.y), gelu_erf(v1.z), gelu_erf(v1.w)};
            }
            if (p.resid) { v0 = r0[it] + p.alpha * v0; v1 = r1[it] + p.alpha * v1; }
            if (p.out_f32) {
                float* o = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            } else {
                bf16x8 o = {f2bf(v0.x), f2bf(v0.y), f2bf(v0.z), f2bf(v0.w), f2bf(v1.x), f2bf(v1.y), f2bf(v1.z), f2bf(v1.w)};
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = o;
            }
        }
    } else {
        for (int id = tid; id < ROWS * TBN; id += NTHR) {
            const int rl = id / TBN, cl = id % TBN;
            const int m = m_base + rl, n = n0 + cl;
            if (m >= p.M || n >= p.N) continue;
            const long nc = p.col_T ? (long)(n / p.col_T) * p.col_Tp + (n % p.col_T) : n;
            float v = ct[rl * TBN + cl];
            if (p.bias_mode == 1) v += p.bias[n];
            else if (p.bias_mode == 2) v += p.bias[m];
            if (p.act == 1) v = gelu_erf(v);
            if (p.resid) v = p.resid[(long)m * p.ldr + n] + p.alpha * v;
            if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + nc] = v;
            else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + nc] = f2bf(v);
        }
    }
}

// Symmetric kernel: every wave both issues its share of the LDS-DMA pieces and runs MFMAs.
//   <128,128,2,2,S=2>: 4 waves, 64 KiB ring -> two blocks per CU (the second block's main loop covers the first one's
//                      prologue/epilogue);  <256,256,2,4,S=2>: 8 waves (wave tile 128x64), 128 KiB ring, half the L2->LDS
//                      bytes per flop — for GEMMs whose 256x256 tile count still fills the chip.
template <int TBM, int TBN, int CWM, int CWN, int STAGES, bool CONV>
__global__ __launch_bounds__(CWM * CWN * 64, (TBM == 128 && TBN == 128 && STAGES == 2) ? 2 : 1) void gemm_glds_kernel(GemmArgs p) {
    constexpr int NW = CWM * CWN, NTH = NW * 64;
    constexpr int WM = TBM / CWM, WN = TBN / CWN, MI = WM / 32, NI = WN / 32;
    constexpr int STG = (TBM + TBN) * BK * 2, ABYTES = TBM * BK * 2;
    constexpr int PPW = (TBM + TBN) / 8 / NW;          // 1-KiB pieces per wave per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CWN, wn = wave % CWN;

    const int ntm = (p.M + TBM - 1) / TBM, ntn = (p.N + TBN - 1) / TBN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the blocks of one XCD walk N fastest within an A row panel
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * TBM, n0 = tn * TBN;

    // ---- per-lane source pointers of this wave's pieces (piece = 8 rows x 128 B; pieces [0, TBM/8) are A rows)
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* src[PPW];
    int c_ti[PPW], c_fi[PPW], c_lc[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int g = wave * PPW + q;
        const bool isA = g < TBM / 8;
        const int row = (isA ? g : g - TBM / 8) * 8 + prow;
        const int lc = pc ^ fsw(row);
        if (isA) {
            const int m = min(m0 + row, p.M - 1);
            if (CONV) {
                const int fo = m % p.Fout, to = (m / p.Fout) % p.Tout, b = m / (p.Fout * p.Tout);
                c_ti[q] = to * p.stride - p.pad_t;
                c_fi[q] = fo * p.stride - p.pad_f;
                c_lc[q] = lc * 8;
                src[q] = p.A + (long)b * p.Tin * p.Fin * p.Cin;
            } else {
                src[q] = p.A + (long)m * p.lda + lc * 8;
            }
        } else {
            const int n = min(n0 + row, p.N - 1);
            src[q] = p.W + (long)n * p.ldw + lc * 8;
        }
    }
    auto issue = [&](int kt, int stage) {
        char* sbase = smem + stage * STG + wave * PPW * 1024;
        int kh = 0, kw = 0, c0 = 0;
        if (CONV) {                                        // Cin % 64 == 0: one tap per K tile
            const int k0 = kt * BK;
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int g = wave * PPW + q;
            const bf16_t* sp;
            if (CONV && g < TBM / 8) {
                const int ti = c_ti[q] + kh, fi = c_fi[q] + kw;
                const bool ok = ti >= 0 && ti < p.Tin && fi >= 0 && fi < p.Fin;
                sp = ok ? src[q] + ((long)ti * p.Fin + fi) * p.Cin + c0 + c_lc[q] : reinterpret_cast<const bf16_t*>(&g_zero_page);
            } else {
                sp = src[q] + kt * BK;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(sbase + q * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int lr = lane & 31, lh = lane >> 5;
    const int krot = p.krot ? (int)((unsigned)(tm * 5 + tn * 3) % (unsigned)nk) : 0;
    auto ktile = [&](int t) { int v = t + krot; return v >= nk ? v - nk : v; };

    issue(ktile(0), 0);
    if (STAGES == 3 && nk > 1) issue(ktile(1), 1);

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt % STAGES;
        // tile kt has landed for THIS wave's pieces once at most the younger tile's pieces are outstanding
        if (STAGES == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");    // every wave's pieces of tile kt landed; stage of tile kt-1 is free
        if (kt + STAGES - 1 < nk && !(p.dbg & 2)) issue(ktile(kt + STAGES - 1), (kt + STAGES - 1) % STAGES);
        if (p.dbg & 1) continue;
        const char* a = smem + stage * STG;
        const char* b = a + ABYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WM + i * 32 + lr;
                fa[i] = *reinterpret_cast<const bf16x8*>(a + row * 128 + (((ks * 2 + lh) ^ fsw(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + lr;
                fb[j] = *reinterpret_cast<const bf16x8*>(b + col * 128 + (((ks * 2 + lh) ^ fsw(col)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                               // all waves done with the ring: reuse it as the fp32 C tile

    // ---- epilogue: raw accumulators through LDS in bands of 128 rows, then coalesced rows out (bias/GELU/residual fused)
    float* ct = reinterpret_cast<float*>(smem);
    constexpr int BAND = (TBN <= 128) ? 128 : 64;     // band x TBN x 4 B <= 64 KiB
#pragma unroll 1
    for (int band = 0; band < TBM / BAND; ++band) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cl = wn * WN + j * 32 + lr;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (rl / BAND == band) ct[(rl - band * BAND) * TBN + cl] = acc[i][j][r];
                }
            }
        }
        __syncthreads();
        epilogue_out<BAND, TBN, NTH>(p, ct, m0 + band * BAND, n0, tid);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Loader/consumer variant: NL loader waves do nothing but issue LDS-DMA pieces (an LDS-DMA costs ~60-100 cycles of the
// issuing wave, MI355X_MICROARCH.md), CWM x CWN consumer waves do nothing but ds_read + MFMA.  Each SIMD then holds a
// consumer next to a loader, so DMA issue overlaps MFMA execution instead of serialising with it (one wave per SIMD
// doing both was the limiter of the symmetric kernel above at one block per CU).  One s_barrier per K tile, S-deep ring.
template <int TBM, int TBN, int CWM, int CWN, int NL, int S, bool CONV>
__global__ __launch_bounds__((CWM * CWN + NL) * 64, (S == 2 && TBM == 128) ? 4 : 1) void gemm_lc_kernel(GemmArgs p) {
    constexpr int NC = CWM * CWN;                     // consumer waves
    constexpr int WM = TBM / CWM, WN = TBN / CWN;     // consumer wave tile
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int STG = (TBM + TBN) * BK * 2;
    constexpr int ABYTES = TBM * BK * 2;
    constexpr int PIECES = (TBM + TBN) / 8;           // 1 KiB pieces per K tile
    constexpr int PPW = PIECES / NL;                  // per loader wave
    constexpr int APW = (TBM / 8) / NL;               // A pieces per loader wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool is_loader = wave >= NC;

    const int ntm = (p.M + TBM - 1) / TBM, ntn = (p.N + TBN - 1) / TBN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int nk = p.K / BK;

    f32x16 acc[MI][NI];      // defined on the consumer path only (keeps the loader path's register pressure low)

    if (is_loader) {
        const int lw = wave - NC;
        const int prow = lane >> 3, pc = lane & 7;
        // piece q of this wave: global piece index g = lw * PPW + q ; g < TBM/8 -> A rows, else W rows
        const bf16_t* src[PPW];
        int c_ti[PPW], c_fi[PPW], c_lc[PPW];
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int g = lw * PPW + q;
            const bool isA = g < TBM / 8;
            const int row = (isA ? g : g - TBM / 8) * 8 + prow;
            const int lc = pc ^ fsw(row);
            if (isA) {
                const int m = min(m0 + row, p.M - 1);
                if (CONV) {
                    const int fo = m % p.Fout, to = (m / p.Fout) % p.Tout, b = m / (p.Fout * p.Tout);
                    c_ti[q] = to * p.stride - p.pad_t;
                    c_fi[q] = fo * p.stride - p.pad_f;
                    c_lc[q] = lc * 8;
                    src[q] = p.A + (long)b * p.Tin * p.Fin * p.Cin;
                } else {
                    src[q] = p.A + (long)m * p.lda + lc * 8;
                }
            } else {
                const int n = min(n0 + row, p.N - 1);
                src[q] = p.W + (long)n * p.ldw + lc * 8;
            }
        }
        auto issue = [&](int kt, int stage) {
            char* sbase = smem + stage * STG + lw * PPW * 1024;
            int kh = 0, kw = 0, c0 = 0;
            if (CONV) {
                const int k0 = kt * BK;
                const int tap = k0 / p.Cin;
                c0 = k0 - tap * p.Cin;
                kh = tap / p.KW;
                kw = tap - kh * p.KW;
            }
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                const int g = lw * PPW + q;
                const bf16_t* sp;
                if (CONV && g < TBM / 8) {
                    const int ti = c_ti[q] + kh, fi = c_fi[q] + kw;
                    const bool ok = ti >= 0 && ti < p.Tin && fi >= 0 && fi < p.Fin;
                    sp = ok ? src[q] + ((long)ti * p.Fin + fi) * p.Cin + c0 + c_lc[q] : reinterpret_cast<const bf16_t*>(&g_zero_page);
                } else {
                    sp = src[q] + kt * BK;
                }
                __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(sbase + q * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < S - 1; ++t)
            if (t < nk) issue(t, t);
        for (int kt = 0; kt < nk; ++kt) {
            // tiles kt+1 .. kt+S-2 may stay in flight; tile kt must have landed
            if (S >= 3 && kt + S - 2 < nk) {
                if (S == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            if (kt + S - 1 < nk && !(p.dbg & 2)) issue(kt + S - 1, (kt + S - 1) % S);
        }
    } else {
        const int wm = wave / CWN, wn = wave % CWN;
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_barrier" ::: "memory");
            if (p.dbg & 1) continue;
            const char* a = smem + (kt % S) * STG;
            const char* b = a + ABYTES;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * WM + i * 32 + lr;
                    fa[i] = *reinterpret_cast<const bf16x8*>(a + row * 128 + (((ks * 2 + lh) ^ fsw(row)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = wn * WN + j * 32 + lr;
                    fb[j] = *reinterpret_cast<const bf16x8*>(b + col * 128 + (((ks * 2 + lh) ^ fsw(col)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: raw fp32 accumulators through LDS in bands of 128 rows (the ring is free now); all waves store
    float* ct = reinterpret_cast<float*>(smem);
    constexpr int NTHR = (NC + NL) * 64;
    constexpr int BAND = 128;
#pragma unroll 1
    for (int band = 0; band < TBM / BAND; ++band) {
        if (!is_loader) {
            const int wm = wave / CWN, wn = wave % CWN;
            const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * WN + j * 32 + lr;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int rbase = wm * WM + i * 32;
                    if (rbase / BAND != band) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = rbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        ct[(rl - band * BAND) * TBN + cl] = acc[i][j][r];
                    }
                }
            }
        }
        __syncthreads();
        epilogue_out<BAND, TBN, NTHR>(p, ct, m0 + band * BAND, n0, tid);
        __syncthreads();
    }
}

int g_variant = 0;   // 0: symmetric kernel above; 1: loader/consumer 128x128; 2: loader/consumer 256x128 where it fills the chip
int g_stages = 2;
int g_dbg = 0;
int g_krot = 0;   // measured: rotating the K loop start per block does not help (no channel camping on these shapes)   // tuning knob (mi_gemm_set_stages), default chosen from measurements

}  // namespace

extern "C" void mi_gemm_set_stages(int stages) { g_stages = (stages == 2) ? 2 : 3; }
extern "C" void mi_gemm_set_krot(int on) { g_krot = on; }
extern "C" void mi_gemm_set_variant(int v) { g_variant = v; }
extern "C" void mi_gemm_set_debug(int v) { g_dbg = v; }   // timing experiments only: 1 = no MFMA, 2 = no loads (wrong results)

bool gemm_glds_supported(const GemmArgs& a, bool conv) {
    if ((a.K % BK) != 0 || a.M <= 0 || a.N <= 0) return false;
    if (((uintptr_t)a.W & 15) || (a.ldw % 8) != 0) return false;
    if (conv) { if ((a.Cin % BK) != 0 || ((uintptr_t)a.A & 15)) return false; }
    else if (((uintptr_t)a.A & 15) || (a.lda % 8) != 0) return false;
    if (((uintptr_t)a.C & 15) || (a.resid && ((uintptr_t)a.resid & 15))) return false;
    return true;
}

int gemm_glds_launch(const GemmArgs& a_in, bool conv, hipStream_t stream) {
    if (!gemm_glds_supported(a_in, conv)) return MI_ERR_UNSUPPORTED;
    GemmArgs a = a_in;
    a.krot = g_krot;
    a.dbg = g_dbg;
    if (g_variant >= 1 && g_variant <= 3) {
        const bool big = g_variant >= 2 && cdiv(a.M, 256) * cdiv(a.N, 128) >= 224;
        if (big) {
            constexpr int S = 3;
            const size_t l = (size_t)S * (256 + 128) * BK * 2;
            const int g = cdiv(a.M, 256) * cdiv(a.N, 128);
            if (conv) hipLaunchKernelGGL((gemm_lc_kernel<256, 128, 4, 2, 4, S, true>), dim3(g), dim3(768), l, stream, a);
            else hipLaunchKernelGGL((gemm_lc_kernel<256, 128, 4, 2, 4, S, false>), dim3(g), dim3(768), l, stream, a);
        } else if (g_variant == 3) {                       // 2-deep ring: 64 KiB -> two blocks (16 waves) per CU
            constexpr int S = 2;
            const size_t l = (size_t)S * STAGE_BYTES;
            const int g = cdiv(a.M, 128) * cdiv(a.N, 128);
            if (conv) hipLaunchKernelGGL((gemm_lc_kernel<128, 128, 2, 2, 4, S, true>), dim3(g), dim3(512), l, stream, a);
            else hipLaunchKernelGGL((gemm_lc_kernel<128, 128, 2, 2, 4, S, false>), dim3(g), dim3(512), l, stream, a);
        } else {
            constexpr int S = 3;
            const size_t l = (size_t)S * STAGE_BYTES;
            const int g = cdiv(a.M, 128) * cdiv(a.N, 128);
            if (conv) hipLaunchKernelGGL((gemm_lc_kernel<128, 128, 2, 2, 4, S, true>), dim3(g), dim3(512), l, stream, a);
            else hipLaunchKernelGGL((gemm_lc_kernel<128, 128, 2, 2, 4, S, false>), dim3(g), dim3(512), l, stream, a);
        }
        return MI_OK;
    }
    // symmetric kernels: 256x256 tiles when they still give every CU a block, else 128x128 at two blocks per CU
    const int t256 = cdiv(a.M, 256) * cdiv(a.N, 256);
    if (g_variant != 5 && t256 >= 200 && (a.N % 256) == 0) {
        const size_t l = (size_t)2 * (256 + 256) * BK * 2;
        if (conv) hipLaunchKernelGGL((gemm_glds_kernel<256, 256, 2, 4, 2, true>), dim3(t256), dim3(512), l, stream, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<256, 256, 2, 4, 2, false>), dim3(t256), dim3(512), l, stream, a);
        return MI_OK;
    }
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int stages = g_stages;
    const size_t lds = (size_t)(stages == 2 ? 2 : 3) * STAGE_BYTES;       // >= 64 KiB: also holds the fp32 C tile
    if (stages == 2) {
        if (conv) hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 2, true>), dim3(grid), dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 2, false>), dim3(grid), dim3(NT), lds, stream, a);
    } else {
        if (conv) hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 3, true>), dim3(grid), dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 3, false>), dim3(grid), dim3(NT), lds, stream, a);
    }
    return MI_OK;
}
