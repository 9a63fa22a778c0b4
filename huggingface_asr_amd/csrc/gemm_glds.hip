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

template <int STAGES, bool CONV>
__global__ __launch_bounds__(NT, STAGES == 2 ? 2 : 1) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the blocks of one XCD walk N fastest within an A row panel
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane source pointers of this wave's 4+4 pieces (piece = 8 rows x 128 B)
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
    int a_ti[4], a_fi[4], a_lc[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + prow;
        const int lc = pc ^ fsw(row);                       // logical chunk this lane fetches
        const int m = min(m0 + row, p.M - 1);
        if (CONV) {
            const int fo = m % p.Fout, to = (m / p.Fout) % p.Tout, b = m / (p.Fout * p.Tout);
            a_ti[i] = to * p.stride - p.pad_t;
            a_fi[i] = fo * p.stride - p.pad_f;
            a_lc[i] = lc * 8;
            asrc[i] = p.A + (long)b * p.Tin * p.Fin * p.Cin;
            a_ok[i] = true;
        } else {
            asrc[i] = p.A + (long)m * p.lda + lc * 8;
        }
        const int n = min(n0 + row, p.N - 1);
        wsrc[i] = p.W + (long)n * p.ldw + lc * 8;
    }

    auto issue = [&](int kt, int stage) {
        char* sbase = smem + stage * STAGE_BYTES + (wave * 4) * 1024;
        int kh = 0, kw = 0, c0 = 0;
        if (CONV) {                                        // Cin % 64 == 0: one tap per K tile
            const int k0 = kt * BK;
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src;
            if (CONV) {
                const int ti = a_ti[i] + kh, fi = a_fi[i] + kw;
                const bool ok = ti >= 0 && ti < p.Tin && fi >= 0 && fi < p.Fin;
                src = ok ? asrc[i] + ((long)ti * p.Fin + fi) * p.Cin + c0 + a_lc[i]
                         : reinterpret_cast<const bf16_t*>(&g_zero_page);
            } else {
                src = asrc[i] + kt * BK;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(sbase + A_BYTES + i * 1024), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int lr = lane & 31, lh = lane >> 5;
    // K-loop rotation: blocks start at different K tiles so that concurrently running blocks do not all stream the
    // same 128-B column slab (same address bits -> same L2/HBM channels) at the same time.
    const int krot = p.krot ? (int)((unsigned)(tm * 5 + tn * 3) % (unsigned)nk) : 0;
    auto ktile = [&](int t) { int v = t + krot; return v >= nk ? v - nk : v; };

    issue(ktile(0), 0);
    if (STAGES == 3 && nk > 1) issue(ktile(1), 1);

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt % STAGES;
        // tile kt has landed for THIS wave's pieces once at most the younger tile's 8 pieces are outstanding
        if (STAGES == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");    // every wave's pieces of tile kt landed; stage of tile kt-1 is free
        if (kt + STAGES - 1 < nk) issue(ktile(kt + STAGES - 1), (kt + STAGES - 1) % STAGES);
        const char* a = smem + stage * STAGE_BYTES;
        const char* b = a + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + lr;
                fa[i] = *reinterpret_cast<const bf16x8*>(a + row * 128 + (((ks * 2 + lh) ^ fsw(row)) << 4));
                const int col = wn * 64 + i * 32 + lr;
                fb[i] = *reinterpret_cast<const bf16x8*>(b + col * 128 + (((ks * 2 + lh) ^ fsw(col)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                               // all waves done with the ring: reuse it as the fp32 C tile

    // ---- epilogue part 1: acc (+ per-column bias, activation) -> LDS fp32 [128][128]
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cl = wn * 64 + j * 32 + lr;
        const int n = n0 + cl;
        const float bcol = (p.bias_mode == 1 && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bcol;
                if (p.bias_mode == 2) v += p.bias[min(m0 + rl, p.M - 1)];
                if (p.act == 1) v = gelu_erf(v);
                ct[rl * BN + cl] = v;
            }
    }
    __syncthreads();

    // ---- epilogue part 2: coalesced rows out (16 B per lane) with the residual folded in
    const bool vec_ok = (p.col_T == 0) && (n0 + BN <= p.N) &&
                        (p.out_f32 ? ((p.ldc & 3) == 0) : ((p.ldc & 7) == 0)) && (!p.resid || (p.ldr & 3) == 0);
    if (vec_ok) {
        const int ch = tid & 15;                       // 8-column chunk
        const int n = n0 + ch * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = (tid >> 4) + 16 * i;
            const int m = m0 + rl;
            if (m >= p.M) break;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ct + rl * BN + ch * 8);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ct + rl * BN + ch * 8 + 4);
            if (p.resid) {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n + 4);
                v0 = r0 + p.alpha * v0;
                v1 = r1 + p.alpha * v1;
            }
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
        const int cl = tid & 127;
        const int n = n0 + cl;
        if (n < p.N) {
            const long nc = p.col_T ? (long)(n / p.col_T) * p.col_Tp + (n % p.col_T) : n;
            for (int rl = tid >> 7; rl < BM; rl += 2) {
                const int m = m0 + rl;
                if (m >= p.M) break;
                float v = ct[rl * BN + cl];
                if (p.resid) v = p.resid[(long)m * p.ldr + n] + p.alpha * v;
                if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + nc] = v;
                else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + nc] = f2bf(v);
            }
        }
    }
}

int g_stages = 2;
int g_krot = 0;   // measured: rotating the K loop start per block does not help (no channel camping on these shapes)   // tuning knob (mi_gemm_set_stages), default chosen from measurements

}  // namespace

extern "C" void mi_gemm_set_stages(int stages) { g_stages = (stages == 2) ? 2 : 3; }
extern "C" void mi_gemm_set_krot(int on) { g_krot = on; }

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
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int stages = g_stages;
    const size_t lds = (size_t)(stages == 2 ? 2 : 3) * STAGE_BYTES;       // >= 64 KiB: also holds the fp32 C tile
    if (stages == 2) {
        if (conv) hipLaunchKernelGGL((gemm_glds_kernel<2, true>), dim3(grid), dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<2, false>), dim3(grid), dim3(NT), lds, stream, a);
    } else {
        if (conv) hipLaunchKernelGGL((gemm_glds_kernel<3, true>), dim3(grid), dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<3, false>), dim3(grid), dim3(NT), lds, stream, a);
    }
    return MI_OK;
}
