// 256x256 output tile variant of the bf16 MFMA GEMM for gfx950 (fast path of mi_gemm_bf16 for wide outputs):
//
//   C[M,N] = epi(A[M,K] · W[N,K]^T),   N % 256 == 0, K % 32 == 0, at least ~one tile per CU
//
// Why a second kernel: per flop a 256² tile moves half the L2->LDS bytes of the 128² kernel (gemm_glds.hip), and what limits that
// kernel on this model's shapes is the latency of the LDS-DMA stream, not the MFMA rate.  Structure:
//   * 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers, 2 waves per SIMD);
//   * K tile = 32 (64-B rows), LDS ring of FOUR 32-KiB stages: three K tiles (96 KiB) are in flight while one is consumed —
//     1.5x the bytes in flight of the 128² kernel at half the bytes per flop;
//   * `global_load_lds_dwordx4`: a wave-instruction fills 16 rows x 64 B; the 16-B chunk index is XOR-swizzled with (row >> 2) & 3 on the
//     SOURCE side and again on the ds_read_b128 (rows 4 apart would otherwise share a bank group);
//   * one counted `s_waitcnt vmcnt(8)` + raw `s_barrier` per K tile (never vmcnt(0) in the steady state); the stage consumed in the
//     previous iteration is refilled right after the barrier;
//   * epilogue from the accumulators, one 32x32 tile at a time (bias / GELU / residual fused), the MFMAs are issued as W·A^T so a lane owns
//     an output ROW: fp32 as 16-B stores, bf16 widened to 16 B per lane with v_permlane32_swap — same as gemm_glds.hip.
#include "gemm_args.hpp"

namespace {

constexpr int T = 256, KT = 32, ST = 4;                   // tile, K tile, ring stages
constexpr int STG_BYTES = 2 * T * KT * 2;                 // 32 KiB: A rows then W rows
constexpr int A_BYTES_ = T * KT * 2;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int sw4(int row) { return (row >> 2) & 3; }

__global__ __launch_bounds__(512, 1) void gemm_256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;              // 2 x 4 waves: rows wm*128.., cols wn*64..
    const int ntm = (p.M + T - 1) / T, ntn = p.N / T;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (blocks of one XCD walk N fastest within an A row panel)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * T, n0 = tn * T;

    // this wave's 4 pieces per K tile: piece g = wave*4 + q; g < 16: A rows g*16.., else W rows (g-16)*16..; lane -> (row = lane>>2, slot = lane&3)
    const int prow = lane >> 2, slot = lane & 3;
    const bf16_t* src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = wave * 4 + q;
        const int row = (g & 15) * 16 + prow;
        const int c = slot ^ sw4(row);
        if (g < 16) src[q] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + c * 8;
        else src[q] = p.W + (long)(n0 + row) * p.ldw + c * 8;
    }
    auto issue = [&](int kt, int stage) {
        char* sbase = smem + stage * STG_BYTES + wave * 4 * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[q] + kt * KT), (lptr_t)(sbase + q * 1024), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / KT;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
        if (t < nk) issue(t, t);

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & (ST - 1);
        // tile kt has landed for THIS wave's pieces once at most the two younger in-flight tiles' pieces (4 each) are outstanding
        const int younger = min(ST - 2, nk - 1 - kt);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // every wave's pieces of tile kt landed; the stage of tile kt-1 is free
        if (kt + ST - 1 < nk) issue(kt + ST - 1, (kt + ST - 1) & (ST - 1));
        const char* a = smem + stage * STG_BYTES;
        const char* b = a + A_BYTES_;
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            bf16x8 fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 128 + i * 32 + lr;
                fa[i] = *reinterpret_cast<const bf16x8*>(a + row * 64 + (((ks * 2 + lh) ^ sw4(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 64 + j * 32 + lr;
                fb[j] = *reinterpret_cast<const bf16x8*>(b + col * 64 + (((ks * 2 + lh) ^ sw4(col)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // C^T tile: lane = m, regs = n
        }
    }

    // ---- epilogue: lane owns output row m = lane & 31 of the tile; register group g4 holds columns nb + 8 g4 + 4 lh + 0..3.
    // Only the vector path exists here (gemm_256_supported guarantees aligned bias / leading dimensions and no column remap).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nb = n0 + wn * 64 + j * 32;
        f32x4 bc[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            bc[g4] = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * g4 + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + lr;
            const bool mok = m < p.M;
            const float brow = (p.bias_mode == 2 && mok) ? p.bias[m] : 0.f;
            f32x4 v[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                v[g4] = f32x4{acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                v[g4] += bc[g4];
                v[g4] += brow;
                if (p.act == 1) v[g4] = f32x4{gelu_erf(v[g4].x), gelu_erf(v[g4].y), gelu_erf(v[g4].z), gelu_erf(v[g4].w)};
                else if (p.act == 2) v[g4] = f32x4{gelu_tanh(v[g4].x), gelu_tanh(v[g4].y), gelu_tanh(v[g4].z), gelu_tanh(v[g4].w)};
                if (p.resid && mok) v[g4] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + 8 * g4 + 4 * lh) + p.alpha * v[g4];
            }
            if (p.out_f32) {
                if (mok) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + nb + 8 * g4 + 4 * lh) = v[g4];
                }
            } else {
                unsigned ux[4], uy[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const bf16x2 lo = {f2bf(v[g4].x), f2bf(v[g4].y)}, hi = {f2bf(v[g4].z), f2bf(v[g4].w)};
                    ux[g4] = __builtin_bit_cast(unsigned, lo);
                    uy[g4] = __builtin_bit_cast(unsigned, hi);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {           // all lanes take part in the swap (EXEC full), stores are masked
                    const auto sx = __builtin_amdgcn_permlane32_swap(ux[2 * k], ux[2 * k + 1], false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(uy[2 * k], uy[2 * k + 1], false, false);
                    if (mok) {
                        const uint4 o = {sx[0], sy[0], sx[1], sy[1]};
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + nb + 16 * k + 8 * lh) = o;
                    }
                }
            }
        }
    }
}

}  // namespace

bool gemm_256_supported(const GemmArgs& a) {
    if ((a.K % KT) != 0 || (a.N % T) != 0 || a.M <= 0) return false;
    if (((uintptr_t)a.W & 15) || (a.ldw % 8) != 0 || ((uintptr_t)a.A & 15) || (a.lda % 8) != 0) return false;
    if (((uintptr_t)a.C & 15) || (a.resid && ((uintptr_t)a.resid & 15))) return false;
    if (a.col_T != 0 || (a.bias_mode == 1 && ((uintptr_t)a.bias & 15)) || (a.out_f32 ? (a.ldc & 3) : (a.ldc & 7)) || (a.resid && (a.ldr & 3))) return false;
    return true;
}

int gemm_256_launch(const GemmArgs& a, hipStream_t stream) {
    if (!gemm_256_supported(a)) return MI_ERR_UNSUPPORTED;
    const int grid = cdiv(a.M, T) * (a.N / T);
    hipLaunchKernelGGL(gemm_256_kernel, dim3(grid), dim3(512), (size_t)ST * STG_BYTES, stream, a);
    return MI_OK;
}
