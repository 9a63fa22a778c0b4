// Strided, batched bf16 MFMA GEMM for the small per-(utterance, head) products of the attention backward pass
// (scores / probabilities are materialised only there; the forward keeps its fused LDS kernel, attention.hip):
//
//   C[z1,z2][m][n] = alpha * sum_k A[z1,z2][m][k] * B[z1,z2][n][k]  (+ C)
//
// Every operand is addressed by element strides, so Q·Kᵀ, P·V, Pᵀ·dO, dSᵀ·Q and the batch-reduced dBDᵀ·(q+v) (K runs over
// (b, t) with one uniform stride) are the same kernel.  One of (row stride, k stride) must be 1 per operand:
//   * k-contiguous operands are staged [row][k] (16-B loads / stores) and read as two 8-B halves per lane,
//   * row-contiguous operands are staged as they lie in memory, [k][row] (16-B loads / stores), and read with the transposing
//     LDS read ds_read_b64_tr_b16 (no scalar transposing writes);
// both read paths deliver the same k-permutation inside a 16-step ({4h..4h+3} ∪ {8+4h..8+4h+3}, h = lane >> 5), so any mix of
// operand layouts multiplies correctly.  Tile 64x64x32, 4 waves (2x2) of one v_mfma_f32_32x32x16_bf16 tile each,
// register-staged prefetch of the next K tile.
#include "common.hpp"

namespace {

struct BgArgs {
    const bf16_t* A; long a_z1, a_z2, a_m, a_k;
    const bf16_t* B; long b_z1, b_z2, b_n, b_k;
    void* C; long c_z1, c_z2, c_m;
    int out_f32, accumulate;
    float alpha;
    int Z2, M, N, K;
};

constexpr int TK = 32;
// Block tile TMB x TMB (64: one 32 x 32 MFMA tile per wave — round 1; 128: 2 x 2 tiles per wave, four MFMAs per k-step and barrier pair instead of one — round 3: the
// attention backward's ten products per layer are 250 x 250 x 128 / 250 x 128 x 250 problems, 128 of them per launch, and ran at ~100 TFLOP/s on the small tile).
template <int TMB> struct BgTile {
    static constexpr int LD0 = TK + 8;        // [row][k] image: 80-B rows
    static constexpr int LD1 = TMB + 8;       // [k][row] image: (TMB + 8) * 2-B rows (4 consecutive k rows land on disjoint bank groups)
    static constexpr int ELEMS = (TMB * LD0 > TK * LD1) ? TMB * LD0 : TK * LD1;
    static constexpr int NP = TMB / 64;       // staging passes per thread (8 elements each), MFMA tiles per wave and dimension
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// stage pass `ps` of a TMB x 32 tile of an operand whose rows are `rows` (bound R) into registers (8 elements per thread and pass)
template <int MODE, int TMB>   // MODE 0: k-contiguous (s_k == 1), 1: row-contiguous (s_r == 1)
__device__ __forceinline__ bf16x8 stage_load(const bf16_t* base, long s_r, long s_k, int r0, int R, int k0, int K, int tid, int ps) {
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE == 0) {
        const int row = r0 + ps * 64 + (tid >> 2), k = k0 + (tid & 3) * 8;
        if (row < R && k < K) {
            const bf16_t* p = base + (long)row * s_r + k;
            if (k + 8 <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) v = *reinterpret_cast<const bf16x8*>(p);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) if (k + j < K) v[j] = p[j];
        }
    } else {
        const int k = k0 + (tid >> 3), row = r0 + ps * 64 + (tid & 7) * 8;
        if (k < K && row < R) {
            const bf16_t* p = base + (long)k * s_k + row;
            if (row + 8 <= R && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) v = *reinterpret_cast<const bf16x8*>(p);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) if (row + j < R) v[j] = p[j];
        }
    }
    return v;
}
template <int MODE, int TMB>
__device__ __forceinline__ void stage_store(bf16_t* s, const bf16x8& v, int tid, int ps) {
    if (MODE == 0) *reinterpret_cast<bf16x8*>(s + (ps * 64 + (tid >> 2)) * BgTile<TMB>::LD0 + (tid & 3) * 8) = v;
    else *reinterpret_cast<bf16x8*>(s + (tid >> 3) * BgTile<TMB>::LD1 + ps * 64 + (tid & 7) * 8) = v;
}
// MFMA operand fragment of the 32 rows starting at rb, 16-step ks (0 / 1) of the staged K tile
template <int MODE, int TMB>
__device__ __forceinline__ bf16x8 frag(const bf16_t* s, int rb, int ks, int lane) {
    if (MODE == 0) {
        const int row = rb + (lane & 31), h = lane >> 5;
        const s16x4 lo = *reinterpret_cast<const s16x4*>(s + row * BgTile<TMB>::LD0 + ks * 16 + 4 * h);
        const s16x4 hi = *reinterpret_cast<const s16x4*>(s + row * BgTile<TMB>::LD0 + ks * 16 + 8 + 4 * h);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
        const int col = rb + (g & 1) * 16 + 4 * p4;
        const int k0 = ks * 16 + 4 * (g >> 1) + q4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + k0 * BgTile<TMB>::LD1 + col));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + (k0 + 8) * BgTile<TMB>::LD1 + col));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int MA, int MB, int TMB>
__global__ __launch_bounds__(256) void bgemm_kernel(BgArgs p) {
    using Tl = BgTile<TMB>;
    constexpr int NP = Tl::NP;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * Tl::ELEMS];
    bf16_t* sA = smem;
    bf16_t* sB = smem + Tl::ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int z1 = blockIdx.z / p.Z2, z2 = blockIdx.z % p.Z2;
    const int m0 = blockIdx.y * TMB, n0 = blockIdx.x * TMB;
    const bf16_t* A = p.A + z1 * p.a_z1 + z2 * p.a_z2;
    const bf16_t* B = p.B + z1 * p.b_z1 + z2 * p.b_z2;
    f32x16 acc[NP][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = (p.K + TK - 1) / TK;
    bf16x8 ra[NP], rb[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        ra[ps] = stage_load<MA, TMB>(A, p.a_m, p.a_k, m0, p.M, 0, p.K, tid, ps);
        rb[ps] = stage_load<MB, TMB>(B, p.b_n, p.b_k, n0, p.N, 0, p.K, tid, ps);
    }
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            stage_store<MA, TMB>(sA, ra[ps], tid, ps);
            stage_store<MB, TMB>(sB, rb[ps], tid, ps);
        }
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                ra[ps] = stage_load<MA, TMB>(A, p.a_m, p.a_k, m0, p.M, (kt + 1) * TK, p.K, tid, ps);
                rb[ps] = stage_load<MB, TMB>(B, p.b_n, p.b_k, n0, p.N, (kt + 1) * TK, p.K, tid, ps);
            }
        }
#pragma unroll
        for (int ks = 0; ks < TK / 16; ++ks) {
            bf16x8 fa[NP], fb[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) fa[i] = frag<MA, TMB>(sA, wm * 32 * NP + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < NP; ++j) fb[j] = frag<MB, TMB>(sB, wn * 32 * NP + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    char* Cb = reinterpret_cast<char*>(p.C);
    const long zoff = z1 * p.c_z1 + z2 * p.c_z2;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int n = n0 + wn * 32 * NP + j * 32 + lr;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * NP + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                const long off = zoff + (long)m * p.c_m + n;
                float v = p.alpha * acc[i][j][r];
                if (p.out_f32) {
                    float* c = reinterpret_cast<float*>(Cb) + off;
                    if (p.accumulate) v += *c;
                    *c = v;
                } else {
                    bf16_t* c = reinterpret_cast<bf16_t*>(Cb) + off;
                    if (p.accumulate) v += bf2f(*c);
                    *c = f2bf(v);
                }
            }
    }
}

template <int TMB>
void bg_launch(const BgArgs& p, int Z, int ma, int mb, hipStream_t st) {
    dim3 grid(cdiv(p.N, TMB), cdiv(p.M, TMB), Z);
    if (ma == 0 && mb == 0) hipLaunchKernelGGL((bgemm_kernel<0, 0, TMB>), grid, dim3(256), 0, st, p);
    else if (ma == 0 && mb == 1) hipLaunchKernelGGL((bgemm_kernel<0, 1, TMB>), grid, dim3(256), 0, st, p);
    else if (ma == 1 && mb == 0) hipLaunchKernelGGL((bgemm_kernel<1, 0, TMB>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((bgemm_kernel<1, 1, TMB>), grid, dim3(256), 0, st, p);
}

}  // namespace

extern "C" int mi_bgemm_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k,
                             const void* B, long b_z1, long b_z2, long b_n, long b_k,
                             void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                             int Z1, int Z2, int M, int N, int K, hipStream_t st) {
    MI_ENTER();
    if (Z1 <= 0 || Z2 <= 0 || M <= 0 || N <= 0 || K <= 0 || (long)Z1 * Z2 > 65535) return MI_ERR_ARG;
    if ((a_k != 1 && a_m != 1) || (b_k != 1 && b_n != 1)) return MI_ERR_UNSUPPORTED;
    BgArgs p{(const bf16_t*)A, a_z1, a_z2, a_m, a_k, (const bf16_t*)B, b_z1, b_z2, b_n, b_k, C, c_z1, c_z2, c_m, out_f32, accumulate, alpha, Z2, M, N, K};
    const int ma = a_k == 1 ? 0 : 1, mb = b_k == 1 ? 0 : 1;
    if (M > 64 && N > 64) bg_launch<128>(p, Z1 * Z2, ma, mb, st);         // 2 x 2 MFMA tiles per wave
    else bg_launch<64>(p, Z1 * Z2, ma, mb, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
