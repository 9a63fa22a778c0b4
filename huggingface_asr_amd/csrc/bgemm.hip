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
    // band (mi_bgemm_band_bf16; 0 = off): A is the un-shifted relative-position gradient dBD — K runs over `band_cg` utterances of band_T query rows each, and row i of an
    // utterance is non-zero only in columns m in [band_a - i, band_a - i + band_T) — so an M tile [m0, m0 + TMB) meets only the rows i in
    // [band_a - (m0 + TMB - 1), band_a + band_T - 1 - m0] of every utterance: the K loop walks those (rounded out to k tiles) and skips the all-zero rest, about half.
    int band_T, band_a, band_cg;
    // m_valid (Z2 ints or null): rows m >= m_valid[z2] of A are zero for this batch entry (keys beyond an utterance's length in P^T / dS^T): their M tiles are not
    // multiplied — the block stores zeros (or leaves C alone when accumulating)
    const int* m_valid;
};

// Block tile TMB x TMB x TKB (64 x 64 x 32: one 32 x 32 MFMA tile per wave — round 1; 128 x 128 x 64: 2 x 2 tiles per wave, sixteen MFMAs per barrier pair and eight
// 16-B loads per thread in flight — round 3: the attention backward's products are 250 x 250 x 128 / 250 x 128 x 250 problems, 128 of them per launch).
template <int TMB, int TKB> struct BgTile {
    static constexpr int LD0 = TKB + 8;       // [row][k] image
    static constexpr int LD1 = TMB + 8;       // [k][row] image (4 consecutive k rows land on disjoint bank groups)
    static constexpr int ELEMS = (TMB * LD0 > TKB * LD1) ? TMB * LD0 : TKB * LD1;
    static constexpr int NT = TMB / 64;       // MFMA tiles per wave and dimension
    static constexpr int NP = TMB * TKB / (256 * 8);      // staging passes per thread (8 elements each)
    static constexpr int TPR0 = TKB / 8, RPP0 = 256 / TPR0;     // k-contiguous operand: threads per row, rows per pass
    static constexpr int TPR1 = TMB / 8, KPP1 = 256 / TPR1;     // row-contiguous operand: threads per k row, k rows per pass
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// stage pass `ps` of a TMB x TKB tile of an operand whose rows are `rows` (bound R) into registers (8 elements per thread and pass)
template <int MODE, int TMB, int TKB>   // MODE 0: k-contiguous (s_k == 1), 1: row-contiguous (s_r == 1)
__device__ __forceinline__ bf16x8 stage_load(const bf16_t* base, long s_r, long s_k, int r0, int R, int k0, int K, int tid, int ps) {
    using Tl = BgTile<TMB, TKB>;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE == 0) {
        const int row = r0 + ps * Tl::RPP0 + tid / Tl::TPR0, k = k0 + (tid % Tl::TPR0) * 8;
        if (row < R && k < K) {
            const bf16_t* p = base + (long)row * s_r + k;
            if (k + 8 <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) v = *reinterpret_cast<const bf16x8*>(p);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) if (k + j < K) v[j] = p[j];
        }
    } else {
        const int k = k0 + ps * Tl::KPP1 + tid / Tl::TPR1, row = r0 + (tid % Tl::TPR1) * 8;
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
// Branch-free form for operands the host found 16-B-row-aligned with readable row padding (mi_bgemm_bf16): one unconditional vector load per pass (address here) — rows past
// the bound re-read row 0 (their products are never stored), k past K is masked to zero in registers — so the compiler can count vmcnt over a deep prefetch
// instead of waiting at every conditional.
__device__ __forceinline__ bf16x8 mask_tail(bf16x8 v, int nvalid) {          // keep elements [0, nvalid)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = nvalid >= 2 * d + 2 ? w[d] : (nvalid == 2 * d + 1 ? (w[d] & 0xFFFFu) : 0u);
    return __builtin_bit_cast(bf16x8, w);
}
template <int MODE, int TMB, int TKB>
__device__ __forceinline__ const bf16_t* stage_addr(const bf16_t* base, long s_r, long s_k, int r0, int R, int k0, int K, int tid, int ps) {
    using Tl = BgTile<TMB, TKB>;
    if (MODE == 0) {
        const int row = r0 + ps * Tl::RPP0 + tid / Tl::TPR0, k = k0 + (tid % Tl::TPR0) * 8;
        return base + (long)(row < R ? row : 0) * s_r + (k < K ? k : 0);
    } else {
        const int k = k0 + ps * Tl::KPP1 + tid / Tl::TPR1, row = r0 + (tid % Tl::TPR1) * 8;
        return base + (long)(k < K ? k : 0) * s_k + (row < R ? row : 0);
    }
}
// the k-tail mask is applied when the stage goes to LDS — touching the registers any earlier would make the compiler wait for the load right where it was issued
template <int MODE, int TMB, int TKB>
__device__ __forceinline__ bf16x8 stage_mask(const bf16x8& v, int k0, int K, int tid, int ps) {
    using Tl = BgTile<TMB, TKB>;
    if (MODE == 0) return mask_tail(v, K - (k0 + (tid % Tl::TPR0) * 8));
    return mask_tail(v, (k0 + ps * Tl::KPP1 + tid / Tl::TPR1) < K ? 8 : 0);
}
template <int MODE, int TMB, int TKB>
__device__ __forceinline__ void stage_store(bf16_t* s, const bf16x8& v, int tid, int ps) {
    using Tl = BgTile<TMB, TKB>;
    if (MODE == 0) *reinterpret_cast<bf16x8*>(s + (ps * Tl::RPP0 + tid / Tl::TPR0) * Tl::LD0 + (tid % Tl::TPR0) * 8) = v;
    else *reinterpret_cast<bf16x8*>(s + (ps * Tl::KPP1 + tid / Tl::TPR1) * Tl::LD1 + (tid % Tl::TPR1) * 8) = v;
}
// MFMA operand fragment of the 32 rows starting at rb, 16-step ks of the staged K tile
template <int MODE, int TMB, int TKB>
__device__ __forceinline__ bf16x8 frag(const bf16_t* s, int rb, int ks, int lane) {
    using Tl = BgTile<TMB, TKB>;
    if (MODE == 0) {
        const int row = rb + (lane & 31), h = lane >> 5;
        const s16x4 lo = *reinterpret_cast<const s16x4*>(s + row * Tl::LD0 + ks * 16 + 4 * h);
        const s16x4 hi = *reinterpret_cast<const s16x4*>(s + row * Tl::LD0 + ks * 16 + 8 + 4 * h);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
        const int col = rb + (g & 1) * 16 + 4 * p4;
        const int k0 = ks * 16 + 4 * (g >> 1) + q4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + k0 * Tl::LD1 + col));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + (k0 + 8) * Tl::LD1 + col));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int MA, int MB, int TMB, int TK, bool FAST>
__global__ __launch_bounds__(256) void bgemm_kernel(BgArgs p) {
    using Tl = BgTile<TMB, TK>;
    constexpr int NP = Tl::NT, NS = Tl::NP;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * Tl::ELEMS];
    bf16_t* sA = smem;
    bf16_t* sB = smem + Tl::ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int z1 = blockIdx.z / p.Z2, z2 = blockIdx.z % p.Z2;
    const int m0 = blockIdx.y * TMB, n0 = blockIdx.x * TMB;
    const bf16_t* A = p.A + z1 * p.a_z1 + z2 * p.a_z2;
    const bf16_t* B = p.B + z1 * p.b_z1 + z2 * p.b_z2;
    const bool dead = p.m_valid != nullptr && m0 >= p.m_valid[z2];        // block-uniform: an all-zero M tile
    f32x16 acc[NP][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = dead ? 0 : (p.K + TK - 1) / TK;
    // Register staging NST k-tiles deep: a block's k-tiles are a chain of dependent HBM round trips (load -> LDS -> MFMA), and with one tile in flight the chain
    // — not bandwidth or the MFMAs — set the time (19-22 us for a 2-GFLOP launch whatever the tile).  The LDS image stays single; the stores into it wait only for
    // the oldest stage's loads (the compiler counts vmcnt over the plain loads).
    auto mma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < TK / 16; ++ks) {
            bf16x8 fa[NP], fb[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) fa[i] = frag<MA, TMB, TK>(sA, wm * 32 * NP + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < NP; ++j) fb[j] = frag<MB, TMB, TK>(sB, wn * 32 * NP + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (FAST) {
        // Two k-tiles of register staging in flight, issued as inline asm with hand-counted waits.  A block's k-tiles are a chain of dependent HBM round trips
        // (load -> LDS -> MFMA) and with one tile in flight that chain — not bandwidth, not the MFMAs — set the time (19-22 us for a 2-GFLOP launch whatever the
        // tile).  Left to the compiler the prefetch does not survive: it needs a proven lower bound on the younger loads for every `vmcnt(N)`, conditional loads
        // give it none, and it reorders the prologue's loads so that the loop header drains the queue.  Every stage's loads are therefore unconditional (k tiles
        // past K re-read k = 0 and are masked to zero on their way to LDS), NS per operand, and a stage waits with vmcnt(2 NS): exactly the younger stage.
        bf16x8 ra[2][NS], rb[2][NS];
        // with a band (BgArgs) only the k tiles that hold a row this M tile's columns can meet are walked, in order, each once: the sums are those of the full walk
        // minus tiles of zeros — the same bits
        const int ilo = p.band_T > 0 ? max(p.band_a - (m0 + TMB - 1), 0) : 0, ihi = p.band_T > 0 ? min(p.band_a + p.band_T - 1 - m0, p.band_T - 1) : 0;
        auto needed = [&](int t) -> bool {                   // block-uniform
            if (p.band_T <= 0) return true;
            const int a = t * TK, b = min(a + TK, p.K) - 1;
            for (int u = a / p.band_T; u <= b / p.band_T; ++u)
                if (max(a, u * p.band_T + ilo) <= min(b, u * p.band_T + ihi)) return true;
            return false;
        };
        auto next_needed = [&](int t) { while (t < nk && !needed(t)) ++t; return t < nk ? t : nk; };
        auto issue = [&](int s, int k0) {
#pragma unroll
            for (int ps = 0; ps < NS; ++ps) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[s][ps]) : "v"(stage_addr<MA, TMB, TK>(A, p.a_m, p.a_k, m0, p.M, k0, p.K, tid, ps)) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[s][ps]) : "v"(stage_addr<MB, TMB, TK>(B, p.b_n, p.b_k, n0, p.N, k0, p.K, tid, ps)) : "memory");
            }
        };
        int tq[2];
        tq[0] = next_needed(0);
        tq[1] = tq[0] < nk ? next_needed(tq[0] + 1) : nk;
        if (nk > 0) {                                          // (a dead M tile loads nothing)
            issue(0, tq[0] * TK);
            issue(1, tq[1] * TK);
        }
        while (tq[0] < nk || tq[1] < nk) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kt = tq[s];
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NS) : "memory");
#pragma unroll
                for (int ps = 0; ps < NS; ++ps) {
                    asm volatile("" : "+v"(ra[s][ps]), "+v"(rb[s][ps]));          // valid only behind the wait
                    stage_store<MA, TMB, TK>(sA, stage_mask<MA, TMB, TK>(ra[s][ps], kt * TK, p.K, tid, ps), tid, ps);
                    stage_store<MB, TMB, TK>(sB, stage_mask<MB, TMB, TK>(rb[s][ps], kt * TK, p.K, tid, ps), tid, ps);
                }
                __syncthreads();
                const int last = max(tq[0], tq[1]);
                tq[s] = last < nk ? next_needed(last + 1) : nk;
                issue(s, tq[s] * TK);                  // (a tile index of nk: k past K — the loads re-read k = 0 and are masked to zero, nothing is multiplied)
                if (kt < nk) mma_tile();               // block-uniform
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        bf16x8 ra[NS], rb[NS];
        if (nk > 0) {
#pragma unroll
            for (int ps = 0; ps < NS; ++ps) {
                ra[ps] = stage_load<MA, TMB, TK>(A, p.a_m, p.a_k, m0, p.M, 0, p.K, tid, ps);
                rb[ps] = stage_load<MB, TMB, TK>(B, p.b_n, p.b_k, n0, p.N, 0, p.K, tid, ps);
            }
        }
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int ps = 0; ps < NS; ++ps) {
                stage_store<MA, TMB, TK>(sA, ra[ps], tid, ps);
                stage_store<MB, TMB, TK>(sB, rb[ps], tid, ps);
            }
            __syncthreads();
            if (kt + 1 < nk) {
#pragma unroll
                for (int ps = 0; ps < NS; ++ps) {
                    ra[ps] = stage_load<MA, TMB, TK>(A, p.a_m, p.a_k, m0, p.M, (kt + 1) * TK, p.K, tid, ps);
                    rb[ps] = stage_load<MB, TMB, TK>(B, p.b_n, p.b_k, n0, p.N, (kt + 1) * TK, p.K, tid, ps);
                }
            }
            mma_tile();
            __syncthreads();
        }
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

template <int TMB, int TK, bool FAST>
void bg_launch(const BgArgs& p, int Z, int ma, int mb, hipStream_t st) {
    dim3 grid(cdiv(p.N, TMB), cdiv(p.M, TMB), Z);
    if (ma == 0 && mb == 0) hipLaunchKernelGGL((bgemm_kernel<0, 0, TMB, TK, FAST>), grid, dim3(256), 0, st, p);
    else if (ma == 0 && mb == 1) hipLaunchKernelGGL((bgemm_kernel<0, 1, TMB, TK, FAST>), grid, dim3(256), 0, st, p);
    else if (ma == 1 && mb == 0) hipLaunchKernelGGL((bgemm_kernel<1, 0, TMB, TK, FAST>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((bgemm_kernel<1, 1, TMB, TK, FAST>), grid, dim3(256), 0, st, p);
}

// an operand qualifies for the branch-free loads when every 16-B chunk the kernel may touch is aligned and inside a row the caller owns: base and all non-unit
// strides multiples of 8 elements, and the contiguous extent either a multiple of 8 or followed by row padding (outer stride >= extent rounded up to 8)
bool bg_operand_fast(const void* base, long z1, long z2, long s_row, long s_k, int R, int K) {
    const long outer = s_k == 1 ? s_row : s_k;
    const int extent = s_k == 1 ? K : R;
    if ((reinterpret_cast<uintptr_t>(base) & 15) || (z1 % 8) || (z2 % 8) || (outer % 8) || outer < 0) return false;
    return (extent % 8 == 0) || outer >= (extent + 7) / 8 * 8;
}

}  // namespace

extern "C" int mi_bgemm_band_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k, const void* B, long b_z1, long b_z2, long b_n, long b_k,
                                  void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                                  int Z1, int Z2, int M, int N, int K, int band_T, int band_a, int band_cg, hipStream_t st);
extern "C" int mi_bgemm_sparse_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k, const void* B, long b_z1, long b_z2, long b_n, long b_k,
                                    void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                                    int Z1, int Z2, int M, int N, int K, int band_T, int band_a, int band_cg, const int* m_valid, hipStream_t st);
extern "C" int mi_bgemm_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k,
                             const void* B, long b_z1, long b_z2, long b_n, long b_k,
                             void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                             int Z1, int Z2, int M, int N, int K, hipStream_t st) {
    MI_ENTER();
    if (Z1 <= 0 || Z2 <= 0 || M <= 0 || N <= 0 || K <= 0 || (long)Z1 * Z2 > 65535) return MI_ERR_ARG;
    if ((a_k != 1 && a_m != 1) || (b_k != 1 && b_n != 1)) return MI_ERR_UNSUPPORTED;
    return mi_bgemm_band_bf16(A, a_z1, a_z2, a_m, a_k, B, b_z1, b_z2, b_n, b_k, C, c_z1, c_z2, c_m, out_f32, accumulate, alpha, Z1, Z2, M, N, K, 0, 0, 0, st);
}

// mi_bgemm_bf16 for an A operand that is banded along K (BgArgs): K = band_cg * band_T rows, row (u, i) non-zero only in columns [band_a - i, band_a - i + band_T).  The
// all-zero k tiles are skipped when the operands qualify for the branch-free loads (otherwise the whole K range is walked: same result).  band_T = 0: no band.
extern "C" int mi_bgemm_band_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k,
                                  const void* B, long b_z1, long b_z2, long b_n, long b_k,
                                  void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                                  int Z1, int Z2, int M, int N, int K, int band_T, int band_a, int band_cg, hipStream_t st) {
    return mi_bgemm_sparse_bf16(A, a_z1, a_z2, a_m, a_k, B, b_z1, b_z2, b_n, b_k, C, c_z1, c_z2, c_m, out_f32, accumulate, alpha, Z1, Z2, M, N, K, band_T, band_a, band_cg, nullptr, st);
}
// ... and with m_valid (Z2 ints, device, or NULL): rows m >= m_valid[z2] of A are zero for batch entry z2 — their M tiles are stored as zeros without being multiplied
extern "C" int mi_bgemm_sparse_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k,
                                    const void* B, long b_z1, long b_z2, long b_n, long b_k,
                                    void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha,
                                    int Z1, int Z2, int M, int N, int K, int band_T, int band_a, int band_cg, const int* m_valid, hipStream_t st) {
    MI_ENTER();
    if (Z1 <= 0 || Z2 <= 0 || M <= 0 || N <= 0 || K <= 0 || (long)Z1 * Z2 > 65535) return MI_ERR_ARG;
    if ((a_k != 1 && a_m != 1) || (b_k != 1 && b_n != 1)) return MI_ERR_UNSUPPORTED;
    if (band_T < 0 || (band_T > 0 && (band_cg <= 0 || (long)band_cg * band_T != K))) return MI_ERR_ARG;
    BgArgs p{(const bf16_t*)A, a_z1, a_z2, a_m, a_k, (const bf16_t*)B, b_z1, b_z2, b_n, b_k, C, c_z1, c_z2, c_m, out_f32, accumulate, alpha, Z2, M, N, K, band_T, band_a, band_cg, m_valid};
    const int ma = a_k == 1 ? 0 : 1, mb = b_k == 1 ? 0 : 1;
    const bool fast = bg_operand_fast(A, a_z1, a_z2, a_m, a_k, M, K) && bg_operand_fast(B, b_z1, b_z2, b_n, b_k, N, K);
    const long blocks128 = (long)cdiv(M, 128) * cdiv(N, 128) * Z1 * Z2;
    if (M > 64 && N > 64 && blocks128 >= 256) {                    // 2 x 2 MFMA tiles per wave, when that still gives every CU a block
        if (fast) bg_launch<128, 64, true>(p, Z1 * Z2, ma, mb, st);
        else bg_launch<128, 64, false>(p, Z1 * Z2, ma, mb, st);
    } else {
        if (fast) bg_launch<64, 32, true>(p, Z1 * Z2, ma, mb, st);
        else bg_launch<64, 32, false>(p, Z1 * Z2, ma, mb, st);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}
