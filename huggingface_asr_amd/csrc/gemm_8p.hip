// 256x256x64 eight-wave "phase-interleaved" bf16 MFMA GEMM for gfx950 (wide-N fast path of mi_gemm_bf16).
//
//   C[M,N] (bf16) = act(A[M,K] · W[N,K]^T + bias)      N % 256 == 0, K % 64 == 0, K >= 128
//
// One workgroup of 8 waves (2 x 4) per 256 x 256 output tile; a wave owns 128 x 64 = 8 x 4 MFMA 16x16x32 tiles (128 accumulator
// registers).  K tiles of 64 live in a two-deep LDS ring (2 x 64 KiB); each K tile is staged as FOUR half-tiles of 16 KiB by
// `global_load_lds_dwordx4` (two 1-KiB pieces per wave and half-tile):
//     Ha0 / Ha1 = the A rows of every wave's first / second 64-row half,   Hb0 / Hb1 = the W rows of its first / second 32 columns.
// A K tile is consumed in four PHASES, one output quadrant (64 x 32 per wave = 16 MFMAs) each:
//     p1: read B0, A0 -> quadrant (0,0)     p2: read B1 -> (0,1)     p3: read A1 -> (1,1)     p4: (nothing to read) -> (1,0)
// and every phase is   { ds_reads ; stage ONE half-tile ; counted vmcnt } s_barrier { lgkmcnt(0) ; 16 MFMAs } s_barrier.
// The two wave rows run ONE barrier apart (waves w and w+4 share a SIMD): while one row's waves issue their 16 MFMAs the other
// row's waves read LDS and issue the next loads, so the matrix pipe of every SIMD always has a wave to take from.
//
// Staging schedule (the order never changes, so a counted vmcnt retires exactly what the next phase reads):
//     tile t:  p1 stages Hb1(t+1)   p2 stages Ha1(t+1)   p3 stages Ha0(t+2)   p4 stages Hb0(t+2)
//   WAR: a region is re-staged >= 2 phases after its last read (reads of phase g are retired by lgkmcnt(0) behind the barrier that
//        opens that phase's MFMA segment; the lagging wave row is one barrier behind) — Ha0/Hb0 read in p1, re-staged in p3/p4;
//        Hb1 read in p2, re-staged in p1 of the next tile; Ha1 read in p3, re-staged in p2 of the next tile.
//   RAW: a half-tile is read one phase after the wait that retires it: `vmcnt(8)` (four younger half-tiles stay in flight = 64 KiB
//        per CU) in p4 (-> Ha0/Hb0 of t+1), p1 (-> Hb1 of t) and p2 (-> Ha1 of t), each in front of the phase's first barrier.
//   The last two K tiles run with their stagings peeled off and the counts lowered accordingly (8,8,4 / 2,0).
//
// LDS image per K tile: A rows [256][128 B] then W rows [256][128 B]; pieces are lane-linear (8 rows x 128 B), the bank swizzle
// (16-B chunk ^ ((row>>1)&7): conflict-free ds_read_b128 for the 16x16x32 fragment pattern) is applied to the per-lane SOURCE
// address and again on the read.  MFMAs are issued as W·A^T, so a lane owns an output ROW (lane & 15) and its four registers are
// four consecutive COLUMNS.  Epilogue: bias / activation on the accumulators, bf16 rows through a wave-private 16-KiB LDS region
// (the ring is dead by then), out as whole 128-B lines (8 lanes x 16 B per row).
#include "gemm_args.hpp"
#include <type_traits>

// A/B switch of tools/gemm_floor_ab.sh only (never set in the product build): 1 = the kernels' K loops WITHOUT their LDS fragment reads and MFMAs (what the
// LDS-DMA staging alone costs: the ingest floor of the schedule), 2 = WITHOUT the staging (what the reads + MFMAs alone cost), 3 = staging + MFMAs without the fragment
// reads, 4 = staging + fragment reads without the MFMAs (which of the two the staging does not overlap with).  Results are garbage in every mode but 0.
#ifndef GEMM_FLOOR
#define GEMM_FLOOR 0
#endif
// which 128 x 128 form `ring = 0` (every product call site) means: 0 = the register-pipelined one, 1 = the loader / consumer form from K = 2048 on (A/B builds: tools/gemm128l_step_ab.sh)
#ifndef GEMM128_LOADER
#define GEMM128_LOADER 0
#endif

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TB = 256, BK = 64;
constexpr int BUF = 65536, BOFF = 32768;

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __attribute__((aligned(16))) uint4 g_zero_page8p = {0u, 0u, 0u, 0u};      // source of implicit-im2col zero padding

struct Ctx {
    const char* A; const char* W;
    unsigned offA[2][2], offB[2][2];     // per-lane source byte offsets of this wave's two pieces of Ha(mq) / Hb(nq), K tile 0
    int cti[2][2], cfi[2][2];            // CONV: input (time, freq) of tap (0,0) for the lane's output row; offA = batch base + swizzled chunk
    unsigned dA[2][2], dB[2][2];         // wave-uniform LDS byte offsets of those pieces inside a ring buffer
    unsigned aRd[2], bRd[2];             // per-lane LDS byte offsets of the fragment reads (k-step 0 / 1), ring buffer 0
};

// CONV: A is a channels-last activation (B, Tin, Fin, Cin), row m = (b, to, fo), k = (kh*KW + kw)*Cin + c with Cin % 64 == 0, so a K tile is one
// tap and a 64-channel slice: the per-lane source row moves with the tap, rows that fall into the zero padding read a 16-B zero page.
// OUT32: fp32 output (the CTC head) — N need not be a multiple of the tile: W rows beyond N are clamped, columns beyond N never stored.
// GATED (GatedConv2d as one GEMM, extractors.py:23-32): the W rows are packed [conv c0..c0+31 ; gate c0..c0+31] per 64, so a wave's 64 columns are 32 output channels
// twice — accumulator columns j = 0,1 the conv, j = 2,3 the gate of the SAME channels in the same lane — and the epilogue writes act((conv + b) * sigmoid(gate + bg)) as (M, N/2).
// LNF (LayerNorm folded into the GEMM, gemm_args.hpp): the prologue reduces the per-row partial (sum, sumsq) pairs of the tile's 256 rows to (rstd, rstd * mean) in a 2-KiB LDS
// table behind the ring — issued before the first K tile is consumed, so its memory round trip hides under the ring's fill — and the epilogue computes
// rstd * acc - rstd * mean * s_n + bias_n instead of acc + bias_n.
template <bool CONV, int ACT, bool OUT32 = false, bool GATED = false, bool LNF = false, bool LSE = false>     // ACT: 0 none, 1 erf-GELU, 2 tanh-GELU — compile-time, so the epilogue is straight-line code with many independent chains in flight
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntm = (p.M + TB - 1) / TB, ntn = (p.N + TB - 1) / TB;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the blocks of one XCD walk N fastest within an A row panel
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / ntn) * TB, n0 = (bid % ntn) * TB;

    Ctx c;
    c.A = reinterpret_cast<const char*>(p.A);
    c.W = reinterpret_cast<const char*>(p.W);
    {
        const int prow = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ra = wr * 128 + h * 64 + (2 * wc + e) * 8;
                const int rb = (wave >> 1) * 64 + h * 32 + (2 * (wave & 1) + e) * 8;
                const int lcA = pc ^ (((ra + prow) >> 1) & 7), lcB = pc ^ (((rb + prow) >> 1) & 7);
                if constexpr (CONV) {
                    const int m = min(m0 + ra + prow, p.M - 1);
                    const int fo = m % p.Fout, to = (m / p.Fout) % p.Tout, b = m / (p.Fout * p.Tout);
                    c.cti[h][e] = to * p.stride - p.pad_t;
                    c.cfi[h][e] = fo * p.stride_f - p.pad_f;
                    c.offA[h][e] = (unsigned)b * (unsigned)(p.Tin * p.Fin * p.Cin * 2) + lcA * 16;
                } else {
                    c.offA[h][e] = (unsigned)min(m0 + ra + prow, p.M - 1) * (unsigned)(p.lda * 2) + lcA * 16;
                }
                c.offB[h][e] = (unsigned)min(n0 + rb + prow, p.N - 1) * (unsigned)(p.ldw * 2) + lcB * 16;
                c.dA[h][e] = ra * 128;
                c.dB[h][e] = BOFF + rb * 128;
            }
        const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
        const unsigned low = fr * 128 + ((fq ^ swz) << 4);
        c.aRd[0] = wr * 128 * 128 + low;
        c.aRd[1] = c.aRd[0] ^ 64;
        c.bRd[0] = BOFF + wc * 64 * 128 + low;
        c.bRd[1] = c.bRd[0] ^ 64;
    }
    auto stageA = [&](int kt, unsigned buf, int mq) {
        if (GEMM_FLOOR == 2) return;
        if constexpr (CONV) {
            const int ntap = p.K / p.Cin;
            const int chunk = kt / ntap, tap = kt - chunk * ntap, c0 = chunk * BK;      // K tiles in channel-slice-major order (see conv_k)
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ti = c.cti[mq][e] + kh, fi = c.cfi[mq][e] + kw;
                const bool ok = (unsigned)ti < (unsigned)p.Tin && (unsigned)fi < (unsigned)p.Fin;
                const char* sp = ok ? c.A + (c.offA[mq][e] + (unsigned)(((ti * p.Fin + fi) * p.Cin + c0) * 2)) : reinterpret_cast<const char*>(&g_zero_page8p);
                __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(smem + buf + c.dA[mq][e]), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                __builtin_amdgcn_global_load_lds((gptr_t)(c.A + c.offA[mq][e] + (unsigned)kt * 128u), (lptr_t)(smem + buf + c.dA[mq][e]), 16, 0, 0);
        }
    };
    // CONV: the K tiles are visited 64-channel slice by slice, the KH*KW (= K / Cin) taps of a slice back to back (k offset = tap * Cin + slice * 64; the sum over K does not care
    // about the order).  The taps of one slice re-read the same input patch (13 output rows x 19 columns of a tile touch 27 x 39 input positions = 135 KB per slice):
    // consecutive K tiles keep it in L2.  Tap-major order re-read the whole 540-KB patch of every tile 2.2x from beyond L2 (PMC: 690 MB fetched per launch for a 320-MB input).
    auto conv_k = [&](int kt) { const int ntap = p.K / p.Cin; const int chunk = kt / ntap; return (kt - chunk * ntap) * p.Cin + chunk * BK; };
    auto stageB = [&](int kt, unsigned buf, int nq) {
        if (GEMM_FLOOR == 2) return;
        const unsigned kb = CONV ? (unsigned)conv_k(kt) * 2u : (unsigned)kt * 128u;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            __builtin_amdgcn_global_load_lds((gptr_t)(c.W + c.offB[nq][e] + kb), (lptr_t)(smem + buf + c.dB[nq][e]), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb[2][2][2];        // A fragments of the current 64-row half; W fragments of both 32-column halves

    auto readA = [&](unsigned cb, int mq) {
        if (GEMM_FLOOR == 1 || GEMM_FLOOR == 3) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                fa[i][s] = *reinterpret_cast<const bf16x8*>(smem + (cb + c.aRd[s]) + (mq * 64 + i * 16) * 128);
    };
    auto readB = [&](unsigned cb, int nq) {
        if (GEMM_FLOOR == 1 || GEMM_FLOOR == 3) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                fb[nq][j][s] = *reinterpret_cast<const bf16x8*>(smem + (cb + c.bRd[s]) + (nq * 32 + j * 16) * 128);
    };
    auto quad = [&](int mq, int nq) {
        if (GEMM_FLOOR == 4) {               // keep the fragment reads alive without the MFMAs
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i][s]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[nq][j][s]));
            }
        }
        if (GEMM_FLOOR == 1 || GEMM_FLOOR == 4) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mq * 4 + i][nq * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nq][j][s], fa[i][s], acc[mq * 4 + i][nq * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // one K tile = four phases; S* = stage in that phase, W* = vmcnt count in front of that phase's first barrier (-1: none)
    auto tile = [&](int kt, unsigned cb, auto S1, auto S2, auto S3, auto S4, auto W1, auto W2, auto W4) {
        // p1
        readB(cb, 0);
        __builtin_amdgcn_sched_barrier(0);
        readA(cb, 0);
        if constexpr (decltype(S1)::value) stageB(kt + 1, cb ^ BUF, 1);
        wait_vm<decltype(W1)::value>();
        barrier();
        wait_lds();
        quad(0, 0);
        barrier();
        // p2
        readB(cb, 1);
        if constexpr (decltype(S2)::value) stageA(kt + 1, cb ^ BUF, 1);
        wait_vm<decltype(W2)::value>();
        barrier();
        wait_lds();
        quad(0, 1);
        barrier();
        // p3
        readA(cb, 1);
        if constexpr (decltype(S3)::value) stageA(kt + 2, cb, 0);
        barrier();
        wait_lds();
        quad(1, 1);
        barrier();
        // p4
        if constexpr (decltype(S4)::value) stageB(kt + 2, cb, 0);
        wait_vm<decltype(W4)::value>();
        barrier();
        quad(1, 0);
        barrier();
    };
    using T = std::true_type;
    using F = std::false_type;

    const int nk = p.K / BK;
    // LNF: the row-statistic partials are requested FIRST — the oldest entries of the in-order vmcnt queue — and only summed behind the first counted wait of the
    // pipeline (which retires them on the way), so the kernel never waits for them alone; two lanes per row, each takes half of the row's partial pairs.
    // The loads are inline asm: the compiler's own waitcnt insertion would put a vmcnt(0) in front of their first use (it does not count across the hand-placed
    // waits), the explicit vmcnt(8) in front of the first barrier retires them.  Branch-free: every lane issues four 16-B loads (indices clamped), masks are applied at the sums.
    f32x4 lnv[4];
    if constexpr (LNF) {
        const int row = tid >> 1, half = tid & 1;
        const int n4 = p.ln_npart == 1 ? 1 : p.ln_npart >> 2;      // f32x4 records per lane: ln_npart pairs = ln_npart / 2 float4, split over the two lanes (1 pair: lane 0 alone)
        const f32x4* s4 = reinterpret_cast<const f32x4*>(p.ln_stats + (long)min(m0 + row, p.M - 1) * LN_STATS_STRIDE) + (p.ln_npart == 1 ? 0 : half * n4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4* q = s4 + min(j, n4 - 1);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lnv[j]) : "v"(q) : "memory");
        }
    }
    // prologue: Ha0(0) Hb0(0) Hb1(0) Ha1(0) Ha0(1) Hb0(1) — the steady-state order
    stageA(0, 0, 0); stageB(0, 0, 0); stageB(0, 0, 1); stageA(0, 0, 1); stageA(1, BUF, 0); stageB(1, BUF, 0);
    wait_vm<8>();
    barrier();
    float ln_r = 0.f, ln_t = 0.f;        // LNF: (rstd, rstd * mean) of row tid >> 1, kept in two registers across the K loop; shared through LDS in the epilogue
    if constexpr (LNF) {
        const int n4 = p.ln_npart == 1 ? 1 : p.ln_npart >> 2;
        const float w1 = n4 > 1 ? 1.f : 0.f, w2 = n4 > 2 ? 1.f : 0.f, w3 = n4 > 3 ? 1.f : 0.f;          // records beyond n4 were clamped re-loads: weight 0
        const float wz = p.ln_npart == 1 ? 0.f : 1.f;                                                     // one pair: only (.x, .y) of record 0 ...
        const float wl = (p.ln_npart == 1 && (tid & 1)) ? 0.f : 1.f;                                      // ... and only on the row's first lane
        float sm = wl * (((lnv[0].x + wz * lnv[0].z) + w1 * (lnv[1].x + lnv[1].z)) + (w2 * (lnv[2].x + lnv[2].z) + w3 * (lnv[3].x + lnv[3].z)));
        float sq = wl * (((lnv[0].y + wz * lnv[0].w) + w1 * (lnv[1].y + lnv[1].w)) + (w2 * (lnv[2].y + lnv[2].w) + w3 * (lnv[3].y + lnv[3].w)));
        sm += dpp_f32<0xB1, 0xF>(0.f, sm);                  // quad_perm [1,0,3,2]: lane ^ 1 holds the row's other half
        sq += dpp_f32<0xB1, 0xF>(0.f, sq);
        const float mean = sm / (float)p.K;
        ln_r = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.f) + p.ln_eps);
        ln_t = ln_r * mean;
    }
    if (wr == 1) barrier();              // the second wave row runs one barrier behind
    unsigned cb = 0;
    int kt = 0;
    for (; kt < nk - 2; ++kt) {
        tile(kt, cb, T{}, T{}, T{}, T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});
        cb ^= BUF;
    }
    tile(kt, cb, T{}, T{}, F{}, F{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    cb ^= BUF;
    tile(kt + 1, cb, F{}, F{}, F{}, F{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{});
    if (wr == 0) barrier();              // barrier counts match; every wave's last LDS reads are retired

    // ---- epilogue: bias / activation, bf16 rows through a wave-private LDS region, whole 128-B lines out.  Done in four blocks of 32 rows so that the
    // stores of one block drain while the next block's activation (VALU-bound: ~16 instructions per element for erf-GELU) is evaluated.
    char* reg = smem + wave * 16384;
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const int nb = n0 + wc * 64;
    if constexpr (OUT32) {
        // fp32 rows (256 B per row of the wave tile): two halves of 64 rows through the same 16-KiB region, 16 lanes x 16 B per row on the way out
        const int n4 = (p.N + 3) & ~3;                      // ldc >= n4 (checked by the launcher): a 16-B store may cover the row's own padding columns
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // the bias goes into the accumulators first: no bias registers live across the two halves
            const int n = nb + j * 16 + fq * 4;
            f32x4 bj = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias_mode == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.N) bj[e] = p.bias[n + e];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += bj;
        }
        const int r16 = lane >> 4, c16 = lane & 15;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = ii * 16 + fr;                                  // row within the half
                    *reinterpret_cast<f32x4*>(reg + row * 256 + (((j * 4 + fq) ^ (row & 15)) << 4)) = acc[half * 4 + ii][j];
                }
            // Residual form (the N = 512 GEMMs of the encoder on this tile: 64 blocks per launch — the throughput mode, where other steps' kernels own the rest of the
            // chip): out = resid + alpha * (acc + bias), and the LayerNorm-fold producer's extras (bf16 copy, per-row partial statistics over the wave's 64 columns).
            // The half's 16 residual vectors are requested once its accumulators have gone to LDS (their registers are free) and land under the LDS round trip.
            f32x4 rr[16];
            if (p.resid) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int m = min(m0 + wr * 128 + half * 64 + u * 4 + r16, p.M - 1);
                    rr[u] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + c16 * 4);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int row = u * 4 + r16;
                f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 256 + ((c16 ^ (row & 15)) << 4));
                const int m = m0 + wr * 128 + half * 64 + row, n = nb + c16 * 4;
                if (p.resid) v = rr[u] + p.alpha * v;
                if (m < p.M && n < n4) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = v;
                if (p.C2 && m < p.M) *reinterpret_cast<bf16x4*>(p.C2 + (long)m * p.ldc2 + n) = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                if (p.stats_out) {                          // 16 lanes hold a row's 64 columns: one (sum, sumsq) pair per row and wave column
                    float sm = (v.x + v.y) + (v.z + v.w), sq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    sm += dpp_f32<0xB1, 0xF>(0.f, sm); sq += dpp_f32<0xB1, 0xF>(0.f, sq);
                    sm += dpp_f32<0x4E, 0xF>(0.f, sm); sq += dpp_f32<0x4E, 0xF>(0.f, sq);
                    sm += dpp_f32<0x141, 0xF>(0.f, sm); sq += dpp_f32<0x141, 0xF>(0.f, sq);
                    sm += dpp_f32<0x140, 0xF>(0.f, sm); sq += dpp_f32<0x140, 0xF>(0.f, sq);
                    if (c16 == 0 && m < p.M) *reinterpret_cast<f32x2*>(p.stats_out + (long)m * LN_STATS_STRIDE + (((n0 >> 8) << 2) + wc) * 2) = f32x2{sm, sq};
                }
                if constexpr (LSE) {                        // the row's 64 columns of this wave block -> (max, sum exp(x - max)); columns >= N do not count
                    float xe[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) xe[e] = (n + e < p.N) ? xe[e] : -INFINITY;
                    float mx = fmaxf(fmaxf(xe[0], xe[1]), fmaxf(xe[2], xe[3]));
                    mx = fmaxf(mx, dpp_f32<0xB1, 0xF>(mx, mx)); mx = fmaxf(mx, dpp_f32<0x4E, 0xF>(mx, mx));
                    mx = fmaxf(mx, dpp_f32<0x141, 0xF>(mx, mx)); mx = fmaxf(mx, dpp_f32<0x140, 0xF>(mx, mx));
                    float se = 0.f;
                    if (mx > -INFINITY) se = (__expf(xe[0] - mx) + __expf(xe[1] - mx)) + (__expf(xe[2] - mx) + __expf(xe[3] - mx));
                    se += dpp_f32<0xB1, 0xF>(0.f, se); se += dpp_f32<0x4E, 0xF>(0.f, se);
                    se += dpp_f32<0x141, 0xF>(0.f, se); se += dpp_f32<0x140, 0xF>(0.f, se);
                    if (c16 == 0 && m < p.M) *reinterpret_cast<f32x2*>(p.lse_part + (long)m * p.lse_ld + ((n0 >> 6) + wc) * 2) = f32x2{mx, se};
                }
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four rows in flight at a time: all 128 accumulator registers are still live in the first half
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the second half overwrites the region
        }
        return;
    }
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        b4[j] = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + nb + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LNF) {
        // acc <- rstd * acc - rstd * mean * s_n: the folded LayerNorm; the bias (W beta + b) is added with b4 below.  The 256 (rstd, rstd * mean) pairs go through a
        // 2-KiB table behind the ring (every wave needs the rows of its wave row)
        if ((tid & 1) == 0) *reinterpret_cast<f32x2*>(smem + 2 * BUF + (tid >> 1) * 8) = f32x2{ln_r, ln_t};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_colsum + nb + j * 16 + fq * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 rt = *reinterpret_cast<const f32x2*>(smem + 2 * BUF + (wr * 128 + i * 16 + fr) * 8);
                acc[i][j] = f32x4{fmaf(rt.x, acc[i][j].x, -rt.y * s4.x), fmaf(rt.x, acc[i][j].y, -rt.y * s4.y), fmaf(rt.x, acc[i][j].z, -rt.y * s4.z), fmaf(rt.x, acc[i][j].w, -rt.y * s4.w)};
            }
        }
    }
    const int prow = lane >> 3, pc = lane & 7;
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
    if constexpr (GATED) {
        // 32 output channels per wave and row: 64-B rows through the wave-private region (128 rows x 64 B), out as 4 lanes x 16 B per row
        const int ob = (n0 >> 1) + wc * 32;
        const int r4 = lane >> 2, c4 = lane & 3;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = blk * 2 + ii;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 z = acc[i][j] + b4[j], g = acc[i][j + 2] + b4[j + 2];
                    f32x4 v = f32x4{z.x * sigmoid_f(g.x), z.y * sigmoid_f(g.y), z.z * sigmoid_f(g.z), z.w * sigmoid_f(g.w)};
                    if constexpr (ACT == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                    const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                    const int row = i * 16 + fr;
                    *reinterpret_cast<bf16x4*>(reg + row * 64 + (((j * 2 + (fq >> 1)) ^ ((row >> 2) & 3)) << 4) + (fq & 1) * 8) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = blk * 2; u < blk * 2 + 2; ++u) {
                const int row = u * 16 + r4;
                const uint4 v = *reinterpret_cast<const uint4*>(reg + row * 64 + ((c4 ^ ((row >> 2) & 3)) << 4));
                const int m = m0 + wr * 128 + row;
                if (m < p.M) *reinterpret_cast<uint4*>(C + (long)m * p.ldc + ob + c4 * 8) = v;
            }
        }
        return;
    }
    if constexpr (ACT >= 3) {
        // Training epilogues (gemm_args.hpp): the bf16 rows come back from the staging region as 8 consecutive columns of one row per lane — the layout the
        // element-wise kernels work in — and are finished there: times act'(saved pre-activation) (3), or stored and also activated into C2 (4).
        const bool drop = p.drop_p > 0.f;
        const float inv_keep = drop ? 1.f / (1.f - p.drop_p) : 1.f;
        const int n8 = p.N >> 3;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            bf16x8 ax[4];
            if constexpr (ACT == 3) {                      // the four pre-activation vectors of this block's rows: in flight while the block is staged
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const int m = m0 + wr * 128 + (blk * 4 + uu) * 8 + prow;
                    ax[uu] = *reinterpret_cast<const bf16x8*>(p.aux + (long)min(m, p.M - 1) * p.ldaux + nb + pc * 8);
                }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = blk * 2 + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[i][j] + b4[j];
                    const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                    const int row = i * 16 + fr;
                    *reinterpret_cast<bf16x4*>(reg + row * 128 + (((j * 2 + (fq >> 1)) ^ swz) << 4) + (fq & 1) * 8) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int row = (blk * 4 + uu) * 8 + prow;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(reg + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4));
                const int m = m0 + wr * 128 + row;
                float ks[8];
                if (drop) {
                    const unsigned long long quad0 = ((unsigned long long)m * n8 + ((nb >> 3) + pc)) << 1;      // logical index m * N + n of the first element, over 4
                    mask_keep4(p.drop_key, quad0, p.drop_p, inv_keep, ks);
                    mask_keep4(p.drop_key, quad0 + 1, p.drop_p, inv_keep, ks + 4);
                }
                bf16x8 o;
                if constexpr (ACT == 3) {
                    float gr[8];
                    if (p.aux_kind == 1) {                 // block-uniform branch: ONE of the two derivatives is evaluated (a select would compute both)
#pragma unroll
                        for (int e = 0; e < 8; ++e) gr[e] = gelu_erf_grad(bf2f(ax[uu][e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gr[e] = gelu_tanh_grad(bf2f(ax[uu][e]));
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = drop ? bf2f(f2bf(bf2f(v[e]) * ks[e])) : bf2f(v[e]);
                        o[e] = f2bf(g * gr[e]);
                    }
                    if (m < p.M) *reinterpret_cast<bf16x8*>(C + (long)m * p.ldc + nb + pc * 8) = o;
                } else {
                    float ac[8];
                    if (p.aux_kind == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ac[e] = gelu_erf(bf2f(v[e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ac[e] = gelu_tanh(bf2f(v[e]));
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bf16_t r = f2bf(ac[e]);
                        if (drop) r = f2bf(bf2f(r) * ks[e]);
                        o[e] = r;
                    }
                    if (m < p.M) {
                        *reinterpret_cast<bf16x8*>(C + (long)m * p.ldc + nb + pc * 8) = v;
                        *reinterpret_cast<bf16x8*>(p.C2 + (long)m * p.ldc2 + nb + pc * 8) = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = blk * 2 + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if constexpr (ACT == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if constexpr (ACT == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                const int row = i * 16 + fr;
                *reinterpret_cast<bf16x4*>(reg + row * 128 + (((j * 2 + (fq >> 1)) ^ swz) << 4) + (fq & 1) * 8) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = blk * 4; u < blk * 4 + 4; ++u) {
            const int row = u * 8 + prow;
            const uint4 v = *reinterpret_cast<const uint4*>(reg + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4));
            const int m = m0 + wr * 128 + row;
            if (m < p.M) *reinterpret_cast<uint4*>(C + (long)m * p.ldc + nb + pc * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// 128 x 128 x 64 variant for the N = 512 GEMMs (FFN out, attention out, cgMLP out, merge): 8 waves (2 x 4), wave tile 64 x 32 =
// 4 x 2 MFMA tiles, ONE phase per K tile { 12 ds_reads ; stage K tile t+3 (4 pieces per wave) ; vmcnt ; lgkmcnt(0) } s_barrier
// { 16 MFMAs } s_barrier on a FOUR-deep ring of 32-KiB K tiles; the two wave rows again run one barrier apart.
//   WAR: the reads of phase t are retired (lgkmcnt(0)) BEFORE the barrier that ends its read segment, so buffer (t-1) & 3 may be
//        re-staged in phase t (one phase after its last read) by either wave row.
//   RAW: K tile t+1 (staged in phase t-2) is retired by `vmcnt(8)` in phase t — the stagings of phases t-1 and t stay in flight
//        (64 KiB per CU) — and read in phase t+1.  Tail: vmcnt(4) in phase nk-3, vmcnt(0) in phase nk-2.
// Epilogue: fp32 (+ residual, read at kernel start in the store layout) or bf16 rows through a wave-private LDS region, out as
// 128-B (fp32) / 64-B (bf16) row segments.
constexpr int B128_BUF = 32768, B128_BOFF = 16384;

template <int NB>     // ring depth (K tiles of 32 KiB): 4 -> three tiles requested ahead (measured: a 5-deep ring, all 160 KiB of LDS, is 7 % SLOWER at K = 2048)
__global__ __launch_bounds__(512, 2) void gemm8p128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntm = (p.M + 127) / 128, ntn = p.N / 128;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / ntn) * 128, n0 = (bid % ntn) * 128;
    const int prow = lane >> 3, pc = lane & 7;

    // residual rows in the store layout (8 lanes x 16 B per row), requested before anything else
    const int nb = n0 + wc * 32;
    f32x4 rres[8];
    const bool use_res = p.out_f32 && p.resid != nullptr;
    if (use_res) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = min(m0 + wr * 64 + u * 8 + prow, p.M - 1);
            rres[u] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + pc * 4);
        }
    }

    // staging: waves 0-3 bring the A rows, waves 4-7 the W rows; 4 pieces of 8 rows per wave and K tile
    const char* src_base = reinterpret_cast<const char*>(wave < 4 ? (const void*)p.A : (const void*)p.W);
    unsigned off[4], dst[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r0 = (4 * (wave & 3) + e) * 8;
        const int lc = pc ^ (((r0 + prow) >> 1) & 7);
        if (wave < 4) off[e] = (unsigned)min(m0 + r0 + prow, p.M - 1) * (unsigned)(p.lda * 2) + lc * 16;
        else off[e] = (unsigned)min(n0 + r0 + prow, p.N - 1) * (unsigned)(p.ldw * 2) + lc * 16;
        dst[e] = (wave < 4 ? 0 : B128_BOFF) + r0 * 128;
    }
    auto stage = [&](int kt) {
        const unsigned buf = (unsigned)(NB == 4 ? (kt & 3) : kt % NB) * B128_BUF;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            __builtin_amdgcn_global_load_lds((gptr_t)(src_base + off[e] + (unsigned)kt * 128u), (lptr_t)(smem + buf + dst[e]), 16, 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const unsigned low = fr * 128 + ((fq ^ swz) << 4);
    const unsigned aRd0 = wr * 64 * 128 + low, bRd0 = B128_BOFF + wc * 32 * 128 + low;

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb[2][2];
    auto phase = [&](int kt, auto DO_STAGE, auto WAIT) {
        const unsigned cb = (unsigned)(NB == 4 ? (kt & 3) : kt % NB) * B128_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[j][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + bRd0) ^ (s * 64)) + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) fa[i][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + aRd0) ^ (s * 64)) + i * 16 * 128);
        if constexpr (decltype(DO_STAGE)::value) stage(kt + NB - 1);
        wait_vm<decltype(WAIT)::value>();
        wait_lds();
        barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][s], fa[i][s], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        barrier();
    };
    const int nk = p.K / BK;
#pragma unroll
    for (int t = 0; t < NB - 1; ++t) stage(t);
    wait_vm<4 * (NB - 2)>();
    barrier();
    if (wr == 1) barrier();
    int kt = 0;
    for (; kt < nk - (NB - 1); ++kt) phase(kt, std::true_type{}, std::integral_constant<int, 4 * (NB - 2)>{});
    if constexpr (NB == 5) { phase(kt, std::false_type{}, std::integral_constant<int, 8>{}); ++kt; }
    phase(kt, std::false_type{}, std::integral_constant<int, 4>{});
    phase(kt + 1, std::false_type{}, std::integral_constant<int, 0>{});
    phase(kt + 2, std::false_type{}, std::integral_constant<int, -1>{});
    if (wr == 0) barrier();

    // ---- epilogue
    char* reg = smem + wave * 8192;                       // 64 rows x 128 B
    f32x4 b4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        b4[j] = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + nb + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.out_f32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if (p.act == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if (p.act == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const int row = i * 16 + fr;
                *reinterpret_cast<f32x4*>(reg + row * 128 + (((j * 4 + fq) ^ swz) << 4)) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = u * 8 + prow;
            f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4));
            if (use_res) v = rres[u] + p.alpha * v;
            const int m = m0 + wr * 64 + row;
            if (m < p.M) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + nb + pc * 4) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if (p.act == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if (p.act == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                const int row = i * 16 + fr;
                *reinterpret_cast<bf16x4*>(reg + row * 128 + (((j * 2 + (fq >> 1)) ^ (swz & 3)) << 4) + (fq & 1) * 8) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        const int r4 = lane >> 2, c4 = lane & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = u * 16 + r4;
            const uint4 v = *reinterpret_cast<const uint4*>(reg + row * 128 + ((c4 ^ ((row >> 1) & 3)) << 4));
            const int m = m0 + wr * 64 + row;
            if (m < p.M) *reinterpret_cast<uint4*>(C + (long)m * p.ldc + nb + c4 * 8) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// 128 x 128 x 64, register-pipelined form: the fragments of K tile t+1 are read (12 ds_read_b128 per wave) WHILE the 16 MFMAs of K tile t
// issue — two fragment sets in registers — so a phase is as long as its MFMAs, not as its LDS reads + their latency; ONE barrier per K tile,
// all eight waves symmetric.  Ring of four 32-KiB K tiles, four tiles requested ahead:
//   phase t:  stage K tile t+4 into buffer t & 3 (tile t was read into registers in phase t-1; those reads were retired before that phase's barrier)
//             read K tile t+1 -> fragment set (t+1) & 1      ||      MFMAs of K tile t from fragment set t & 1
//             vmcnt(8): tile t+2 has landed (t+3, t+4 stay in flight)  ;  lgkmcnt(0)  ;  s_barrier
// Tail: no staging from phase nk-4 on, counts 4 / 0 / none.  Needs an even number of K tiles (static fragment-set indices, loop unrolled by 2), nk >= 4.
__global__ __launch_bounds__(512, 2) void gemm8p128p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntm = (p.M + 127) / 128, ntn = p.N / 128;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / ntn) * 128, n0 = (bid % ntn) * 128;
    const int prow = lane >> 3, pc = lane & 7;

    const int nb = n0 + wc * 32;
    f32x4 rres[8];
    const bool use_res = p.out_f32 && p.resid != nullptr;
    if (use_res) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = min(m0 + wr * 64 + u * 8 + prow, p.M - 1);
            rres[u] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + pc * 4);
        }
    }
    const char* src_base = reinterpret_cast<const char*>(wave < 4 ? (const void*)p.A : (const void*)p.W);
    unsigned off[4], dst[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r0 = (4 * (wave & 3) + e) * 8;
        const int lc = pc ^ (((r0 + prow) >> 1) & 7);
        if (wave < 4) off[e] = (unsigned)min(m0 + r0 + prow, p.M - 1) * (unsigned)(p.lda * 2) + lc * 16;
        else off[e] = (unsigned)min(n0 + r0 + prow, p.N - 1) * (unsigned)(p.ldw * 2) + lc * 16;
        dst[e] = (wave < 4 ? 0 : B128_BOFF) + r0 * 128;
    }
    auto stage = [&](int kt) {
        if (GEMM_FLOOR == 2) return;
        const unsigned buf = (unsigned)(kt & 3) * B128_BUF;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            __builtin_amdgcn_global_load_lds((gptr_t)(src_base + off[e] + (unsigned)kt * 128u), (lptr_t)(smem + buf + dst[e]), 16, 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const unsigned low = fr * 128 + ((fq ^ swz) << 4);
    const unsigned aRd0 = wr * 64 * 128 + low, bRd0 = B128_BOFF + wc * 32 * 128 + low;

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4][2], fb[2][2][2];
    auto read_frags = [&](int kt, auto SET) {
        if (GEMM_FLOOR == 1 || GEMM_FLOOR == 3) return;
        constexpr int S = decltype(SET)::value;
        const unsigned cb = (unsigned)(kt & 3) * B128_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[S][j][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + bRd0) ^ (s * 64)) + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) fa[S][i][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + aRd0) ^ (s * 64)) + i * 16 * 128);
    };
    auto mfmas = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        if (GEMM_FLOOR == 4) {               // keep the fragment reads alive without the MFMAs
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[S][i][s]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[S][j][s]));
            }
        }
        if (GEMM_FLOOR == 1 || GEMM_FLOOR == 4) return;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][j][s], fa[S][i][s], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // CUR = fragment set of K tile kt; DO_STAGE: request K tile kt+4; DO_READ: read K tile kt+1; WAIT: vmcnt count (-1 none)
    auto phase = [&](int kt, auto CUR, auto DO_STAGE, auto DO_READ, auto WAIT) {
        constexpr int C = decltype(CUR)::value;
        if constexpr (decltype(DO_STAGE)::value) stage(kt + 4);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(DO_READ)::value) read_frags(kt + 1, std::integral_constant<int, 1 - C>{});
        mfmas(CUR);
        if constexpr (decltype(DO_READ)::value) {
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<decltype(WAIT)::value>();
        if constexpr (decltype(DO_READ)::value) wait_lds();
        barrier();
    };
    using T = std::true_type;
    using F = std::false_type;
    const int nk = p.K / BK;
    stage(0); stage(1); stage(2); stage(3);
    wait_vm<12>();
    barrier();
    read_frags(0, I0{});
    wait_vm<8>();
    wait_lds();
    barrier();
    int kt = 0;
    for (; kt < nk - 4; kt += 2) {
        phase(kt, I0{}, T{}, T{}, std::integral_constant<int, 8>{});
        phase(kt + 1, I1{}, T{}, T{}, std::integral_constant<int, 8>{});
    }
    phase(kt, I0{}, F{}, T{}, std::integral_constant<int, 4>{});
    phase(kt + 1, I1{}, F{}, T{}, std::integral_constant<int, 0>{});
    phase(kt + 2, I0{}, F{}, T{}, std::integral_constant<int, -1>{});
    phase(kt + 3, I1{}, F{}, F{}, std::integral_constant<int, -1>{});

    // ---- epilogue (as gemm8p128_kernel)
    char* reg = smem + wave * 8192;
    f32x4 b4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        b4[j] = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + nb + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.out_f32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if (p.act == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if (p.act == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const int row = i * 16 + fr;
                *reinterpret_cast<f32x4*>(reg + row * 128 + (((j * 4 + fq) ^ swz) << 4)) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = u * 8 + prow;
            f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4));
            const int m = m0 + wr * 64 + row;
            if (p.drop_p > 0.f) {            // dropout of the linear's output before the residual add (training: hidden / final dropout): mask of mi_dropout_add_f32 for (m, n)
                float k4[4];
                mask_keep4(p.drop_key, (unsigned long long)m * (unsigned)(p.N >> 2) + (unsigned)((nb >> 2) + pc), p.drop_p, 1.f / (1.f - p.drop_p), k4);
                v = f32x4{v.x * k4[0], v.y * k4[1], v.z * k4[2], v.w * k4[3]};
            }
            if (use_res) v = rres[u] + p.alpha * v;
            if (m < p.M) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + nb + pc * 4) = v;
            // LayerNorm-fold producer (gemm_args.hpp): the bf16 copy of the stored row segment and its partial (sum, sumsq) — 8 lanes hold a row's 32 columns
            if (p.C2 && m < p.M) *reinterpret_cast<bf16x4*>(p.C2 + (long)m * p.ldc2 + nb + pc * 4) = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            if (p.stats_out) {
                float sm = (v.x + v.y) + (v.z + v.w), sq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                sm += dpp_f32<0xB1, 0xF>(0.f, sm); sq += dpp_f32<0xB1, 0xF>(0.f, sq);          // lane ^ 1
                sm += dpp_f32<0x4E, 0xF>(0.f, sm); sq += dpp_f32<0x4E, 0xF>(0.f, sq);          // lane ^ 2
                sm += dpp_f32<0x141, 0xF>(0.f, sm); sq += dpp_f32<0x141, 0xF>(0.f, sq);        // row_half_mirror: the other quad of the 8 lanes
                if (pc == 0 && m < p.M) *reinterpret_cast<f32x2*>(p.stats_out + (long)m * LN_STATS_STRIDE + (((n0 >> 7) << 2) + wc) * 2) = f32x2{sm, sq};
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if (p.act == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if (p.act == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
                const int row = i * 16 + fr;
                *reinterpret_cast<bf16x4*>(reg + row * 128 + (((j * 2 + (fq >> 1)) ^ (swz & 3)) << 4) + (fq & 1) * 8) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        const int r4 = lane >> 2, c4 = lane & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = u * 16 + r4;
            uint4 v = *reinterpret_cast<const uint4*>(reg + row * 128 + ((c4 ^ ((row >> 1) & 3)) << 4));
            const int m = m0 + wr * 64 + row;
            if (p.drop_p > 0.f) {            // dropout on the bf16 rows (mask and rounding of mi_dropout's bf16 form for element m * N + n)
                bf16x8 e = __builtin_bit_cast(bf16x8, v);
                const unsigned long long quad0 = ((unsigned long long)m * (unsigned)(p.N >> 3) + (unsigned)((nb >> 3) + c4)) << 1;
                float k8[8];
                mask_keep4(p.drop_key, quad0, p.drop_p, 1.f / (1.f - p.drop_p), k8);
                mask_keep4(p.drop_key, quad0 + 1, p.drop_p, 1.f / (1.f - p.drop_p), k8 + 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = f2bf(bf2f(e[q]) * k8[q]);
                v = __builtin_bit_cast(uint4, e);
            }
            if (m < p.M) *reinterpret_cast<uint4*>(C + (long)m * p.ldc + nb + c4 * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// 128 x 128 x 64, loader / consumer form (round 5).  tools/gemm_floor_ab.sh on the pipelined form above: its K loop costs 0.44 us per K tile, its staging alone 0.28, its
// fragment reads + MFMAs alone 0.22 (the matrix pipe's own time) — and ANY TWO of {staging, reads, MFMAs} overlap perfectly (0.26 - 0.29) while all three together do not:
// a wave that has a phase's LDS-DMA pieces to issue sits in their issue (100 - 185 cycles per piece inside a phase that also reads LDS, MI355X_MICROARCH.md) instead of
// feeding the matrix pipe, and with all eight waves symmetric both waves of a SIMD do so at the same time.  Here the roles are split: waves 0-3 (one per SIMD) own a
// 64 x 64 quarter of the tile each — fragment sets of K tiles t and t+1 in registers, 16 ds_read_b128 under 32 MFMAs, never a vector-memory instruction in the loop —
// and waves 4-7 (their SIMD partners) issue the ring's LDS-DMA pieces (waves 4, 5 the A rows, 6, 7 the W rows: 8 pieces per wave and K tile) behind a counted vmcnt.
// NOT the product's choice (see gemm_8p128_launch: it loses inside the step); kept as variant 43 — the measured answer to "would dedicated loader waves lift the K loop".
// Same ring (four 32-KiB K tiles), same one barrier per K tile, same K order and MFMA shape as the pipelined form: the same bits.
//   loader, phase t:   stage K tile t+4 into buffer t & 3 (its reads were retired before the barrier that ended phase t-1); vmcnt(16): tile t+2 has landed; s_barrier
//   consumer, phase t: read K tile t+1 -> fragment set (t+1) & 1  ||  MFMAs of K tile t from set t & 1;  lgkmcnt(0);  s_barrier
// Epilogue by all eight waves: the consumers leave acc + bias as an fp32 tile in LDS (the ring is dead; 16-B chunk ^ (row & 31): conflict-free both ways), then every wave
// finishes 16 rows — residual rows requested at kernel start (the loaders' registers are free, the consumers stay under 256), whole 512-B / 256-B row segments out.
__global__ __launch_bounds__(512, 2) void gemm8p128l_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (p.M + 127) / 128, ntn = p.N / 128;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / ntn) * 128, n0 = (bid % ntn) * 128;
    const int nk = p.K / BK;
#ifdef GEMM128_STAMPS
    const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif

    // final pass: wave w finishes rows [16 w, 16 w + 16) of the tile, two rows per instruction (32 lanes x 16 B = one 512-B fp32 row)
    const int er = lane >> 5, ec = lane & 31;
    f32x4 rres[8];
    const bool use_res = p.out_f32 && p.resid != nullptr;
    if (use_res) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = min(m0 + wave * 16 + u * 2 + er, p.M - 1);
            rres[u] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n0 + ec * 4);
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wr = (wave >> 1) & 1, wc = wave & 1;          // consumers: 64-row / 64-column half of the tile
    const int fr = lane & 15, fq = lane >> 4;

    if (wave >= 4) {
        // ---------------- loader
        const int prow = lane >> 3, pc = lane & 7;
        const bool isA = wave < 6;
        const char* src_base = reinterpret_cast<const char*>(isA ? (const void*)p.A : (const void*)p.W);
        unsigned off[8], dst[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r0 = (8 * (wave & 1) + e) * 8;
            const int lc = pc ^ (((r0 + prow) >> 1) & 7);
            if (isA) off[e] = (unsigned)min(m0 + r0 + prow, p.M - 1) * (unsigned)(p.lda * 2) + lc * 16;
            else off[e] = (unsigned)min(n0 + r0 + prow, p.N - 1) * (unsigned)(p.ldw * 2) + lc * 16;
            dst[e] = (isA ? 0 : B128_BOFF) + r0 * 128;
        }
        auto stage = [&](int kt) {
            if (GEMM_FLOOR == 2) return;
            const unsigned buf = (unsigned)(kt & 3) * B128_BUF;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __builtin_amdgcn_global_load_lds((gptr_t)(src_base + off[e] + (unsigned)kt * 128u), (lptr_t)(smem + buf + dst[e]), 16, 0, 0);
        };
        stage(0); stage(1); stage(2); stage(3);
        wait_vm<24>();                      // tile 0 (and the residual rows, older still) has landed
        barrier();
        wait_vm<16>();                      // tile 1
        barrier();
        int kt = 0;
#ifdef GEMM128_STAMPS                       // instrumented build (tools/gemm128l_stamps.py): where a loader wave's K loop goes — issuing, waiting for its pieces to land, waiting at the barrier
        unsigned long long t_iss = 0, t_vm = 0, t_bar = 0;
        const unsigned long long t_begin = __builtin_readcyclecounter();
        for (; kt < nk - 4; ++kt) {
            const unsigned long long a0 = __builtin_readcyclecounter();
            stage(kt + 4);
            const unsigned long long a1 = __builtin_readcyclecounter();
            wait_vm<16>();
            const unsigned long long a2 = __builtin_readcyclecounter();
            barrier();
            const unsigned long long a3 = __builtin_readcyclecounter();
            t_iss += a1 - a0; t_vm += a2 - a1; t_bar += a3 - a2;
        }
        if (wave == 4 && lane == 0) {
            float* dbg = reinterpret_cast<float*>(p.C) + (long)p.M * p.ldc * (p.out_f32 ? 1 : 0) + blockIdx.x * 8;       // fp32 output only: row M.. of the (padded) output buffer
            if (p.out_f32) { dbg[0] = (float)t_iss; dbg[1] = (float)t_vm; dbg[2] = (float)t_bar; dbg[3] = (float)(__builtin_readcyclecounter() - t_begin); }
        }
#else
        for (; kt < nk - 4; ++kt) {
            stage(kt + 4);
            wait_vm<16>();                  // tiles kt+3, kt+4 stay in flight: kt+2 has landed
            barrier();
        }
#endif
        wait_vm<8>();  barrier();           // phase nk-4: tile nk-2
        wait_vm<0>();  barrier();           // phase nk-3: tile nk-1
        barrier();                          // phase nk-2
        barrier();                          // phase nk-1
    } else {
        // ---------------- consumer
        const int swz = (fr >> 1) & 7;
        const unsigned low = fr * 128 + ((fq ^ swz) << 4);
        const unsigned aRd0 = wr * 64 * 128 + low, bRd0 = B128_BOFF + wc * 64 * 128 + low;
        bf16x8 fa[2][4][2], fb[2][4][2];
        auto read_frags = [&](int kt, auto SET) {
            if (GEMM_FLOOR == 1 || GEMM_FLOOR == 3) return;
            constexpr int S = decltype(SET)::value;
            const unsigned cb = (unsigned)(kt & 3) * B128_BUF;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) fb[S][j][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + bRd0) ^ (s * 64)) + j * 16 * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) fa[S][i][s] = *reinterpret_cast<const bf16x8*>(smem + ((cb + aRd0) ^ (s * 64)) + i * 16 * 128);
        };
        auto mfmas = [&](auto SET) {
            constexpr int S = decltype(SET)::value;
            if (GEMM_FLOOR == 1) return;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][j][s], fa[S][i][s], acc[i][j], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
#ifdef GEMM128_STAMPS
        unsigned long long s_lds = 0, s_bar = 0;
#endif
        auto phase = [&](int kt, auto CUR, auto DO_READ) {
            constexpr int C = decltype(CUR)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(DO_READ)::value) read_frags(kt + 1, std::integral_constant<int, 1 - C>{});
            mfmas(CUR);
            if constexpr (decltype(DO_READ)::value) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef GEMM128_STAMPS
            const unsigned long long c0 = __builtin_readcyclecounter();       // (s_memtime + lgkmcnt(0): behind the phase's last MFMA issue, so c1 - c0 is the wait for the LDS reads)
            if constexpr (decltype(DO_READ)::value) wait_lds();
            const unsigned long long c1 = __builtin_readcyclecounter();
            barrier();
            const unsigned long long c2 = __builtin_readcyclecounter();
            s_lds += c1 - c0; s_bar += c2 - c1;
#else
            if constexpr (decltype(DO_READ)::value) wait_lds();
            barrier();
#endif
        };
        using T = std::true_type;
        using F = std::false_type;
        barrier();
        read_frags(0, I0{});
        wait_lds();
        barrier();
        int kt = 0;
        for (; kt < nk - 2; kt += 2) {
            phase(kt, I0{}, T{});
            phase(kt + 1, I1{}, T{});
        }
        phase(kt, I0{}, T{});
        phase(kt + 1, I1{}, F{});
#ifdef GEMM128_STAMPS
        if (wave == 0 && lane == 0 && p.out_f32) {
            float* dbg = reinterpret_cast<float*>(p.C) + (long)p.M * p.ldc + blockIdx.x * 8;
            dbg[4] = (float)s_lds; dbg[5] = (float)s_bar; dbg[6] = (float)(__builtin_readcyclecounter() - t_kernel0);      // kernel start -> end of the K loop
        }
#endif
    }

    // ---- epilogue: consumers -> fp32 tile [128][512 B] in LDS, 16-B chunk c of row r at chunk c ^ (r & 31)
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wc * 64 + j * 16 + fq * 4;
            const f32x4 b4 = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][j] + b4;
                if (p.act == 1) v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                else if (p.act == 2) v = f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
                const int row = wr * 64 + i * 16 + fr;
                const int ch = wc * 16 + j * 4 + fq;
                *reinterpret_cast<f32x4*>(smem + row * 512 + ((ch ^ (row & 31)) << 4)) = v;
            }
        }
    }
    __syncthreads();
    if (p.out_f32) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = wave * 16 + u * 2 + er;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 512 + ((ec ^ (row & 31)) << 4));
            const int m = m0 + row, n = n0 + ec * 4;
            if (p.drop_p > 0.f) {            // dropout of the linear's output before the residual add (training: hidden / final dropout): mask of mi_dropout_add_f32 for (m, n)
                float k4[4];
                mask_keep4(p.drop_key, (unsigned long long)m * (unsigned)(p.N >> 2) + (unsigned)(n >> 2), p.drop_p, 1.f / (1.f - p.drop_p), k4);
                v = f32x4{v.x * k4[0], v.y * k4[1], v.z * k4[2], v.w * k4[3]};
            }
            if (use_res) v = rres[u] + p.alpha * v;
            if (m < p.M) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = v;
            if (p.C2 && m < p.M) *reinterpret_cast<bf16x4*>(p.C2 + (long)m * p.ldc2 + n) = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            if (p.stats_out) {               // LayerNorm-fold producer: per-row partial (sum, sumsq) per 32 columns — 8 lanes hold them, same order as the pipelined form
                float sm = (v.x + v.y) + (v.z + v.w), sq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                sm += dpp_f32<0xB1, 0xF>(0.f, sm); sq += dpp_f32<0xB1, 0xF>(0.f, sq);
                sm += dpp_f32<0x4E, 0xF>(0.f, sm); sq += dpp_f32<0x4E, 0xF>(0.f, sq);
                sm += dpp_f32<0x141, 0xF>(0.f, sm); sq += dpp_f32<0x141, 0xF>(0.f, sq);
                if ((ec & 7) == 0 && m < p.M) *reinterpret_cast<f32x2*>(p.stats_out + (long)m * LN_STATS_STRIDE + (((n0 >> 7) << 2) + (ec >> 3)) * 2) = f32x2{sm, sq};
            }
        }
    } else {
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = wave * 16 + u * 2 + er;
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 512 + ((ec ^ (row & 31)) << 4));
            const int m = m0 + row, n = n0 + ec * 4;
            bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            if (p.drop_p > 0.f) {            // dropout on the bf16 rows (mask and rounding of mi_dropout's bf16 form for element m * N + n)
                float k4[4];
                mask_keep4(p.drop_key, ((unsigned long long)m * (unsigned)p.N + (unsigned)n) >> 2, p.drop_p, 1.f / (1.f - p.drop_p), k4);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = f2bf(bf2f(o[q]) * k4[q]);
            }
            if (m < p.M) *reinterpret_cast<bf16x4*>(C + (long)m * p.ldc + n) = o;
        }
    }
}

}  // namespace

bool gemm_8p_supported(const GemmArgs& a, bool conv) {
    if (a.M <= 0 || a.N <= 0 || (a.K % BK) != 0 || a.K < 2 * BK) return false;
    if (a.col_T || a.bias_mode == 2) return false;
    if (a.resid && (!a.out_f32 || (a.N % TB) != 0 || ((uintptr_t)a.resid & 15) || (a.ldr % 4) != 0)) return false;        // residual: the fp32 form only, whole tiles
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.C & 15) || (a.ldw % 8)) return false;
    if (a.out_f32) {                 // fp32 output (CTC head; the residual GEMMs in throughput mode): no activation, 16-B stores may touch the row's padding up to the next multiple of 4
        if (conv || a.act != 0 || (a.ldc % 4) != 0 || a.ldc < ((a.N + 3) & ~3)) return false;
    } else {
        if ((a.N % TB) != 0 || (a.ldc % 8) != 0) return false;
        if (a.bias_mode == 1 && ((uintptr_t)a.bias & 15)) return false;
    }
    if (a.gated && (!conv || a.out_f32 || a.act != 1)) return false;
    if (a.ln_stats) {                // folded LayerNorm: plain bf16-out GEMM, act none / erf-GELU, 16-B aligned column vectors, a partial count the prologue can split over two lanes
        if (conv || a.out_f32 || a.gated || a.act > 1 || a.bias_mode != 1 || !a.ln_colsum || ((uintptr_t)a.ln_colsum & 15) || ((uintptr_t)a.ln_stats & 15)) return false;
        if (!(a.ln_npart == 1 || (a.ln_npart >= 4 && a.ln_npart <= 16 && (a.ln_npart % 4) == 0))) return false;
    }
    if (a.act >= 3) {                // training epilogues: plain bf16-out GEMM, 16-B aligned rows of the extra operand
        if (conv || a.out_f32 || a.gated || a.ln_stats || a.resid || a.col_T || a.act > 4 || (a.aux_kind != 1 && a.aux_kind != 2) || a.drop_p < 0.f || a.drop_p >= 1.f) return false;
        if (a.act == 3 && (!a.aux || ((uintptr_t)a.aux & 15) || (a.ldaux % 8))) return false;
        if (a.act == 4 && (!a.C2 || ((uintptr_t)a.C2 & 15) || (a.ldc2 % 8))) return false;
    } else if (a.C2 || a.stats_out) {            // LayerNorm-fold producer on this tile: the fp32 residual form, at most 16 partial pairs per row (one per 64 columns)
        if (!a.out_f32 || !a.resid || a.N > 1024 || (a.C2 && (((uintptr_t)a.C2 & 7) || (a.ldc2 % 4))) || ((uintptr_t)a.stats_out & 7)) return false;
    }
    if (a.lse_part && (conv || !a.out_f32 || a.resid || a.C2 || a.stats_out || a.act != 0 || ((uintptr_t)a.lse_part & 7) || (a.lse_ld % 2) || a.lse_ld < 8 * cdiv(a.N, TB))) return false;
    if ((long)a.N * a.ldw * 2 >= (1l << 32)) return false;                                             // 32-bit source offsets
    if (conv) {
        if ((a.Cin % BK) != 0 || a.Fout <= 0 || a.Tout <= 0) return false;
        if ((long)cdiv(a.M, (long)a.Fout * a.Tout) * a.Tin * a.Fin * a.Cin * 2 >= (1l << 32)) return false;
    } else if ((a.lda % 8) || (long)a.M * a.lda * 2 >= (1l << 32)) return false;
    return true;
}

// kernel attributes are set once per (kernel, device) through common.hpp's table (a function-local static configured only the device current at the first call)
template <typename K>
static bool set_lds_attr(K kernel, int bytes) { return ensure_dynamic_lds_ptr(reinterpret_cast<const void*>(kernel), (size_t)bytes); }     // per (kernel, device)

int gemm_8p_launch(const GemmArgs& a, bool conv, hipStream_t stream) {
    using kern_t = void (*)(GemmArgs);
    static const kern_t kerns[2][3] = {{gemm8p_kernel<false, 0>, gemm8p_kernel<false, 1>, gemm8p_kernel<false, 2>},
                                       {gemm8p_kernel<true, 0>, gemm8p_kernel<true, 1>, gemm8p_kernel<true, 2>}};
    // (the kernel actually launched below is one of these six; configuring only that one keeps the per-call cost at one table lookup)
    if (a.act >= 0 && a.act <= 2 && !set_lds_attr(kerns[conv ? 1 : 0][a.act], 2 * BUF)) return MI_ERR_LAUNCH;
    if (a.act < 0 || a.act > 4) return MI_ERR_ARG;
    const int grid = cdiv(a.M, TB) * cdiv(a.N, TB);
    if (a.act >= 3) {
        const bool attr_t = set_lds_attr(gemm8p_kernel<false, 3>, 2 * BUF) && set_lds_attr(gemm8p_kernel<false, 4>, 2 * BUF);
        (void)attr_t;
        if (a.act == 3) launch_dense(PF_8P, gemm8p_kernel<false, 3>, dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
        else launch_dense(PF_8P_GELU, gemm8p_kernel<false, 4>, dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
        return MI_OK;
    }
    if (a.out_f32 && a.lse_part) {
        const bool attr_e = set_lds_attr(gemm8p_kernel<false, 0, true, false, false, true>, 2 * BUF);
        (void)attr_e;
        launch_dense(PF_8P_OUT32, gemm8p_kernel<false, 0, true, false, false, true>, dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
        return MI_OK;
    }
    if (a.out_f32) {
        const bool attr32 = set_lds_attr(gemm8p_kernel<false, 0, true>, 2 * BUF);
        (void)attr32;
        launch_dense(PF_8P_OUT32, gemm8p_kernel<false, 0, true>, dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
        return MI_OK;
    }
    if (a.ln_stats) {
        const bool attr_l = set_lds_attr(gemm8p_kernel<false, 0, false, false, true>, 2 * BUF + 2048) && set_lds_attr(gemm8p_kernel<false, 1, false, false, true>, 2 * BUF + 2048);
        (void)attr_l;
        if (a.act) launch_dense(PF_8P_GELU, gemm8p_kernel<false, 1, false, false, true>, dim3(grid), dim3(512), (size_t)2 * BUF + 2048, stream, a);
        else launch_dense(PF_8P, gemm8p_kernel<false, 0, false, false, true>, dim3(grid), dim3(512), (size_t)2 * BUF + 2048, stream, a);
        return MI_OK;
    }
    if (a.gated) {
        const bool attr_g = set_lds_attr(gemm8p_kernel<true, 1, false, true>, 2 * BUF);
        (void)attr_g;
        launch_dense(PF_8P_CONV, gemm8p_kernel<true, 1, false, true>, dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
        return MI_OK;
    }
    launch_dense(conv ? PF_8P_CONV : (a.act ? PF_8P_GELU : PF_8P), kerns[conv ? 1 : 0][a.act], dim3(grid), dim3(512), (size_t)2 * BUF, stream, a);
    return MI_OK;
}

bool gemm_8p128_supported(const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || (a.N % 128) != 0 || (a.K % BK) != 0 || a.K < 5 * BK) return false;
    if (a.col_T || a.bias_mode == 2) return false;
    if (a.resid && !a.out_f32) return false;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.C & 15) || (a.lda % 8) || (a.ldw % 8)) return false;
    if (a.out_f32 ? (a.ldc % 4) != 0 : (a.ldc % 8) != 0) return false;
    if (a.resid && (((uintptr_t)a.resid & 15) || (a.ldr % 4) != 0)) return false;
    if (a.bias_mode == 1 && ((uintptr_t)a.bias & 15)) return false;
    if ((long)a.M * a.lda * 2 >= (1l << 32) || (long)a.N * a.ldw * 2 >= (1l << 32)) return false;
    if (a.ln_stats) return false;                // consumer side lives in the 256x256 kernel
    if ((a.C2 || a.stats_out) && (!a.out_f32 || a.N > 512 || (a.K % 128) != 0 || (a.C2 && (((uintptr_t)a.C2 & 7) || (a.ldc2 % 4))) || ((uintptr_t)a.stats_out & 7))) return false;
    return true;
}

int gemm_8p128_launch(const GemmArgs& a, int ring, hipStream_t stream) {
    const bool attr_set = set_lds_attr(gemm8p128_kernel<4>, 4 * B128_BUF);
    (void)attr_set;
    const int grid = cdiv(a.M, 128) * (a.N / 128);
    // the product's form for an even number of K tiles is the register-pipelined one.  The loader / consumer form wins the isolated replay from K = 2048 on (24.8 -> 24.1 us,
    // 47.4 -> 43.6 at K = 5120: its K loop runs at the staging floor) but LOSES inside the step, where its launches meet cold caches and its block-wide epilogue is exposed:
    // same-box A/B of the dispatch rule "loader form from K = 2048 on" (tools/gemm128l_step_ab.sh, three alternations): forward one step at a time 5.01 -> 5.08 ms, base
    // training step 18.05 -> 18.37 ms.  It stays reachable as variant 43 (tests, A/B); GEMM128_LOADER = 1 builds a library that takes it from K = 2048 on.
    if (ring == 0) ring = (GEMM128_LOADER == 1 && a.K >= 2048) ? 2 : 3;
    if (ring == 2) {          // loader / consumer form (even number of K tiles, at least four)
        const bool attr_l = set_lds_attr(gemm8p128l_kernel, 4 * B128_BUF);
        (void)attr_l;
        launch_dense(PF_8P128, gemm8p128l_kernel, dim3(grid), dim3(512), (size_t)4 * B128_BUF, stream, a);
        return MI_OK;
    }
    if (ring == 3) {          // register-pipelined form (even number of K tiles)
        const bool attr_p = set_lds_attr(gemm8p128p_kernel, 4 * B128_BUF);
        (void)attr_p;
        launch_dense(PF_8P128, gemm8p128p_kernel, dim3(grid), dim3(512), (size_t)4 * B128_BUF, stream, a);
        return MI_OK;
    }
    launch_dense(PF_8P128, gemm8p128_kernel<4>, dim3(grid), dim3(512), (size_t)4 * B128_BUF, stream, a);
    return MI_OK;
}
