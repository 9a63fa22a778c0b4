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


// Symmetric kernel: every wave both issues its share of the LDS-DMA pieces and runs MFMAs.
//   <128,128,2,2,S=2>: 4 waves, 64 KiB ring -> two blocks per CU (the second block's main loop covers the first one's
//                      prologue/epilogue);  <256,256,2,4,S=2>: 8 waves (wave tile 128x64), 128 KiB ring, half the L2->LDS
//                      bytes per flop — for GEMMs whose 256x256 tile count still fills the chip.
template <int TBM, int TBN, int CWM, int CWN, int STAGES, bool CONV>
__device__ __forceinline__ void gemm_glds_tile(const GemmArgs& p, int bid, char* smem, int next_bid, bool prologue_done) {
    constexpr int NW = CWM * CWN, NTH = NW * 64;
    constexpr int WM = TBM / CWM, WN = TBN / CWN, MI = WM / 32, NI = WN / 32;
    constexpr int STG = (TBM + TBN) * BK * 2, ABYTES = TBM * BK * 2;
    constexpr int PPW = (TBM + TBN) / 8 / NW;          // 1-KiB pieces per wave per K tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CWN, wn = wave % CWN;

    const int ntm = (p.M + TBM - 1) / TBM, ntn = (p.N + TBN - 1) / TBN;
    const int nwg = ntm * ntn;
    // XCD-aware bijective remap: the blocks of one XCD walk N fastest within an A row panel
    auto remap = [&](int id) {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, idx = id >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    bid = remap(bid);
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
                c_fi[q] = fo * p.stride_f - p.pad_f;
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
        int kh = 0, kw = 0, c0 = 0, kofs = kt * BK;
        if (CONV) {                                        // Cin % 64 == 0: one tap per K tile; tiles visited 64-channel slice by slice, as gemm8p_kernel does (same
            const int ntap = p.K / p.Cin;                  // accumulation order in both kernels: a batch of one and a batch of 32 give the same bits)
            const int chunk = kt / ntap, tap = kt - chunk * ntap;
            c0 = chunk * BK;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
            kofs = tap * p.Cin + c0;
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
                sp = src[q] + kofs;
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
    auto ktile = [&](int t) { return t; };

    if (!prologue_done) {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) issue(ktile(t), t);
    }

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt % STAGES;
        // tile kt has landed for THIS wave's pieces once at most the younger in-flight tiles' pieces are outstanding
        const int younger = min(STAGES - 2, nk - 1 - kt);
        static_assert(STAGES <= 8 && 6 * PPW <= 63, "vmcnt immediates below");
        switch (younger) {                                  // the count is an immediate: one case per depth the ring can have
            case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PPW) : "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PPW) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");    // every wave's pieces of tile kt landed; stage of tile kt-1 is free
        if (kt + STAGES - 1 < nk) issue(ktile(kt + STAGES - 1), (kt + STAGES - 1) % STAGES);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // C^T tile: lane = m, regs = n
        }
    }
    // ---- cross-tile prefetch (persistent grid): the first K tile of this block's NEXT output tile is requested before the epilogue,
    // so its L2->LDS latency hides under the epilogue's stores.  Stage 0 is free once every wave has left the main loop (barrier).
    auto prefetch_next = [&]() {
        if (CONV || STAGES != 2 || next_bid < 0) return;
        const int nb_ = remap(next_bid);
        const int nm0 = (nb_ / ntn) * TBM, nn0 = (nb_ % ntn) * TBN;
        asm volatile("s_barrier" ::: "memory");
        char* sbase = smem + wave * PPW * 1024;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int g = wave * PPW + q;
            const bool isA = g < TBM / 8;
            const int row = (isA ? g : g - TBM / 8) * 8 + prow;
            const int lc = pc ^ fsw(row);
            const bf16_t* sp = isA ? p.A + (long)min(nm0 + row, p.M - 1) * p.lda + lc * 8 : p.W + (long)min(nn0 + row, p.N - 1) * p.ldw + lc * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(sbase + q * 1024), 16, 0, 0);
        }
    };
    // ---- epilogue straight from the accumulators, no LDS, no block barrier: the MFMAs were issued as W·A^T, so a lane owns
    // output ROW m = lane&31 and its registers walk the COLUMNS: 4 consecutive n per register group -> fp32 leaves as 16-B
    // stores, bf16 is widened to 16 B per lane with v_permlane32_swap (halves hold n+0..3 / n+4..7 of the same row).
    const bool vec_ok = (p.col_T == 0) && (n0 + TBN <= p.N) && ((p.bias_mode != 1) || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
                        (p.out_f32 ? ((p.ldc & 3) == 0) : ((p.ldc & 7) == 0)) && (!p.resid || (p.ldr & 3) == 0);
    // bias / residual values are fetched ahead of the stores: C may alias resid (in-place x += ...), so the compiler cannot
    // hoist loads above earlier stores by itself and every tile would otherwise pay a full memory latency.
    constexpr bool SMALL = (MI * NI <= 4);            // big wave tiles (128 accumulator registers) have no room for look-ahead
    f32x4 bcol[SMALL ? NI : 1][4];
    auto load_bias = [&](int j, f32x4 (&bc)[4]) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            bc[g4] = (p.bias_mode == 1) ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WN + j * 32 + 8 * g4 + 4 * lh)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if (vec_ok && SMALL) {
#pragma unroll
        for (int j = 0; j < NI; ++j) load_bias(j, bcol[SMALL ? j : 0]);
    }
    auto load_resid_ij = [&](int i, int j, f32x4 (&rr)[4]) {
        const int m = m0 + wm * WM + i * 32 + lr;
        const int nb = n0 + wn * WN + j * 32;
        if (m < p.M) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) rr[g4] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + 8 * g4 + 4 * lh);
        }
    };
    auto load_resid = [&](int t, f32x4 (&rr)[4]) {
        const int i = t / NI, j = t % NI;
        const int m = m0 + wm * WM + i * 32 + lr;
        const int nb = n0 + wn * WN + j * 32;
        if (m < p.M) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) rr[g4] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + nb + 8 * g4 + 4 * lh);
        }
    };
    f32x4 rcur[4], rnext[4];
    const bool use_res = vec_ok && p.resid != nullptr;
    if (SMALL && use_res) load_resid(0, rcur);
    prefetch_next();            // after the epilogue's own early loads: a later load would have to wait behind the prefetch (in-order vmcnt)
#pragma unroll
    for (int t = 0; t < MI * NI; ++t) {
        // SMALL: row-major tile order with all biases preloaded; else column-block-major so one bias fetch serves MI tiles
        const int i = SMALL ? t / NI : t % MI, j = SMALL ? t % NI : t / MI;
        if (!SMALL && vec_ok && i == 0) load_bias(j, bcol[0]);
        const int m = m0 + wm * WM + i * 32 + lr;
        const bool mok = m < p.M;
        const float brow = (p.bias_mode == 2 && mok) ? p.bias[m] : 0.f;
        const int nb = n0 + wn * WN + j * 32;
        if (vec_ok) {
            if (SMALL && use_res && t + 1 < MI * NI) load_resid(t + 1, rnext);
            if (!SMALL && use_res) load_resid_ij(i, j, rcur);
            f32x4 v[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                v[g4] = f32x4{acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                v[g4] += bcol[SMALL ? j : 0][g4];
                v[g4] += brow;
                if (p.act == 1) v[g4] = f32x4{gelu_erf(v[g4].x), gelu_erf(v[g4].y), gelu_erf(v[g4].z), gelu_erf(v[g4].w)};
                else if (p.act == 2) v[g4] = f32x4{gelu_tanh(v[g4].x), gelu_tanh(v[g4].y), gelu_tanh(v[g4].z), gelu_tanh(v[g4].w)};
                if (use_res) v[g4] = rcur[g4] + p.alpha * v[g4];
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
            if (SMALL && use_res) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) rcur[g4] = rnext[g4];
            }
        } else if (mok) {
            // generic edge path (odd N / unaligned ld / column remap): element stores, correct but not coalesced
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nb + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n >= p.N) continue;
                const long nc = p.col_T ? (long)(n / p.col_T) * p.col_Tp + (n % p.col_T) : n;
                float vv = acc[i][j][e] + ((p.bias_mode == 1) ? p.bias[n] : brow);
                if (p.act == 1) vv = gelu_erf(vv);
                    else if (p.act == 2) vv = gelu_tanh(vv);
                if (p.resid) vv = p.resid[(long)m * p.ldr + n] + p.alpha * vv;
                if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + nc] = vv;
                else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + nc] = f2bf(vv);
            }
        }
    }
}

// Grid-stride over output tiles: with gridDim.x == number of tiles this is the plain one-tile-per-block launch; with a persistent grid
// (two blocks per CU) a block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... — half the workgroup dispatches for the same work.
template <int TBM, int TBN, int CWM, int CWN, int STAGES, bool CONV>
__global__ __launch_bounds__(CWM * CWN * 64, (TBM <= 128 && TBN <= 128 && STAGES == 2) ? 2 : 1) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = ((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
    const bool pf = !CONV && STAGES == 2;
    for (int tile = blockIdx.x; tile < nwg; tile += gridDim.x) {
        const bool first = tile == (int)blockIdx.x;
        if (!first && !pf) asm volatile("s_barrier" ::: "memory");     // every wave is done reading the previous tile's last LDS stage
        const int next = (pf && tile + (int)gridDim.x < nwg) ? tile + (int)gridDim.x : -1;
        gemm_glds_tile<TBM, TBN, CWM, CWN, STAGES, CONV>(p, tile, smem, next, pf && !first);
    }
}
}  // namespace

bool gemm_glds_supported(const GemmArgs& a, bool conv) {
    if ((a.K % BK) != 0 || a.M <= 0 || a.N <= 0) return false;
    if (((uintptr_t)a.W & 15) || (a.ldw % 8) != 0) return false;
    if (conv) { if ((a.Cin % BK) != 0 || ((uintptr_t)a.A & 15)) return false; }
    else if (((uintptr_t)a.A & 15) || (a.lda % 8) != 0) return false;
    if (((uintptr_t)a.C & 15) || (a.resid && ((uintptr_t)a.resid & 15))) return false;
    return true;
}

// Kernel selection.  a.variant (mi_gemm_bf16_v / mi_conv2d_cl_bf16_v; 0 from every product call site) exists for A/B runs and for tests that must reach a kernel
// at a size its default dispatch would not pick:  40 = phase kernels wherever supported, 41 = never (this file's kernels only), 32 = this file's 32x64 small-M tiles, 42 = 128x128 phase kernel wherever
// supported, 47 = its two-segment form, 30 = this file's 128x128 tiles, 31 = the same, one block per tile.
int gemm_glds_launch(const GemmArgs& a, bool conv, hipStream_t stream) {
    if (!gemm_glds_supported(a, conv)) return MI_ERR_UNSUPPORTED;
    const bool tail_split = a.variant != 48;                  // 48 = the product's dispatch without the tail-round split below (callers that keep several batches in flight)
    const int v = a.variant == 48 ? 0 : a.variant;
    const bool phase_ok = v != 41 && v != 30 && v != 31 && v != 32;
    // wide-N bf16-out GEMMs (FFN in, cgMLP in, QKV) and the implicit-GEMM conv: 256x256 tiles on the phase-interleaved schedule (gemm_8p.hip) once the tiles fill half the chip
    const int t256 = cdiv(a.M, 256) * cdiv(a.N, 256);
    // Wave quantization (round 4): t256 tiles run in ceil(t256 / 256) rounds of one tile per CU; when the last round would be less than half full its rows go to the
    // 128 x 128 phase kernel instead — quarter-size tiles, so the tail costs a quarter of a round or less.  Whisper-small at 16 x 30 s (M = 24 000): N = 768 is 282
    // tiles = TWO rounds for 1.1 rounds of work, N = 2304 / 3072 are 3.3 / 4.4 rounds.  Same K order and MFMA shape in both kernels: the same bits as one launch.
    // Same-box A/B on config 4 (tools/tail_split_ab.sh): one batch at a time 8.94 -> 8.72 ms (dense kernel time 5.53 -> 5.05 ms per step); with three batches in
    // flight the other batches' blocks already fill the tail and the extra launches cost 3 % (7.53 -> 7.77 ms): such callers pass variant 48.
#ifndef HFASR_NO_TAIL_SPLIT                                   // (A/B builds of tools/tail_split_ab.sh only)
    if (tail_split && !conv && v == 0 && t256 > 256 && (t256 % 256) > 0 && (t256 % 256) < 128 && a.act <= 2 && a.drop_p == 0.f && !a.ln_stats && !a.gated && gemm_8p_supported(a, false)) {
        const int ntn = cdiv(a.N, 256);
        const int mt_main = (t256 / 256) * 256 / ntn;                     // M tiles that fill whole rounds
        const int m_main = mt_main * 256;
        if (mt_main > 0 && m_main < a.M) {
            GemmArgs a1 = a, a2 = a;
            a1.M = m_main;
            a2.M = a.M - m_main;
            a2.A = a.A + (long)m_main * a.lda;
            a2.C = reinterpret_cast<char*>(a.C) + (long)m_main * a.ldc * (a.out_f32 ? 4 : 2);
            if (a.resid) a2.resid = a.resid + (long)m_main * a.ldr;
            if (a.C2) a2.C2 = a.C2 + (long)m_main * a.ldc2;
            if (a.stats_out) a2.stats_out = a.stats_out + (long)m_main * LN_STATS_STRIDE;
            if (gemm_8p128_supported(a2) && gemm_8p_supported(a1, false)) {
                const int rc = gemm_8p_launch(a1, false, stream);
                if (rc != MI_OK) return rc;
                return gemm_8p128_launch(a2, (a2.K % 128) == 0 ? 0 : 4, stream);
            }
        }
    }
#endif
    if (phase_ok && v != 42 && v != 43 && v != 47 && gemm_8p_supported(a, conv) && (t256 >= 128 || v == 40)) return gemm_8p_launch(a, conv, stream);
    // N = 512-class GEMMs: 128x128 tiles (fp32 + residual or bf16 out); register-pipelined form for an even number of K tiles
    if (!conv && phase_ok && gemm_8p128_supported(a) && (cdiv(a.M, 128) * (a.N / 128) >= 128 || v == 40 || v == 42 || v == 43 || v == 47))
        return gemm_8p128_launch(a, (v != 47 && (a.K % 128) == 0) ? (v == 43 ? 2 : (v == 42 ? 3 : 0)) : 4, stream);       // 43 / 42 = the loader-consumer / the pipelined form whatever the default (A/B)
    // Everything else (small problems, the CTC head's odd N, K < 320): this file's LDS-DMA kernel.  Short launches — up to ~48 K steps of 128 x 128 work per CU — run
    // 128 x 64 output tiles, 48 KiB of LDS -> THREE persistent blocks per CU (blocks in flight beat bytes per flop when operands arrive cold); longer ones
    // (the CTC head: 79) and the conv GEMM keep 128 x 128 tiles, two persistent blocks per CU, the next tile's first K tile prefetched under the epilogue.
    const long steps_per_cu = (long)cdiv(a.M, BM) * cdiv(a.N, BN) * (a.K / BK) / 256;
    // A handful of rows (one utterance: M = 250 frames; the decoder's prompt): 128 x 64 tiles would give 16-64 blocks, each walking K with ONE tile in flight — the launch is
    // a chain of L2 round trips on a sixth of the chip (16 us per GEMM of the bs = 1 encoder).  32 x 64 tiles on two waves, four stages (three K tiles in flight, 48 KiB):
    // 4x the blocks, a third of the exposed round trips; same accumulation order, hence the same bits.  v == 32 forces it.  (Eight stages for the K = 2048 ones measured
    // slower, 13.6 vs 12.2 us, and 64 x 64 tiles on four waves the same, 12.6: and one accumulator per k-step (no dependent MFMA chain)
    // slower too, 12.9: the 0.24 us per K tile of that loop is none of tiles in flight, pieces per wave or MFMA latency — what is left is the barrier + wait per tile.)
    if (!conv && v != 30 && v != 31 && (v == 32 || (a.M <= 2048 && cdiv(a.M, 128) * cdiv(a.N, 64) < 128))) {
        const int g = cdiv(a.M, 32) * cdiv(a.N, 64);
        launch_dense(PF_GLDS, gemm_glds_kernel<32, 64, 1, 2, 4, false>, dim3(g), dim3(128), (size_t)4 * (32 + 64) * BK * 2, stream, a);
        return MI_OK;
    }
    if (!conv && v != 30 && v != 31 && steps_per_cu <= 48) {
        int g = cdiv(a.M, 128) * cdiv(a.N, 64);
        if (g > 768) g = 768;
        launch_dense(PF_GLDS, gemm_glds_kernel<128, 64, 2, 2, 2, false>, dim3(g), dim3(NT), (size_t)2 * (128 + 64) * BK * 2, stream, a);
        return MI_OK;
    }
    int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    if (v != 31 && grid > 512) grid = 512;
    const size_t lds = (size_t)2 * STAGE_BYTES;
    if (conv) launch_dense(PF_GLDS, gemm_glds_kernel<128, 128, 2, 2, 2, true>, dim3(grid), dim3(NT), lds, stream, a);
    else launch_dense(PF_GLDS, gemm_glds_kernel<128, 128, 2, 2, 2, false>, dim3(grid), dim3(NT), lds, stream, a);
    return MI_OK;
}
