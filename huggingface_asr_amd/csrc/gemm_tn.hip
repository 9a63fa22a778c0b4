// Weight-gradient GEMM for gfx950:   dW[n][k] += sum_m dY[m][n] * X[m][k]      (dY (M,N) bf16, X (M,K) bf16, dW (N,K) fp32)
//
// The contraction runs over the ROWS of both operands (M = B*T' is long, the output N x K is small), so neither operand is
// K-contiguous for the MFMA: instead of materialising dYᵀ / Xᵀ in HBM, the [64 m][128 cols] tiles go to LDS as they lie in
// memory (`global_load_lds_dwordx4`, 16-B chunk index XOR-swizzled with the row on the SOURCE side) and the MFMA fragments are
// fetched with the transposing LDS read `ds_read_b64_tr_b16` (a 16-lane group turns a 4 x 16 block around: lane i gets the 4
// consecutive m of column i).  Both operands use the same m-permutation inside a 16-step, so the products line up.
// Output tiles are few (N*K/128² = 16 ... 160), so M is split over `splits` blocks per tile: each writes its fp32 partial tile
// to a slab, `slab_reduce_kernel` adds the slabs into dW (splits == 1: accumulate in place).
// Block 128 x 128, 4 waves (2x2) of 64 x 64, 2-stage LDS ring (64 KiB -> two blocks per CU).  A 128 (n) x 64 (k) form (48 KiB, three per CU; variant 1 of
// mi_gemm_tn_bf16) exists for A/B: unlike the forward GEMM it is slower here (more slabs to write and reduce).
// Round 2, measured with in-kernel stamps (8000 x 2048 x 512, 16 stages per block, two blocks per CU): a stage costs a wave ~2800 cycles = 10 waiting for its tile,
// 76 at the barrier, ~1180 ISSUING its 8 LDS-DMA pieces and ~1420 on 32 transposed reads + 16 MFMAs; the epilogue (64 dword stores per lane) 9200.  The tile is never
// late: the cost of a DMA instruction is back-pressure from the CU's vector-memory path, i.e. the kernel takes in 64 KiB per CU and stage pair = 48 GB/s per CU, 70 % of
// the 66-73 GB/s a CU ingests from L2 at all (MI355X guide, gather-into-LDS table).  At 128 x 128 tiles (64 FLOP per ingested byte) that ceiling is ~1.15 PFLOP/s for
// the chip, and what is left below it is the epilogue and the slab reduce.  Consequences, all measured: a FOUR-deep ring with one block per CU is slower (64 vs 48 us:
// one wave per SIMD exposes the LDS-read latency, and nothing was waiting on the ring); keeping a split's tiles on one XCD (blockIdx % 8) changes nothing (L2 fills are
// not the limit); an eight-wave register-pipelined form was slower while the transposed reads were still 4-way bank-conflicted (fixed since: `swz`, -13 %).  The next
// step is a 256 x 256 output tile (128 FLOP per byte, as the forward GEMM's gemm8p_kernel), at the price of twice the slab bytes.
#include "common.hpp"
#include <utility>

namespace {

constexpr int TN_T = 128, TN_KM = 64;                   // dY tile: 64 m x 128 n;  X tile: 64 m x XW k (XW = 128: 32 KiB per stage, 64: 24 KiB)

__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct TnArgs {
    const bf16_t* Y; long ldy;
    const bf16_t* X; long ldx;
    float* out; long ldo;          // splits == 1: dW (accumulated);  else: slab base, slab s at out + s * slab_stride
    long slab_stride;
    float* db;                     // optional: db[n] += sum_m dY[m][n] (bias gradient), taken from the dY tiles already staged in LDS
    float* db_slab;                // splits > 1: split s leaves its column sums in db_slab[s * N + n] (plain stores; slab_reduce_kernel adds the splits in order).  splits == 1:
                                   // exactly one block owns a column, which adds into db directly — no float atomics either way, the bias gradients are bit-reproducible
    int M, N, K, n_store, splits, rows_per_split;
    // conv != 0: X is not a matrix but a channels-last activation (B, Tin, Fin, Cin) and row m = (b, to, fo), column k = (kh * KW + kw) * Cin + c address
    // X[b][to * cst - cpt + kh][fo * cst - cpf + kw][c] (zero outside): the im2col operand of a Conv2d weight gradient, gathered by the LDS-DMA source addresses
    // instead of being written to HBM first (Cin a multiple of the k tile: a tile's columns stay inside one tap)
    int conv, Tin, Fin, Cin, Tout, Fout, KW, cst, cpt, cpf;
    int overwrite;                 // splits == 1: dW = / db = instead of += (the caller knows the targets hold zeros: a training step's first backward after zero_grad) — the
                                   // epilogue then only stores; it otherwise ends with a dependent load -> add -> store of the whole output tile
};

// 16-B chunk swizzle of a tile row.  256-B rows (16 chunks): chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)) — the transposed read's 32-lane half takes 4 consecutive rows x
// 64 B, and on 256-B rows every row starts on bank 0: the row's low two bits must move the 64-B group, or the four rows collide (4-way: what `chunk ^ (row & 15)`, the
// swizzle for ds_read_b128 row reads, leaves; CDNA guide, "one image for row reads and transposed reads", image (b)).  128-B rows (A/B variant): chunk ^ (row & 7).
template <int CH>
__device__ __forceinline__ int swz(int row) {
    return CH >= 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (row & (CH - 1));       // 512-B rows (32 chunks, the 256-wide dY tile) take the 256-B rows' pattern: the high chunk bit stays
}

// The transposed LDS reads are INLINE ASM with hand-placed lgkmcnt waits (round 3).  Through the builtin the compiler cannot tell them from the destinations of the
// LDS-DMA stagings issued just before (next stage, other ring buffer) and puts `s_waitcnt vmcnt(0)` in front of the first read of every stage: the prefetch then
// never overlaps the stage's own MFMAs — a block was a chain of full memory round trips (3.2 us per stage for a lone block; what round 2's stamps read as "1180 cycles
// issuing 8 DMA pieces"), and only the second block of the CU hid any of it.
template <int ROWB>                                       // bytes per tile row (256: 128 columns, 128: 64 columns)
__device__ __forceinline__ void tr_frag_issue(const char* tile, int cb, int s, int lane, s16x4& lo, s16x4& hi) {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int col = cb + (g & 1) * 16 + 4 * p4;
    const int lc = col >> 3, within = (col & 7) * 2;
    const int k0 = 16 * s + 4 * (g >> 1) + q4, k1 = k0 + 8;
    const unsigned a0 = (unsigned)(size_t)(tile + k0 * ROWB + ((lc ^ swz<ROWB / 16>(k0)) << 4) + within);      // addrspace(3) pointers are 32-bit offsets
    const unsigned a1 = (unsigned)(size_t)(tile + k1 * ROWB + ((lc ^ swz<ROWB / 16>(k1)) << 4) + within);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
}
// The same read from a lane address computed ONCE per block (stage 0, k-step 0) plus an immediate: the swizzle of a row depends on (row & 15) only and a k-step moves
// the rows by 16, so step s of a stage is the byte offset s * 16 * ROWB — the `offset:` field of the instruction, no VALU work between the MFMAs
template <int OFF>
__device__ __forceinline__ void tr_frag_issue_at(unsigned a0, unsigned a1, s16x4& lo, s16x4& hi) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF) : "memory");
}
template <int ROWB>
__device__ __forceinline__ void tr_frag_addr(unsigned tile_off, int cb, int lane, unsigned& a0, unsigned& a1) {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int col = cb + (g & 1) * 16 + 4 * p4;
    const int lc = col >> 3, within = (col & 7) * 2;
    const int k0 = 4 * (g >> 1) + q4, k1 = k0 + 8;
    a0 = tile_off + k0 * ROWB + ((lc ^ swz<ROWB / 16>(k0)) << 4) + within;
    a1 = tile_off + k1 * ROWB + ((lc ^ swz<ROWB / 16>(k1)) << 4) + within;
}
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

__device__ __forceinline__ bf16x8 tr_frag_join(s16x4& lo, s16x4& hi) {
    asm volatile("" : "+v"(lo), "+v"(hi));                // the asm outputs are valid only behind the wait: keep the compiler from reading them earlier
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NS = depth of the LDS ring (see the grouped launch below for what it is worth per tile).
// TNN = output tile rows (n): 128 (4 waves, 2 x 2 of 64 x 64) or 256 (8 waves, 4 x 2: the grouped launch — 85 FLOP per ingested byte instead of 64, two waves per SIMD).
template <int TNN, int XW, int NS, bool CONVOK = true, int KM = TN_KM>       // KM = rows (m) per stage: 64, or 32 (finer stages: a deeper ring in the same LDS)
//       // CONVOK = false: the gathered-operand form (p.conv) is compiled out
__device__ __forceinline__ void tn_tile(const TnArgs& pin, const int bid) {
    TnArgs p = pin;
    if constexpr (!CONVOK) p.conv = 0;
    constexpr int XB = XW * 2, XCH = XW / 8;                  // X tile row bytes, 16-B chunks per row
    constexpr int YB = TNN * 2, YCH = TNN / 8;                // dY tile row bytes / chunks
    constexpr int NW = (TNN / 64) * 2, NT = 64 * NW;          // waves (n x k = TNN/64 x 2), threads
    constexpr int YP = KM * YB / 1024, XP = KM * XB / 1024, PPW = (YP + XP) / NW;   // 1-KiB pieces per stage: dY, X, per wave
    constexpr int STAGE = KM * (YB + XB);
    constexpr int NJ = XW / 64;                               // 32-column k blocks per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int ntn = (p.N + TNN - 1) / TNN, ntk = (p.K + XW - 1) / XW;
    const int ntiles = ntn * ntk;
    const int split = bid / ntiles, tile = bid % ntiles;
    const int n0 = (tile / ntk) * TNN, k0 = (tile % ntk) * XW;
    const int m_lo = split * p.rows_per_split, m_hi = min(p.M, m_lo + p.rows_per_split);
    const int nit = (m_hi - m_lo + KM - 1) / KM;

    // piece g of the stage: 1 KiB of one operand tile — pieces [0, YP) dY, then X.  Per piece a lane keeps ONE register (its chunk's column within the operand's rows,
    // negative when outside the matrix); the row within the stage and the operand's base / row stride are recomputed from wave-uniform values at issue time
    // (eight 64-bit source pointers per lane pushed the 256 x 256 form over its 256-register budget: the pointers spilled and every reload waited for vmcnt(0)).
    constexpr int RPY = 1024 / YB, RPX = 1024 / XB;           // rows per piece
    const int rsubY = lane / YCH, rsubX = lane / XCH;
    auto coloff_of = [&](int q) {
        const int g = wave * PPW + q;
        if (g < YP) {
            const int row = g * RPY + rsubY, cs = lane % YCH;
            const int c = (cs ^ swz<YCH>(row)) * 8;          // source chunk for LDS slot cs
            return n0 + c < p.N ? n0 + c : -1;
        }
        const int row = (g - YP) * RPX + rsubX, cs = lane % XCH;
        const int c = (cs ^ swz<XCH>(row)) * 8;
        return k0 + c < p.K ? (p.conv ? k0 % p.Cin + c : k0 + c) : -1;
    };
    // the gathered form addresses every stage through these; the plain-matrix kernels (CONVOK = false) need them for a ragged last stage only and recompute them
    // there: eight registers that the second fragment set of the 64 x 128 wave tile takes instead
    int coloff[CONVOK ? PPW : 1];
    if constexpr (CONVOK) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) coloff[q] = coloff_of(q);
    }
    const int ctap = p.conv ? k0 / p.Cin : 0, ckh = p.conv ? ctap / p.KW : 0, ckw = p.conv ? ctap % p.KW : 0;       // block-uniform
    // conv: (b, to, fo) of the wave's FIRST X row of the stage, decoded once and advanced by the stage's 64 rows with carries (no divisions between a barrier and the
    // DMA it releases); a wave's pieces are consecutive rows, so piece q is that position + q * RPX with carries (three registers instead of three per piece)
    int cb0 = 0, cto0 = 0, cfo0 = 0;
    const int adv_f = KM % max(p.Fout, 1), adv_t = KM / max(p.Fout, 1);
    if (p.conv) {
        const unsigned ctf = (unsigned)(p.Tout * p.Fout);
        const int g0 = max(wave * PPW - YP, 0);
        const unsigned um = (unsigned)(m_lo + g0 * RPX + rsubX);
        const unsigned bq = um / ctf, rem = um - bq * ctf, to = rem / (unsigned)p.Fout;
        cb0 = (int)bq; cto0 = (int)to; cfo0 = (int)(rem - to * (unsigned)p.Fout);
    }
    auto issue_part = [&](int it, int stage, int qlo, int qhi) {
        char* sbase = smem + stage * STAGE + wave * PPW * 1024;
        const int mb = m_lo + it * KM;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            if (q < qlo || q >= qhi) continue;
            const int colq = CONVOK ? coloff[CONVOK ? q : 0] : coloff_of(q);
            const int g = wave * PPW + q;                    // wave-uniform
            const bool isY = g < YP;
            const int m = mb + (isY ? g * RPY + rsubY : (g - YP) * RPX + rsubX);
            const bf16_t* base = isY ? p.Y : p.X;
            const long ld = isY ? p.ldy : p.ldx;
            const bf16_t* sp = (m < m_hi && colq >= 0) ? base + (long)m * ld + colq : reinterpret_cast<const bf16_t*>(&g_zero16);
            if (p.conv && !isY) {                            // gathered im2col row (wave-uniform branch); stages are issued in order, so the carried position is this stage's
                int fo = cfo0 + (g - max(wave * PPW - YP, 0) - YP) * RPX, to = cto0, bb = cb0;
                while (fo >= p.Fout) { fo -= p.Fout; ++to; }
                while (to >= p.Tout) { to -= p.Tout; ++bb; }
                const int ti = to * p.cst - p.cpt + ckh, fi = fo * p.cst - p.cpf + ckw;
                const bool in = m < m_hi && colq >= 0 && ti >= 0 && ti < p.Tin && fi >= 0 && fi < p.Fin;
                sp = in ? p.X + (((long)bb * p.Tin + ti) * p.Fin + fi) * p.Cin + colq : reinterpret_cast<const bf16_t*>(&g_zero16);
            }
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(sbase + q * 1024), 16, 0, 0);
        }
        if (p.conv && qhi == PPW) {
            cfo0 += adv_f; cto0 += adv_t;
            if (cfo0 >= p.Fout) { cfo0 -= p.Fout; ++cto0; }
            while (cto0 >= p.Tout) { cto0 -= p.Tout; ++cb0; }
        }
    };
    // A plain matrix operand goes through a BUFFER descriptor rebuilt per stage from scalars: base = operand + first row of the stage, records = the bytes up to the
    // split's last row — rows past it read as zero (the hardware's range check is the ragged last stage), and the per-lane part of a piece's source is one 32-bit
    // byte offset fixed for the whole block (row within the stage x row stride + column; a column outside the matrix is clamped to the tile's first one: what it
    // brings in only reaches output rows / columns that are never stored).  The per-lane 64-bit pointer arithmetic of `issue_part` (~10 VALU operations per piece)
    // was 40 % of the kernel's VALU instructions, which share the issue port with the MFMAs.
    constexpr bool FAST = !(CONVOK && XW >= 256);             // (the gathered 256 x 256 form has no registers for the offsets and never takes this path)
    unsigned poff[FAST ? PPW : 1];
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const bool fast_ok = FAST && !p.conv && (long)p.M * max(p.ldy, p.ldx) * 2 < (1l << 32) - 4096;
    if constexpr (FAST) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int g = wu * PPW + q;
            const bool isY = g < YP;
            const int r = isY ? g * RPY + rsubY : (g - YP) * RPX + rsubX;
            const int cq = coloff_of(q);
            const int c = cq >= 0 ? cq : (isY ? n0 : k0);
            poff[q] = (unsigned)(((long)r * (isY ? p.ldy : p.ldx) + c) * 2);
        }
    }
    auto issue = [&](int it, int stage) {
        const int mb = m_lo + it * KM;
        if (FAST && fast_ok) {
            char* sbase = smem + stage * STAGE + wu * PPW * 1024;
            const long left = m_hi - mb;                                            // > 0: the stage exists
            const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.Y)) + (long)mb * p.ldy * 2, 0, (int)(unsigned)(left * p.ldy * 2), 0x00020000);
            const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.X)) + (long)mb * p.ldx * 2, 0, (int)(unsigned)(left * p.ldx * 2), 0x00020000);
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                if (wu * PPW + q < YP) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lptr_t)(sbase + q * 1024), 16, (int)poff[FAST ? q : 0], 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(sbase + q * 1024), 16, (int)poff[FAST ? q : 0], 0, 0, 0);
            }
        } else issue_part(it, stage, 0, PPW);
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient: the blocks of the first k-tile column also sum their dY tile over m (thread = column tid & 127, row half tid >> 7)
    // (from the dY FRAGMENTS the MFMAs take anyway: a lane of an A operand holds eight m of one column n, `v_dot2_f32_bf16` against (1, 1) adds them two at a time into
    // fp32 — 8 VALU instructions per k-step in the waves of the first k half, no LDS traffic.  Reading the column out of the LDS tile, 32 `ds_read_u16` per thread
    // and stage, made these blocks the slowest of their launch: +14 ... +34 % on the whole grouped launch when every problem has a bias.)
    const bool do_db = p.db != nullptr && (tile % ntk) == 0;
    const bool dbw = __builtin_amdgcn_readfirstlane((int)(do_db && wk == 0)) != 0;
    float bsum[2] = {0.f, 0.f};

    unsigned fya0[2], fya1[2], fxa0[NJ], fxa1[NJ];              // lane addresses of the fragment reads in stage 0, k-step 0 (LDS byte offsets)
    {
        const unsigned sm = (unsigned)(size_t)smem;
#pragma unroll
        for (int i = 0; i < 2; ++i) tr_frag_addr<YB>(sm, wn * 64 + i * 32, lane, fya0[i], fya1[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) tr_frag_addr<XB>(sm + KM * YB, wk * (XW / 2) + j * 32, lane, fxa0[j], fxa1[j]);
    }
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nit) issue(s0, s0);
    for (int it = 0; it < nit; ++it) {
        const int stage = it % NS;
        // stage `it` has landed when at most the (NS - 2) younger stagings are still in flight (in-order vmcnt); near the end fewer were issued: wait for everything
        if (NS > 2 && it + NS - 2 < nit) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                        // every wave is done with stage it - 1: its buffer takes stage it + NS - 1
        if (it + NS - 1 < nit) issue(it + NS - 1, (it + NS - 1) % NS);
        const char* ty = smem + stage * STAGE;
        const char* tx = ty + KM * YB;
        // 4 k-steps of 16 rows; the fragments of step s + 1 are requested before the MFMAs of step s issue (two register sets, counted lgkmcnt)
        constexpr int NR = 2 * (2 + NJ);                      // LDS reads per k-step and lane
        constexpr int NB = (XW >= 256 && CONVOK) ? 1 : 2;    // fragment register sets: the gathered form on the 64 x 128 wave tile has no room for a second one
        s16x4 ylo[NB][2], yhi[NB][2], xlo[NB][NJ], xhi[NB][NJ];
        auto rd = [&](int buf, auto sc) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) tr_frag_issue_at<s * 16 * YB>(fya0[i], fya1[i], ylo[buf][i], yhi[buf][i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) tr_frag_issue_at<s * 16 * XB>(fxa0[j], fxa1[j], xlo[buf][j], xhi[buf][j]);
        };
        if constexpr (NB == 2) rd(0, std::integral_constant<int, 0>{});
        static_for(std::make_integer_sequence<int, KM / 16>{}, [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (NB == 2 && s + 1 < KM / 16) {
                rd((s + 1) & 1, std::integral_constant<int, s + 1>{});
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");      // step s's reads are the older NR
            } else {
                if constexpr (NB == 1) rd(0, sc);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            bf16x8 fy[2], fx[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) fy[i] = tr_frag_join(ylo[s & (NB - 1)][i], yhi[s & (NB - 1)][i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fx[j] = tr_frag_join(xlo[s & (NB - 1)][j], xhi[s & (NB - 1)][j]);
            if (dbw) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(fy[i], fy[i], 0, 1), one2, bsum[i], false);
                    bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(fy[i], fy[i], 2, 3), one2, bsum[i], false);
                    bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(fy[i], fy[i], 4, 5), one2, bsum[i], false);
                    bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(fy[i], fy[i], 6, 7), one2, bsum[i], false);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);   // rows n (regs), cols k (lanes)
            // the MFMAs of a step stay between its wait and the next step's: left free, the scheduler sinks the third step's below the fourth's reads and
            // lgkmcnt(0) (the asm statements keep their order, plain instructions float past them) — 16 MFMAs in a row behind a full LDS wait at the end of every stage
            __builtin_amdgcn_sched_barrier(0);
        });
        {                                                    // the read addresses follow the ring in place (no per-stage copies: registers)
            const int step = stage == NS - 1 ? -(NS - 1) * STAGE : STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) { fya0[i] += step; fya1[i] += step; }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { fxa0[j] += step; fxa1[j] += step; }
        }
    }
    if (do_db) {
        asm volatile("s_barrier" ::: "memory");                          // every wave is done with the LDS tiles
        float* red = reinterpret_cast<float*>(smem);
        if (dbw) {                                                       // column n of the tile: its two m halves (lane, lane + 32) side by side
#pragma unroll
            for (int i = 0; i < 2; ++i) red[(wn * 64 + i * 32 + (lane & 31)) * 2 + (lane >> 5)] = bsum[i];
        }
        __syncthreads();
        if (tid < TNN && n0 + tid < p.n_store) {
            const float v = red[2 * tid] + red[2 * tid + 1];
            if (p.db_slab) p.db_slab[(long)split * p.N + n0 + tid] = v;
            else if (p.overwrite) p.db[n0 + tid] = v;
            else p.db[n0 + tid] += v;
        }
    }
    // C layout: col (k) = lane & 31, row (n) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int lr = lane & 31, lh = lane >> 5;
    float* outp = p.out + (p.splits > 1 ? (long)split * p.slab_stride : 0L);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = k0 + wk * (XW / 2) + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n >= p.n_store) continue;
                float* o = outp + (long)n * p.ldo + k;
                if (p.splits > 1 || p.overwrite) *o = acc[i][j][r];
                else *o += acc[i][j][r];
            }
        }
}

template <int XW>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs p) { tn_tile<TN_T, XW, 2>(p, blockIdx.x); }
// one long problem on the 256 x 256 tile of the grouped launch (8 waves, split over M)
template <bool CONV>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(TnArgs p) { tn_tile<256, 256, 2, CONV>(p, blockIdx.x); }

// Grouped form: the weight-gradient GEMMs of one encoder layer (up to TN_GROUP problems: dY_i^T X_i -> dW_i) as ONE launch.  A layer's ten dW GEMMs together have
// ~230 output tiles of 256 x 128 — enough to fill the chip WITHOUT splitting M: every block contracts over all rows of its problem and adds its tile into dW in place,
// so there are no slabs and no reduce pass (launched one by one, each problem needs 8-32 M-splits to fill 512 blocks: 164 launches x (32 us GEMM + 9 us slab reduce)
// per training step).  Block -> (problem, tile) by the prefix table of tile counts.
constexpr int TN_GROUP = 48;
// 64-B problem descriptors: 48 of them + the tile prefix table stay under the 4-KiB kernel-argument segment (the small encoder's layers have ~50 tiles each: its
// problems are gathered over several layers, and the attention decoder's 46 over its whole backward, before a launch fills the chip)
struct TnDesc { const bf16_t* Y; const bf16_t* X; float* out; float* db; int ldy, ldx, ldo, M, N, K, n_store, overwrite; };
struct TnGroup { TnDesc a[TN_GROUP]; int tile0[TN_GROUP + 1]; int n; };

constexpr int TN_GROUP_N = 256;                            // 256 (n) x XW (k) tiles, 8 waves: one block per CU, two waves per SIMD
// XW = 128: 48-KiB stages, wave tiles 64 x 64 (two LDS fragment reads per MFMA), THREE stages.  XW = 256: 64-KiB stages, wave tiles 64 x 128 (1.5 reads per MFMA), 128
// instead of 85 FLOP per ingested byte, two stages (three would be 192 KiB) — used whenever the recorded problems still give (nearly) every CU a block: two base-size
// layers per launch.
// Round 4 (tools/tn_group_depth.py: whole launches, every problem with its bias gradient; profiles/r04_m): PMC on the round-3 kernel showed MFMA busy 36 % of its CUs,
// LDS <= 27 %, waves 47 % of their time in s_waitcnt — and 5.9 VALU instructions per MFMA.  Three causes, fixed in this order:
//  (1) the address of every transposed read and the 64-bit source pointer of every LDS-DMA piece were recomputed per use (the inline-asm reads hide their address
//      pattern from the compiler): now 12 lane addresses per block + the instruction's offset field, and a buffer descriptor per stage built in SALU;
//  (2) the scheduler sank the third k-step's MFMAs below the fourth's reads and its lgkmcnt(0) (sched_barrier per step), and the 64 x 128 wave tile had one fragment
//      set (a second one fits once (1) freed the registers): each wave now covers its own LDS latency;
//  (3) the bias gradient read its dY column out of LDS (32 ds_read_u16 per thread and stage): the blocks that own one — half of them at d = 256 — set the launch's length.
// With those gone the ring depth shows (it did not before): 256 x 128, small encoder x 4 layers at config 3's 48 000 rows 1407 -> 1060 us (three stages; 1197 with
// two), two base layers on 256 x 256 329 -> 292 us (832 TFLOP/s).  The 128-wide form now sits at the chip's L2 -> LDS ingest (232 CUs x 48 KiB / 1.41 us = 7.9 TB/s;
// the forward GEMM at 8192^3 takes in 10.5); the 256-wide one at 6.4 TB/s is bound by its two-stage ring (a stage's DMA has one stage of compute to land).
// Measured and not kept: 32-row stages (4 x 32 KiB / 6 x 24 KiB rings: 300 / 361 us, twice the barriers), three 48-row stages for the 256 x 256 tile (144 KiB: 318 vs 307 us), spreading a stage's DMA instructions over its k-steps
// (334 us / 467 us: the issue back-pressure round 2's stamps showed was the compiler's vmcnt(0), not the queue).
template <int XW, int NS>
__global__ __launch_bounds__(512) void gemm_tn_group_kernel(TnGroup g) {
    // XCD-aware tile order: the hardware deals consecutive block ids round-robin over the 8 XCDs (each with its own L2), so XCD x works through one contiguous run of
    // the tile list — the tiles of one or two problems, which share their dY / X stage tiles — instead of every XCD pulling every problem's operands (worth 1.6 %)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int lo = 0, hi = g.n - 1;                                // the last problem whose first tile is <= bid
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (bid >= g.tile0[mid]) lo = mid; else hi = mid - 1;
    }
    // the problem index is wave-uniform: say so, and take a COPY of the descriptor — indexed in place, its fields come back as vector memory loads inside the pipeline,
    // each with a `s_waitcnt vmcnt(0)` in front of the LDS-DMA instruction that uses it (seen in the ISA: one full wait per staging instruction)
    const int iu = __builtin_amdgcn_readfirstlane(lo);
    const TnDesc q = g.a[iu];
    TnArgs p;
    p.Y = q.Y; p.ldy = q.ldy; p.X = q.X; p.ldx = q.ldx; p.out = q.out; p.ldo = q.ldo; p.slab_stride = 0; p.db = q.db; p.db_slab = nullptr;
    p.M = q.M; p.N = q.N; p.K = q.K; p.n_store = q.n_store; p.splits = 1; p.rows_per_split = (q.M + TN_KM - 1) / TN_KM * TN_KM;
    p.conv = 0; p.Tin = p.Fin = p.Cin = p.Tout = p.Fout = p.KW = 1; p.cst = 1; p.cpt = p.cpf = 0; p.overwrite = q.overwrite;
    tn_tile<TN_GROUP_N, XW, NS, false>(p, bid - __builtin_amdgcn_readfirstlane(g.tile0[iu]));
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ out, long ldo, const float* __restrict__ slabs, long slab_stride,
                                                           int splits, int rows, int cols, float* __restrict__ db, const float* __restrict__ db_slab, int N) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        float s = 0.f;
        for (int q = 0; q < splits; ++q) s += slabs[q * slab_stride + i];
        out[(long)r * ldo + c] += s;
    }
    if (db && blockIdx.x == 0)                                   // the bias gradient's per-split column sums, added in split order
        for (int n = threadIdx.x; n < rows; n += 256) {
            float s = 0.f;
            for (int q = 0; q < splits; ++q) s += db_slab[(long)q * N + n];
            db[n] += s;
        }
}

}  // namespace

// M splits: two (128-wide X tiles, 64 KiB of LDS) or three (64-wide, 48 KiB) blocks per CU on 256 CUs — 512 blocks per launch measured as the optimum (round 1)
// variant 2: the grouped launch's 256 x 256 tile (one block per CU: 256 blocks) for a single long problem — the Conv2d #2 and CTC-head weight gradients
static int tn_splits(int M, int N, int K, int variant) {
    const int xw = variant == 1 ? 64 : 128;
    const int tiles = variant == 2 ? cdiv(N, 256) * cdiv(K, 256) : cdiv(N, TN_T) * cdiv(K, xw);
    int s = (variant == 2 ? 256 : variant == 1 ? 768 : 512) / tiles;
    const int max_s = cdiv(M, 4 * TN_KM);              // at least 4 K-iterations per block
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}
// the long single problems: at least 32 Ki rows and a full 256 x 256 tile
static bool tn_big(int M, int N, int K) { return M >= 32768 && N >= 256 && K >= 256; }
extern "C" size_t mi_gemm_tn_workspace_bytes(int M, int N, int K) {       // enough for either variant
    size_t best = 0;
    for (int v = 0; v < 3; ++v) {
        const int s = tn_splits(M, N, K, v);
        if (s > 1 && (size_t)s * N * (K + 1) * sizeof(float) > best) best = (size_t)s * N * (K + 1) * sizeof(float);      // dW slabs + the bias gradient's column-sum slabs
    }
    return best;
}

// dW (n_store, K) fp32 (row stride ldo) += dY[:, :N]^T · X;  N, K % 8 == 0, rows of dY / X 16-B aligned; n_store <= N
// db (optional, n_store floats): bias gradient db[n] += sum_m dY[m][n], fused (per-split column sums, added in split order by the slab reduce: no atomics)
// variant: 0 = the product's tile (128 x 128), 1 = 128 (n) x 64 (k) for A/B.  Per call: the library keeps no kernel-selection state.
extern "C" int mi_gemm_tn_bf16(const void* dY, long ldy, const void* X, long ldx, float* dW, long ldo, float* db, int M, int N, int K, int n_store,
                               void* workspace, size_t workspace_bytes, int variant, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || (N % 8) || (K % 8) || (ldy % 8) || (ldx % 8) || n_store > N || n_store <= 0 || variant < 0 || variant > 1) return MI_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(X) & 15)) return MI_ERR_ARG;
    const bool big = variant == 0 && tn_big(M, N, K);
    const int splits = tn_splits(M, N, K, big ? 2 : variant);
    if (splits > 1 && workspace_bytes < (size_t)splits * N * (K + 1) * sizeof(float)) return MI_ERR_ARG;
    TnArgs p{};
    p.Y = (const bf16_t*)dY; p.ldy = ldy; p.X = (const bf16_t*)X; p.ldx = ldx;
    p.M = M; p.N = N; p.K = K; p.n_store = n_store; p.splits = splits; p.db = db;
    p.rows_per_split = cdiv(cdiv(M, splits), TN_KM) * TN_KM;
    if (splits > 1) { p.out = (float*)workspace; p.ldo = K; p.slab_stride = (long)N * K; p.db_slab = db ? (float*)workspace + (size_t)splits * N * K : nullptr; }
    else { p.out = dW; p.ldo = ldo; p.slab_stride = 0; }
    const int tiles = big ? cdiv(N, 256) * cdiv(K, 256) : cdiv(N, TN_T) * cdiv(K, variant == 1 ? 64 : 128);
    if (big) {
        const size_t lds = (size_t)2 * TN_KM * 1024;
        if (!ensure_dynamic_lds<1>(reinterpret_cast<const void*>(gemm_tn_big_kernel<false>), lds)) return MI_ERR_LAUNCH;
        hipLaunchKernelGGL(gemm_tn_big_kernel<false>, dim3(tiles * splits), dim3(512), lds, st, p);
    } else if (variant == 0) hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(tiles * splits), dim3(256), (size_t)2 * TN_KM * (TN_T * 2 + 256), st, p);
    else hipLaunchKernelGGL(gemm_tn_kernel<64>, dim3(tiles * splits), dim3(256), (size_t)2 * TN_KM * (TN_T * 2 + 128), st, p);
    MI_CHECK_LAUNCH();
    if (splits > 1) {
        const long total = (long)n_store * K;
        const long g = (total + 255) / 256;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, dW, ldo, (const float*)workspace, p.slab_stride,
                           splits, n_store, K, db, p.db_slab, N);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}

// Conv2d weight gradient without an im2col buffer: dW (n_store, KH*KW*Cin) fp32 += dY[:, :N]^T · im2col(x), x (B, Tin, Fin, Cin) bf16 channels-last, rows of dY = (b, to, fo),
// k = (kh * KW + kw) * Cin + c; square stride, pads (pad_t, pad_f) on the leading side (the causal front end passes 2 * pad).  Cin % 128 == 0.  Workspace as for
// mi_gemm_tn_bf16 with M = B * Tout * Fout, K = KH * KW * Cin.  replaces: the weight gradient autograd derives for extractors.py:82-89's second Conv2d.
extern "C" int mi_conv2d_wgrad_cl_bf16(const void* dY, long ldy, const void* x, float* dW, long ldo, float* db, int B, int Tin, int Fin, int Cin, int KH, int KW,
                                       int stride, int pad_t, int pad_f, int Tout, int Fout, int N, int n_store, void* workspace, size_t workspace_bytes, hipStream_t st) {
    MI_ENTER();
    const long Ml = (long)B * Tout * Fout;
    const int K = KH * KW * Cin;
    if (B <= 0 || Tin <= 0 || Fin <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || Tout <= 0 || Fout <= 0 || N <= 0 || (N % 8) || (ldy % 8) || n_store > N || n_store <= 0) return MI_ERR_ARG;
    if ((Cin % 128) || Ml >= (1l << 31) || (long)B * Tin * Fin * Cin >= (1l << 40)) return MI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || !dW) return MI_ERR_ARG;
    const int M = (int)Ml;
    const bool big = tn_big(M, N, K) && (Cin % 256) == 0;
    const int splits = tn_splits(M, N, K, big ? 2 : 0);
    if (splits > 1 && workspace_bytes < (size_t)splits * N * (K + 1) * sizeof(float)) return MI_ERR_ARG;
    TnArgs p{};
    p.Y = (const bf16_t*)dY; p.ldy = ldy; p.X = (const bf16_t*)x; p.ldx = Cin;
    p.M = M; p.N = N; p.K = K; p.n_store = n_store; p.splits = splits; p.db = db;
    p.rows_per_split = cdiv(cdiv(M, splits), TN_KM) * TN_KM;
    if (splits > 1) { p.out = (float*)workspace; p.ldo = K; p.slab_stride = (long)N * K; p.db_slab = db ? (float*)workspace + (size_t)splits * N * K : nullptr; }
    else { p.out = dW; p.ldo = ldo; p.slab_stride = 0; }
    p.conv = 1; p.Tin = Tin; p.Fin = Fin; p.Cin = Cin; p.Tout = Tout; p.Fout = Fout; p.KW = KW; p.cst = stride; p.cpt = pad_t; p.cpf = pad_f;
    const int tiles = big ? cdiv(N, 256) * cdiv(K, 256) : cdiv(N, TN_T) * cdiv(K, 128);
    if (big) {
        const size_t lds = (size_t)2 * TN_KM * 1024;
        if (!ensure_dynamic_lds<2>(reinterpret_cast<const void*>(gemm_tn_big_kernel<true>), lds)) return MI_ERR_LAUNCH;
        hipLaunchKernelGGL(gemm_tn_big_kernel<true>, dim3(tiles * splits), dim3(512), lds, st, p);
    } else hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(tiles * splits), dim3(256), (size_t)2 * TN_KM * (TN_T * 2 + 256), st, p);
    MI_CHECK_LAUNCH();
    if (splits > 1) {
        const long total = (long)n_store * K;
        const long g = (total + 255) / 256;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, dW, ldo, (const float*)workspace, p.slab_stride, splits, n_store, K, db, p.db_slab, N);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}

// The grouped form (see gemm_tn_group_kernel): n <= 48 problems, each dW_i (n_store_i, K_i) fp32 (row stride ldo_i) += dY_i[:, :N_i]^T X_i with optional bias gradient db_i,
// ONE launch, no M-split, no workspace.  tile_k: k extent of the 256-row output tiles — 128, 256, or 0 = 256 when that still gives >= 200 tiles, else 128.  Arrays of length n on the HOST.  Same operand constraints as mi_gemm_tn_bf16.  Meant for >= ~256 output tiles in total
// (sum of ceil(N_i / 128) * ceil(K_i / 128)); with fewer the chip is under-filled: use mi_gemm_tn_bf16 per problem.
// overwrite != 0: every dW_i / db_i is WRITTEN (= instead of +=): for targets the caller knows to hold zeros — the first backward after the gradients were cleared, each
// target the output of exactly one problem; the kernel's epilogue then has no read of the output tile in it.
extern "C" int mi_gemm_tn_group_ow_bf16(int n, const void* const* dY, const long* ldy, const void* const* X, const long* ldx, float* const* dW, const long* ldo,
                                        float* const* db, const int* M, const int* N, const int* K, const int* n_store, int tile_k, int overwrite, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || n > TN_GROUP) return MI_ERR_ARG;
    TnGroup g{};
    g.n = n;
    long t256 = 0;
    for (int i = 0; i < n; ++i) {
        if (M[i] <= 0 || N[i] <= 0 || K[i] <= 0 || (N[i] % 8) || (K[i] % 8) || (ldy[i] % 8) || (ldx[i] % 8) || n_store[i] > N[i] || n_store[i] <= 0) return MI_ERR_ARG;
        if ((reinterpret_cast<uintptr_t>(dY[i]) & 15) || (reinterpret_cast<uintptr_t>(X[i]) & 15) || !dW[i]) return MI_ERR_ARG;
        if (ldy[i] >= (1l << 31) || ldx[i] >= (1l << 31) || ldo[i] >= (1l << 31) || ldy[i] <= 0 || ldx[i] <= 0 || ldo[i] <= 0) return MI_ERR_ARG;
        t256 += (long)cdiv(N[i], TN_GROUP_N) * cdiv(K[i], 256);
    }
    if (tile_k != 0 && tile_k != 128 && tile_k != 256) return MI_ERR_ARG;
    const int xw = tile_k ? tile_k : (t256 >= 200 ? 256 : 128);
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        TnDesc& p = g.a[i];
        p.Y = (const bf16_t*)dY[i]; p.ldy = (int)ldy[i]; p.X = (const bf16_t*)X[i]; p.ldx = (int)ldx[i];
        p.M = M[i]; p.N = N[i]; p.K = K[i]; p.n_store = n_store[i]; p.db = db ? db[i] : nullptr; p.overwrite = overwrite ? 1 : 0;
        p.out = dW[i]; p.ldo = (int)ldo[i];
        g.tile0[i] = tiles;
        tiles += cdiv(N[i], TN_GROUP_N) * cdiv(K[i], xw);
    }
    g.tile0[n] = tiles;
    const int ns = xw == 256 ? 2 : 3;
    const size_t lds = (size_t)ns * TN_KM * (TN_GROUP_N * 2 + xw * 2);          // 2 x 64 KiB / 3 x 48 KiB: one block per CU
    if (xw == 256) {
        if (!ensure_dynamic_lds<3>(reinterpret_cast<const void*>(gemm_tn_group_kernel<256, 2>), lds)) return MI_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_tn_group_kernel<256, 2>), dim3(tiles), dim3(512), lds, st, g);
    } else {
        if (!ensure_dynamic_lds<4>(reinterpret_cast<const void*>(gemm_tn_group_kernel<128, 3>), lds)) return MI_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_tn_group_kernel<128, 3>), dim3(tiles), dim3(512), lds, st, g);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_gemm_tn_group_bf16(int n, const void* const* dY, const long* ldy, const void* const* X, const long* ldx, float* const* dW, const long* ldo,
                                     float* const* db, const int* M, const int* N, const int* K, const int* n_store, int tile_k, hipStream_t st) {
    return mi_gemm_tn_group_ow_bf16(n, dY, ldy, X, ldx, dW, ldo, db, M, N, K, n_store, tile_k, 0, st);
}
