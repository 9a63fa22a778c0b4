// Fused self-attention for the E-Branchformer encoder on gfx950 (wave64, v_mfma_f32_32x32x16_bf16).
//
// Computes, per (batch, head), reference e_branchformer.py:105-138 with the Transformer-XL relative
// position term of tf wav2vec2_conformer :528-565 restated as the direct index map
//     score[i][j] = ( (q_i+u)·k_j + (q_i+v)·p[T-1-i+j] ) / sqrt(hd)          
// (p = linear_pos(rel-pos table), (2T-1, d)), key-padding mask (tf:667-672), optional causal mask
// (e_branchformer.py:119-124), softmax, and probs·V — without materialising the (B,H,T,T) scores.
//
// Orientation: everything is computed TRANSPOSED so that the query index lives on the MFMA lane:
//   S^T[key][query] = K_tile · (Q+u)^T ,  G^T[c][query] = P_band · (Q+v)^T ,  O^T[hd][query] += V^T_tile · P^T
// The softmax reduction over keys is then over a lane's own accumulator registers (+ one cross-half
// shuffle), the online-softmax rescale is a per-lane scalar, and the exponentiated S^T accumulators are
// directly the B operand of the PV product (no LDS transpose).  The rel-shift BD[i][j] = G[i][j-i+31]
// is a per-query (per-lane) skew done through a wave-private LDS row.
// One wave owns 32 queries; operands (K rows, P rows, V^T rows) are L2-resident and loaded straight
// into MFMA fragments.  V^T (d, B*Tp) is produced key-contiguous by the V-projection GEMM.
#include "common.hpp"

namespace {

struct AttnArgs {
    const bf16_t* q; long ldq;           // (B*T, .) query projections, head h at columns [h*HD, (h+1)*HD)
    const bf16_t* k; long ldk;           // (B*T, .) key projections
    const bf16_t* vt; long ldvt; int Tp; // V^T: row = h*HD + c, column = b*Tp + t  (zero beyond T)
    const bf16_t* pos; long ldp;         // (2T-1, .) projected relative positions (REL)
    const float* bias_u; const float* bias_v;   // (H, HD)  (REL)
    const int* lengths;                  // (B) number of valid keys, or null
    bf16_t* out; long ldo;               // (B*T, .) context
    int B, T, H; float scale; int causal;
    int Tk;                              // LDS-staged kernel: number of key/value rows per batch (0 = T); queries are T
    long kv_bstride;                     // elements between consecutive batches of k / v (0 = Tk * ld): KV caches
    float* lse;                          // LDS-staged kernel: (B, H, T) log2-domain log-sum-exp of the scaled scores — written by the forward when non-null, read by the backward
    // backward form (attn_lds_kernel<.., BW = true>): `out` is the forward's context (read), and
    const bf16_t* dctx; long ldd;        // (B*T, .) gradient of the context
    bf16_t* prob; bf16_t* ds; long ldsr; // (H, B, T, ldsr) probabilities and score gradients dS = P (dP - delta) scale; ldsr a multiple of 32, every column written
    bf16_t* dbd; long ldbd; int pad;     // (H, B, T, ldbd): dBD[i][T-1-i+j + pad] = dS[i][j] (the inverse of the rel-shift), zero elsewhere; ldbd a multiple of 32
    bf16_t* dq; long lddq;               // (B*T, .) gradient of the query projection: dS K (+ dBD P with relative positions), accumulated over the walk
    float* dsum_u; float* dsum_v;        // (B, 4 ceil(T/128), H*HD): per-wave column sums of dS K and dBD P — the pos_bias_u / pos_bias_v gradients once summed
    // attention-probability dropout (e_branchformer.py:132) in the LDS-staged kernel, training forward and backward: the counter-based mask of dropout.hip for the
    // logical element ((h * B + b) * T + i) * Tk + j of the (H, B, T, Tk) probabilities — the same mask the generic softmax kernels (attn_bwd.hip) and the host twin draw
    float drop_p; unsigned long long drop_key;
    bf16_t* qu_out; bf16_t* qv_out; long ldqb;   // BW + REL, optional: (B*T, ldqb) bf16 rows q + pos_bias_u / q + pos_bias_v of this head's columns, as the walk's A fragments hold them
                                         // (the operands of the dK and d(positions) products that follow: no pass of their own)
    long ldsum;                          // row stride of dsum_u / dsum_v in floats (0 = H*HD); 2 H*HD when the caller interleaves them as rows of [u | v] (one deferred reduction for both)
#ifdef ATTN_STAMPS
    unsigned long long* stamps;          // instrumented build only (tools/attn_stamps.py): per (block, wave) 32 shader-clock readings
#endif
    int gx; unsigned mgx, mgy;            // eight-wave form (1-D grid): query blocks per (batch, head) and ceil(2^32 / gx), ceil(2^32 / H) — exact quotients for block ids < 2^16
    int variant;                         // 0: the product's choice; 1: the four-wave LDS-staged forward; 2: the eight-wave form (1, 2: A/B only, tools/attn_ab.py)
    // BW, round 5: the zeros nobody reads are not written.  The caller guarantees that (a) dbd is a buffer it zero-filled ONCE and that only this entry writes — a row's
    // relative positions outside its wave's MAXIMAL band (every key valid) are the same for every launch, so they stay zero —, and (b) columns of P / dS from the key
    // length rounded up to 128 on are never read (mi_bgemm_sparse_bf16's m_valid skips those M tiles).  The walk then zeroes only what an earlier launch with longer
    // utterances may have written: the rest of the maximal band in dbd, the rest of the last 128-key tile in P / dS.
    int sparse;
};

// 16-B chunk swizzle of the LDS-staged kernel's tiles (applied on the DMA source and on every read).  256-B rows (hd 128): the image that is conflict-free for ds_read_b128
// row reads (K, position rows, Q) AND for the transposed reads of V (CDNA guide, "one image for row reads and transposed reads", image (b)); chunk ^ (row & 15) leaves the
// transposed read 4-way (four consecutive rows of a 32-lane half all start on bank 0 and stay inside one 64-B group).  128-B rows (hd 64): chunk ^ (row & 7).
template <int CH>
__device__ __forceinline__ int tswz(int row) {
    return CH == 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (row & (CH - 1));
}
constexpr int SKEW_LD = 66;   // words per query row of the skew scratch: reads conflict-free, writes 2-way (free)
constexpr int OST_B = 272;    // LDS-staged kernel, output staging: bytes per query row (hd 128 -> 256 B + 16: rows stay 16-B aligned for ds_read_b128, writes 2-way)
constexpr int WSCR_B = 32 * OST_B;   // per-wave scratch of the LDS-staged kernel: skew rows (32 x 66 words) | Q tile (8 KiB, prologue) | output rows (epilogue)

__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// keep factors (0 or 1 / (1 - p)) of a lane's 16 score registers of one 32-key step: register e is key j0 + crow(e, half); idx0 = logical index of key j0 of the
// lane's query row.  One hash per QUAD of logical indices (common.hpp): a run of four consecutive keys takes one hash, two when it does not start on a quad
// (`unaligned`, wave-uniform: Tk not a multiple of 4 — then a row's first key sits anywhere in its quad).
__device__ __forceinline__ void keep16(unsigned long long key, float p, unsigned long long idx0, int half, bool unaligned, float (&ks)[16]) {
    const float inv = 1.f / (1.f - p);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const unsigned long long s = idx0 + 8 * g4 + 4 * half, q0 = s >> 2;
        if (!unaligned) { mask_keep4(key, q0, p, inv, &ks[4 * g4]); continue; }
        const unsigned long long h0 = mask_hash(key, q0), h1 = mask_hash(key, q0 + 1);
        const int o = (int)(s & 3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = o + k;
            ks[4 * g4 + k] = mask_u01(j < 4 ? h0 : h1, j & 3) >= p ? inv : 0.f;
        }
    }
}

template <int HD, bool REL>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    constexpr int KS = HD / 16;                 // k-steps over the head dimension
    constexpr int NTO = (HD + 31) / 32;         // 32-row tiles of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int i0 = (blockIdx.x * 4 + wave) * 32;
    if (i0 >= p.T) return;
    float* skew = reinterpret_cast<float*>(smem) + wave * 32 * SKEW_LD;

    const int T = p.T;
    const int len = p.lengths ? min(p.lengths[b], T) : T;
    const int iq = min(i0 + r, T - 1);
    const long rowq = (long)b * T + iq;

    bf16x8 qu[KS], qv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = head * HD + ks * 16 + h2 * 8;
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(p.q + rowq * p.ldq + c);
        if (REL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f(raw[j]);
                qu[ks][j] = f2bf(f + p.bias_u[c + j]);
                qv[ks][j] = f2bf(f + p.bias_v[c + j]);
            }
        } else {
            qu[ks] = raw;
        }
    }

    f32x16 O[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] = 0.f;
    float m = -1e30f, l = 0.f;

    const int kend = p.causal ? min(len, i0 + 32) : len;
    for (int j0 = 0; j0 < kend; j0 += 32) {
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
        {
            const long rowk = (long)b * T + min(j0 + r, T - 1);
            const bf16_t* kp = p.k + rowk * p.ldk + head * HD + h2 * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[ks], S, 0, 0, 0);
            }
        }
        if (REL) {
            const int rbase = T - 1 - i0 - 31 + j0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x16 G;
#pragma unroll
                for (int e = 0; e < 16; ++e) G[e] = 0.f;
                const int pr = min(max(rbase + 32 * g + r, 0), 2 * T - 2);
                const bf16_t* pp = p.pos + (long)pr * p.ldp + head * HD + h2 * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pp + ks * 16);
                    G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[ks], G, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) skew[r * SKEW_LD + 32 * g + crow(e, h2)] = G[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int e = 0; e < 16; ++e) S[e] += skew[r * SKEW_LD + crow(e, h2) - r + 31];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        // scale, mask, online softmax (lane = query; keys across registers and the two lane halves)
        float mx = -1e30f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jj = j0 + crow(e, h2);
            const bool dead = (jj >= len) || (p.causal && jj > i0 + r);
            S[e] = dead ? -INFINITY : S[e] * p.scale;
            mx = fmaxf(mx, S[e]);
        }
        mx = half_swap_max(mx);
        const float mnew = fmaxf(m, mx);
        const float alpha = __expf(m - mnew);
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            S[e] = __expf(S[e] - mnew);
            ls += S[e];
        }
        ls = half_swap_sum(ls);
        l = l * alpha + ls;
        m = mnew;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[t][e] *= alpha;
        // P^T as the B operand: registers 8s..8s+7 are k-step s; element j of half h is key 16s+8(j>>2)+4h+(j&3)
        bf16x8 pb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[s][j] = f2bf(S[8 * s + j]);
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const int hr = t * 32 + r;                       // row of V^T within the head
            const bool hok = hr < HD;
            const bf16_t* vp = p.vt + (long)(head * HD + (hok ? hr : 0)) * p.ldvt + (long)b * p.Tp + j0 + 4 * h2;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 va = {0, 0, 0, 0, 0, 0, 0, 0};
                if (hok) {
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vp + 16 * s);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vp + 16 * s + 8);
                    va = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[s], O[t], 0, 0, 0);
            }
        }
    }

    if (i0 + r < T) {
        const float inv = 1.f / l;
        bf16_t* op = p.out + ((long)b * T + i0 + r) * p.ldo + head * HD;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = t * 32 + 8 * g4 + 4 * h2;
                if (c < HD) {
                    bf16x4 o = {f2bf(O[t][4 * g4 + 0] * inv), f2bf(O[t][4 * g4 + 1] * inv),
                                f2bf(O[t][4 * g4 + 2] * inv), f2bf(O[t][4 * g4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(op + c) = o;
                }
            }
    }
}


// ======================================================================================================================
// LDS-staged kernel (head size 64 / 128): a block = 4 waves = 128 consecutive queries of one (batch, head).
//  * K, V (both [key][hd] row-major straight out of the fused QKV projection) and the relative-position rows are staged
//    per 32-key step with LDS-DMA one step ahead (2-slot rings for K/V, 6-slot ring for the position rows) and shared by
//    the four waves: 4x less L2 traffic than one wave per 32 queries, and no global-load latency inside a step.
//  * V needs no transposed copy in HBM: the PV product's A operand (V^T) is read with ds_read_b64_tr_b16.
//  * BD band reuse: for a wave the band of step t is [rb, rb+64); its upper half is the lower half of step t+1, so only
//    ONE new 32-row G tile (8 MFMAs at hd=128) is computed per step and the other is carried in registers.
//  * LDS images are lane-linear per DMA piece; the 16-B-chunk XOR swizzle (tswz) is applied to the
//    source address and to every read (conflict-free ds_read_b128 / tr reads).
//
// BW = true is the first half of the training backward (round 3): the same walk over the keys recomputes S^T, takes P = 2^(S - lse) with the forward's row
// log-sum-exp, adds dP^T = V_tile · dctx^T (V rows read like K rows) and leaves P, dS = P (dP - delta) scale and the un-shifted dBD in HBM as bf16 — in place of
// three batched GEMMs with fp32 outputs (Q K^T, Q P^T, dctx V^T: 130 MB per layer) and the two softmax passes over them.  delta_i = dctx_i · ctx_i.
constexpr int STG_B = 80;     // BW: bytes per query row of the P / dS staging tiles (32 keys = 64 B + 16)
constexpr int BAND_B = 144;   // BW: bytes per query row of the dBD band buffer (two 32-column blocks = 128 B + 16)
template <int HD, bool REL, bool BW>
__global__ __launch_bounds__(256, 1) void attn_lds_kernel(AttnArgs p) {
    constexpr int KS = HD / 16, NTO = HD / 32;
    constexpr int ROWB = HD * 2, NCH = ROWB / 16;            // bytes per row, 16-B chunks per row
    constexpr int TILEB = 32 * ROWB;                          // one 32-row tile
    constexpr int RPP = 1024 / ROWB;                          // rows per 1-KiB DMA piece
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / 4;    // per tile / per wave
    constexpr int PRING = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                          // [2][TILEB]
    char* sV = sK + 2 * TILEB;                                // [2][TILEB]
    char* sP = sV + 2 * TILEB;                                // [PRING][TILEB]
    float* skew_all = reinterpret_cast<float*>(sP + PRING * TILEB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave-uniform: ring slots and band blocks stay on the scalar unit
    const int r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int ib = blockIdx.x * 128, i0 = ib + wave * 32;
    char* wscr = reinterpret_cast<char*>(skew_all) + wave * WSCR_B;       // this wave's private scratch
    float* skew = reinterpret_cast<float*>(wscr);
    char* band = reinterpret_cast<char*>(skew_all) + 4 * WSCR_B + wave * (32 * BAND_B);      // BW only (the launch sizes the LDS for it)
    const int T = p.T;                                        // queries per batch
    const int Tk = p.Tk > 0 ? p.Tk : p.T;                     // keys per batch (cross-attention / KV cache: Tk != T)
    const int coff = Tk - T;                                  // causal: query i sees keys <= i + coff
    const int len = p.lengths ? min(p.lengths[b], Tk) : Tk;
    const int kend = p.causal ? min(len, ib + 128 + coff) : len;
    const int nkt = (kend + 31) / 32;
    const int rb0 = T - 1 - ib - 31;                          // band base of wave 0 at step 0 (REL needs Tk == T)

    // ---- DMA issue: this wave's PPW pieces of a 32-row tile.  Everything that does not change from step to step (the piece's row within the tile, the swizzled
    // source column, the operand's batch base) is folded into one per-lane pointer per piece and operand before the loop; a step costs clamp + one 64-bit mad per piece.
    const int prow = lane / NCH, pc = lane % NCH;
    int prw[PPW];
    const char *kB[PPW], *vB[PPW], *pB[PPW];
    const long kvoff = p.kv_bstride ? (long)b * p.kv_bstride : 0;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int row = (wave * PPW + q) * RPP + prow;        // 0..31
        const int colb = (head * HD + (pc ^ tswz<NCH>(row)) * 8) * 2;
        prw[q] = row;
        kB[q] = reinterpret_cast<const char*>(p.k + (p.kv_bstride ? kvoff : (long)b * Tk * p.ldk)) + colb;
        vB[q] = reinterpret_cast<const char*>(p.vt + (p.kv_bstride ? kvoff : (long)b * Tk * p.ldvt)) + colb;       // p.vt = V [key][hd]
        pB[q] = reinterpret_cast<const char*>(p.pos) + colb;
    }
    const unsigned ldkB = (unsigned)p.ldk * 2u, ldvB = (unsigned)p.ldvt * 2u, ldpB = (unsigned)p.ldp * 2u;     // row strides in bytes (< 2^32: checked on the host)
    auto issue_rows = [&](const char* const (&base)[PPW], unsigned ldB, int row0, int rmax, char* dst) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const unsigned gr = (unsigned)min(max(row0 + prw[q], 0), rmax);
            const char* src = base[q] + (unsigned long)gr * ldB;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (wave * PPW + q) * 1024), 16, 0, 0);
        }
    };
    int pslot_dma = (1 + PRING * 4) % PRING;                  // ring slot of position block it + 1 (the one the step's DMA fills), advanced by hand: no modulo in the loop
    auto issue_step = [&](int it) {
        issue_rows(kB, ldkB, 32 * it, Tk - 1, sK + (it & 1) * TILEB);
        issue_rows(vB, ldvB, 32 * it, Tk - 1, sV + (it & 1) * TILEB);
        if (REL) issue_rows(pB, ldpB, rb0 + 32 * it + 32, 2 * T - 2, sP + pslot_dma * TILEB);
    };

    // ---- prologue: position blocks -4..-1 and the step-0 tiles, then this wave's 32 query rows — staged like a K tile (whole 1-KiB pieces = 4 or 8 full rows per
    // instruction, swizzled) into the wave's private scratch and read back as MFMA fragments: a lane-per-row global read of Q touches 32 lines per instruction.
    if (REL) {
#pragma unroll
        for (int k = -4; k < 0; ++k)
            issue_rows(pB, ldpB, rb0 + 32 * k + 32, 2 * T - 2, sP + ((k + PRING * 4) % PRING) * TILEB);
    }
    pslot_dma = (0 + PRING * 4) % PRING;
    issue_step(0);
    pslot_dma = (1 + PRING * 4) % PRING;
    {
        const char* qb = reinterpret_cast<const char*>(p.q + (long)b * T * p.ldq);
        const unsigned ldqB = (unsigned)p.ldq * 2u;
#pragma unroll
        for (int piece = 0; piece < PIECES; ++piece) {
            const int row = piece * RPP + prow;
            const int colb = (head * HD + (pc ^ tswz<NCH>(row)) * 8) * 2;
            const char* src = qb + (unsigned long)(unsigned)min(i0 + row, T - 1) * ldqB + colb;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                             (__attribute__((address_space(3))) void*)(wscr + piece * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // own pieces only: the scratch is wave-private (the block-wide wait + barrier for the shared tiles follows below)
    bf16x8 qu[KS], qv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = head * HD + ks * 16 + h2 * 8;
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(wscr + r * ROWB + (((ks * 2 + h2) ^ tswz<NCH>(r)) << 4));
        if (REL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f(raw[j]);
                qu[ks][j] = f2bf(f + p.bias_u[c + j]);
                qv[ks][j] = f2bf(f + p.bias_v[c + j]);
            }
        } else {
            qu[ks] = raw;
        }
        if constexpr (BW && REL) {
            if (p.qu_out && i0 + r < T) {
                const long o = ((long)b * T + i0 + r) * p.ldqb + c;
                *reinterpret_cast<bf16x8*>(p.qu_out + o) = qu[ks];
                *reinterpret_cast<bf16x8*>(p.qv_out + o) = qv[ks];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fragments are in registers before the scratch is reused as the skew rows

    bf16x8 dc[BW ? KS : 1];                                  // BW: dctx rows of this wave's queries as the B operand of dP^T = V · dctx^T
    float delta = 0.f, lse2 = 0.f;
    if constexpr (BW) {
        auto stage_rows = [&](const bf16_t* base, long ld) {         // this wave's 32 rows of a (B*T, ld) operand -> scratch, as Q above
            const char* qb = reinterpret_cast<const char*>(base + (long)b * T * ld);
            const unsigned ldB = (unsigned)ld * 2u;
#pragma unroll
            for (int piece = 0; piece < PIECES; ++piece) {
                const int row = piece * RPP + prow;
                const int colb = (head * HD + (pc ^ tswz<NCH>(row)) * 8) * 2;
                const char* src = qb + (unsigned long)(unsigned)min(i0 + row, T - 1) * ldB + colb;
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                                 (__attribute__((address_space(3))) void*)(wscr + piece * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        stage_rows(p.dctx, p.ldd);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dc[ks] = *reinterpret_cast<const bf16x8*>(wscr + r * ROWB + (((ks * 2 + h2) ^ tswz<NCH>(r)) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stage_rows(p.out, p.ldo);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 cx = *reinterpret_cast<const bf16x8*>(wscr + r * ROWB + (((ks * 2 + h2) ^ tswz<NCH>(r)) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) delta = fmaf(bf2f(dc[ks][j]), bf2f(cx[j]), delta);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        delta = half_swap_sum(delta);
        lse2 = p.lse[((long)b * p.H + head) * T + min(i0 + r, T - 1)];
        // both blocks of the band buffer start out zero
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = q * 8 + (lane >> 3), ch = lane & 7;
            *reinterpret_cast<bf16x8*>(band + row * BAND_B + ch * 16) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // BW: where this wave's rows live in the three outputs; the band's block `it` is columns [cb0 + 32 it, +32) of dBD
    const long orow0 = ((long)head * p.B + b) * T + i0;
    const unsigned long long drop_row0 = (unsigned long long)(((long)head * p.B + b) * T + min(i0 + r, T - 1)) * (unsigned long long)Tk;     // logical index of this lane's query row, key 0
    const int cb0 = T - 32 + p.pad - i0;                     // a multiple of 32 (the host chose pad so)
    const int srow = lane >> 2, sch = lane & 3;              // staging read-back: 16 rows x four 16-B chunks per instruction

    f32x16 O[BW ? 1 : NTO];
#pragma unroll
    for (int t = 0; t < (BW ? 1 : NTO); ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] = 0.f;
    float m = -1e30f, l = 0.f;                               // running maximum in the exp2 domain (scores * scale * log2 e)
    f32x16 Gc;                                               // carried G tile (lower half of this step's band)
#pragma unroll
    for (int e = 0; e < 16; ++e) Gc[e] = 0.f;

    // per-lane LDS offsets that do not change from step to step: K / P fragment rows, V^T transposed-read addresses
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r * ROWB + (((ks * 2 + h2) ^ tswz<NCH>(r)) << 4);
    int voff[NTO][2];
    {
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int col = t * 32 + (g & 1) * 16 + 4 * p4;             // hd column of this lane's 4 elements
                const int krow = 8 * hi + 4 * (g >> 1) + q4;                // key row within a 16-key half (the half adds 16 rows: a multiple of NCH, the swizzle is unchanged)
                voff[t][hi] = krow * ROWB + (((col >> 3) ^ tswz<NCH>(krow)) << 4) + (col & 7) * 2;
            }
    }

    // BW: dQu^T[c][query] += K^T · dS^T and dQv^T[c][query] += P_band^T · dG^T — the forward's O^T += V^T · P^T with other operands (transposed LDS reads of a
    // [row][hd] tile as A, 32 x 32 accumulator registers re-used as the B operand)
    f32x16 Ou[BW ? NTO : 1], Ov[BW && REL ? NTO : 1];
    if constexpr (BW) {
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                Ou[t][e] = 0.f;
                if (REL) Ov[t][e] = 0.f;
            }
    }
    auto tr_mma = [&](const char* tile, const bf16x8 (&bb)[2], f32x16* acc) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const unsigned vb = (unsigned)(size_t)tile;
        s16x4 lo[NTO][2], hi[NTO][2];
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[t][s2]) : "v"(vb + voff[t][0]), "n"(s2 * 16 * ROWB) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[t][s2]) : "v"(vb + voff[t][1]), "n"(s2 * 16 * ROWB) : "memory");
            }
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            if (t == NTO - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (t == NTO - 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (t == NTO - 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                asm volatile("" : "+v"(lo[t][s2]), "+v"(hi[t][s2]));
                const s16x8 v8 = {lo[t][s2][0], lo[t][s2][1], lo[t][s2][2], lo[t][s2][3], hi[t][s2][0], hi[t][s2][1], hi[t][s2][2], hi[t][s2][3]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), bb[s2], acc[t], 0, 0, 0);
            }
        }
    };
    // the band block's 32 relative positions of this lane's query as the B operand (k-step s2, element j <-> band row 16 s2 + 8 (j >> 2) + 4 h2 + (j & 3))
    auto band_operand = [&](int blk, bf16x8 (&gb)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x4 a = *reinterpret_cast<const bf16x4*>(band + r * BAND_B + blk * 64 + (16 * s2 + 4 * h2) * 2);
            const bf16x4 c = *reinterpret_cast<const bf16x4*>(band + r * BAND_B + blk * 64 + (16 * s2 + 8 + 4 * h2) * 2);
            gb[s2] = bf16x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        }
    };
    auto gtile = [&](int slot) {                             // G^T tile of the position block in ring slot `slot` against (q+v)
        f32x16 G;
#pragma unroll
        for (int e = 0; e < 16; ++e) G[e] = 0.f;
        const char* pb = sP + slot * TILEB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pb + foff[ks]);
            G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[ks], G, 0, 0, 0);
        }
        return G;
    };
    // wave w's band at step it = position blocks (it - w - 1) [lower, carried] and (it - w) [upper, new]
    int pslot_new = (0 - wave + PRING * 4) % PRING;
    int pslot_low = 0;
    if (REL && nkt > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        Gc = gtile((0 - wave - 1 + PRING * 4) % PRING);
    }
    const float sc2 = p.scale * 1.4426950408889634f;
    bf16x8 st_p[2], st_s[2], st_b[2];                         // BW: a step's output rows, held until the next step has issued its DMA
    auto flush_rows = [&](int its) {                          // the stores of step `its`
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = q * 16 + srow;
            const int cb = cb0 + 32 * its;
            if (i0 + row < T) {
                *reinterpret_cast<bf16x8*>(p.prob + (orow0 + row) * p.ldsr + 32 * its + sch * 8) = st_p[q];
                *reinterpret_cast<bf16x8*>(p.ds + (orow0 + row) * p.ldsr + 32 * its + sch * 8) = st_s[q];
                if (REL && cb >= 0 && cb < (int)p.ldbd) *reinterpret_cast<bf16x8*>(p.dbd + (orow0 + row) * p.ldbd + cb + sch * 8) = st_b[q];
            }
        }
    };

    for (int it = 0; it < nkt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (it + 1 < nkt) issue_step(it + 1);
        pslot_dma = pslot_dma + 1 == PRING ? 0 : pslot_dma + 1;
        if constexpr (BW) {
            if (it > 0) flush_rows(it - 1);
            asm volatile("" ::: "memory");
        }
        const int j0 = 32 * it;
        // ---- S^T = K_tile · (Q+u)^T
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
        {
            const char* kb = sK + (it & 1) * TILEB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + foff[ks]);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[ks], S, 0, 0, 0);
            }
        }
        if (REL) {
            const f32x16 Gn = gtile(pslot_new);
            pslot_low = pslot_new == 0 ? PRING - 1 : pslot_new - 1;         // the carried (lower) block's slot: BW multiplies its rows into dQv
            pslot_new = pslot_new + 1 == PRING ? 0 : pslot_new + 1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                skew[r * SKEW_LD + crow(e, h2)] = Gc[e];
                skew[r * SKEW_LD + 32 + crow(e, h2)] = Gn[e];
            }
            Gc = Gn;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; ++e) S[e] += skew[r * SKEW_LD + crow(e, h2) - r + 31];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- scale (into the exp2 domain), mask, online softmax (lane = query).  Masks only on tiles that touch the key length or the causal diagonal (wave-uniform).
        const bool edge = (j0 + 32 > len) || (p.causal && j0 + 31 > i0 + coff);
        float mx = -1e30f;
        if (edge) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int jj = j0 + crow(e, h2);
                const bool dead = (jj >= len) || (p.causal && jj > i0 + r + coff);
                S[e] = dead ? -INFINITY : S[e] * sc2;
                mx = fmaxf(mx, S[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[e] *= sc2;
                mx = fmaxf(mx, S[e]);
            }
        }
        if constexpr (BW) {
            // ---- P = 2^(S - lse) (dead keys: S = -inf -> 0); dP^T = V_tile · dctx^T; dS = P (dP - delta) scale
            f32x16 D;
#pragma unroll
            for (int e = 0; e < 16; ++e) D[e] = 0.f;
            {
                const char* vb = sV + (it & 1) * TILEB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + foff[ks]);
                    D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dc[ks], D, 0, 0, 0);
                }
            }
            if (p.drop_p > 0.f) {
                // the forward multiplied V by P keep / (1 - p): dctx V^T is the gradient of THAT, delta = dctx · ctx already is sum_j P_j keep_j dP_j, and the P that
                // leaves for dV = P^T dctx is the dropped one
                float ks[16];
                keep16(p.drop_key, p.drop_p, drop_row0 + (unsigned long long)j0, h2, (Tk & 3) != 0, ks);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pr = __builtin_amdgcn_exp2f(S[e] - lse2);
                    D[e] = pr * (D[e] * ks[e] - delta) * p.scale;
                    S[e] = pr * ks[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S[e] = __builtin_amdgcn_exp2f(S[e] - lse2);
                    D[e] = S[e] * (D[e] - delta) * p.scale;
                }
            }
            {
                bf16x8 db[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) db[s2][j] = f2bf(D[8 * s2 + j]);
                tr_mma(sK + (it & 1) * TILEB, db, Ou);
            }
            // ---- stage the two 32 x 32 tiles as [query][key] rows in the (now idle) skew scratch, and drop dS into the band buffer at column key - query + 31
            // of block (it & 1): the lower block is complete after this step (its other half came from step it - 1), the upper one is finished by step it + 1.
            char* stP = wscr;
            char* stS = wscr + 32 * STG_B;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const bf16x4 pv = {f2bf(S[4 * g4 + 0]), f2bf(S[4 * g4 + 1]), f2bf(S[4 * g4 + 2]), f2bf(S[4 * g4 + 3])};
                const bf16x4 dv = {f2bf(D[4 * g4 + 0]), f2bf(D[4 * g4 + 1]), f2bf(D[4 * g4 + 2]), f2bf(D[4 * g4 + 3])};
                *reinterpret_cast<bf16x4*>(stP + r * STG_B + (8 * g4 + 4 * h2) * 2) = pv;
                *reinterpret_cast<bf16x4*>(stS + r * STG_B + (8 * g4 + 4 * h2) * 2) = dv;
#pragma unroll
                for (int j = 0; REL && j < 4; ++j) {
                    const int bc = (8 * g4 + 4 * h2 + j - r + 31 + 32 * (it & 1)) & 63;
                    *reinterpret_cast<bf16_t*>(band + r * BAND_B + bc * 2) = dv[j];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if constexpr (REL) {
                bf16x8 gb[2];
                band_operand(it & 1, gb);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                tr_mma(sP + pslot_low * TILEB, gb, Ov);
            }
            // back out of the staging as whole 64-B row segments, into registers: the global stores are issued at the top of the NEXT step, behind its DMA —
            // stores count in vmcnt like loads, and issued here they would be waited for by the very next instruction (the step's vmcnt(0)).
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = q * 16 + srow;
                st_p[q] = *reinterpret_cast<const bf16x8*>(stP + row * STG_B + sch * 16);
                st_s[q] = *reinterpret_cast<const bf16x8*>(stS + row * STG_B + sch * 16);
                char* bp = band + row * BAND_B + (it & 1) * 64 + sch * 16;
                st_b[q] = *reinterpret_cast<const bf16x8*>(bp);
                *reinterpret_cast<bf16x8*>(bp) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};        // same lane, same address: this block is the upper one of step it + 1
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
            mx = half_swap_max(mx);
            // Lazy rescale: the running maximum is a reference point, not a value — any m with S - m bounded gives the same quotient O / l.  It moves (and O, l are rescaled:
            // 64 accumulator registers) only when some query's maximum grew by more than 2^11; otherwise p = 2^(S - m) <= 2^11 with the stale m, exact in fp32 / bf16 range.
            if (__builtin_amdgcn_ballot_w64(mx > m + 11.f) != 0) {
                const float mnew = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                l *= alpha;
                m = mnew;
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[t][e] *= alpha;
            }
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[e] = __builtin_amdgcn_exp2f(S[e] - m);
                ls += S[e];
            }
            l += half_swap_sum(ls);
            if (p.drop_p > 0.f) {                            // probability dropout (training forward): the normaliser keeps every key, the PV product the survivors / (1 - p)
                float ks[16];
                keep16(p.drop_key, p.drop_p, drop_row0 + (unsigned long long)j0, h2, (Tk & 3) != 0, ks);
#pragma unroll
                for (int e = 0; e < 16; ++e) S[e] *= ks[e];
            }
            bf16x8 pb[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[s2][j] = f2bf(S[8 * s2 + j]);
            // ---- O^T += V^T · P^T ; V^T fragments by transposed LDS reads of the [key][hd] tile.
            // The reads are issued as inline asm with their own lgkmcnt waits: through the builtin the compiler cannot tell them from the LDS-DMA destinations of step
            // it + 1 (issued at the top of this step) and puts `s_waitcnt vmcnt(0)` in front of them — the prefetch then has to land within ~60 % of a step.
            {
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const unsigned vb = (unsigned)(size_t)(sV + (it & 1) * TILEB);          // LDS byte address (addrspace(3) pointers are 32-bit offsets)
                s16x4 lo[NTO][2], hi[NTO][2];
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[t][s2]) : "v"(vb + voff[t][0]), "n"(s2 * 16 * ROWB) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[t][s2]) : "v"(vb + voff[t][1]), "n"(s2 * 16 * ROWB) : "memory");
                    }
#pragma unroll
                for (int t = 0; t < NTO; ++t) {
                    // tile t's four reads are the oldest outstanding: 4 * (NTO - 1 - t) younger ones may still be in flight
                    if (t == NTO - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else if (t == NTO - 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else if (t == NTO - 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        // the asm outputs are only valid after the wait above: keep the compiler from reading them earlier
                        asm volatile("" : "+v"(lo[t][s2]), "+v"(hi[t][s2]));
                        const s16x8 v8 = {lo[t][s2][0], lo[t][s2][1], lo[t][s2][2], lo[t][s2][3], hi[t][s2][0], hi[t][s2][1], hi[t][s2][2], hi[t][s2][3]};
                        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pb[s2], O[t], 0, 0, 0);
                    }
                }
            }
        }
    }

    if constexpr (BW) {
        // ---- the band's last block (upper half of the last step), then zeros over every column block of this wave's rows that no step wrote:
        // keys past the length / the causal edge in P and dS, relative positions outside this wave's band in dBD.
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (nkt > 0) flush_rows(nkt - 1);
        const int nbd = (int)p.ldbd >> 5, nbs = (int)p.ldsr >> 5;
        const int blo = cb0 >> 5, bhi = nkt > 0 ? blo + nkt : blo - 1;          // band blocks [blo, bhi] were (or, for bhi, are now) written from the buffer
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = q * 16 + srow;
            if (i0 + row >= T) continue;
            if (REL && nkt > 0) {
                const int cb = cb0 + 32 * nkt;
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(band + row * BAND_B + (nkt & 1) * 64 + sch * 16);
                if (cb >= 0 && cb < (int)p.ldbd) *reinterpret_cast<bf16x8*>(p.dbd + (orow0 + row) * p.ldbd + cb + sch * 8) = bv;
            }
            const int bmax = blo + ((T + 31) >> 5);                    // last block of this wave's band when every key is valid
            for (int blk = p.sparse ? max(blo, 0) : 0; REL && blk < (p.sparse ? min(nbd, bmax + 1) : nbd); ++blk)
                if (blk < blo || blk > bhi) *reinterpret_cast<bf16x8*>(p.dbd + (orow0 + row) * p.ldbd + blk * 32 + sch * 8) = z8;
            for (int blk = nkt; blk < ((p.sparse && !p.causal) ? min(nbs, (nkt + 3) & ~3) : nbs); ++blk) {      // (causal: a query block's future keys are zeros the dV / dK products DO read)
                *reinterpret_cast<bf16x8*>(p.prob + (orow0 + row) * p.ldsr + blk * 32 + sch * 8) = z8;
                *reinterpret_cast<bf16x8*>(p.ds + (orow0 + row) * p.ldsr + blk * 32 + sch * 8) = z8;
            }
        }
        // ---- the last band block's share of dQv (its position rows are the last step's upper block, still in the ring)
        if constexpr (REL) {
            if (nkt > 0) {
                bf16x8 gb[2];
                band_operand(nkt & 1, gb);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                tr_mma(sP + (pslot_new == 0 ? PRING - 1 : pslot_new - 1) * TILEB, gb, Ov);
            }
        }
        // ---- column sums of dQu / dQv over this wave's valid queries (the bias gradients' partials): 64 channels at a time through the scratch as fp32 rows
        const int rows_valid = min(32, T - i0);
        const long psum0 = ((long)b * (gridDim.x * 4) + blockIdx.x * 4 + wave) * (p.ldsum ? p.ldsum : (long)p.H * HD) + (long)head * HD;
        auto colsum_out = [&](const f32x16* acc, float* dst) {
#pragma unroll
            for (int hh = 0; hh < HD / 64; ++hh) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 v4 = {acc[2 * hh + tt][4 * g4 + 0], acc[2 * hh + tt][4 * g4 + 1], acc[2 * hh + tt][4 * g4 + 2], acc[2 * hh + tt][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(wscr + r * OST_B + (tt * 32 + 8 * g4 + 4 * h2) * 4) = v4;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                float su = 0.f;
                for (int rr = 0; rr < rows_valid; ++rr) su += *reinterpret_cast<const float*>(wscr + rr * OST_B + lane * 4);
                dst[psum0 + hh * 64 + lane] = su;
            }
        };
        if constexpr (REL) {
            colsum_out(Ou, p.dsum_u);
            colsum_out(Ov, p.dsum_v);
        }
        // ---- dQ rows (dQu + dQv) as bf16, staged like the forward's output rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = t * 32 + 8 * g4 + 4 * h2;
                bf16x4 o;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) o[k4] = f2bf(REL ? Ou[t][4 * g4 + k4] + Ov[REL ? t : 0][4 * g4 + k4] : Ou[t][4 * g4 + k4]);
                *reinterpret_cast<bf16x4*>(wscr + r * OST_B + c * 2) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            constexpr int LPR = ROWB / 16, RPI = 64 / LPR;
            const int orow = lane / LPR, och = lane % LPR;
#pragma unroll
            for (int q = 0; q < 32 / RPI; ++q) {
                const int row = q * RPI + orow;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(wscr + row * OST_B + och * 16);
                if (i0 + row < T) *reinterpret_cast<bf16x8*>(p.dq + ((long)b * T + i0 + row) * p.lddq + head * HD + och * 8) = v;
            }
        }
    } else {
        // ---- epilogue: a lane owns a query ROW of O (hd values spread over its accumulators); row-per-lane global stores touch 64 lines per instruction and are
        // store-issue-bound.  Rows go through the wave's scratch instead and leave as whole 16-B-per-lane rows (64 lanes = 1 KiB = 4 or 8 complete rows per store).
        {
            const float inv = 1.f / l;
            if (p.lse && h2 == 0 && i0 + r < T) p.lse[((long)b * p.H + head) * T + i0 + r] = m + __builtin_amdgcn_logf(l);      // log2 domain (v_log_f32)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = t * 32 + 8 * g4 + 4 * h2;
                    const bf16x4 o = {f2bf(O[t][4 * g4 + 0] * inv), f2bf(O[t][4 * g4 + 1] * inv), f2bf(O[t][4 * g4 + 2] * inv), f2bf(O[t][4 * g4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(wscr + r * OST_B + c * 2) = o;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            constexpr int LPR = ROWB / 16, RPI = 64 / LPR;       // lanes per row, rows per store instruction
            const int orow = lane / LPR, och = lane % LPR;
#pragma unroll
            for (int q = 0; q < 32 / RPI; ++q) {
                const int row = q * RPI + orow;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(wscr + row * OST_B + och * 16);
                if (i0 + row < T) *reinterpret_cast<bf16x8*>(p.out + ((long)b * T + i0 + row) * p.ldo + head * HD + och * 8) = v;
            }
        }
    }
}


// ======================================================================================================================
// Eight-wave form of the LDS-staged forward (round 4; head size 64 / 128).  A block = 8 waves = 128 consecutive queries of one (batch, head), TWO waves per SIMD:
// waves w and w + 4 own the SAME 32 queries (query group qi = w & 3) and split the keys by tile parity (s = w >> 2 takes the tiles t with t & 1 == s), each with its own
// flash state (m, l, O); the pair is merged once at the end.  The two halves run HALF A TILE APART: a tile is two phases,
//     P1(t): S^T = K_t (Q + u)^T, both G tiles of the band, the rel-shift's select + write               (24 MFMAs at head size 128, few VALU)
//     P2(t): shifted band read back, scale, mask, online softmax, O^T += V_t^T P^T                        (8 MFMAs, the VALU work)
// and in interval i (one s_barrier each) the half with s == (i & 1) runs P1(i) while the other runs P2(i - 1): on every SIMD one wave feeds the matrix pipe while
// its partner does the soft-max — with one wave per SIMD (rounds 1-3) every segment ran at 1.5-3x its issue count because nothing else was there to issue.
//  * Per interval ONE K tile, ONE V tile and ONE position block are requested (a 1-KiB LDS-DMA piece per wave and operand at head size 128), two intervals ahead of
//    their first use: K / V rings of 3 tiles (4 without the position ring), position ring of 7 blocks; the wait at the top of an interval is a counted vmcnt that
//    leaves the previous interval's requests in flight.
//  * Relative positions: block k = rows RB0 + 32 k of the projected table; (qi, t) needs k = t - qi (lower G tile) and t - qi + 1 (upper): P1(t) reads blocks t - 3 .. t + 1.
//    A wave computes BOTH G tiles (the carried tile of the four-wave form would belong to the partner: 32 instead of 24 MFMAs per tile).  The rel-shift no longer needs a
//    64-column fp32 row per query: of the lower tile's element c and the upper tile's element c exactly one is inside the band's parallelogram (lower iff c >= 31 - r),
//    and both land on key (c + r + 1) & 31 — one select (lane masks in SGPRs), ONE ds_write_b32 per register into a 32-column row, and the shifted tile comes back as
//    four ds_read_b128: 4.5 KiB of scratch per wave instead of 8.5.
//  * LDS map: [K ring][V ring][position ring][8 skew scratches, pair-adjacent][8 x 512 B].  Q is staged into the pair's two scratches (9 KiB >= one 32-row tile), the
//    head's pos_bias_u / pos_bias_v into the 512-B areas (per-lane global loads of them cost 3 us of every launch in rounds 1-3); after the loop the ring area is the
//    exchange region of the merge: each wave hands the partner the half of O^T (hd rows) it does not keep, so both combine, normalise and store one half of the head.
//  * blockIdx -> (query block, head, batch) is XCD-aware: the hardware deals consecutive workgroups round-robin to the 8 XCDs, so consecutive LOGICAL blocks (the query
//    blocks of one (batch, head), which read the same K / V / position rows) are given ids that are congruent mod 8: K and V reach one L2 once per (batch, head).
#ifdef ATTN_STAMPS
#define STAMP(i) do { if (lane == 0) sp[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
constexpr int SC8_LD = 36;                       // words per query row of a wave's skew scratch (32 + 4: rows stay 16-B aligned for ds_read_b128)
constexpr int SC8_B = 32 * SC8_LD * 4;           // 4608 B per wave; re-used as the output staging (32 rows x (hd / 2 x 2 B + 16))
template <int HD, bool REL, bool DROP>
__global__ __launch_bounds__(512) void attn8_kernel(AttnArgs p) {
    constexpr int KS = HD / 16, NTO = HD / 32, NH = NTO / 2;
    constexpr int ROWB = HD * 2, NCH = ROWB / 16;
    constexpr int TILEB = 32 * ROWB;
    constexpr int RPP = 1024 / ROWB;                          // rows per 1-KiB DMA piece
    constexpr int PIECES = TILEB / 1024;                      // per 32-row tile: 8 (hd 128: one per wave) / 4 (hd 64)
    constexpr int KD = REL ? 3 : 4;                           // K / V ring depth
    constexpr int FD = KS < 6 ? KS : 6;                      // P1: MFMA-fragment ds_read_b128 in flight per chain
    constexpr int PRING = 7;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = sK + KD * TILEB;
    char* sP = sV + KD * TILEB;
    char* scr_all = sP + (REL ? PRING * TILEB : 0);
    float* ml_all = reinterpret_cast<float*>(scr_all + 8 * SC8_B);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = wave & 3, s = wave >> 2;
    const int r = lane & 31, h2 = lane >> 5;
    // XCD-aware block order (see above): hardware id L runs on XCD L % 8; logical id = (L % 8) * (N / 8) + L / 8.  Quotients by host-made reciprocals: no division, no
    // dependent scalar load in front of the prologue's requests
    int head, b, ib;
    {
        const unsigned N = gridDim.x;
        unsigned L = blockIdx.x;
        if ((N & 7) == 0) L = (L & 7) * (N >> 3) + (L >> 3);
        const unsigned q1 = p.gx == 1 ? L : __umulhi(L, p.mgx);               // L / gx   (ceil(2^32 / 1) does not fit 32 bits)
        const unsigned q2 = p.H == 1 ? q1 : __umulhi(q1, p.mgy);              // L / (gx H)
        ib = (int)(L - q1 * (unsigned)p.gx) * 128;
        head = (int)(q1 - q2 * (unsigned)p.H);
        b = (int)q2;
    }
    const int i0 = ib + qi * 32;
    char* scr = scr_all + (qi * 2 + s) * SC8_B;               // pair-adjacent: the pair's two scratches are one 9-KiB area (Q staging)
#ifdef ATTN_STAMPS
    unsigned long long* sp = p.stamps + ((long)blockIdx.x * 8 + wave) * 32;
    STAMP(0);
#endif
    const int T = p.T;
    const int Tk = p.Tk > 0 ? p.Tk : p.T;
    const int coff = Tk - T;
    const int RB0 = T - 1 - ib - 31;                          // first row of position block 0 (REL needs Tk == T)

    // ---- DMA addressing: this wave's piece of a 32-row tile is the same for K, V and the position block
    const int prow = lane / NCH, pc = lane % NCH;
    const int piece = wave & (PIECES - 1);
    const bool doK = PIECES == 8 || wave < 4, doV = PIECES == 8 || wave >= 4, doP = REL && (PIECES == 8 || wave < 4);
    const int trow = piece * RPP + prow;                      // 0..31
    const long kvoff = p.kv_bstride ? (long)b * p.kv_bstride : 0;
    const int colb = (head * HD + (pc ^ tswz<NCH>(trow)) * 8) * 2;
    const char* kB = reinterpret_cast<const char*>(p.k + (p.kv_bstride ? kvoff : (long)b * Tk * p.ldk)) + colb;
    const char* vB = reinterpret_cast<const char*>(p.vt + (p.kv_bstride ? kvoff : (long)b * Tk * p.ldvt)) + colb;
    const char* pB = reinterpret_cast<const char*>(p.pos) + colb;
    const unsigned ldkB = (unsigned)p.ldk * 2u, ldvB = (unsigned)p.ldvt * 2u, ldpB = (unsigned)p.ldp * 2u;
    auto issue_k = [&](int t, int slot) {
        const unsigned gr = (unsigned)min(32 * t + trow, Tk - 1);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(kB + (unsigned long)gr * ldkB),
                                         (__attribute__((address_space(3))) void*)(sK + slot * TILEB + piece * 1024), 16, 0, 0);
    };
    auto issue_v = [&](int t, int slot) {
        const unsigned gr = (unsigned)min(32 * t + trow, Tk - 1);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(vB + (unsigned long)gr * ldvB),
                                         (__attribute__((address_space(3))) void*)(sV + slot * TILEB + piece * 1024), 16, 0, 0);
    };
    auto issue_p = [&](int k, int slot) {
        const unsigned gr = (unsigned)min(max(RB0 + 32 * k + trow, 0), 2 * T - 2);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(pB + (unsigned long)gr * ldpB),
                                         (__attribute__((address_space(3))) void*)(sP + slot * TILEB + piece * 1024), 16, 0, 0);
    };

    // ---- prologue: the pair's Q tile first (it is needed first), the head's position biases, then K tiles 0 .. KD - 2, V tiles 0 .. KD - 3, position blocks -3 .. 2
    char* qst = scr_all + qi * 2 * SC8_B;
    {
        const char* qb = reinterpret_cast<const char*>(p.q + (long)b * T * p.ldq);
        const unsigned ldqB = (unsigned)p.ldq * 2u;
#pragma unroll
        for (int pp = 0; pp < PIECES / 2; ++pp) {
            const int pq = s * (PIECES / 2) + pp;
            const int row = pq * RPP + prow;
            const int cq = (head * HD + (pc ^ tswz<NCH>(row)) * 8) * 2;
            const char* src = qb + (unsigned long)(unsigned)min(i0 + row, T - 1) * ldqB + cq;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                             (__attribute__((address_space(3))) void*)(qst + pq * 1024), 16, 0, 0);
        }
    }
    if (REL && wave < 2) {                                    // wave 0: pos_bias_u[head], wave 1: pos_bias_v[head] -> ml_all[0 .. HD), ml_all[256 .. 256 + HD) (lanes past HD / 4 re-load the start)
        const float* bsrc = (wave == 0 ? p.bias_u : p.bias_v) + head * HD + (lane % (HD / 4)) * 4;
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)bsrc,
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(ml_all) + wave * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < KD - 1; ++t) if (doK) issue_k(t, t);
#pragma unroll
    for (int t = 0; t < KD - 2; ++t) if (doV) issue_v(t, t);
    if (REL) {
#pragma unroll
        for (int k = -3; k <= 2; ++k) if (doP) issue_p(k, (k + PRING) % PRING);
    }
    STAMP(1);
    asm volatile("" ::: "memory");
    // the key length is a dependent load from HBM: asked for only now, behind the prologue's requests (it sat in front of them for 1 - 2 us of every launch)
    const int len = p.lengths ? min(p.lengths[b], Tk) : Tk;
    const int kend = p.causal ? min(len, ib + 128 + coff) : len;
    const int nkt = (kend + 31) / 32;
    // per-lane constants of the loop while the prologue's loads fly
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r * ROWB + (((ks * 2 + h2) ^ tswz<NCH>(r)) << 4);
    int voff[NTO][2];
    {
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int col = t * 32 + (g & 1) * 16 + 4 * p4;
                const int krow = 8 * hi + 4 * (g >> 1) + q4;
                voff[t][hi] = krow * ROWB + (((col >> 3) ^ tswz<NCH>(krow)) << 4) + (col & 7) * 2;
            }
    }
    // rel-shift: register e of a G tile is band column c = crow(e, h2) (lower tile) or 32 + c (upper); the lower one is inside the band iff c >= 31 - r; both belong to key (c + r + 1) & 31
    bool low_in[REL ? 16 : 1];
    unsigned wadr[REL ? 16 : 1];
    if constexpr (REL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = crow(e, h2);
            low_in[e] = c >= 31 - r;
            wadr[e] = (unsigned)(size_t)scr + (unsigned)((r * SC8_LD + ((c + r + 1) & 31)) * 4);
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(foff[ks]));        // computed under the prologue's loads, not behind them
#pragma unroll
    for (int t = 0; t < NTO; ++t) asm volatile("" : "+v"(voff[t][0]), "+v"(voff[t][1]));
    if constexpr (REL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(wadr[e]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(2);
    asm volatile("s_barrier" ::: "memory");
    STAMP(3);
    bf16x8 qu[KS], qv[REL ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qst + foff[ks]);
        if (REL) {
            const int c = ks * 16 + h2 * 8;
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(ml_all + c), u1 = *reinterpret_cast<const f32x4*>(ml_all + c + 4);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ml_all + 256 + c), v1 = *reinterpret_cast<const f32x4*>(ml_all + 256 + c + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f(raw[j]);
                qu[ks][j] = f2bf(f + (j < 4 ? u0[j & 3] : u1[j & 3]));
                qv[ks][j] = f2bf(f + (j < 4 ? v0[j & 3] : v1[j & 3]));
            }
        } else {
            qu[ks] = raw;
        }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // Q and the biases are in registers before the first interval's barrier lets the skew writes into the staging area
    const unsigned long long drop_row0 = (unsigned long long)(((long)head * p.B + b) * T + min(i0 + r, T - 1)) * (unsigned long long)Tk;
    f32x16 O[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] = 0.f;
    float m = -1e30f, l = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    int pend = 0;                                             // LDS-DMA pieces this wave requested in the previous interval
    int iv = 0;                                               // interval counter: every wave passes exactly nkt + 1 interval heads (barriers)
    auto head_of_interval = [&]() {
        if (iv < 4) STAMP(4 + 3 * iv);
        // everything requested two or more intervals ago has landed.  No lgkmcnt wait: every read of a shared tile was consumed by an MFMA behind a counted wait, and what
        // may still be queued are the wave's own skew writes, which its own reads of the next interval follow in order.
        if (pend >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (iv < 4) STAMP(5 + 3 * iv);
        asm volatile("s_barrier" ::: "memory");
        if (iv < 4) STAMP(6 + 3 * iv);
    };
    // this interval's requests (K tile iv + KD - 1, V tile iv + KD - 2, position block iv + 3); called once per interval, BEHIND the phase's first LDS reads so that their
    // round trip passes under the address arithmetic and the issue of the pieces (100 - 185 cycles each inside a busy phase)
    auto issue_dma_part = [&](int part) {                   // part 0: K, 1: V, 2: position block; part 0 opens the interval's count, part 2 closes the interval
        if (part == 0) { pend = 0; if (doK && iv + KD - 1 < nkt) { issue_k(iv + KD - 1, (iv + KD - 1) % KD); ++pend; } }
        if (part == 1) { if (doV && iv + KD - 2 < nkt) { issue_v(iv + KD - 2, (iv + KD - 2) % KD); ++pend; } }
        if (part == 2) { if (doP && iv + 3 <= nkt) { issue_p(iv + 3, (iv + 3) % PRING); ++pend; } ++iv; }
    };
    auto issue_dma = [&]() { issue_dma_part(0); issue_dma_part(1); issue_dma_part(2); };
    if (s == 1) {
        __builtin_amdgcn_s_setprio(1);                        // waves 4-7 are the younger ones on their SIMDs and lose every issue arbitration to their partners: one static raise, no per-phase flips
        head_of_interval();                                   // the odd half starts half a tile late
        issue_dma();
    }
    int kvslot = s;                                           // t % KD, advanced by 2 per tile
    int slo = (s - qi + PRING) % PRING;                       // ring slot of position block t - qi, advanced by 2 per tile
    for (int t = s; t < nkt; t += 2) {
        const int j0 = 32 * t;
        const bool live = i0 < T && !(p.causal && j0 > i0 + 31 + coff);      // wave-uniform
        // ---- P1(t)
        head_of_interval();
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
        if (live) {
            // ONE stream of MFMAs (REL: lower G tile, upper G tile, S; else S) whose A fragments (rows of [32][hd] LDS tiles) are requested FD deep by hand with counted
            // waits: left to the compiler every use of a fragment is a full `s_waitcnt lgkmcnt(0)`, and a drain at the head of each chain exposes a loaded LDS round trip.
            constexpr int NF = REL ? 3 * KS : KS;
            const unsigned bK = (unsigned)(size_t)(sK + kvslot * TILEB);
            const unsigned bPl = (unsigned)(size_t)(sP + slo * TILEB), bPu = (unsigned)(size_t)(sP + (slo + 1 == PRING ? 0 : slo + 1) * TILEB);
            bf16x8 f[FD];
            f32x16 Gl = S, Gu = S;                            // zeros
#pragma unroll
            for (int n = 0; n < FD; ++n) {
                const unsigned base = !REL ? bK : (n / KS == 0 ? bPl : (n / KS == 1 ? bPu : bK));
                asm volatile("ds_read_b128 %0, %1" : "=v"(f[n]) : "v"(base + foff[n % KS]) : "memory");
            }
            if (t == s + 2) STAMP(16);
            issue_dma_part(0);                                // one piece now, the others a third and two thirds into the stream: eight waves issuing three pieces each right behind the barrier queue up (~400 cycles)
            if (t == s + 2) STAMP(17);
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                if (n == NF / 3) issue_dma_part(1);
                if (n == 2 * NF / 3) issue_dma_part(2);
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[n % FD]) : "n"(NF - 1 - n < FD - 1 ? NF - 1 - n : FD - 1) : "memory");
                if (!REL || n / KS == 2) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[n % FD], qu[n % KS], S, 0, 0, 0);
                else if (n / KS == 0) Gl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[n % FD], qv[REL ? n % KS : 0], Gl, 0, 0, 0);
                else Gu = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[n % FD], qv[REL ? n % KS : 0], Gu, 0, 0, 0);
                if (n + FD < NF) {
                    const int n2 = n + FD;
                    const unsigned base = !REL ? bK : (n2 / KS == 0 ? bPl : (n2 / KS == 1 ? bPu : bK));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(f[n % FD]) : "v"(base + foff[n2 % KS]) : "memory");
                }
            }
            if (t == s + 2) STAMP(18);
            if constexpr (REL) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = low_in[e] ? Gl[e] : Gu[e];
                    asm volatile("ds_write_b32 %0, %1" :: "v"(wadr[e]), "v"(v) : "memory");
                }
            }
        } else {
            issue_dma();
        }
        if (t < 4) STAMP(22);
        // ---- P2(t)
        head_of_interval();
        if (live) {
            // the shifted band (four 16-B pieces of this lane's row) and the V^T fragments are requested first, then this interval's DMA pieces are issued under their round trip
            f32x4 bd[REL ? 4 : 1];
            if constexpr (REL) {
                const unsigned sb = (unsigned)(size_t)scr + (unsigned)((r * SC8_LD + 4 * h2) * 4);
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bd[a]) : "v"(sb), "n"(32 * a) : "memory");
            }
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            s16x4 lo[NTO][2], hi[NTO][2];
            {
                const unsigned vb = (unsigned)(size_t)(sV + kvslot * TILEB);
#pragma unroll
                for (int tt = 0; tt < NTO; ++tt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[tt][s2]) : "v"(vb + voff[tt][0]), "n"(s2 * 16 * ROWB) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[tt][s2]) : "v"(vb + voff[tt][1]), "n"(s2 * 16 * ROWB) : "memory");
                    }
            }
            if (t == s + 2) STAMP(19);
            issue_dma_part(0);
            if (t == s + 2) STAMP(20);
            if constexpr (REL) {
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bd[0]), "+v"(bd[1]), "+v"(bd[2]), "+v"(bd[3]) : "n"(4 * NTO > 15 ? 15 : 4 * NTO) : "memory");      // the band only: the V^T reads stay in flight
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int k = 0; k < 4; ++k) S[4 * a + k] += bd[a][k];
            }
            // soft-max in the exp2 domain with the scale folded into ONE fma per element: p = 2^(S sc2 - m); the row maximum is taken on the raw scores (sc2 > 0)
            const bool edge = (j0 + 32 > len) || (p.causal && j0 + 31 > i0 + coff);
            float mx = -1e30f;
            if (edge) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jj = j0 + crow(e, h2);
                    const bool dead = (jj >= len) || (p.causal && jj > i0 + r + coff);
                    S[e] = dead ? -INFINITY : S[e];
                    mx = fmaxf(mx, S[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[e]);
            }
            mx = half_swap_max(mx) * sc2;
            if (__builtin_amdgcn_ballot_w64(mx > m + 11.f) != 0) {       // lazy rescale (see the four-wave form)
                const float mnew = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                l *= alpha;
                m = mnew;
#pragma unroll
                for (int tt = 0; tt < NTO; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[tt][e] *= alpha;
            }
            issue_dma_part(1);
            float ls = 0.f;
            const float negm = -m;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[e], sc2, negm));
                ls += S[e];
            }
            l += half_swap_sum(ls);
            issue_dma_part(2);
            if constexpr (DROP) {
                float ks[16];
                keep16(p.drop_key, p.drop_p, drop_row0 + (unsigned long long)j0, h2, (Tk & 3) != 0, ks);
#pragma unroll
                for (int e = 0; e < 16; ++e) S[e] *= ks[e];
            }
            bf16x8 pb[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[s2][j] = f2bf(S[8 * s2 + j]);
#ifdef ATTN_STAMPS
            asm volatile("" : "+v"(pb[0]), "+v"(pb[1]));
            if (t == s + 2) STAMP(21);
#endif
#pragma unroll
            for (int tt = 0; tt < NTO; ++tt) {
                if (tt == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // requested a soft-max ago
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    asm volatile("" : "+v"(lo[tt][s2]), "+v"(hi[tt][s2]));
                    const s16x8 v8 = {lo[tt][s2][0], lo[tt][s2][1], lo[tt][s2][2], lo[tt][s2][3], hi[tt][s2][0], hi[tt][s2][1], hi[tt][s2][2], hi[tt][s2][3]};
                    O[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pb[s2], O[tt], 0, 0, 0);
                }
            }
        } else {
            issue_dma();
        }
        if (t < 4) STAMP(23);
        kvslot = kvslot + 2 >= KD ? kvslot + 2 - KD : kvslot + 2;
        slo = slo + 2 >= PRING ? slo + 2 - PRING : slo + 2;
    }
    while (iv <= nkt) { head_of_interval(); issue_dma(); }    // the half that finishes first keeps the barrier count

    // ---- merge of the pair: the half of O^T this wave does not keep, and its (m, l), go to the partner through LDS (the ring area is free behind the barrier)
    STAMP(24);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    STAMP(25);
    {
        f32x4* xo = reinterpret_cast<f32x4*>(smem) + wave * (NH * 4 * 64) + lane;         // [tile][register quad][lane] 16-B pieces: conflict-free ds_write_b128 / ds_read_b128
        float* mlw = ml_all + wave * 128;
        mlw[lane] = m;
        mlw[64 + lane] = l;
#pragma unroll
        for (int tt = 0; tt < NH; ++tt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = s == 0 ? O[NH + tt][4 * g4 + k] : O[tt][4 * g4 + k];
                xo[(tt * 4 + g4) * 64] = v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(26);
    asm volatile("s_barrier" ::: "memory");
    STAMP(27);
    {
        const int pw = wave ^ 4;
        const f32x4* xi = reinterpret_cast<const f32x4*>(smem) + pw * (NH * 4 * 64) + lane;
        const float pm = ml_all[pw * 128 + lane], pl = ml_all[pw * 128 + 64 + lane];
        const float mn = fmaxf(m, pm);
        const float a = __builtin_amdgcn_exp2f(m - mn), pa = __builtin_amdgcn_exp2f(pm - mn);
        const float lt = l * a + pl * pa;
        const float inv = 1.f / lt;
        if (p.lse && s == 0 && h2 == 0 && i0 + r < T) p.lse[((long)b * p.H + head) * T + i0 + r] = mn + __builtin_amdgcn_logf(lt);      // log2 domain
        constexpr int STG = HD + 16;                          // bytes per staged row: hd / 2 bf16 + 16
        const float ca = a * inv, cp = pa * inv;
#pragma unroll
        for (int tt = 0; tt < NH; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 X = xi[(tt * 4 + g4) * 64];
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float mine = s == 0 ? O[tt][4 * g4 + k] : O[NH + tt][4 * g4 + k];
                    o[k] = f2bf(mine * ca + X[k] * cp);
                }
                *reinterpret_cast<bf16x4*>(scr + r * STG + (tt * 32 + 8 * g4 + 4 * h2) * 2) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        constexpr int LPR = (HD / 2 * 2) / 16, RPI = 64 / LPR;       // lanes per half row, rows per store instruction
        const int orow = lane / LPR, och = lane % LPR;
#pragma unroll
        for (int q = 0; q < 32 / RPI; ++q) {
            const int row = q * RPI + orow;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(scr + row * STG + och * 16);
            if (i0 + row < T) *reinterpret_cast<bf16x8*>(p.out + ((long)b * T + i0 + row) * p.ldo + head * HD + s * (HD / 2) + och * 8) = v;
        }
        STAMP(28);
#ifdef ATTN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(29);
#endif
    }
}

template <int HD, bool REL, bool DROP>
int launch_attn8_inst(const AttnArgs& a_in, hipStream_t stream) {
    AttnArgs a = a_in;
    a.gx = cdiv(a.T, 128);
    const long nblk = (long)a.gx * a.H * a.B;
    if (nblk >= (1l << 16)) return MI_ERR_UNSUPPORTED;        // the reciprocal quotients are exact below 2^16 blocks (the caller falls back to the four-wave form)
    a.mgx = (unsigned)(((1ull << 32) + (unsigned)a.gx - 1) / (unsigned)a.gx);
    a.mgy = (unsigned)(((1ull << 32) + (unsigned)a.H - 1) / (unsigned)a.H);
    dim3 grid((unsigned)nblk), block(512);
    const size_t lds = (size_t)(REL ? 6 + 7 : 8) * 32 * HD * 2 + 8 * SC8_B + 8 * 512;      // K + V rings (3 + 3 | 4 + 4 tiles), position ring, skew scratches, (m, l) / bias area
    if (!ensure_dynamic_lds<HD * 4 + REL * 2 + DROP>(reinterpret_cast<const void*>(attn8_kernel<HD, REL, DROP>), lds)) return MI_ERR_LAUNCH;
    hipLaunchKernelGGL((attn8_kernel<HD, REL, DROP>), grid, block, lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
template <int HD>
int launch_attn8(const AttnArgs& a, bool rel, hipStream_t stream) {
    const bool drop = a.drop_p > 0.f;
    if (rel) return drop ? launch_attn8_inst<HD, true, true>(a, stream) : launch_attn8_inst<HD, true, false>(a, stream);
    return drop ? launch_attn8_inst<HD, false, true>(a, stream) : launch_attn8_inst<HD, false, false>(a, stream);
}

template <int HD>
int launch_lds(const AttnArgs& a, bool rel, hipStream_t stream) {
    // The product's choice (variant 0): the eight-wave form, except with relative positions at head size 64 — there the four-wave form already runs two workgroups per CU
    // (75 KiB of LDS each) on 24 instead of 32 MFMAs per tile and measures 10 % faster (tools/attn_ab.py: 89 vs 100 us at 96 x 500 frames, 20.4 vs 22.5 us at 32 x 250 x 8 heads).
    if (a.variant == 2 || (a.variant == 0 && !(HD == 64 && rel))) {
        const int rc = launch_attn8<HD>(a, rel, stream);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    dim3 grid(cdiv(a.T, 128), a.H, a.B), block(256);
    const size_t lds = (size_t)(2 + 2 + 6) * 32 * HD * 2 + 4 * WSCR_B;
    if (rel) hipLaunchKernelGGL((attn_lds_kernel<HD, true, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_lds_kernel<HD, false, false>), grid, block, lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HD>
int launch_lds_bw(const AttnArgs& a, bool rel, hipStream_t stream) {
    dim3 grid(cdiv(a.T, 128), a.H, a.B), block(256);
    const size_t lds = (size_t)(2 + 2 + 6) * 32 * HD * 2 + 4 * WSCR_B + 4 * 32 * BAND_B;
    if (!ensure_dynamic_lds<1000 + HD * 2 + 1>(reinterpret_cast<const void*>(attn_lds_kernel<HD, true, true>), lds) ||
        !ensure_dynamic_lds<1000 + HD * 2>(reinterpret_cast<const void*>(attn_lds_kernel<HD, false, true>), lds)) return MI_ERR_LAUNCH;
    if (rel) hipLaunchKernelGGL((attn_lds_kernel<HD, true, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_lds_kernel<HD, false, true>), grid, block, lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HD>
int launch_hd(const AttnArgs& a, bool rel, hipStream_t stream) {
    dim3 grid(cdiv(cdiv(a.T, 32), 4), a.H, a.B), block(256);
    const size_t lds = rel ? 4 * 32 * SKEW_LD * sizeof(float) : 0;
    if (rel) hipLaunchKernelGGL((attn_kernel<HD, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_kernel<HD, false>), grid, block, lds, stream, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace

// q,k: (B*T, ld) bf16 ; vt: (H*hd, ldvt) bf16 with column b*Tp + t (Tp >= T rounded up to 32, zero padded);
// pos: (2T-1, ldp) bf16 or null (no relative term); lengths (B) int32 or null; out (B*T, ldo) bf16.
extern "C" int mi_attention_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, int Tp,
                                 const void* pos, long ldp, const float* bias_u, const float* bias_v,
                                 const int* lengths, void* out, long ldo, int B, int T, int H, int hd,
                                 float scale, int causal, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || H <= 0) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldvt % 4) || (Tp % 32) || Tp < T || (ldo % 4)) return MI_ERR_ARG;
    if (pos && ((ldp % 8) || !bias_u || !bias_v)) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)vt, ldvt, Tp, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)out, ldo, B, T, H, scale, causal};
    const bool rel = pos != nullptr;
    switch (hd) {
        case 16: return launch_hd<16>(a, rel, stream);
        case 32: return launch_hd<32>(a, rel, stream);
        case 64: return launch_hd<64>(a, rel, stream);
        case 128: return launch_hd<128>(a, rel, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}

// LDS-staged form: q, k, v are all (B*T, ld) bf16 row-major (columns of one fused QKV projection); hd in {64, 128}.
extern "C" int mi_attention_qkv_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                     const void* pos, long ldp, const float* bias_u, const float* bias_v,
                                     const int* lengths, void* out, long ldo, int B, int T, int Tk, long kv_bstride, int H, int hd,
                                     float scale, int causal, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || Tk < 0) return MI_ERR_ARG;
    if (pos && Tk != 0 && Tk != T) return MI_ERR_ARG;                    // the relative term needs a square score matrix
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || ((uintptr_t)out & 15)) return MI_ERR_ARG;                 // 16-B row stores
    if (ldq >= (1l << 30)) return MI_ERR_ARG;
    if (ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30) || ldp >= (1l << 30)) return MI_ERR_ARG;      // the kernel keeps row strides as 32-bit byte counts
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || (kv_bstride % 8)) return MI_ERR_ARG;
    if (pos && ((ldp % 8) || ((uintptr_t)pos & 15) || !bias_u || !bias_v)) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)out, ldo, B, T, H, scale, causal, Tk, kv_bstride};
    const bool rel = pos != nullptr;
    switch (hd) {
        case 64: return launch_lds<64>(a, rel, stream);
        case 128: return launch_lds<128>(a, rel, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}

// A/B form of the entry above: variant 0 = the product's kernel choice, 1 = the four-wave LDS-staged forward of rounds 1-3, 2 = the eight-wave form of round 4
// (tools/attn_ab.py; no product call site passes a non-zero variant).
extern "C" int mi_attention_qkv_bf16_v(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                       const void* pos, long ldp, const float* bias_u, const float* bias_v,
                                       const int* lengths, void* out, long ldo, int B, int T, int Tk, long kv_bstride, int H, int hd,
                                       float scale, int causal, int variant, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || Tk < 0 || variant < 0 || variant > 2) return MI_ERR_ARG;
    if (pos && Tk != 0 && Tk != T) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || ((uintptr_t)out & 15)) return MI_ERR_ARG;
    if (ldq >= (1l << 30)) return MI_ERR_ARG;
    if (ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30) || ldp >= (1l << 30)) return MI_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || (kv_bstride % 8)) return MI_ERR_ARG;
    if (pos && ((ldp % 8) || ((uintptr_t)pos & 15) || !bias_u || !bias_v)) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)out, ldo, B, T, H, scale, causal, Tk, kv_bstride};
    a.variant = variant;
    switch (hd) {
        case 64: return launch_lds<64>(a, pos != nullptr, stream);
        case 128: return launch_lds<128>(a, pos != nullptr, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}

#ifdef ATTN_STAMPS
extern "C" int mi_attention_qkv_stamps(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* pos, long ldp, const float* bias_u, const float* bias_v,
                                       const int* lengths, void* out, long ldo, int B, int T, int H, int hd, float scale, int causal, unsigned long long* stamps, hipStream_t stream) {
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)out, ldo, B, T, H, scale, causal, 0, 0};
    a.stamps = stamps;
    return hd == 64 ? launch_lds<64>(a, pos != nullptr, stream) : launch_lds<128>(a, pos != nullptr, stream);
}
#endif

// The same kernel, also leaving the rows' log-sum-exp (log2 domain, of the scaled scores) in lse (B, H, T) fp32 for mi_attention_qkv_bwd_probs.
extern "C" int mi_attention_qkv_lse_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                         const void* pos, long ldp, const float* bias_u, const float* bias_v,
                                         const int* lengths, void* out, long ldo, float* lse, int B, int T, int H, int hd,
                                         float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || !lse || drop_p < 0.f || drop_p >= 1.f) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || ((uintptr_t)out & 15)) return MI_ERR_ARG;
    if (ldq >= (1l << 30) || ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30) || ldp >= (1l << 30)) return MI_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15)) return MI_ERR_ARG;
    if (pos && ((ldp % 8) || ((uintptr_t)pos & 15) || !bias_u || !bias_v)) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)out, ldo, B, T, H, scale, causal, 0, 0, lse};
    a.drop_p = drop_p; a.drop_key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    switch (hd) {
        case 64: return launch_lds<64>(a, pos != nullptr, stream);
        case 128: return launch_lds<128>(a, pos != nullptr, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}

// First half of the attention backward (self-attention, fused QKV operand; hd in {64, 128}): from q, k, v, the projected positions, the forward's context and row
// log-sum-exp, and dctx -> prob, ds (H, B, T, ldsr) bf16 and dbd (H, B, T, ldbd) bf16 with dbd[i][T-1-i+j + pad] = ds[i][j]; and the query gradient itself,
// dq (B*T, lddq) bf16 = dS K + dBD P, with the per-wave column sums of its two terms in dsum_u / dsum_v (B, 4 ceil(T/128), H*hd) fp32.  ldsr, ldbd multiples of 32,
// ldsr >= T rounded up to 32, ldbd >= pad + 2T - 1, (T - 32 + pad) % 32 == 0: a wave's band of relative positions then starts on a 64-B boundary of its rows.
// Every element of the three outputs is written (zeros where no key / relative position contributes).
// qu_out / qv_out (both or neither; with pos only): (B*T, ldqb) bf16 = q + pos_bias_u / q + pos_bias_v, written by the walk's prologue from its A fragments.
// ... _f: the same with `flags`.  bit 0 (sparse writes): zeros nobody reads are not written — dbd must be a buffer the caller zero-filled once and that only this entry
// writes (relative positions outside a row's maximal band stay zero from launch to launch), and columns of prob / ds from the key length rounded up to 128 on must never be
// read (mi_bgemm_sparse_bf16 with m_valid = lengths does not).  At BASELINE config 3 (clips of 1-20 s padded to 20 s) that is 42 % of the walk's 800 MB of writes per launch.
extern "C" int mi_attention_qkv_bwd_probs_f(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                            const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                            const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                            void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                            void* dq, long lddq, float* dsum_u, float* dsum_v, void* qu_out, void* qv_out, long ldqb,
                                            int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, int flags, hipStream_t stream);
extern "C" int mi_attention_qkv_bwd_probs_qb(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                             const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                             const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                             void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                             void* dq, long lddq, float* dsum_u, float* dsum_v, void* qu_out, void* qv_out, long ldqb,
                                             int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t stream) {
    return mi_attention_qkv_bwd_probs_f(q, ldq, k, ldk, v, ldv, pos, ldp, bias_u, bias_v, lengths, ctx, ldo, dctx, ldd, lse, prob, ds, ldsr, dbd, ldbd, pad, dq, lddq,
                                        dsum_u, dsum_v, qu_out, qv_out, ldqb, B, T, H, hd, scale, causal, drop_p, seed, stream_id, 0, stream);
}
extern "C" int mi_attention_qkv_bwd_probs_f(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                            const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                            const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                            void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                            void* dq, long lddq, float* dsum_u, float* dsum_v, void* qu_out, void* qv_out, long ldqb,
                                            int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, int flags, hipStream_t stream) {
    MI_ENTER();
    if (flags & ~1) return MI_ERR_ARG;
    if ((qu_out != nullptr) != (qv_out != nullptr) || (qu_out && (!pos || (ldqb % 8) || ldqb < (long)H * hd || (((uintptr_t)qu_out | (uintptr_t)qv_out) & 15)))) return MI_ERR_ARG;
    if (B <= 0 || T <= 0 || H <= 0 || !lse || !ctx || !dctx || !prob || !ds || !dq || drop_p < 0.f || drop_p >= 1.f) return MI_ERR_ARG;
    if ((lddq % 8) || lddq >= (1l << 30) || ((uintptr_t)dq & 15) || (pos && (!dsum_u || !dsum_v))) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || (ldd % 8)) return MI_ERR_ARG;
    if (ldq >= (1l << 30) || ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30) || ldp >= (1l << 30) || ldo >= (1l << 30) || ldd >= (1l << 30)) return MI_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)prob | (uintptr_t)ds | (uintptr_t)dbd) & 15)) return MI_ERR_ARG;
    if ((ldsr % 32) || ldsr < (T + 31) / 32 * 32) return MI_ERR_ARG;
    if (pos) {
        if ((ldp % 8) || ((uintptr_t)pos & 15) || !bias_u || !bias_v || !dbd) return MI_ERR_ARG;
        if ((ldbd % 32) || pad < 0 || pad >= 32 || ((T - 32 + pad) % 32) || ldbd < pad + 2 * T - 1 || ldbd >= (1l << 30)) return MI_ERR_ARG;
    }
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, (const bf16_t*)pos, ldp,
               bias_u, bias_v, lengths, (bf16_t*)const_cast<void*>(ctx), ldo, B, T, H, scale, causal, 0, 0, const_cast<float*>(lse),
               (const bf16_t*)dctx, ldd, (bf16_t*)prob, (bf16_t*)ds, ldsr, (bf16_t*)dbd, pos ? ldbd : 0, pos ? pad : 0,
               (bf16_t*)dq, lddq, dsum_u, dsum_v, drop_p, ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed};
    a.ldsum = (dsum_u && dsum_v == dsum_u + (long)H * hd) ? 2l * H * hd : 0;      // [u | v] rows of one (rows, 2 H hd) buffer: see the header
    a.qu_out = (bf16_t*)qu_out; a.qv_out = (bf16_t*)qv_out; a.ldqb = ldqb;
    a.sparse = flags & 1;
    switch (hd) {
        case 64: return launch_lds_bw<64>(a, pos != nullptr, stream);
        case 128: return launch_lds_bw<128>(a, pos != nullptr, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}
extern "C" int mi_attention_qkv_bwd_probs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                          const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                          const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                          void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                          void* dq, long lddq, float* dsum_u, float* dsum_v,
                                          int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t stream) {
    return mi_attention_qkv_bwd_probs_qb(q, ldq, k, ldk, v, ldv, pos, ldp, bias_u, bias_v, lengths, ctx, ldo, dctx, ldd, lse, prob, ds, ldsr, dbd, ldbd, pad, dq, lddq,
                                         dsum_u, dsum_v, nullptr, nullptr, 0, B, T, H, hd, scale, causal, drop_p, seed, stream_id, stream);
}

// The two training entries above for Tq != Tk and separate q / k / v operands (no relative positions): the GPT-2 decoder's causal self-attention (Tq = Tk = U) and its
// cross-attention over the encoder frames (Tq = U, Tk = T', `lengths` = valid keys), multi_head_gpt2.py:80-170 — which rounds 2-3 trained through materialised scores
// (batched GEMM, soft-max, batched GEMM: three launches forward, seven backward).  lse (B, H, Tq); prob / ds (H, B, Tq, ldsr), ldsr a multiple of 32 >= Tk rounded up to 32.
extern "C" int mi_attention_x_lse_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const int* lengths, void* out, long ldo, float* lse,
                                       int B, int Tq, int Tk, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0 || !lse || drop_p < 0.f || drop_p >= 1.f) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || ((uintptr_t)out & 15)) return MI_ERR_ARG;
    if (ldq >= (1l << 30) || ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30)) return MI_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15)) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, nullptr, 0, nullptr, nullptr, lengths, (bf16_t*)out, ldo, B, Tq, H, scale, causal, Tk, 0, lse};
    a.drop_p = drop_p; a.drop_key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    switch (hd) {
        case 64: return launch_lds<64>(a, false, stream);
        case 128: return launch_lds<128>(a, false, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}
extern "C" int mi_attention_x_bwd_probs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const int* lengths,
                                        const void* ctx, long ldo, const void* dctx, long ldd, const float* lse, void* prob, void* ds, long ldsr, void* dq, long lddq,
                                        int B, int Tq, int Tk, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0 || !lse || !ctx || !dctx || !prob || !ds || !dq || drop_p < 0.f || drop_p >= 1.f) return MI_ERR_ARG;
    if ((lddq % 8) || lddq >= (1l << 30) || ((uintptr_t)dq & 15)) return MI_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || (ldd % 8)) return MI_ERR_ARG;
    if (ldq >= (1l << 30) || ldk <= 0 || ldv <= 0 || ldk >= (1l << 30) || ldv >= (1l << 30) || ldo >= (1l << 30) || ldd >= (1l << 30)) return MI_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)prob | (uintptr_t)ds) & 15)) return MI_ERR_ARG;
    if ((ldsr % 32) || ldsr < (Tk + 31) / 32 * 32) return MI_ERR_ARG;
    AttnArgs a{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, 0, nullptr, 0, nullptr, nullptr, lengths, (bf16_t*)const_cast<void*>(ctx), ldo, B, Tq, H, scale, causal,
               Tk, 0, const_cast<float*>(lse), (const bf16_t*)dctx, ldd, (bf16_t*)prob, (bf16_t*)ds, ldsr, nullptr, 0, 0, (bf16_t*)dq, lddq, nullptr, nullptr, drop_p,
               ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed};
    switch (hd) {
        case 64: return launch_lds_bw<64>(a, false, stream);
        case 128: return launch_lds_bw<128>(a, false, stream);
        default: return MI_ERR_UNSUPPORTED;
    }
}
