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
};

constexpr int SKEW_LD = 66;   // words per query row of the skew scratch: reads conflict-free, writes 2-way (free)

__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

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
//  * LDS images are lane-linear per DMA piece; the 16-B-chunk XOR swizzle (chunk ^ (row & (chunks-1))) is applied to the
//    source address and to every read (conflict-free ds_read_b128 / tr reads).
template <int HD, bool REL>
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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int ib = blockIdx.x * 128, i0 = ib + wave * 32;
    float* skew = skew_all + wave * 32 * SKEW_LD;
    const int T = p.T;                                        // queries per batch
    const int Tk = p.Tk > 0 ? p.Tk : p.T;                     // keys per batch (cross-attention / KV cache: Tk != T)
    const int coff = Tk - T;                                  // causal: query i sees keys <= i + coff
    const int len = p.lengths ? min(p.lengths[b], Tk) : Tk;
    const int kend = p.causal ? min(len, ib + 128 + coff) : len;
    const int nkt = (kend + 31) / 32;
    const int rb0 = T - 1 - ib - 31;                          // band base of wave 0 at step 0 (REL needs Tk == T)

    // ---- DMA issue helpers (this wave's PPW pieces of a 32-row tile)
    const int prow = lane / NCH, pc = lane % NCH;
    auto issue_rows = [&](const bf16_t* base, long ld, int row0, int rmax, char* dst) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = wave * PPW + q;
            const int row = piece * RPP + prow;               // 0..31
            const int lc = pc ^ (row & (NCH - 1));
            const int gr = min(max(row0 + row, 0), rmax);
            const bf16_t* src = base + (long)gr * ld + head * HD + lc * 8;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_step = [&](int it) {
        issue_rows(p.k + (p.kv_bstride ? (long)b * p.kv_bstride : (long)b * Tk * p.ldk), p.ldk, 32 * it, Tk - 1, sK + (it & 1) * TILEB);
        issue_rows(p.vt + (p.kv_bstride ? (long)b * p.kv_bstride : (long)b * Tk * p.ldvt), p.ldvt, 32 * it, Tk - 1, sV + (it & 1) * TILEB);   // p.vt = V [key][hd]
        if (REL) issue_rows(p.pos, p.ldp, rb0 + 32 * it + 32, 2 * T - 2, sP + ((it + PRING * 4) % PRING) * TILEB);
    };

    // ---- prologue: Q fragments (+u / +v), step 0 tiles, position blocks -4..-1
    const int iq = min(i0 + r, T - 1);
    bf16x8 qu[KS], qv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = head * HD + ks * 16 + h2 * 8;
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(p.q + ((long)b * T + iq) * p.ldq + c);
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
    if (REL) {
#pragma unroll
        for (int k = -4; k < 0; ++k)
            issue_rows(p.pos, p.ldp, rb0 + 32 * k + 32, 2 * T - 2, sP + ((k + PRING * 4) % PRING) * TILEB);
    }
    issue_step(0);

    f32x16 O[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] = 0.f;
    float m = -1e30f, l = 0.f;
    f32x16 Gc;                                               // carried G tile (lower half of this step's band)
#pragma unroll
    for (int e = 0; e < 16; ++e) Gc[e] = 0.f;

    auto gtile = [&](int blk) {                              // G^T tile of position block `blk` against (q+v)
        f32x16 G;
#pragma unroll
        for (int e = 0; e < 16; ++e) G[e] = 0.f;
        const char* pb = sP + ((blk + PRING * 4) % PRING) * TILEB + r * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pb + (((ks * 2 + h2) ^ (r & (NCH - 1))) << 4));
            G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[ks], G, 0, 0, 0);
        }
        return G;
    };

    for (int it = 0; it < nkt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (it + 1 < nkt) issue_step(it + 1);
        const int j0 = 32 * it;
        // ---- S^T = K_tile · (Q+u)^T
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
        {
            const char* kb = sK + (it & 1) * TILEB + r * ROWB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + (((ks * 2 + h2) ^ (r & (NCH - 1))) << 4));
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[ks], S, 0, 0, 0);
            }
        }
        if (REL) {
            // wave w's band at step it = position blocks (it - w - 1) [lower, carried] and (it - w) [upper, new]
            if (it == 0) Gc = gtile(it - wave - 1);
            const f32x16 Gn = gtile(it - wave);
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
        // ---- scale, mask, online softmax (lane = query)
        float mx = -1e30f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jj = j0 + crow(e, h2);
            const bool dead = (jj >= len) || (p.causal && jj > i0 + r + coff);
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
        bf16x8 pb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[s][j] = f2bf(S[8 * s + j]);
        // ---- O^T += V^T · P^T ; V^T fragments by transposed LDS reads of the [key][hd] tile
        {
            const char* vb = sV + (it & 1) * TILEB;
            const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                const int col = t * 32 + (g & 1) * 16 + 4 * p4;             // hd column of this lane's 4 elements
                const int lc = col >> 3, within = (col & 7) * 2;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    s16x4 lo, hi;
                    {
                        const int krow = 16 * s + 4 * (g >> 1) + q4;
                        const char* a = vb + krow * ROWB + ((lc ^ (krow & (NCH - 1))) << 4) + within;
                        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
                    }
                    {
                        const int krow = 16 * s + 8 + 4 * (g >> 1) + q4;
                        const char* a = vb + krow * ROWB + ((lc ^ (krow & (NCH - 1))) << 4) + within;
                        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
                    }
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pb[s], O[t], 0, 0, 0);
                }
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
                bf16x4 o = {f2bf(O[t][4 * g4 + 0] * inv), f2bf(O[t][4 * g4 + 1] * inv),
                            f2bf(O[t][4 * g4 + 2] * inv), f2bf(O[t][4 * g4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + c) = o;
            }
    }
}

template <int HD>
int launch_lds(const AttnArgs& a, bool rel, hipStream_t stream) {
    dim3 grid(cdiv(a.T, 128), a.H, a.B), block(256);
    const size_t lds = (size_t)(2 + 2 + 6) * 32 * HD * 2 + 4 * 32 * SKEW_LD * sizeof(float);
    if (rel) hipLaunchKernelGGL((attn_lds_kernel<HD, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_lds_kernel<HD, false>), grid, block, lds, stream, a);
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
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 4)) return MI_ERR_ARG;
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
