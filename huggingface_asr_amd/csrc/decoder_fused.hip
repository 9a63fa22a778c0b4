// GPT-2 decoder token step, fused form (round 5): THREE launches per layer instead of eight.
//
// The launch-per-op step (decoder_step.hip) is a chain of 66 dependent launches of <= 8 rows, each ~4.5 us of launch floor for ~2 us of work (DESIGN.md "Decoder / joint
// decoding"); a persistent one-launch step lost to its grid barriers.  This form cuts the chain without any grid barrier by moving every reduction that spans workgroups
// into the NEXT launch's prologue:
//   A (row m, head h):  x -> ln_1 -> q, k, v of head h (K/V cache append) -> causal attention over the cache -> ctx_h · Wo[:, h]^T  = partial[h][m][:]   (d-wide, fp32)
//   B (row m, head h):  x += bo + sum_h partial_A  -> ln_cross -> q of head h -> cross-attention over the cached encoder K/V -> ctx_h · Wco[:, h]^T = partial[h][m][:]
//   C (row m, 64 hidden columns j):      x += bco + sum_h partial_B -> ln_2 -> gelu_new(x · Wfc[j]^T + b) -> h_j · Wpr[:, j]^T = partial[j][m][:]
//   (next layer's A, or the final launch F: x += bpr + sum_j partial_C -> ...)
// The residual stream is therefore kept "pending": x_cur = x_base + bias + sum_p partial[p], summed in a fixed order (partials split over the four waves in index order,
// wave sums added in wave order) by every consumer block for the row(s) it needs; one designated block per row stores x_cur as the next base (ping-pong buffers).
// Rounding points are the launch-per-op step's (bf16 operands of every linear, bf16 q / k / v / context / hidden, fp32 residual stream, soft-max and LayerNorm); only the
// order of the fp32 sums differs (per-head / per-slice partial dot products), so the two forms agree to fp32 rounding of bf16-sized terms, not bit for bit.
// Everything a block reads that does not depend on an earlier phase of the same launch — the cache's K / V rows, its weight rows — is requested before the prologue's
// dependent loads are consumed, so a launch is ~two memory round trips deep.
//
// Eligible: one new token per row (U = 1), M = B <= 8 rows, head size 64, d <= 512 a multiple of 64; everything else runs the launch-per-op step.
// Replaces per token what the reference runs through transformers' GPT2Model.forward with a KV cache (multi_head_gpt2.py:80-170, tf gpt2 modeling :262-310).
#include "common.hpp"
#include "../../include/hfasr_hip.h"

namespace {

constexpr int FD_MAXM = 8, FD_MAXD = 512, FD_HD = 64;

struct Pending {                 // x_cur[m] = base[m] + pbias + sum_{p < P} part[p][m]      (P = 0: x_cur = base, nothing stored)
    const float* base; const float* part; const float* pbias; int P; float* next;      // part: (P, M, d) fp32; next: (M, d) — the designated block stores x_cur there
    // the first launch of a token step embeds the tokens itself (ids != null; base unused): x_cur[m] = wte[ids[m]] * emb_scale + pos_row — mi_embed_tokens' arithmetic —
    // and the designated block stores it (the cross-attention launch of layer 0 reads it back as its base)
    const long* ids; const float* wte; const float* pos_row; float emb_scale; int V;
};

// 8 consecutive elements k0 = 8 * lane of row m, for every wave of the block (all four take part: the partials are split over the waves, wave sums go through `wsum`).
// row_issue requests the loads (the caller then requests whatever else it can — weights, cache rows — so that one round trip serves them all), row_finish sums, exchanges
// and leaves x_cur's 8 elements in xv.  row_finish contains one __syncthreads (skipped when P == 0: uniform).
struct RowLoads { f32x4 b0, b1, c0, c1, v0[8], v1[8]; };
__device__ __forceinline__ void row_issue(const Pending& pd, int m, int M, int d, int lane, int wave, RowLoads& r) {
    const int k0 = lane * 8;
    const int ko = k0 < d ? k0 : 0;
    if (pd.ids) {
        long id = pd.ids[m];
        id = id < 0 ? 0 : (id >= pd.V ? pd.V - 1 : id);
        const float* er = pd.wte + id * d + ko;
        r.b0 = *reinterpret_cast<const f32x4*>(er); r.b1 = *reinterpret_cast<const f32x4*>(er + 4);
        r.c0 = *reinterpret_cast<const f32x4*>(pd.pos_row + ko); r.c1 = *reinterpret_cast<const f32x4*>(pd.pos_row + ko + 4);
        return;
    }
    const float* br = pd.base + (long)m * d + ko;
    r.b0 = *reinterpret_cast<const f32x4*>(br); r.b1 = *reinterpret_cast<const f32x4*>(br + 4);
    if (pd.P > 0) {
        const int per = (pd.P + 3) >> 2;
        const int p0 = wave * per;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                        // unconditional, clamped (a load under a condition is its own basic block and costs a full wait at the join)
            const int p = min(p0 + i, pd.P - 1);
            const float* pr = pd.part + ((long)p * M + m) * d + ko;
            r.v0[i] = *reinterpret_cast<const f32x4*>(pr); r.v1[i] = *reinterpret_cast<const f32x4*>(pr + 4);
        }
        r.c0 = *reinterpret_cast<const f32x4*>(pd.pbias + ko); r.c1 = *reinterpret_cast<const f32x4*>(pd.pbias + ko + 4);
    }
}
__device__ __forceinline__ void row_finish(const Pending& pd, const RowLoads& r, int m, int d, float (*wsum)[FD_MAXD], int lane, int wave, bool store, float (&xv)[8]) {
    const int k0 = lane * 8;
    const bool on = k0 < d;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 b0 = on ? r.b0 : z, b1 = on ? r.b1 : z;
    if (pd.ids) {
        b0 = b0 * pd.emb_scale + r.c0; b1 = b1 * pd.emb_scale + r.c1;
        if (!on) { b0 = z; b1 = z; }
        if (store && wave == 0 && on) { *reinterpret_cast<f32x4*>(pd.next + (long)m * d + k0) = b0; *reinterpret_cast<f32x4*>(pd.next + (long)m * d + k0 + 4) = b1; }
    } else if (pd.P > 0) {
        const int per = (pd.P + 3) >> 2;
        const int p0 = wave * per, p1 = min(pd.P, p0 + per);
        f32x4 a0 = z, a1 = z;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (p0 + i < p1) { a0 += r.v0[i]; a1 += r.v1[i]; }
        if (on) { *reinterpret_cast<f32x4*>(&wsum[wave][k0]) = a0; *reinterpret_cast<f32x4*>(&wsum[wave][k0 + 4]) = a1; }
        __syncthreads();
        if (on) {
            b0 += r.c0; b1 += r.c1;
#pragma unroll
            for (int w = 0; w < 4; ++w) { b0 += *reinterpret_cast<const f32x4*>(&wsum[w][k0]); b1 += *reinterpret_cast<const f32x4*>(&wsum[w][k0 + 4]); }
            if (store && wave == 0) { *reinterpret_cast<f32x4*>(pd.next + (long)m * d + k0) = b0; *reinterpret_cast<f32x4*>(pd.next + (long)m * d + k0 + 4) = b1; }
        }
    }
    xv[0] = b0.x; xv[1] = b0.y; xv[2] = b0.z; xv[3] = b0.w; xv[4] = b1.x; xv[5] = b1.y; xv[6] = b1.z; xv[7] = b1.w;
}

// LayerNorm of the row a wave holds as 8 elements per lane -> bf16 (the operand precision of the linear that follows)
__device__ __forceinline__ bf16x8 ln_row(const float (&xv)[8], const float* g, const float* b, float eps, int d, int lane) {
    const int k0 = lane * 8;
    const bool on = k0 < d;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += on ? xv[j] : 0.f;
    const float mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float a = on ? xv[j] - mean : 0.f; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
    bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + k0), g1 = *reinterpret_cast<const f32x4*>(g + k0 + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(b + k0), c1 = *reinterpret_cast<const f32x4*>(b + k0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf((xv[j] - mean) * rstd * gg[j] + cc[j]);
    }
    return o;
}

__device__ __forceinline__ float dot8(const bf16x8& w8, const bf16x8& x8, float a) {
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 0, 1), __builtin_shufflevector(x8, x8, 0, 1), a, false);
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 2, 3), __builtin_shufflevector(x8, x8, 2, 3), a, false);
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 4, 5), __builtin_shufflevector(x8, x8, 4, 5), a, false);
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 6, 7), __builtin_shufflevector(x8, x8, 6, 7), a, false);
    return a;
}

// 16 output columns of a linear for the ONE row the wave holds (x8: its 8 elements per lane): rows `wrow(i)` of W (ld = d) are requested by cols16_issue — before the
// prologue's barrier, so that they travel with the residual rows — and consumed by cols16_compute, which leaves result i in lane i
template <typename RowFn>
__device__ __forceinline__ void cols16_issue(const bf16_t* W, int d, RowFn wrow, int lane, bf16x8 (&wv)[16]) {
    const int k0 = lane * 8;
    const bool on = k0 < d;
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = *reinterpret_cast<const bf16x8*>(W + (long)wrow(i) * d + (on ? k0 : 0));
}
__device__ __forceinline__ float cols16_compute(const bf16x8 (&wv)[16], int d, const bf16x8& x8, int lane) {
    const bool on = lane * 8 < d;
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = wave_sum(on ? dot8(wv[i], x8, 0.f) : 0.f);
        mine = lane == i ? v : mine;
    }
    return mine;
}

// attention of one (row, head) over `nkeys` keys (64-wide heads): 8 lanes per key, the four waves split the keys, fp32 soft-max — the body of decode_attn_kernel.
// Batch 0 (keys < 256) comes in registers (k8 / v8: requested by the caller before its prologue); key index `xkey` (>= 0: the token being appended) takes xk / xv instead
// of the cache row.  Leaves the normalised context (64 values, bf16-rounded) in ctxs.  Contains __syncthreads.
struct AttnSrc { const bf16_t* kb; const bf16_t* vb; long ldkv; };     // per-lane bases (row 0 of this batch and head, column chunk c included)
__device__ __forceinline__ void attn_issue(const AttnSrc& s, int key0, int g, int nload, bf16x8 (&k8)[8], bf16x8 (&v8)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int key = key0 + i * 32 + g;
        const long off = (long)(key < nload ? key : 0) * s.ldkv;
        k8[i] = *reinterpret_cast<const bf16x8*>(s.kb + off);
        v8[i] = *reinterpret_cast<const bf16x8*>(s.vb + off);
    }
}
__device__ __forceinline__ void attn_rows(const AttnSrc& s, bf16x8 (&k8)[8], bf16x8 (&v8)[8], const bf16x8& q8, int nkeys, int nload, int xkey, const bf16x8& xk, const bf16x8& xv,
                                          float scale, float* red, float* rs, float* rm, float* ctxs, int lane, int wave) {
    const int g = lane >> 3, c = lane & 7;
    float mrun = -INFINITY, lsum = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int key0 = wave * 8; key0 < nkeys; key0 += 256) {               // wave-uniform
        if (key0 >= 256) attn_issue(s, key0, g, nload, k8, v8);
        float sv[8], bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = key0 + i * 32 + g;
            if (key == xkey) { k8[i] = xk; v8[i] = xv; }
            float sd = dot8(k8[i], q8, 0.f);
            sd += dpp_f32<0xB1, 0xF>(0.f, sd);
            sd += dpp_f32<0x4E, 0xF>(0.f, sd);
            sd += dpp_f32<0x141, 0xF>(0.f, sd);
            sv[i] = key < nkeys ? sd * scale : -INFINITY;
            bm = fmaxf(bm, sv[i]);
        }
        bm = wave_max(bm);
        const float mnew = fmaxf(mrun, bm);
        const float f = mrun == -INFINITY ? 0.f : __expf(mrun - mnew);
        lsum *= f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= f;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pr = sv[i] == -INFINITY ? 0.f : __expf(sv[i] - mnew);
            lsum += pr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(pr, bf2f(v8[i][j]), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(wave * 8 + g) * FD_HD + c * 8 + j] = acc[j];
    if (c == 0) rs[wave * 8 + g] = lsum;
    if (lane == 0) rm[wave] = mrun;
    __syncthreads();
    if (wave == 0) {
        float gm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) gm = fmaxf(gm, rm[w]);
        float tot = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = rm[w] == -INFINITY ? 0.f : __expf(rm[w] - gm);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                tot = fmaf(rs[w * 8 + k], f, tot);
                o = fmaf(red[(w * 8 + k) * FD_HD + lane], f, o);
            }
        }
        ctxs[lane] = bf2f(f2bf(o * (tot > 0.f ? 1.f / tot : 0.f)));
    }
    __syncthreads();
}

// partial output projection: out[n] = sum_c ctx[c] * W[n][col0 + c] for n < d (8 lanes per weight row, 8 rows per instruction, the waves split the rows) -> po[n] in LDS
__device__ __forceinline__ void proj_issue(const bf16_t* W, long ldw, int col0, int d, int lane, int wave, bf16x8 (&wv)[16]) {
    const int r8 = lane >> 3, c = lane & 7;
    const int rows = d >> 2;                                  // rows per wave
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = wave * rows + min(i * 8 + r8, rows - 1);
        wv[i] = *reinterpret_cast<const bf16x8*>(W + (long)n * ldw + col0 + c * 8);
    }
}
__device__ __forceinline__ void proj_rows(const bf16x8 (&wv)[16], const float* ctxs, float* po, int d, int lane, int wave) {
    const int r8 = lane >> 3, c = lane & 7;
    const int rows = d >> 2;
    bf16x8 c8;
#pragma unroll
    for (int j = 0; j < 8; ++j) c8[j] = f2bf(ctxs[c * 8 + j]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = dot8(wv[i], c8, 0.f);
        v += dpp_f32<0xB1, 0xF>(0.f, v);
        v += dpp_f32<0x4E, 0xF>(0.f, v);
        v += dpp_f32<0x141, 0xF>(0.f, v);
        if (c == 0 && i * 8 + r8 < rows) po[wave * rows + i * 8 + r8] = v;
    }
}

struct FusedArgs {
    Pending pd;
    const float* ln_g; const float* ln_b; float eps;
    const bf16_t* Win; const float* bin;           // A: Wqkv (3d, d) / bqkv;  B: Wq (d, d) / bq;  C: Wfc (4d, d) / bfc
    const bf16_t* Wout;                            // A: Wo (d, d);  B: Wco (d, d);  C: Wpr (d, 4d)
    float* part_out;                               // (P_out, M, d)
    bf16_t* kc; bf16_t* vc; int past, Lmax;        // A: caches (B, Lmax, d)
    const bf16_t* ckv; int T_enc; const int* enc_len;   // B: (B * T_enc, 2d) [K | V]
    int M, d, H; float scale;
};

// ---- A: self-attention block of one (row, head)
__global__ __launch_bounds__(256) void fused_self_kernel(FusedArgs p) {
    __shared__ __attribute__((aligned(16))) float wsum[4][FD_MAXD];
    __shared__ float qkvs[3 * FD_HD], red[4 * 8 * FD_HD], rs[32], rm[4], ctxs[FD_HD];
    __shared__ __attribute__((aligned(16))) float po[FD_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x / p.H, h = blockIdx.x - m * p.H, d = p.d;
    const int g = lane >> 3, c = lane & 7;
    // one round trip for everything that does not depend on this launch's own results: the row of the stream with its partials, the first 16 of this wave's 48
    // weight rows, the cache rows of the keys already there
    RowLoads rl;
    row_issue(p.pd, m, p.M, d, lane, wave, rl);
    auto wrow = [&](int c0) { return [=](int i) { const int cc = c0 + i; return (cc >> 6) * d + h * FD_HD + (cc & 63); }; };
    bf16x8 wv[16];
    cols16_issue(p.Win, d, wrow(wave * 48), lane, wv);
    float bias3[3];                                           // the three rounds' biases of this lane's column, requested here: a load under `lane < 16` inside a round is a basic
#pragma unroll                                                // block of its own and its wait (vmcnt(0)) also drains the next round's weight rows
    for (int rnd = 0; rnd < 3; ++rnd) {
        const int cc = wave * 48 + rnd * 16 + (lane & 15);
        bias3[rnd] = p.bin[(cc >> 6) * d + h * FD_HD + (cc & 63)];
    }
    // (all 48 weight rows up front — 370 registers, no spills — measured SLOWER, 13.5 -> 14.1 ms greedy same box: more loads queued in front of the ones the prologue waits for)
    AttnSrc src{p.kc + (long)m * p.Lmax * d + h * FD_HD + c * 8, p.vc + (long)m * p.Lmax * d + h * FD_HD + c * 8, (long)d};
    bf16x8 k8[8], v8[8];
    attn_issue(src, wave * 8, g, p.past, k8, v8);
    float xv[8];
    row_finish(p.pd, rl, m, d, wsum, lane, wave, h == 0, xv);
    const bf16x8 x8 = ln_row(xv, p.ln_g, p.ln_b, p.eps, d, lane);
    // q, k, v of this head: 192 columns, 48 per wave in three rounds of 16 (the next round's rows requested before this round's sums); column cc -> row (cc / 64) * d + h * 64 + cc % 64 of Wqkv
#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
        const int c0 = wave * 48 + rnd * 16;
        bf16x8 wn[16];
        if (rnd < 2) cols16_issue(p.Win, d, wrow(c0 + 16), lane, wn);
        const float v = cols16_compute(wv, d, x8, lane);
        if (lane < 16) qkvs[c0 + lane] = bf2f(f2bf(v + bias3[rnd]));
        if (rnd < 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = wn[i];
        }
    }
    bf16x8 wo[16];
    proj_issue(p.Wout, d, h * FD_HD, d, lane, wave, wo);      // under the attention
    __syncthreads();
    if (threadIdx.x < 128) {                                   // the new token's K / V row of this head -> caches (position `past`)
        const int which = threadIdx.x >> 6, e = threadIdx.x & 63;
        (which ? p.vc : p.kc)[((long)m * p.Lmax + p.past) * d + h * FD_HD + e] = f2bf(qkvs[FD_HD * (1 + which) + e]);
    }
    bf16x8 q8, nk, nv;
#pragma unroll
    for (int j = 0; j < 8; ++j) { q8[j] = f2bf(qkvs[c * 8 + j]); nk[j] = f2bf(qkvs[FD_HD + c * 8 + j]); nv[j] = f2bf(qkvs[2 * FD_HD + c * 8 + j]); }
    attn_rows(src, k8, v8, q8, p.past + 1, p.past, p.past, nk, nv, p.scale, red, rs, rm, ctxs, lane, wave);
    proj_rows(wo, ctxs, po, d, lane, wave);
    __syncthreads();
    float* out = p.part_out + ((long)h * p.M + m) * d;
    for (int i = threadIdx.x * 2; i < d; i += 512) *reinterpret_cast<f32x2*>(out + i) = *reinterpret_cast<const f32x2*>(po + i);
}

// ---- B: cross-attention block of one (row, head)
__global__ __launch_bounds__(256) void fused_cross_kernel(FusedArgs p) {
    __shared__ __attribute__((aligned(16))) float wsum[4][FD_MAXD];
    __shared__ float qs[FD_HD], red[4 * 8 * FD_HD], rs[32], rm[4], ctxs[FD_HD];
    __shared__ __attribute__((aligned(16))) float po[FD_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x / p.H, h = blockIdx.x - m * p.H, d = p.d;
    const int g = lane >> 3, c = lane & 7;
    const int nkeys = p.enc_len ? min(p.enc_len[m], p.T_enc) : p.T_enc;
    RowLoads rl;
    row_issue(p.pd, m, p.M, d, lane, wave, rl);
    const int c0 = wave * 16;
    bf16x8 wv[16];
    cols16_issue(p.Win, d, [&](int i) { return h * FD_HD + c0 + i; }, lane, wv);
    const float bq = p.bin[h * FD_HD + c0 + (lane & 15)];     // (requested with everything else: see fused_self_kernel)
    AttnSrc src{p.ckv + (long)m * p.T_enc * 2 * d + h * FD_HD + c * 8, p.ckv + (long)m * p.T_enc * 2 * d + d + h * FD_HD + c * 8, (long)2 * d};
    bf16x8 k8[8], v8[8];
    attn_issue(src, wave * 8, g, nkeys, k8, v8);
    float xv[8];
    row_finish(p.pd, rl, m, d, wsum, lane, wave, h == 0, xv);
    const bf16x8 x8 = ln_row(xv, p.ln_g, p.ln_b, p.eps, d, lane);
    {
        const float v = cols16_compute(wv, d, x8, lane);
        if (lane < 16) qs[c0 + lane] = bf2f(f2bf(v + bq));
    }
    bf16x8 wo[16];
    proj_issue(p.Wout, d, h * FD_HD, d, lane, wave, wo);
    __syncthreads();
    bf16x8 q8;
#pragma unroll
    for (int j = 0; j < 8; ++j) q8[j] = f2bf(qs[c * 8 + j]);
    attn_rows(src, k8, v8, q8, nkeys, nkeys, -1, q8, q8, p.scale, red, rs, rm, ctxs, lane, wave);
    proj_rows(wo, ctxs, po, d, lane, wave);
    __syncthreads();
    float* out = p.part_out + ((long)h * p.M + m) * d;
    for (int i = threadIdx.x * 2; i < d; i += 512) *reinterpret_cast<f32x2*>(out + i) = *reinterpret_cast<const f32x2*>(po + i);
}

// ---- C: 64 hidden columns of the MLP for one row (grid: J x M; the first form — every row in one block of 32 — spent its time in M x 16 dependent wave sums per wave
// on an eighth of the chip: 20.7 us per launch at W = 5)
__global__ __launch_bounds__(256) void fused_mlp_kernel(FusedArgs p) {
    __shared__ __attribute__((aligned(16))) float wsum[4][FD_MAXD];
    __shared__ float hs[FD_HD];
    __shared__ __attribute__((aligned(16))) float po[FD_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = p.d >> 4, m = blockIdx.x / J, j = blockIdx.x - m * J, d = p.d;      // (m, j) with j fastest: the blocks that stream the same weight slice share blockIdx % 8, i.e. an XCD's L2
    RowLoads rl;
    row_issue(p.pd, m, p.M, d, lane, wave, rl);
    bf16x8 wf[16];
    cols16_issue(p.Win, d, [&](int i) { return j * FD_HD + wave * 16 + i; }, lane, wf);
    bf16x8 wp[16];
    proj_issue(p.Wout, 4 * d, j * FD_HD, d, lane, wave, wp);
    const float bfc = p.bin[j * FD_HD + wave * 16 + (lane & 15)];
    float xv[8];
    row_finish(p.pd, rl, m, d, wsum, lane, wave, j == 0, xv);
    const bf16x8 x8 = ln_row(xv, p.ln_g, p.ln_b, p.eps, d, lane);
    const float v = cols16_compute(wf, d, x8, lane);
    if (lane < 16) hs[wave * 16 + lane] = bf2f(f2bf(gelu_tanh(v + bfc)));
    __syncthreads();
    proj_rows(wp, hs, po, d, lane, wave);
    __syncthreads();
    float* out = p.part_out + ((long)j * p.M + m) * d;
    for (int i = threadIdx.x * 2; i < d; i += 512) *reinterpret_cast<f32x2*>(out + i) = *reinterpret_cast<const f32x2*>(po + i);
}

// ---- F: the stream after the last layer -> ln_f -> bf16 rows for the lm head
__global__ __launch_bounds__(256) void fused_final_kernel(FusedArgs p, bf16_t* hid) {
    __shared__ __attribute__((aligned(16))) float wsum[4][FD_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x, d = p.d;
    RowLoads rl;
    row_issue(p.pd, m, p.M, d, lane, wave, rl);
    float xv[8];
    row_finish(p.pd, rl, m, d, wsum, lane, wave, false, xv);
    const bf16x8 x8 = ln_row(xv, p.ln_g, p.ln_b, p.eps, d, lane);
    if (wave == 0 && lane * 8 < d) *reinterpret_cast<bf16x8*>(hid + (long)m * d + lane * 8) = x8;
}

}  // namespace

bool gpt2_step_fused_ok(const mi_gpt2_config& c, int B, int U) {
    return U == 1 && B >= 1 && B <= FD_MAXM && c.H >= 1 && c.d == c.H * FD_HD && c.d <= FD_MAXD && c.d >= 64;
}

// floats of workspace the fused form needs beyond the launch-per-op step's: two bases (M, d) and two partial buffers (Pmax, M, d), Pmax = max(H, 4 d / 64)
size_t gpt2_step_fused_floats(const mi_gpt2_config& c, int M) {
    const size_t pmax = (size_t)(c.H > c.d / 16 ? c.H : c.d / 16);
    return 2 * (size_t)M * c.d + 2 * pmax * M * c.d;
}

// ids: (M) int64 new tokens, embedded by the first launch (wte = weights[0] scaled by emb_scale, + position row `past` of weights[1]); fws: gpt2_step_fused_floats(c, M)
// floats; hid: (M, d) bf16 out = ln_f(stream) for the lm head.  Weight table as mi_gpt2_step.
int gpt2_step_fused(const mi_gpt2_config& c, const void* const* weights, const long* ids, float emb_scale, int M, int past, int Lmax, void* const* kcache, void* const* vcache,
                    const void* const* cross_kv, int T_enc, const int* enc_len, float* fws, bf16_t* hid, hipStream_t st) {
    const int d = c.d, H = c.H, J = d / 16;
    const size_t pmax = (size_t)(H > J ? H : J);
    float* xb[2] = {fws, fws + (size_t)M * d};
    float* pb[2] = {fws + 2 * (size_t)M * d, fws + 2 * (size_t)M * d + pmax * M * d};
    auto Lw = [&](int l, int i) { return weights[5 + l * 18 + i]; };
    auto Lf = [&](int l, int i) { return (const float*)weights[5 + l * 18 + i]; };
    Pending pd{nullptr, nullptr, nullptr, 0, nullptr, ids, (const float*)weights[0], (const float*)weights[1] + (long)past * d, emb_scale, c.V};
    int xi = 0, pi = 0;
    auto advance = [&](float* part_written, const float* bias, int P) {           // the launch just enqueued wrote `part_written`; if it had something to fold in (partials, or the embedding), it also stored pd.next
        if (pd.P > 0 || pd.ids) pd.base = pd.next;
        pd.ids = nullptr;
        pd.part = part_written; pd.pbias = bias; pd.P = P;
        pd.next = xb[xi]; xi ^= 1;
    };
    pd.next = xb[xi]; xi ^= 1;
    const float scale = 1.0f / sqrtf((float)FD_HD);
    for (int l = 0; l < c.L; ++l) {
        FusedArgs a{};
        a.M = M; a.d = d; a.H = H; a.scale = scale; a.eps = c.eps;
        a.pd = pd; a.ln_g = Lf(l, 0); a.ln_b = Lf(l, 1); a.Win = (const bf16_t*)Lw(l, 2); a.bin = Lf(l, 3); a.Wout = (const bf16_t*)Lw(l, 4);
        a.part_out = pb[pi]; a.kc = (bf16_t*)kcache[l]; a.vc = (bf16_t*)vcache[l]; a.past = past; a.Lmax = Lmax;
        hipLaunchKernelGGL(fused_self_kernel, dim3(M * H), dim3(256), 0, st, a);
        advance(pb[pi], Lf(l, 5), H); pi ^= 1;
        FusedArgs b{};
        b.M = M; b.d = d; b.H = H; b.scale = scale; b.eps = c.eps;
        b.pd = pd; b.ln_g = Lf(l, 6); b.ln_b = Lf(l, 7); b.Win = (const bf16_t*)Lw(l, 8); b.bin = Lf(l, 9); b.Wout = (const bf16_t*)Lw(l, 10);
        b.part_out = pb[pi]; b.ckv = (const bf16_t*)cross_kv[l]; b.T_enc = T_enc; b.enc_len = enc_len;
        hipLaunchKernelGGL(fused_cross_kernel, dim3(M * H), dim3(256), 0, st, b);
        advance(pb[pi], Lf(l, 11), H); pi ^= 1;
        FusedArgs m{};
        m.M = M; m.d = d; m.H = H; m.eps = c.eps;
        m.pd = pd; m.ln_g = Lf(l, 12); m.ln_b = Lf(l, 13); m.Win = (const bf16_t*)Lw(l, 14); m.bin = Lf(l, 15); m.Wout = (const bf16_t*)Lw(l, 16);
        m.part_out = pb[pi];
        hipLaunchKernelGGL(fused_mlp_kernel, dim3(J * M), dim3(256), 0, st, m);
        advance(pb[pi], Lf(l, 17), J); pi ^= 1;
    }
    FusedArgs f{};
    f.M = M; f.d = d; f.H = H; f.eps = c.eps; f.pd = pd; f.ln_g = (const float*)weights[2]; f.ln_b = (const float*)weights[3];
    hipLaunchKernelGGL(fused_final_kernel, dim3(M), dim3(256), 0, st, f, hid);
    return MI_OK;
}
