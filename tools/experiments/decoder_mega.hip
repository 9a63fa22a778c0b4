// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3).  Kept as the record of a measured and rejected lever; DESIGN.md "Decoder / joint decoding" has the numbers.
// Result on MI355X, config-5 decoder (8 x 512, W = 5): 0.72 ms per token step against 0.50 ms for the chain of 66 launches it was meant to replace.  Block 0's clock
// readings: a grid barrier costs 3.0-3.2 us in situ (release = L2 write-back, flag store, polls, acquire = L2 invalidate: several trips to the memory side, because the eight XCDs' L2s
// are not coherent with each other), and the first load of a phase's input after the barrier is another ~2 us trip (the invalidate has emptied L2) — LayerNorm staging 5.2 us for two
// rows per wave, a 4-column epilogue (5 DPP wave sums each) 2-2.5 us.  A phase therefore costs what a dependent kernel launch costs (~6-7 us); the launch boundary IS this
// cache maintenance.  Making it pay would need all blocks on ONE XCD with L2 as the coherence point, i.e. relying on block placement for correctness: not done.
//
// The GPT-2 decoder token step as ONE persistent launch (BASELINE config 5: streaming bs = 1 latency).
//
// mi_gpt2_step's chain of 8 launches per layer (csrc/decoder_step.hip; reference: transformers' GPT2Model.forward with a KV cache under
// src/models/decoders/multi_head_gpt2.py:80-170) is latency, not work: every dependent launch costs ~4.5 us on this part before its first instruction does anything
// useful, and the linears' weight rows — which do not depend on the activations — cannot be requested before their launch starts.  Here 40 blocks (one (row, head) attention task each at 5 beams x 8 heads) stay resident for
// the whole step and meet at grid barriers instead of kernel boundaries:
//   * barrier = per-block epoch words, published with an agent-scope release store by thread 0 after __syncthreads and polled by one wave with ONE load per round
//     (no read-modify-write contention): ~2 us for 32-40 blocks against 4.5 us for a kernel boundary (tools/barrier_bench.hip);
//   * every wave requests the weight rows of its NEXT linear before it goes to the barrier, so their HBM / MALL latency runs under the barrier and the LayerNorm;
//   * the attentions are wave-per-key-group GEMVs (8 or 16 lanes per key, fp32 soft-max, the four waves of a block split the keys and merge through LDS).
// Exit condition: a barrier gives up after MG_SPIN_LIMIT polls (or as soon as another block has), raises the error word of the workspace and the block returns — every
// wave reaches the end of the kernel whatever the other blocks do.  The caller checks the word (mi_gpt2_step_status).
#include "common.hpp"
#include "../../include/hfasr_hip.h"

namespace {

constexpr int MG_THREADS = 256, MG_WAVES = 4, MG_BLOCKS = 40, MG_MAXM = 8, MG_MAXL = 16, MG_MAXKEYS = 2048;
constexpr unsigned MG_SPIN_LIMIT = 400000;

struct MegaSync { unsigned flags[64]; unsigned epoch; unsigned error; unsigned pad[62]; };      // the first 512 bytes of the workspace; zero before the first call

struct MegaLayer {
    const float* ln1g; const float* ln1b; const bf16_t* wqkv; const float* bqkv; const bf16_t* wo; const float* bo;
    const float* lncg; const float* lncb; const bf16_t* wq; const float* bq; const bf16_t* wco; const float* bco;
    const float* ln2g; const float* ln2b; const bf16_t* wfc; const float* bfc; const bf16_t* wpr; const float* bpr;
    bf16_t* kc; bf16_t* vc; const bf16_t* ckv;
};

struct MegaArgs {
    MegaLayer L[MG_MAXL];
    const float* wte; const float* pos; const float* lnfg; const float* lnfb; const bf16_t* head;
    const long* ids; const int* enc_len;
    float* x; bf16_t* qkv; bf16_t* ctx; bf16_t* qq; bf16_t* m;
    float* logits; long ld_logits;
    MegaSync* sync;
    unsigned long long* stamps;          // optional: block 0 leaves a 100-MHz clock reading before and after every barrier (HFASR_STEP_STAMPS=1; tools/decode_profile.py)
    int B, U, past, Lmax, T_enc, d, H, nl, V;
    float eps, emb_scale, scale;
};

struct Lin {
    const bf16_t* W; const float* bias; int N, K, act;
    float* out32; long ldo32; const float* resid;      // out32[m][n] = (resid ? resid[m][n] : 0) + v
    bf16_t* out16; long ldo16;
    bf16_t* kc; bf16_t* vc; int dkv, U, past, Lmax;    // optional KV-cache append of columns [dkv, 3 dkv)
};

// ---- grid barrier: false = gave up (error word raised)
__device__ __forceinline__ bool grid_sync(MegaSync* s, unsigned target) {
    __syncthreads();                                                   // every wave's stores have reached L2 (the workgroup fence waits for them)
    __shared__ int ok;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(&s->flags[blockIdx.x], target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool good = true;
        for (;;) {
            const unsigned v = threadIdx.x < gridDim.x ? __hip_atomic_load(&s->flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all((int)(v - target) >= 0)) break;
            ++spins;
            if ((spins & 63) == 0 && __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { good = false; break; }
            if (spins > MG_SPIN_LIMIT) { __hip_atomic_store(&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = false; break; }
        }
        if (threadIdx.x == 0) ok = good ? 1 : 0;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // every wave: nothing it reads from here on may come from a line cached before the barrier
    return ok != 0;
}

// ---- linear phases: a wave owns the columns [gw cpw, (gw + 1) cpw) of the layer and works through them 16 / PARTS at a time; K <= 512 PARTS
struct LinRegs { bf16x8 wv[16]; float bias[16]; };       // what a wave holds of its next columns before it knows their input: weight pieces and biases

// Every load here is unconditional with a clamped address: a load under a condition becomes its own basic block, and the compiler waits for all outstanding loads where
// blocks join — sixteen round trips instead of one.  Columns past the wave's range read a valid row whose result is dropped, pieces past K read the row's start and meet zeros in x.
template <int PARTS>
__device__ __forceinline__ void lin_prefetch(const Lin& s, int col0, int ncols, LinRegs& r) {
    const int lane = threadIdx.x & 63;
    const float* bp = s.bias ? s.bias : reinterpret_cast<const float*>(s.W);          // no bias: some readable words, never used (the epilogue selects)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int c = PARTS == 1 ? u : u / PARTS, part = PARTS == 1 ? 0 : u % PARTS;
        const int k = part * 512 + lane * 8;
        const int col = col0 + c < s.N ? col0 + c : s.N - 1;
        r.wv[u] = *reinterpret_cast<const bf16x8*>(s.W + (long)col * s.K + (k < s.K ? k : 0));
    }
#pragma unroll
    for (int c = 0; c < 16 / PARTS; ++c) r.bias[c] = bp[col0 + c < s.N ? col0 + c : s.N - 1];
    (void)ncols;
}

__device__ __forceinline__ void lin_range(const Lin& s, int& col0, int& col1) {
    const int nw = gridDim.x * MG_WAVES, gw = blockIdx.x * MG_WAVES + (threadIdx.x >> 6);
    const int cpw = (s.N + nw - 1) / nw;
    col0 = gw * cpw;
    col1 = col0 + cpw < s.N ? col0 + cpw : s.N;
}

template <int PARTS>
__device__ __forceinline__ void lin_first(const Lin& s, LinRegs& wv) {                     // request the first chunk of this wave's columns (before the barrier)
    int c0, c1;
    lin_range(s, c0, c1);
    constexpr int COLS = 16 / PARTS;
    lin_prefetch<PARTS>(s, c0, c1 - c0 < COLS ? (c1 - c0 < 0 ? 0 : c1 - c0) : COLS, wv);
}

template <int PARTS>
__device__ __forceinline__ void lin_chunk(const Lin& s, int col0, int ncols, const LinRegs& r, const bf16_t* xs, int M) {
    const int lane = threadIdx.x & 63;
    constexpr int COLS = 16 / PARTS;
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float rsd[COLS];                                                   // residual values of this lane's row: requested now, used after the dot products
#pragma unroll
    for (int c = 0; c < COLS; ++c) {                                   // unconditional, clamped (see lin_prefetch); without a residual the words of the output itself, never used
        const float* rp = s.resid ? s.resid : s.out32;
        rsd[c] = rp ? rp[(long)(lane < M ? lane : 0) * s.ldo32 + (col0 + c < s.N ? col0 + c : s.N - 1)] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < COLS / 4; ++g) {
        if (g * 4 >= ncols) break;                                     // wave-uniform
        float acc[4][MG_MAXM];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MG_MAXM; ++m) acc[c][m] = 0.f;
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            const int k = part * 512 + lane * 8;
#pragma unroll
            for (int m = 0; m < MG_MAXM; ++m) {
                if (m >= M) continue;
                const bf16x8 x8 = k < s.K ? *reinterpret_cast<const bf16x8*>(xs + m * s.K + k) : z8;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bf16x8 w8 = r.wv[PARTS == 1 ? g * 4 + c : c * PARTS + part];
                    float a = acc[c][m];
                    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 0, 1), __builtin_shufflevector(x8, x8, 0, 1), a, false);
                    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 2, 3), __builtin_shufflevector(x8, x8, 2, 3), a, false);
                    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 4, 5), __builtin_shufflevector(x8, x8, 4, 5), a, false);
                    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w8, w8, 6, 7), __builtin_shufflevector(x8, x8, 6, 7), a, false);
                    acc[c][m] = a;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (g * 4 + c >= ncols) break;                             // wave-uniform
            const int n = col0 + g * 4 + c;
#pragma unroll
            for (int m = 0; m < MG_MAXM; ++m)
                if (m < M) acc[c][m] = wave_sum(acc[c][m]);
            if (lane < M) {
                float v = 0.f;
#pragma unroll
                for (int m = 0; m < MG_MAXM; ++m) if (m == lane) v = acc[c][m];
                v += s.bias ? r.bias[g * 4 + c] : 0.f;
                if (s.act == 2) v = gelu_tanh(v);
                if (s.out32) s.out32[(long)lane * s.ldo32 + n] = (s.resid ? rsd[g * 4 + c] : 0.f) + v;
                else {
                    const bf16_t o = f2bf(v);
                    s.out16[(long)lane * s.ldo16 + n] = o;
                    if (s.kc && n >= s.dkv) {
                        const int b = lane / s.U, u = lane - b * s.U;
                        const long row = ((long)b * s.Lmax + s.past + u) * s.dkv;
                        if (n < 2 * s.dkv) s.kc[row + n - s.dkv] = o; else s.vc[row + n - 2 * s.dkv] = o;
                    }
                }
            }
        }
    }
}

template <int PARTS>
__device__ __forceinline__ void lin_run(const Lin& s, LinRegs& wv, const bf16_t* xs, int M) {           // the first chunk is already in wv
    int c0, c1;
    lin_range(s, c0, c1);
    constexpr int COLS = 16 / PARTS;
    for (int c = c0; c < c1; c += COLS) {
        const int nc = c1 - c < COLS ? c1 - c : COLS;
        if (c != c0) lin_prefetch<PARTS>(s, c, nc, wv);
        lin_chunk<PARTS>(s, c, nc, wv, xs, M);
    }
}

// ---- LDS images of a linear's input (bf16, the operand precision of every linear on this path)
// LayerNorm of fp32 rows (K <= 512): rows m = wave, wave + 4.  `emb`: the rows are the token embeddings (layer 0), written to x by block 0 on the way
struct LnRegs { float g[8], b[8]; };
__device__ __forceinline__ void ln_prefetch(const float* g, const float* b, int K, LnRegs& r) {           // the affine pair does not depend on the input either
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int k = lane + 64 * i < K ? lane + 64 * i : 0; r.g[i] = g[k]; r.b[i] = b[k]; }      // clamped, unconditional
}
__device__ __forceinline__ void stage_ln(const MegaArgs& p, const float* x, long ldx, int M, int K, const LnRegs& ln, bf16_t* xs, bool emb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = wave; m < M; m += MG_WAVES) {
        float xv[8];
        float sum = 0.f;
        const float* xr = x + (long)m * ldx;
        const float* er = nullptr; const float* pr = nullptr;
        if (emb) {
            const long id = p.ids[m];
            er = p.wte + id * K;
            pr = p.pos + (long)(p.past + m % p.U) * K;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lane + 64 * i;
            const bool ok = k < K;
            const int kc = ok ? k : 0;
            float v = emb ? fmaf(er[kc], p.emb_scale, pr[kc]) : xr[kc];
            v = ok ? v : 0.f;
            xv[i] = v;
            sum += v;
        }
        if (emb && blockIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int k = lane + 64 * i; if (k < K) p.x[(long)m * K + k] = xv[i]; }
        }
        const float mean = wave_sum(sum) / K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int k = lane + 64 * i; const float a = k < K ? xv[i] - mean : 0.f; q += a * a; }
        const float rstd = rsqrtf(wave_sum(q) / K + p.eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int k = lane + 64 * i; if (k < K) xs[m * K + k] = f2bf((xv[i] - mean) * rstd * ln.g[i] + ln.b[i]); }
    }
}

__device__ __forceinline__ void stage_bf16(const bf16_t* src, long ld, int M, int K, bf16_t* xs) {
    const int k8 = K >> 3;
    for (int i = threadIdx.x; i < M * k8; i += MG_THREADS) {
        const int m = i / k8, c = i - m * k8;
        *reinterpret_cast<bf16x8*>(xs + m * K + c * 8) = *reinterpret_cast<const bf16x8*>(src + (long)m * ld + c * 8);
    }
}

// ---- attention of the new rows over cached keys: task = (row, head); the block's four waves split the keys, LPK = HD / 8 lanes share a key (16-B pieces of its K and V
// rows), a wave takes NB x KPI keys per batch with all 2 NB loads of a lane requested before the first is used (a lone wave has nobody to hide a round trip behind), running
// max / rescale across batches, key slots and waves merged through LDS.
template <int HD>
__device__ __forceinline__ void attn_phase(const bf16_t* q, long ldq, const bf16_t* Kb, const bf16_t* Vb, long ldkv, long bstride, int M, int U, int H, int past, bool self,
                                           const int* enc_len, int T_enc, float scale, bf16_t* ctx, long ldo, float* red, float* rs, float* rm) {
    constexpr int LPK = HD / 8, KPI = 64 / LPK, NB = 8, EPL = HD / 64;       // lanes per key, keys per wave and load round, rounds per batch, output dims per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPK, c = lane % LPK;
    for (int t = blockIdx.x; t < M * H; t += gridDim.x) {             // block-uniform
        const int m = t / H, h = t - m * H, b = m / U, u = m - b * U;
        const int nkeys = self ? past + u + 1 : (enc_len ? (enc_len[b] < T_enc ? enc_len[b] : T_enc) : T_enc);
        const bf16_t* kb = Kb + (long)b * bstride + h * HD + c * 8;
        const bf16_t* vb = Vb + (long)b * bstride + h * HD + c * 8;
        const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(q + (long)m * ldq + h * HD + c * 8);
        float mrun = -INFINITY, lsum = 0.f, acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int key0 = wave * KPI; key0 < nkeys; key0 += MG_WAVES * KPI * NB) {       // wave-uniform
            bf16x8 k8[NB], v8[NB];
            float sv[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int key = key0 + i * MG_WAVES * KPI + g;
                const long off = (long)(key < nkeys ? key : 0) * ldkv;
                k8[i] = *reinterpret_cast<const bf16x8*>(kb + off);
                v8[i] = *reinterpret_cast<const bf16x8*>(vb + off);
            }
            float bm = -INFINITY;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float sd = 0.f;
                sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 0, 1), __builtin_shufflevector(q8, q8, 0, 1), sd, false);
                sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 2, 3), __builtin_shufflevector(q8, q8, 2, 3), sd, false);
                sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 4, 5), __builtin_shufflevector(q8, q8, 4, 5), sd, false);
                sd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(k8[i], k8[i], 6, 7), __builtin_shufflevector(q8, q8, 6, 7), sd, false);
                sd += dpp_f32<0xB1, 0xF>(0.f, sd);                     // the key's LPK lanes: quad, half row (8 lanes) [, row (16 lanes)]
                sd += dpp_f32<0x4E, 0xF>(0.f, sd);
                sd += dpp_f32<0x141, 0xF>(0.f, sd);
                if (LPK == 16) sd += dpp_f32<0x140, 0xF>(0.f, sd);
                sv[i] = (key0 + i * MG_WAVES * KPI + g) < nkeys ? sd * scale : -INFINITY;
                bm = fmaxf(bm, sv[i]);
            }
            bm = wave_max(bm);                                         // finite: the batch's first key slot 0 is < nkeys
            const float mnew = fmaxf(mrun, bm);
            const float f = mrun == -INFINITY ? 0.f : __expf(mrun - mnew);
            lsum *= f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= f;
            mrun = mnew;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float pr = sv[i] == -INFINITY ? 0.f : __expf(sv[i] - mnew);
                lsum += pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(pr, bf2f(v8[i][j]), acc[j]);
            }
        }
        // key slots and waves -> LDS (a wave without keys has mrun = -inf and zeros)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(wave * KPI + g) * HD + c * 8 + j] = acc[j];
        if (c == 0) rs[wave * KPI + g] = lsum;
        if (lane == 0) rm[wave] = mrun;
        __syncthreads();
        if (wave == 0) {
            float gm = -INFINITY;
#pragma unroll
            for (int w = 0; w < MG_WAVES; ++w) gm = fmaxf(gm, rm[w]);
            float tot = 0.f, o[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
            for (int w = 0; w < MG_WAVES; ++w) {
                const float f = rm[w] == -INFINITY ? 0.f : __expf(rm[w] - gm);
#pragma unroll
                for (int k = 0; k < KPI; ++k) {
                    tot = fmaf(rs[w * KPI + k], f, tot);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[e] = fmaf(red[(w * KPI + k) * HD + lane * EPL + e], f, o[e]);
                }
            }
            const float inv = tot > 0.f ? 1.f / tot : 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) ctx[(long)m * ldo + h * HD + lane * EPL + e] = f2bf(o[e] * inv);
        }
        __syncthreads();
    }
}

#define MG_STAMP() do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[nstamp++] = wall_clock64(); } while (0)
#define MG_SYNC() do { MG_STAMP(); if (!grid_sync(p.sync, ++epoch)) return; MG_STAMP(); } while (0)

template <int HD>
__global__ __launch_bounds__(MG_THREADS) void gpt2_step_mega_kernel(MegaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[MG_MAXM * 2048];                              // 32 KiB: the current linear's input rows
    __shared__ float red[MG_WAVES * 64 * 8];                                                       // attention partials: [wave][key slot][dim]  (KPI HD = 512 floats per wave)
    __shared__ float rs[MG_WAVES * 8];
    __shared__ float rm[MG_WAVES];
    const int d = p.d, M = p.B * p.U;
    unsigned epoch = p.sync->epoch;                                    // every block reads it before block 0 can have passed the first barrier
    const unsigned epoch0 = epoch;
    int nstamp = 0;
    MG_STAMP();
    LinRegs wv;
    LnRegs ln;
    Lin s{};
    for (int l = 0; l < p.nl; ++l) {
        const MegaLayer w = p.L[l];
        // P1: ln_1 + fused QKV projection, K / V columns appended to the caches
        s = Lin{w.wqkv, w.bqkv, 3 * d, d, 0, nullptr, 0, nullptr, p.qkv, 3 * d, w.kc, w.vc, d, p.U, p.past, p.Lmax};
        if (l == 0) { lin_first<1>(s, wv); ln_prefetch(w.ln1g, w.ln1b, d, ln); }
        stage_ln(p, p.x, d, M, d, ln, xs, l == 0);
        __syncthreads();
        lin_run<1>(s, wv, xs, M);
        const Lin so{w.wo, w.bo, d, d, 0, p.x, d, p.x, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
        lin_first<1>(so, wv);                                          // c_proj's rows travel under the attention phase
        MG_SYNC();
        // P2: causal self-attention over the cache
        attn_phase<HD>(p.qkv, 3 * d, w.kc, w.vc, d, (long)p.Lmax * d, M, p.U, p.H, p.past, true, nullptr, 0, p.scale, p.ctx, d, red, rs, rm);
        MG_SYNC();
        // P3: x += ctx Wo^T + bo
        stage_bf16(p.ctx, d, M, d, xs);
        __syncthreads();
        lin_run<1>(so, wv, xs, M);
        const Lin sq{w.wq, w.bq, d, d, 0, nullptr, 0, nullptr, p.qq, d, nullptr, nullptr, 0, 0, 0, 0};
        lin_first<1>(sq, wv);
        ln_prefetch(w.lncg, w.lncb, d, ln);
        MG_SYNC();
        // P4: ln_cross_attn + q projection
        stage_ln(p, p.x, d, M, d, ln, xs, false);
        __syncthreads();
        lin_run<1>(sq, wv, xs, M);
        const Lin sc_{w.wco, w.bco, d, d, 0, p.x, d, p.x, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
        lin_first<1>(sc_, wv);
        MG_SYNC();
        // P5: cross-attention over the cached encoder K / V
        attn_phase<HD>(p.qq, d, w.ckv, w.ckv + d, 2 * d, (long)p.T_enc * 2 * d, M, p.U, p.H, 0, false, p.enc_len, p.T_enc, p.scale, p.ctx, d, red, rs, rm);
        MG_SYNC();
        // P6: x += ctx Wco^T + bco
        stage_bf16(p.ctx, d, M, d, xs);
        __syncthreads();
        lin_run<1>(sc_, wv, xs, M);
        const Lin sf{w.wfc, w.bfc, 4 * d, d, 2, nullptr, 0, nullptr, p.m, 4 * d, nullptr, nullptr, 0, 0, 0, 0};
        lin_first<1>(sf, wv);
        ln_prefetch(w.ln2g, w.ln2b, d, ln);
        MG_SYNC();
        // P7: ln_2 + c_fc + gelu_new
        stage_ln(p, p.x, d, M, d, ln, xs, false);
        __syncthreads();
        lin_run<1>(sf, wv, xs, M);
        const Lin sp{w.wpr, w.bpr, d, 4 * d, 0, p.x, d, p.x, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
        lin_first<4>(sp, wv);
        MG_SYNC();
        // P8: x += m Wpr^T + bpr
        stage_bf16(p.m, 4 * d, M, 4 * d, xs);
        __syncthreads();
        lin_run<4>(sp, wv, xs, M);
        if (l + 1 < p.nl) {
            const MegaLayer wn = p.L[l + 1];
            s = Lin{wn.wqkv, wn.bqkv, 3 * d, d, 0, nullptr, 0, nullptr, p.qkv, 3 * d, wn.kc, wn.vc, d, p.U, p.past, p.Lmax};
            ln_prefetch(wn.ln1g, wn.ln1b, d, ln);
        } else {
            s = Lin{p.head, nullptr, p.V, d, 0, p.logits, p.ld_logits, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
            ln_prefetch(p.lnfg, p.lnfb, d, ln);
        }
        lin_first<1>(s, wv);
        MG_SYNC();
    }
    // ln_f on the last new position of every sequence + lm head -> fp32 logits
    stage_ln(p, p.x + (long)(p.U - 1) * d, (long)p.U * d, p.B, d, ln, xs, false);
    __syncthreads();
    lin_run<1>(s, wv, xs, p.B);
    MG_STAMP();
    if (blockIdx.x == 0 && threadIdx.x == 0) p.sync->epoch = epoch;    // nobody reads it again in this launch (epoch0 was read before the first barrier)
    (void)epoch0;
}

}  // namespace

extern "C" size_t mi_gpt2_step_workspace_bytes(const mi_gpt2_config* cfg, int B, int U);

// returns MI_ERR_UNSUPPORTED when the shape is outside the persistent kernel's limits (the caller then takes the launch-per-op path)
int gpt2_step_mega(const mi_gpt2_config& c, const void* const* weights, const long* ids_new, int B, int U, int past, int Lmax, void* const* kcache, void* const* vcache,
                   const void* const* cross_kv, int T_enc, const int* enc_len, float emb_scale, void* sync_words, void* stamps, float* x, bf16_t* qkv, bf16_t* ctx, bf16_t* qq, bf16_t* m,
                   float* logits, long ld_logits, hipStream_t st) {
    const int M = B * U, d = c.d, hd = c.d / c.H;
    if (M > MG_MAXM || c.L > MG_MAXL || d > 512 || (d % 8) || (hd != 64 && hd != 128) || T_enc > MG_MAXKEYS || past + U > MG_MAXKEYS || T_enc <= 0) return MI_ERR_UNSUPPORTED;
    MegaArgs a{};
    for (int l = 0; l < c.L; ++l) {
        const void* const* w = weights + 5 + l * 18;
        a.L[l] = MegaLayer{(const float*)w[0], (const float*)w[1], (const bf16_t*)w[2], (const float*)w[3], (const bf16_t*)w[4], (const float*)w[5],
                           (const float*)w[6], (const float*)w[7], (const bf16_t*)w[8], (const float*)w[9], (const bf16_t*)w[10], (const float*)w[11],
                           (const float*)w[12], (const float*)w[13], (const bf16_t*)w[14], (const float*)w[15], (const bf16_t*)w[16], (const float*)w[17],
                           (bf16_t*)kcache[l], (bf16_t*)vcache[l], (const bf16_t*)cross_kv[l]};
    }
    a.wte = (const float*)weights[0]; a.pos = (const float*)weights[1]; a.lnfg = (const float*)weights[2]; a.lnfb = (const float*)weights[3]; a.head = (const bf16_t*)weights[4];
    a.ids = ids_new; a.enc_len = enc_len; a.x = x; a.qkv = qkv; a.ctx = ctx; a.qq = qq; a.m = m; a.logits = logits; a.ld_logits = ld_logits;
    a.sync = (MegaSync*)sync_words;
    a.stamps = (unsigned long long*)stamps;
    a.B = B; a.U = U; a.past = past; a.Lmax = Lmax; a.T_enc = T_enc; a.d = d; a.H = c.H; a.nl = c.L; a.V = c.V;
    a.eps = c.eps; a.emb_scale = emb_scale; a.scale = 1.0f / sqrtf((float)hd);
    if (hd == 64) hipLaunchKernelGGL(gpt2_step_mega_kernel<64>, dim3(MG_BLOCKS), dim3(MG_THREADS), 0, st, a);
    else hipLaunchKernelGGL(gpt2_step_mega_kernel<128>, dim3(MG_BLOCKS), dim3(MG_THREADS), 0, st, a);
    return MI_OK;
}
