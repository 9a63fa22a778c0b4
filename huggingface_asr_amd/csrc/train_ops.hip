// Backward / optimizer building blocks of the training step on gfx950.
//
// Reference: the training step the reference delegates to torch autograd + HF Trainer (bf16 autocast, fp32 master weights,
// AdamW, clip 1.0 — src/utilities/training_utils.py:93-115, recipes .../train_small_baseline.sh:43,53-58).  Autograd has no
// source to restate: every kernel here is the analytic gradient of the forward kernel it is paired with, checked against
// torch autograd of the CPU oracle in tests/.
//
// Conventions: activations bf16 (row-major, leading dimension in elements), residual stream / parameter gradients fp32.
// Parameter gradients ACCUMULATE (+=, float atomics at block granularity): the step zeroes the flat gradient buffer once.
#include "common.hpp"
#include "../../include/hfasr_hip.h"       // mi_lnred_desc

namespace {

// ------------------------------------------------------------------------------------------------ transpose
// in (M,N) bf16 -> out (N, Mp) bf16 with columns M..Mp-1 zero (Mp = K extent of the GEMM that reads it, % 64 == 0).
// 64 x 64 tiles through LDS, 16-B global accesses on both sides when rows are 16-B aligned (scalar edge path otherwise).
constexpr int TR = 64;
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, long ld_in, bf16_t* __restrict__ out,
                                                         long ld_out, int M, int N, int Mp, int vec) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[TR][TR + 8];
    const int m0 = blockIdx.y * TR, n0 = blockIdx.x * TR;
    const int tid = threadIdx.x;
    if (vec) {
        for (int id = tid; id < TR * 8; id += 256) {
            const int r = id >> 3, ch = id & 7;
            const int m = m0 + r, n = n0 + ch * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < M && n + 8 <= N) v = *reinterpret_cast<const bf16x8*>(in + (long)m * ld_in + n);
            else if (m < M)
                for (int j = 0; j < 8; ++j) if (n + j < N) v[j] = in[(long)m * ld_in + n + j];
            *reinterpret_cast<bf16x8*>(&tile[r][ch * 8]) = v;
        }
        __syncthreads();
        for (int id = tid; id < TR * 8; id += 256) {
            const int r = id >> 3, ch = id & 7;              // output row n0 + r, columns m0 + ch*8 ..
            const int n = n0 + r, m = m0 + ch * 8;
            if (n >= N || m >= Mp) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[ch * 8 + j][r];
            *reinterpret_cast<bf16x8*>(out + (long)n * ld_out + m) = v;      // Mp % 8 == 0 and ld_out % 8 == 0 on this path
        }
        return;
    }
    const int tx = tid & 63, ty = tid >> 6;
    for (int r = ty; r < TR; r += 4) {
        const int m = m0 + r, n = n0 + tx;
        tile[r][tx] = (m < M && n < N) ? in[(long)m * ld_in + n] : (bf16_t)0.f;
    }
    __syncthreads();
    for (int r = ty; r < TR; r += 4) {
        const int n = n0 + r, m = m0 + tx;
        if (n < N && m < Mp) out[(long)n * ld_out + m] = tile[tx][r];
    }
}

// many transposes in one launch (the K-major bf16 copies of every weight matrix after an optimizer step): descriptor i = {in, out, M, N, Mp}
struct TrDesc { const bf16_t* in; bf16_t* out; int M, N, Mp, pad; };
__global__ __launch_bounds__(256) void transpose_many_kernel(const TrDesc* __restrict__ descs, int tiles_x_max) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[TR][TR + 8];
    const TrDesc d = descs[blockIdx.y];
    const int ntx = (d.N + TR - 1) / TR, nty = (d.Mp + TR - 1) / TR;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntx * nty; t += gridDim.x) {
        const int n0 = (t % ntx) * TR, m0 = (t / ntx) * TR;
        __syncthreads();
        for (int id = tid; id < TR * 8; id += 256) {
            const int r = id >> 3, ch = id & 7;
            const int m = m0 + r, n = n0 + ch * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < d.M && n + 8 <= d.N) v = *reinterpret_cast<const bf16x8*>(d.in + (long)m * d.N + n);
            else if (m < d.M)
                for (int j = 0; j < 8; ++j) if (n + j < d.N) v[j] = d.in[(long)m * d.N + n + j];
            *reinterpret_cast<bf16x8*>(&tile[r][ch * 8]) = v;
        }
        __syncthreads();
        for (int id = tid; id < TR * 8; id += 256) {
            const int r = id >> 3, ch = id & 7;
            const int n = n0 + r, m = m0 + ch * 8;
            if (n >= d.N || m >= d.Mp) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[ch * 8 + j][r];
            *reinterpret_cast<bf16x8*>(d.out + (long)n * d.Mp + m) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sums (bias grads)
// stage 1: block (column block, row chunk) leaves its 64 column sums in partial[row chunk][N]; stage 2 (rows_reduce_kernel) adds the chunks in order
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ld, int M, int N, float* __restrict__ partial,
                                                      int rows_per_block) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int m = m0 + ty; m < m1; m += 4) s += (float)x[(long)m * ld + n];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) partial[(long)blockIdx.y * N + n] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// out[n] = bf16(sum over the M rows of x[:, n]), rows added in order by one thread (a handful of partial-sum rows: deterministic, no zero-fill, no fp32 round trip)
__global__ __launch_bounds__(256) void colsum_cast_kernel(const float* __restrict__ x, long ld, int M, int N4, bf16_t* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N4; i += (long)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 4);
        for (int m = 1; m < M; ++m) a += *reinterpret_cast<const f32x4*>(x + (long)m * ld + i * 4);
        const bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
        *reinterpret_cast<bf16x4*>(out + i * 4) = o;
    }
}

// out_a[n] += sum_m a[m][n], out_b[n] += sum_m b[m][n]: two partial-sum matrices of the same shape (a few hundred to a few thousand rows of row stride ld) — the position-bias
// gradients from the attention backward's per-wave partials, without atomics (deterministic) and in one launch.  Block = 16 columns x 16 row groups (round 4: a thread per
// column walking ALL rows alone left 2 x N / 256 blocks on the chip — 169 us per call at config 3's 1536 rows x 256 columns, 4 % of that step).
__global__ __launch_bounds__(256) void colsum2_acc_kernel(const float* __restrict__ a, const float* __restrict__ b, long ld, int M, int N,
                                                           float* __restrict__ out_a, float* __restrict__ out_b) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + tx;
    const float* src = (blockIdx.y ? b : a) + n;
    float* dst = blockIdx.y ? out_b : out_a;
    float s = 0.f;
    if (n < N) {
        int m = ty;
        for (; m + 7 * 16 < M; m += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(m + 16 * u) * ld];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; m < M; m += 16) s += src[(long)m * ld];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][tx];
        dst[n] += t;
    }
}

// ------------------------------------------------------------------------------------------------ activations
// MODE 0: out = act(a);  MODE 1: out = a * act'(b)   (a = dy, b = pre-activation);  kind 1 erf-GELU, 2 tanh-GELU
// DROP: the activation dropout of the FFN (tf wav2vec2_conformer :353) applied in the same pass — forward out = dropout(act(a)), backward
// out = dropout(a) * act'(b) — with the mask of dropout.hip for logical index m * N + n and the same bf16 rounding points as the two-pass form
// (act / dropout each rounded to bf16), so results are bit-identical to running mi_dropout separately.
template <int MODE, bool DROP>
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb,
                                                   bf16_t* __restrict__ out, long ldo, int M, int N8, int kind, float p, unsigned long long key) {
    const long total = (long)M * N8;
    const float inv_keep = DROP ? 1.f / (1.f - p) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N8), c = (int)(i % N8) * 8;
        const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + (long)m * lda + c);
        bf16x8 vb = va, o;
        if (MODE == 1) vb = *reinterpret_cast<const bf16x8*>(b + (long)m * ldb + c);
        float ks[8];
        if (DROP) {
            const unsigned long long quad0 = (unsigned long long)i << 1;          // logical index of the first element = 8 i (rows are N = 8 N8 long)
            mask_keep4(key, quad0, p, inv_keep, ks);
            mask_keep4(key, quad0 + 1, p, inv_keep, ks + 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                const float x = bf2f(va[j]);
                bf16_t r = f2bf(kind == 1 ? gelu_erf(x) : gelu_tanh(x));
                if (DROP) r = f2bf(bf2f(r) * ks[j]);
                o[j] = r;
            } else {
                const float x = bf2f(vb[j]);
                const float g = DROP ? bf2f(f2bf(bf2f(va[j]) * ks[j])) : bf2f(va[j]);
                o[j] = f2bf(g * (kind == 1 ? gelu_erf_grad(x) : gelu_tanh_grad(x)));
            }
        }
        *reinterpret_cast<bf16x8*>(out + (long)m * ldo + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// Optional second output of the kernel: out (M, d) bf16 = alpha * dropout(dx) of the row it has just finished (dx after the accumulate) — the bf16 operand of the
// linear backward that follows (what mi_dropout / the fp32 -> bf16 cast made in a pass of their own); p = 0: plain alpha * dx.  Mask of mi_dropout for element m * d + c.
struct LnCast { bf16_t* out; long ld; float alpha, p; unsigned long long key; };
// one wave per row, 4 consecutive columns per lane and step (16-B / 8-B accesses): c = 4 (lane + 64 j);
//   dx (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;   dgamma += dy * xhat, dbeta += dy (block-reduced, then atomics)
// DUAL: a second LayerNorm of the SAME rows (the layer's two branch norms both read x1, e_branchformer.py:273,292) in the same pass: dx is linear in g = dy * gamma, so the
// pass runs once on g = dy * gamma + dy2 * gamma2 (x, the statistics and the old dx are read once instead of twice); only the affine gradients keep two accumulator sets.
struct LnDual { const float* gamma2; const void* dy2; long lddy2; int dy2_f32; float* partial2; };
template <int NV, bool DUAL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ x, long ldx, int x_bf16, const float* __restrict__ gamma,
                                                      float eps, const void* __restrict__ dy, long lddy, int dy_f32,
                                                      void* __restrict__ dx, long lddx, int dx_bf16, int accumulate,
                                                      float* __restrict__ partial, int M, int d,
                                                      int rows_per_wave, LnCast cst, LnDual du) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const dgamma = partial;                       // non-null: this block's (2d) partial row goes to partial[blockIdx.x]
    float* sg = reinterpret_cast<float*>(smem);          // [4 waves][2d]: every wave's (dgamma | dbeta) partial, summed in wave order below (no atomics: deterministic,
                                                         // and a float atomic add on LDS is a compare-and-swap loop in this build)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d4 = d >> 2;
    constexpr int NV2 = DUAL ? NV : 1;
    f32x4 gm[NV], ag[NV], ab[NV], gm2[NV2], ag2[NV2], ab2[NV2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        gm[j] = (c < d4) ? reinterpret_cast<const f32x4*>(gamma)[c] : z4;
        ag[j] = z4; ab[j] = z4;
        if constexpr (DUAL) { gm2[j] = (c < d4) ? reinterpret_cast<const f32x4*>(du.gamma2)[c] : z4; ag2[j] = z4; ab2[j] = z4; }
    }
    const float inv_d = 1.f / d;
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    // A wave walks its rows (wave_id, wave_id + nwaves, ...) two at a time: the reads of both rows — x, dy and, when accumulating, the old dx —
    // are issued before either row is reduced, so the second row's memory round trip hides behind the first row's arithmetic.
    auto load_row = [&](long r, f32x4 (&xv)[NV], f32x4 (&gv)[NV], f32x4 (&ov)[NV], f32x4 (&hv)[NV2]) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            f32x4 v = z4, g = z4, o = z4;
            if constexpr (DUAL) {
                f32x4 h = z4;
                if (c < d4) {
                    if (du.dy2_f32) h = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(du.dy2) + r * du.lddy2)[c];
                    else { const bf16x4 t = reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(du.dy2) + r * du.lddy2)[c]; h = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])}; }
                }
                hv[j] = h;
            }
            if (c < d4) {
                if (x_bf16) { const bf16x4 t = reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(x) + r * ldx)[c]; v = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])}; }
                else v = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + r * ldx)[c];
                if (dy_f32) g = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + r * lddy)[c];
                else { const bf16x4 t = reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(dy) + r * lddy)[c]; g = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])}; }
                if (accumulate) {
                    if (dx_bf16) { const bf16x4 t = reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(dx) + r * lddx)[c]; o = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])}; }
                    else o = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dx) + r * lddx)[c];
                }
            }
            xv[j] = v; gv[j] = g; ov[j] = o;
        }
    };
    auto do_row = [&](long r, f32x4 (&xv)[NV], f32x4 (&gv)[NV], f32x4 (&ov)[NV], f32x4 (&hv)[NV2]) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += xv[j].x + xv[j].y + xv[j].z + xv[j].w;
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            const f32x4 a = (c < d4) ? xv[j] - mean : z4;
            xv[j] = a; q += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xv[j] *= rstd;                        // xhat
            ag[j] += gv[j] * xv[j];
            ab[j] += gv[j];
            gv[j] *= gm[j];                       // g = dy * gamma
            if constexpr (DUAL) { ag2[j] += hv[j] * xv[j]; ab2[j] += hv[j]; gv[j] += hv[j] * gm2[j]; }
            const f32x4 t = gv[j] * xv[j];
            s1 += gv[j].x + gv[j].y + gv[j].z + gv[j].w; s2 += t.x + t.y + t.z + t.w;
        }
        s1 = wave_sum(s1) * inv_d; s2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            if (c >= d4) continue;
            const f32x4 v = (gv[j] - s1 - xv[j] * s2) * rstd + ov[j];            // ov = 0 unless accumulating
            if (dx_bf16) reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(dx) + r * lddx)[c] = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            else reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dx) + r * lddx)[c] = v;
            if (cst.out) {
                f32x4 k4 = {1.f, 1.f, 1.f, 1.f};
                if (cst.p > 0.f) {
                    float kk[4];
                    mask_keep4(cst.key, ((unsigned long long)r * (unsigned)d + 4u * (unsigned)c) >> 2, cst.p, 1.f / (1.f - cst.p), kk);      // d % 4 == 0: the four columns are one quad
                    k4 = f32x4{kk[0], kk[1], kk[2], kk[3]};
                }
                reinterpret_cast<bf16x4*>(cst.out + r * cst.ld)[c] = bf16x4{f2bf(v.x * cst.alpha * k4.x), f2bf(v.y * cst.alpha * k4.y), f2bf(v.z * cst.alpha * k4.z), f2bf(v.w * cst.alpha * k4.w)};
            }
        }
    };
    for (int i = 0; i < rows_per_wave; i += 2) {
        const long r0 = wave_id + (long)i * nwaves, r1 = r0 + nwaves;          // interleaved rows: neighbouring waves touch neighbouring rows
        if (r0 >= M) break;
        const bool two = (i + 1 < rows_per_wave) && r1 < M;
        f32x4 xa[NV], ga[NV], oa[NV], xb[NV], gb[NV], ob[NV], ha[NV2], hb[NV2];
        load_row(r0, xa, ga, oa, ha);
        if (two) load_row(r1, xb, gb, ob, hb);
        do_row(r0, xa, ga, oa, ha);
        if (two) do_row(r1, xb, gb, ob, hb);
    }
    if (dgamma) {
        float* mine = sg + (size_t)wave * 2 * d;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            if (c < d4) {
                reinterpret_cast<f32x4*>(mine)[c] = ag[j];
                reinterpret_cast<f32x4*>(mine + d)[c] = ab[j];
            }
        }
        __syncthreads();
        float* prow = partial + (long)blockIdx.x * 2 * d;
        for (int c = threadIdx.x; c < 2 * d; c += 256) prow[c] = (sg[c] + sg[2 * d + c]) + (sg[4 * d + c] + sg[6 * d + c]);
        if constexpr (DUAL) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = lane + 64 * j;
                if (c < d4) {
                    reinterpret_cast<f32x4*>(mine)[c] = ag2[j];
                    reinterpret_cast<f32x4*>(mine + d)[c] = ab2[j];
                }
            }
            __syncthreads();
            float* prow2 = du.partial2 + (long)blockIdx.x * 2 * d;
            for (int c = threadIdx.x; c < 2 * d; c += 256) prow2[c] = (sg[c] + sg[2 * d + c]) + (sg[4 * d + c] + sg[6 * d + c]);
        }
    }
}

// dgamma[c] += sum_blk partial[blk][c],  dbeta[c] += sum_blk partial[blk][d + c] for up to 24 LayerNorms (or other row sets, `kind`) in ONE launch (the trainer defers the reductions of a layer's LayerNorm backward passes and flushes them together: a
// reduce of 2 MB is all launch latency).  grid (columns / 16, LayerNorm); a block owns its 16 columns over ALL partial rows (16 row groups x 32 rows), summed in a fixed order: no atomics.
constexpr int LNRED_MAX = 24;            // a pair of layers defers 18 reductions (12 LayerNorm, 4 depthwise, 2 position-bias): one launch per pair
struct LnRedMany { mi_lnred_desc d[LNRED_MAX]; int blk0[LNRED_MAX + 1]; int n; };      // blk0: first block of every entry (64 columns per block): the grid has no empty blocks
__global__ __launch_bounds__(256) void ln_partial_reduce_many_kernel(LnRedMany p) {
    // a block owns 64 columns of ONE entry (a wave reads 256 consecutive bytes of a partial row) over all its partial rows: 4 row groups (one per wave), each walking its
    // rows in order with eight loads in flight.  The grid is the exact block count (prefix table blk0): as (widest entry / 16 columns) x entries it was ~37 000 blocks, most
    // of which found nothing to do — dispatching them, not the 40 MB, was the launch's 16-18 us (64-column blocks alone, and 16 row groups alone, changed nothing).
    __shared__ float red[4][64];
    int e = 0;
    while (e + 1 < p.n && (int)blockIdx.x >= p.blk0[e + 1]) ++e;
    e = __builtin_amdgcn_readfirstlane(e);
    const mi_lnred_desc q = p.d[e];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // 64 columns x 4 row groups
    const int c = ((int)blockIdx.x - p.blk0[e]) * 64 + tx;
    const int ncol = q.kind ? q.d * 32 : 2 * q.d;                      // kind = K > 0: a depthwise conv's (channel, 32) tap-gradient partials (include/hfasr_hip.h)
    float s = 0.f;
    if (c < ncol) {
        const float* src = q.partial + c;
        const long ld = ncol;
        int b = ty;
        for (; b + 7 * 4 < q.nblk; b += 8 * 4) {                      // the sum order is fixed (row group, then rows ascending)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long)(b + j * 4) * ld];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; b < q.nblk; b += 4) s += src[(long)b * ld];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < ncol) {
        const float t = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        if (q.kind) {
            const int ch = c >> 5, k = c & 31;
            if (k < q.kind) q.dgamma[(long)ch * q.kind + k] += t;
            else if (k == 31 && q.dbeta) q.dbeta[ch] += t;
        } else {
            float* o = c < q.d ? q.dgamma + c : q.dbeta + (c - q.d);
            *o += t;
        }
    }
}

// y = LN(x) with precomputed (mean, rstd) per row: the CSGU gate normalisation re-materialised for the backward pass
__global__ __launch_bounds__(256) void ln_apply_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ y, long ldy, int M, int N) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        y[(long)m * ldy + c] = f2bf((bf2f(x[(long)m * ldx + c]) - stats[2 * m]) * stats[2 * m + 1] * gamma[c] + beta[c]);
    }
}

// ------------------------------------------------------------------------------------------------ small element-wise ops
__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ a, const float* __restrict__ b, long n, float alpha) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] += alpha * b[i];
}
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ a, long n, float alpha) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] *= alpha;
}
// a *= *alpha with alpha on the device; nothing is read or written when it is exactly 1 (the d(loss) that `loss.backward()` seeds: the HF route's common case)
__global__ __launch_bounds__(256) void scale_dev_f32_kernel(float* __restrict__ a, long n, const float* __restrict__ alpha) {
    const float s = *alpha;
    if (s == 1.f) return;
    const long n4 = n >> 2;
    f32x4* a4 = reinterpret_cast<f32x4*>(a);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) a4[i] *= s;
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] *= s;
}
// out = bf16(alpha * (a [+ b])) on (M,N) views
__global__ __launch_bounds__(256) void add2_cast_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                         bf16_t* __restrict__ out, long ldo, int M, int N, float alpha) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        float v = a[(long)m * lda + c];
        if (b) v += b[(long)m * ldb + c];
        out[(long)m * ldo + c] = f2bf(alpha * v);
    }
}
// fast form: 8 columns per thread (two 16-B reads per operand, one 16-B write), 32-bit index math; same arithmetic per element
__global__ __launch_bounds__(256) void add2_cast_vec8_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                              bf16_t* __restrict__ out, long ldo, int M, int N8, float alpha) {
    const unsigned total = (unsigned)M * (unsigned)N8;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned m = i / (unsigned)N8, c = (i - m * (unsigned)N8) << 3;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(a + (long)m * lda + c), v1 = *reinterpret_cast<const f32x4*>(a + (long)m * lda + c + 4);
        if (b) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(b + (long)m * ldb + c), w1 = *reinterpret_cast<const f32x4*>(b + (long)m * ldb + c + 4);
            v0 += w0; v1 += w1;
        }
        const bf16x8 o = {f2bf(alpha * v0.x), f2bf(alpha * v0.y), f2bf(alpha * v0.z), f2bf(alpha * v0.w),
                          f2bf(alpha * v1.x), f2bf(alpha * v1.y), f2bf(alpha * v1.z), f2bf(alpha * v1.w)};
        *reinterpret_cast<bf16x8*>(out + (long)m * ldo + c) = o;
    }
}
// out = bf16(x + vec[c])   (q + pos_bias_u / pos_bias_v, flattened (H*hd) vector)
__global__ __launch_bounds__(256) void add_rowvec_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ vec,
                                                          bf16_t* __restrict__ out, long ldo, int M, int N) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        out[(long)m * ldo + c] = f2bf(bf2f(x[(long)m * ldx + c]) + vec[c]);
    }
}
// the same for TWO vectors from one read of x, 8 columns per thread: (q + pos_bias_u, q + pos_bias_v) of the attention backward
__global__ __launch_bounds__(256) void add_rowvec2_vec8_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ u, const float* __restrict__ v,
                                                                bf16_t* __restrict__ ou, bf16_t* __restrict__ ov, long ldo, int M, int N8) {
    const long total = (long)M * N8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N8), c = (int)(i % N8) * 8;
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (long)m * ldx + c);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(u + c), u1 = *reinterpret_cast<const f32x4*>(u + c + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(v + c), v1 = *reinterpret_cast<const f32x4*>(v + c + 4);
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = bf2f(xv[j]);
            a[j] = f2bf(xf + (j < 4 ? u0[j & 3] : u1[j & 3]));
            b[j] = f2bf(xf + (j < 4 ? v0[j & 3] : v1[j & 3]));
        }
        *reinterpret_cast<bf16x8*>(ou + (long)m * ldo + c) = a;
        *reinterpret_cast<bf16x8*>(ov + (long)m * ldo + c) = b;
    }
}
// gating backward of s = r * c:  dr = ds * c,  dc = ds * r
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* __restrict__ ds, long ldds, const bf16_t* __restrict__ c, long ldc,
                                                        const bf16_t* __restrict__ r, long ldr, bf16_t* __restrict__ dr, long lddr,
                                                        bf16_t* __restrict__ dc, long lddc, int M, int N) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float g = bf2f(ds[(long)m * ldds + n]);
        dr[(long)m * lddr + n] = f2bf(g * bf2f(c[(long)m * ldc + n]));
        dc[(long)m * lddc + n] = f2bf(g * bf2f(r[(long)m * ldr + n]));
    }
}
// zero the rows of padded frames: row (b, t) with t >= lengths[b]
__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, long ld, const int* __restrict__ lengths, int T, int M, int N) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        if ((m % T) >= lengths[m / T]) x[(long)m * ld + c] = 0.f;
    }
}

// in-model SpecAugment (tf wav2vec2_conformer `_mask_hidden_states` :1086-1130): rows with time_mask != 0 are REPLACED by the learned
// masked_spec_embed vector, then columns with feat_mask[b][c] != 0 are zeroed.  Backward: masked rows send their gradient to the embed
// vector (column sums, atomics at block granularity) and nothing upstream; masked columns send nothing.
__global__ __launch_bounds__(256) void spec_mask_apply_kernel(float* __restrict__ x, long ld, const unsigned char* __restrict__ tmask,
                                                               const float* __restrict__ embed, const unsigned char* __restrict__ fmask, int T, int M, int N) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), c = (int)(i % N);
        float v = x[(long)m * ld + c];
        if (tmask && tmask[m]) v = embed[c];
        if (fmask && fmask[(long)(m / T) * N + c]) v = 0.f;
        x[(long)m * ld + c] = v;
    }
}
__global__ __launch_bounds__(256) void spec_mask_bwd_kernel(float* __restrict__ dx, long ld, const unsigned char* __restrict__ tmask, float* __restrict__ partial,
                                                             const unsigned char* __restrict__ fmask, int T, int M, int N, int rows_per_block) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int m = m0 + ty; m < m1; m += 4) {
            float g = dx[(long)m * ld + c];
            if (fmask && fmask[(long)(m / T) * N + c]) g = 0.f;          // zeroed AFTER the replacement in the forward
            if (tmask && tmask[m]) { s += g; g = 0.f; }
            dx[(long)m * ld + c] = g;
        }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N && tmask && partial) partial[(long)blockIdx.y * N + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// ------------------------------------------------------------------------------------------------ optimizer
// deterministic sum of squares (every data-parallel rank must derive the SAME clip coefficient from the same reduced gradient, or the
// replicas drift): fixed grid of per-block partials, then one block adds them in a fixed order.  No float atomics.
constexpr int SSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
    __shared__ float part[4];
    float s = 0.f;
    const long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;          // 16-B loads; fixed element -> thread map, so still run-to-run identical
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (part[0] + part[1]) + (part[2] + part[3]);
}
// ---- layer mixing of the CTC fine-tuning head (bestrq.py:239-245): hidden = sum_l softmax(per_layer_weights)_l * hidden_states[l].
// The weights stay on the device: the axpy reads its coefficient from memory, so the step has no host round trip.
__global__ __launch_bounds__(256) void axpy_dev_kernel(float* __restrict__ a, const float* __restrict__ b, long n, const float* __restrict__ alpha, int overwrite) {
    const float w = alpha[0];
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const long n4 = al ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 bv = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 av = overwrite ? f32x4{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<f32x4*>(a)[i];
        av.x += w * bv.x; av.y += w * bv.y; av.z += w * bv.z; av.w += w * bv.w;
        reinterpret_cast<f32x4*>(a)[i] = av;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] = (overwrite ? 0.f : a[i]) + w * b[i];
}
// <a, b> with the partial scheme of sumsq_kernel (fixed element -> thread map, fixed-order final add: run-to-run identical)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ partial) {
    __shared__ float part[4];
    float s = 0.f;
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const long n4 = al ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 u = reinterpret_cast<const f32x4*>(a)[i], v = reinterpret_cast<const f32x4*>(b)[i];
        s += (u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// softmax of a short vector (n <= 1024, one block) and its backward dw += s * (g - <s, g>)
__global__ __launch_bounds__(64) void softmax_vec_kernel(const float* __restrict__ w, int n, float* __restrict__ s) {
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, w[i]);
    m = wave_max(m);
    float z = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) z += expf(w[i] - m);
    z = wave_sum(z);
    for (int i = threadIdx.x; i < n; i += 64) s[i] = expf(w[i] - m) / z;
}
__global__ __launch_bounds__(64) void softmax_vec_bwd_kernel(const float* __restrict__ s, const float* __restrict__ g, int n, float* __restrict__ dw) {
    float sg = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) sg += s[i] * g[i];
    sg = wave_sum(sg);
    for (int i = threadIdx.x; i < n; i += 64) dw[i] += s[i] * (g[i] - sg);
}
// norm = sqrt(sumsq); coef = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_).  skip = 1 when the norm is not finite or above
// `skip_above` (> 0): the reference's GradAwareTrainer drops such a step altogether (training_utils.py:81,101-115: grads set to None).
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float skip_above, float* out /* [norm, coef, skip] */) {
    const float norm = sqrtf(sumsq[0]);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(1.f, max_norm / (norm + 1e-6f));
    const bool skip = !isfinite(norm) || (skip_above > 0.f && norm > skip_above);
    if (skip) coef = 0.f;
    out[0] = norm; out[1] = coef; out[2] = skip ? 1.f : 0.f;
}
// torch.optim.AdamW (decoupled weight decay) on the flat fp32 master buffer; also refreshes the bf16 mirror the GEMMs read
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, const unsigned char* __restrict__ decay, long n,
                                                     float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                     const float* __restrict__ clip /* [norm, coef, skip] or null */, bf16_t* __restrict__ mirror) {
    const float coef = clip ? clip[1] : 1.f;
    const bool skip = clip && (clip[2] != 0.f || !isfinite(clip[0]));
    const float step = lr / bc1, rs2 = rsqrtf(bc2);
    auto one = [&](float& pv, float gi, float& mi, float& vi, bool dec) {
        gi *= coef;
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        if (dec) pv *= (1.f - lr * wd);
        pv -= step * mi / (sqrtf(vi) * rs2 + eps);
    };
    // 16 B per lane and array (the step is pure streaming: 4 reads + 3 writes of the flat buffer + the bf16 mirror)
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                    (!decay || (reinterpret_cast<uintptr_t>(decay) & 3) == 0) && (!mirror || (reinterpret_cast<uintptr_t>(mirror) & 7) == 0);
    const long n4 = al ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 p4 = reinterpret_cast<f32x4*>(p)[i];
        float pv[4] = {p4.x, p4.y, p4.z, p4.w};
        if (!skip) {
            const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i], m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
            const unsigned dm = decay ? reinterpret_cast<const unsigned*>(decay)[i] : 0x01010101u;
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
            float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) one(pv[j], gv[j], mv[j], vv[j], (dm >> (8 * j)) & 0xffu);
            reinterpret_cast<f32x4*>(m)[i] = f32x4{mv[0], mv[1], mv[2], mv[3]};
            reinterpret_cast<f32x4*>(v)[i] = f32x4{vv[0], vv[1], vv[2], vv[3]};
            reinterpret_cast<f32x4*>(p)[i] = f32x4{pv[0], pv[1], pv[2], pv[3]};
        }
        if (mirror) {
            const bf16x2 lo = {f2bf(pv[0]), f2bf(pv[1])}, hi = {f2bf(pv[2]), f2bf(pv[3])};
            reinterpret_cast<uint2*>(mirror)[i] = uint2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        }
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float pv = p[i];
        if (!skip) {
            float mi = m[i], vi = v[i];
            one(pv, g[i], mi, vi, !decay || decay[i]);
            m[i] = mi; v[i] = vi; p[i] = pv;
        }
        if (mirror) mirror[i] = f2bf(pv);
    }
}

int grid_for(long n) { const long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

}  // namespace

extern "C" int mi_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int M, int N, int Mp, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || Mp < M || ld_out < Mp) return MI_ERR_ARG;
    const int vec = ((ld_in % 8) == 0 && (ld_out % 8) == 0 && (Mp % 8) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(N, TR), cdiv(Mp, TR)), dim3(256), 0, st, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, M, N, Mp, vec);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// descs: device array of `count` {in (M,N) contiguous bf16, out (N,Mp) contiguous bf16, M, N, Mp, pad}; N % 8 == 0, Mp % 8 == 0, 16-B aligned
extern "C" int mi_transpose_many_bf16(const void* descs, int count, hipStream_t st) {
    MI_ENTER();
    if (count <= 0 || count > 65535) return MI_ERR_ARG;
    hipLaunchKernelGGL(transpose_many_kernel, dim3(32, count), dim3(256), 0, st, (const TrDesc*)descs, 0);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// out[n] += sum_m x[m,n];  dtype 0 f32, 1 bf16.  workspace: mi_colsum_workspace_floats(M, N) floats (per-chunk partial sums; the chunks are added in a fixed order)
extern "C" size_t mi_colsum_workspace_floats(int M, int N) { return (size_t)cdiv(M, 128) * (size_t)N; }
extern "C" int mi_colsum(const void* x, long ld, int dtype, int M, int N, float* out, float* workspace, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || !workspace) return MI_ERR_ARG;
    const int rpb = 128;
    dim3 grid(cdiv(N, 64), cdiv(M, rpb));
    if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ld, M, N, workspace, rpb);
    else if (dtype == 1) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ld, M, N, workspace, rpb);
    else return MI_ERR_ARG;
    MI_CHECK_LAUNCH();
    rows_reduce_launch(workspace, (int)grid.y, N, EmitAdd{out}, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// out_a (N) += column sums of a (M, N) f32, out_b (N) += column sums of b (M, N) f32 (same shape and row stride): deterministic, one launch
extern "C" int mi_colsum2_acc_f32(const float* a, const float* b, long ld, int M, int N, float* out_a, float* out_b, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || !a || !b || !out_a || !out_b) return MI_ERR_ARG;
    hipLaunchKernelGGL(colsum2_acc_kernel, dim3(cdiv(N, 16), 2), dim3(256), 0, st, a, b, ld, M, N, out_a, out_b);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// out (N) bf16 = column sums of x (M, N) f32, M small (partial sums of a batched product): N % 4 == 0, 16-byte aligned rows
extern "C" int mi_colsum_cast_bf16(const float* x, long ld, int M, int N, void* out, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || (N % 4) || (ld % 4) || !x || !out || ((reinterpret_cast<uintptr_t>(x) & 15) != 0) || ((reinterpret_cast<uintptr_t>(out) & 7) != 0)) return MI_ERR_ARG;
    hipLaunchKernelGGL(colsum_cast_kernel, dim3(grid_for((long)N / 4)), dim3(256), 0, st, x, ld, M, N / 4, (bf16_t*)out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// kind 1 erf-GELU (ACT2FN["gelu"]), 2 tanh-GELU (gelu_new)
extern "C" int mi_act_fwd_bf16(const void* pre, long ldp, void* out, long ldo, int M, int N, int kind, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || (N % 8) || (ldp % 8) || (ldo % 8) || (kind != 1 && kind != 2)) return MI_ERR_ARG;
    hipLaunchKernelGGL((act_kernel<0, false>), dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, (const bf16_t*)pre, ldp, (const bf16_t*)nullptr, 0L, (bf16_t*)out, ldo, M, N / 8, kind, 0.f, 0ull);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// out = dropout(act(pre)) / dx = dropout(dy) * act'(pre): activation + the FFN's activation dropout in one pass (mask of mi_dropout for the (M, N) matrix)
extern "C" int mi_act_dropout_fwd_bf16(const void* pre, long ldp, void* out, long ldo, int M, int N, int kind, float p, unsigned seed, unsigned stream_id,
                                       hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || (N % 8) || (ldp % 8) || (ldo % 8) || (kind != 1 && kind != 2) || p < 0.f || p >= 1.f) return MI_ERR_ARG;
    const unsigned long long key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    hipLaunchKernelGGL((act_kernel<0, true>), dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, (const bf16_t*)pre, ldp, (const bf16_t*)nullptr, 0L, (bf16_t*)out, ldo, M, N / 8, kind, p, key);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_act_dropout_bwd_bf16(const void* dy, long lddy, const void* pre, long ldp, void* dx, long lddx, int M, int N, int kind, float p,
                                       unsigned seed, unsigned stream_id, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || (N % 8) || (ldp % 8) || (lddy % 8) || (lddx % 8) || (kind != 1 && kind != 2) || p < 0.f || p >= 1.f) return MI_ERR_ARG;
    const unsigned long long key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    hipLaunchKernelGGL((act_kernel<1, true>), dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)pre, ldp, (bf16_t*)dx, lddx, M, N / 8, kind, p, key);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_act_bwd_bf16(const void* dy, long lddy, const void* pre, long ldp, void* dx, long lddx, int M, int N, int kind, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || (N % 8) || (ldp % 8) || (lddy % 8) || (lddx % 8) || (kind != 1 && kind != 2)) return MI_ERR_ARG;
    hipLaunchKernelGGL((act_kernel<1, false>), dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)pre, ldp, (bf16_t*)dx, lddx, M, N / 8, kind, 0.f, 0ull);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// workspace: >= mi_layernorm_bwd_workspace_floats(d) floats when dgamma != NULL (per-block partial sums; no float atomics)
extern "C" size_t mi_layernorm_bwd_workspace_floats(int d) { return (size_t)512 * 2 * d; }
static int ln_bwd_launch(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                         void* dx, long lddx, int dx_bf16, int accumulate, float* partial, int* nblk, int M, int d, hipStream_t st, LnCast cst = LnCast{nullptr, 0, 1.f, 0.f, 0ull},
                         LnDual du = LnDual{nullptr, nullptr, 0, 0, nullptr}) {
    if (du.dy2 && (!du.gamma2 || !du.partial2 || !partial || d > 512 || (du.lddy2 % 4) || (reinterpret_cast<uintptr_t>(du.dy2) & (du.dy2_f32 ? 15 : 7)))) return MI_ERR_ARG;
    if (cst.out && ((cst.ld % 4) || (reinterpret_cast<uintptr_t>(cst.out) & 7) || cst.p < 0.f || cst.p >= 1.f || (d % 2))) return MI_ERR_ARG;
    if (M <= 0 || d <= 0 || d > 2048 || (d % 4) || (ldx % 4) || (lddy % 4) || (lddx % 4) || !gamma) return MI_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & (x_bf16 ? 7 : 15)) || (reinterpret_cast<uintptr_t>(dy) & (dy_f32 ? 15 : 7)) ||
        (reinterpret_cast<uintptr_t>(dx) & (dx_bf16 ? 7 : 15))) return MI_ERR_ARG;
    int grid = cdiv(M, 4);
    if (grid > 512) grid = 512;                         // two blocks per CU; each wave walks rows wave_id, wave_id + nwaves, ... (256 / 1024 / 2048 blocks measured slower)
    const int rpw = cdiv(M, (long)grid * 4);
    const size_t lds = (size_t)4 * 2 * d * sizeof(float);
    const int nv = cdiv(d, 256);
#define LN_BWD(NV, DU) hipLaunchKernelGGL((ln_bwd_kernel<NV, DU>), dim3(grid), dim3(256), lds, st, x, ldx, x_bf16, gamma, eps, dy, lddy, dy_f32, dx, lddx, dx_bf16, accumulate, partial, M, d, rpw, cst, du)
    if (du.dy2) { if (nv <= 1) LN_BWD(1, true); else LN_BWD(2, true); }
    else if (nv <= 1) LN_BWD(1, false); else if (nv <= 2) LN_BWD(2, false); else if (nv <= 3) LN_BWD(3, false); else if (nv <= 4) LN_BWD(4, false); else LN_BWD(8, false);
#undef LN_BWD
    if (nblk) *nblk = grid;
    return MI_OK;
}

extern "C" int mi_layernorm_bwd(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                                void* dx, long lddx, int dx_bf16, int accumulate, float* dgamma, float* dbeta, float* workspace, int M, int d,
                                hipStream_t st) {
    MI_ENTER();
    if (dgamma && (!dbeta || !workspace)) return MI_ERR_ARG;
    int grid = 0;
    const int rc = ln_bwd_launch(x, ldx, x_bf16, gamma, eps, dy, lddy, dy_f32, dx, lddx, dx_bf16, accumulate, dgamma ? workspace : nullptr, &grid, M, d, st);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    if (dgamma) {
        LnRedMany one{};
        one.d[0] = mi_lnred_desc{workspace, grid, d, dgamma, dbeta, 0};
        one.n = 1; one.blk0[0] = 0; one.blk0[1] = cdiv(2 * d, 64);
        hipLaunchKernelGGL(ln_partial_reduce_many_kernel, dim3(one.blk0[1]), dim3(256), 0, st, one);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}

// as mi_layernorm_bwd, but the (dgamma | dbeta) partial rows stay in `partial` (*nblk rows of 2 d floats, at most 512: mi_layernorm_bwd_workspace_floats) and are
// NOT reduced here: the caller hands them to mi_ln_partial_reduce_many, several LayerNorms per launch
extern "C" int mi_layernorm_bwd_partial(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                                        void* dx, long lddx, int dx_bf16, int accumulate, float* partial, int* nblk, int M, int d, hipStream_t st) {
    MI_ENTER();
    if (!partial || !nblk) return MI_ERR_ARG;
    const int rc = ln_bwd_launch(x, ldx, x_bf16, gamma, eps, dy, lddy, dy_f32, dx, lddx, dx_bf16, accumulate, partial, nblk, M, d, st);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// as mi_layernorm_bwd_partial, and in the same pass cast (M, d) bf16 = alpha * dropout(dx) of the finished rows: the operand of the linear backward that follows
// (drop_p = 0: the plain scaled cast; mask of mi_dropout for (seed, stream_id), element m * d + c)
extern "C" int mi_layernorm_bwd_partial_cast(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                                             void* dx, long lddx, int dx_bf16, int accumulate, float* partial, int* nblk, void* cast, long ldcast, float alpha,
                                             float drop_p, unsigned seed, unsigned stream_id, int M, int d, hipStream_t st) {
    MI_ENTER();
    if (!partial || !nblk || !cast) return MI_ERR_ARG;
    const int rc = ln_bwd_launch(x, ldx, x_bf16, gamma, eps, dy, lddy, dy_f32, dx, lddx, dx_bf16, accumulate, partial, nblk, M, d, st,
                                 LnCast{(bf16_t*)cast, ldcast, alpha, drop_p, ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed});
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// Two LayerNorms of the same rows x (same eps) in one pass: dx (+)= dLN1/dx · dy + dLN2/dx · dy2; their (dgamma | dbeta) partial rows go to `partial` and `partial2`
// (*nblk rows each).  d <= 512.  cast (may be NULL): the bf16 alpha * dropout(dx) operand of mi_layernorm_bwd_partial_cast.
extern "C" int mi_layernorm_bwd_dual_partial(const void* x, long ldx, int x_bf16, float eps, const float* gamma, const void* dy, long lddy, int dy_f32,
                                             const float* gamma2, const void* dy2, long lddy2, int dy2_f32, void* dx, long lddx, int dx_bf16, int accumulate,
                                             float* partial, float* partial2, int* nblk, void* cast, long ldcast, float alpha, float drop_p, unsigned seed,
                                             unsigned stream_id, int M, int d, hipStream_t st) {
    MI_ENTER();
    if (!partial || !partial2 || !nblk || !dy2 || !gamma2) return MI_ERR_ARG;
    const int rc = ln_bwd_launch(x, ldx, x_bf16, gamma, eps, dy, lddy, dy_f32, dx, lddx, dx_bf16, accumulate, partial, nblk, M, d, st,
                                 LnCast{(bf16_t*)cast, ldcast, alpha, drop_p, ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed},
                                 LnDual{gamma2, dy2, lddy2, dy2_f32, partial2});
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_ln_partial_reduce_many(const mi_lnred_desc* descs, int n, hipStream_t st) {
    MI_ENTER();
    if (!descs || n <= 0 || n > LNRED_MAX) return MI_ERR_ARG;
    LnRedMany p{};
    p.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!descs[i].partial || !descs[i].dgamma || (!descs[i].dbeta && !descs[i].kind) || descs[i].nblk <= 0 || descs[i].d <= 0 || descs[i].kind < 0 || descs[i].kind > 31) return MI_ERR_ARG;
        p.d[i] = descs[i];
        p.blk0[i] = blocks;
        blocks += cdiv(descs[i].kind ? descs[i].d * 32 : 2 * descs[i].d, 64);
    }
    p.blk0[n] = blocks;
    hipLaunchKernelGGL(ln_partial_reduce_many_kernel, dim3(blocks), dim3(256), 0, st, p);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_ln_apply_bf16(const void* x, long ldx, const float* stats, const float* gamma, const float* beta, void* y, long ldy,
                                int M, int N, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(ln_apply_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const bf16_t*)x, ldx, stats, gamma, beta, (bf16_t*)y, ldy, M, N);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_axpy_f32(float* a, const float* b, long n, float alpha, hipStream_t st) {
    MI_ENTER();
    if (n <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, b, n, alpha);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_scale_f32(float* a, long n, float alpha, hipStream_t st) {
    MI_ENTER();
    if (n <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, n, alpha);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_scale_dev_f32(float* a, long n, const float* alpha_dev, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || !alpha_dev || (reinterpret_cast<uintptr_t>(a) & 15)) return MI_ERR_ARG;
    hipLaunchKernelGGL(scale_dev_f32_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, st, a, n, alpha_dev);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_add2_cast_bf16(const float* a, long lda, const float* b, long ldb, void* out, long ldo, int M, int N, float alpha, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0) return MI_ERR_ARG;
    const bool vec = (N % 8) == 0 && (lda % 4) == 0 && (!b || (ldb % 4) == 0) && (ldo % 8) == 0 && (long)M * N < (1L << 31) &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(add2_cast_vec8_kernel, dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, a, lda, b, ldb, (bf16_t*)out, ldo, M, N / 8, alpha);
    else hipLaunchKernelGGL(add2_cast_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, a, lda, b, ldb, (bf16_t*)out, ldo, M, N, alpha);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_add_rowvec_bf16(const void* x, long ldx, const float* vec, void* out, long ldo, int M, int N, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const bf16_t*)x, ldx, vec, (bf16_t*)out, ldo, M, N);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// out_u = bf16(x + u), out_v = bf16(x + v): one pass over x (the element-wise form twice when the rows are not 16-byte aligned)
extern "C" int mi_add_rowvec2_bf16(const void* x, long ldx, const float* u, const float* v, void* out_u, void* out_v, long ldo, int M, int N, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || !x || !u || !v || !out_u || !out_v) return MI_ERR_ARG;
    const bool vec = (N % 8) == 0 && (ldx % 8) == 0 && (ldo % 8) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out_u) | reinterpret_cast<uintptr_t>(out_v)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(add_rowvec2_vec8_kernel, dim3(grid_for((long)M * N / 8)), dim3(256), 0, st, (const bf16_t*)x, ldx, u, v, (bf16_t*)out_u, (bf16_t*)out_v, ldo, M, N / 8);
    else {
        hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const bf16_t*)x, ldx, u, (bf16_t*)out_u, ldo, M, N);
        hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const bf16_t*)x, ldx, v, (bf16_t*)out_v, ldo, M, N);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_gate_bwd_bf16(const void* ds, long ldds, const void* c, long ldc, const void* r, long ldr, void* dr, long lddr,
                                void* dc, long lddc, int M, int N, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const bf16_t*)ds, ldds, (const bf16_t*)c, ldc,
                       (const bf16_t*)r, ldr, (bf16_t*)dr, lddr, (bf16_t*)dc, lddc, M, N);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// frame counts behind the two-layer Conv2d sub-sampling for every utterance in one launch (the dozen one-block torch kernels of the length arithmetic were 70 us of a step):
// inner = min(the count with the convs' padding, tmax) — what the encoder's masks use —, outer = the count without padding — what the CTC loss is given
// (reference: _get_feat_extract_output_lengths with / without padding).  Floor division as torch.div(rounding_mode="floor").
__global__ void subsampled_lengths_kernel(const int* __restrict__ in, int B, int k, int s, int p, int layers, int tmax, int* __restrict__ inner, int* __restrict__ outer) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    auto fdiv = [](int a, int d) { int q = a / d; if ((a % d) != 0 && ((a < 0) != (d < 0))) --q; return q; };
    int li = in[b], lo = li;
    for (int i = 0; i < layers; ++i) { li = fdiv(li + 2 * p - k, s) + 1; lo = fdiv(lo - k, s) + 1; }
    inner[b] = li < tmax ? li : tmax;
    outer[b] = lo;
}
extern "C" int mi_subsampled_lengths_i32(const int* lengths, int B, int kernel, int stride, int pad, int layers, int tmax, int* inner, int* outer, hipStream_t st) {
    MI_ENTER();
    if (!lengths || !inner || !outer || B <= 0 || kernel <= 0 || stride <= 0 || pad < 0 || layers < 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(subsampled_lengths_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, lengths, B, kernel, stride, pad, layers, tmax, inner, outer);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_mask_rows_f32(float* x, long ld, const int* lengths, int T, int M, int N, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || T <= 0 || !lengths) return MI_ERR_ARG;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, x, ld, lengths, T, M, N);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_spec_mask_apply(float* x, long ld, const unsigned char* time_mask, const float* embed, const unsigned char* feat_mask, int T, int M, int N,
                                  hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || T <= 0 || (time_mask && !embed)) return MI_ERR_ARG;
    hipLaunchKernelGGL(spec_mask_apply_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, x, ld, time_mask, embed, feat_mask, T, M, N);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// workspace: ceil(M / 128) * N floats when dembed is given (per-chunk partial sums of the masked rows' gradient, added in a fixed order)
extern "C" int mi_spec_mask_bwd(float* dx, long ld, const unsigned char* time_mask, float* dembed, const unsigned char* feat_mask, int T, int M, int N,
                                float* workspace, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || N <= 0 || T <= 0) return MI_ERR_ARG;      // dembed == NULL: masked rows were replaced by a constant (noise): gradient dropped
    const bool want = time_mask && dembed;
    if (want && !workspace) return MI_ERR_ARG;
    const int rpb = 128;
    hipLaunchKernelGGL(spec_mask_bwd_kernel, dim3(cdiv(N, 64), cdiv(M, rpb)), dim3(256), 0, st, dx, ld, time_mask, want ? workspace : nullptr, feat_mask, T, M, N, rpb);
    MI_CHECK_LAUNCH();
    if (want) {
        rows_reduce_launch(workspace, cdiv(M, rpb), N, EmitAdd{dembed}, st);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}

// sumsq[0] += sum x^2 (caller zeroes sumsq; bit-reproducible);  workspace: 1024 floats;  clip: out = [norm, coef]
extern "C" int mi_sumsq_f32(const float* x, long n, float* sumsq, float* workspace, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || !workspace) return MI_ERR_ARG;
    hipLaunchKernelGGL(sumsq_kernel, dim3(SSQ_BLOCKS), dim3(256), 0, st, x, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, workspace, SSQ_BLOCKS, sumsq);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_axpy_dev_f32(float* a, const float* b, long n, const float* alpha, int overwrite, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || !alpha) return MI_ERR_ARG;
    hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, a, b, n, alpha, overwrite);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_dot_f32(const float* a, const float* b, long n, float* out, float* workspace, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || !workspace) return MI_ERR_ARG;
    hipLaunchKernelGGL(dot_kernel, dim3(SSQ_BLOCKS), dim3(256), 0, st, a, b, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, workspace, SSQ_BLOCKS, out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_softmax_vec_f32(const float* w, int n, float* s, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || n > 1024) return MI_ERR_ARG;
    hipLaunchKernelGGL(softmax_vec_kernel, dim3(1), dim3(64), 0, st, w, n, s);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_softmax_vec_bwd_f32(const float* s, const float* g, int n, float* dw, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || n > 1024) return MI_ERR_ARG;
    hipLaunchKernelGGL(softmax_vec_bwd_kernel, dim3(1), dim3(64), 0, st, s, g, n, dw);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_clip_coef(const float* sumsq, float max_norm, float skip_above, float* norm_coef_skip, hipStream_t st) {
    MI_ENTER();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, st, sumsq, max_norm, skip_above, norm_coef_skip);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
extern "C" int mi_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* decay, long n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, const float* norm_coef, void* mirror_bf16, hipStream_t st) {
    MI_ENTER();
    if (n <= 0 || step < 1) return MI_ERR_ARG;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, decay, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       norm_coef, (bf16_t*)mirror_bf16);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
