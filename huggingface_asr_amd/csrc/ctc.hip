// CTC head tail on gfx950: row log-sum-exp + alpha recursion (forward loss).
//
// Reference: e_branchformer.py:472-488 — labels >= 0 are the targets, log_softmax in fp32,
// F.ctc_loss(blank = last class, reduction = config, zero_infinity = config) with input lengths from the
// UN-padded conv formula (SURVEY.md §8a row 8').  log_softmax is never materialised: the loss only needs
// lse[b,t] and the logits at the blank / label columns, so the (B,T,V+1) tensor is read exactly once.
#include "common.hpp"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void row_lse_kernel(const T* __restrict__ x, long ld, int V, float* __restrict__ lse, int M) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (long)row * ld;
    constexpr int NR = 80;                          // rows up to 5120 columns (the 5001-class CTC head) stay in registers: ONE pass over HBM instead of two
    if (V <= NR * 64) {
        float v[NR];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < V ? (float)xr[c] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) s += __expf(v[i] - mx);        // exp(-inf) = 0 for the padding
        s = wave_sum(s);
        if (lane == 0) lse[row] = mx + __logf(s);
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, (float)xr[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < V; c += 64) s += __expf((float)xr[c] - mx);
    s = wave_sum(s);
    if (lane == 0) lse[row] = mx + __logf(s);
}

__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

// one block per utterance; extended label sequence and alpha (double) in LDS
template <typename T>
__global__ __launch_bounds__(256) void ctc_alpha_kernel(const T* __restrict__ logits, long ld_b, long ld_t,
                                                         const float* __restrict__ lse, int Tmax,
                                                         const long* __restrict__ labels, int U,
                                                         const int* __restrict__ in_len, int blank,
                                                         float* __restrict__ nll, int* __restrict__ tgt_len_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S_max = 2 * U + 1;
    int* s_tl_p = reinterpret_cast<int*>(smem);                       // [4] (all LDS in the dynamic region, 16-B carved)
    int* ext = s_tl_p + 4;                                            // [S_max]
    int* skip = ext + S_max;                                          // [S_max]
    double* alpha = reinterpret_cast<double*>(smem + (((4 + 2 * S_max) * sizeof(int) + 15) / 16) * 16);   // [2][S_max]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int n = 0;
        for (int u = 0; u < U; ++u) {
            const long v = labels[(long)b * U + u];
            if (v >= 0) { ext[2 * n + 1] = (int)v; ++n; }
        }
        for (int s = 0; s <= 2 * n; s += 2) ext[s] = blank;
        for (int s = 0; s <= 2 * n; ++s) skip[s] = (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) ? 1 : 0;
        s_tl_p[0] = n;
    }
    __syncthreads();
    const int tl = s_tl_p[0], S = 2 * tl + 1;
    const int Tb = min(in_len[b], Tmax);
    if (tid == 0) tgt_len_out[b] = tl;
    if (Tb <= 0) { if (tid == 0) nll[b] = (tl == 0) ? 0.f : INFINITY; return; }
    const T* lg = logits + (long)b * ld_b;
    const float* ls = lse + (long)b * Tmax;
    for (int s = tid; s < S; s += 256)
        alpha[s] = (s < 2) ? (double)((float)lg[ext[s]] - ls[0]) : -INFINITY;
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < Tb; ++t) {
        const double* a = alpha + cur * S_max;
        double* an = alpha + (cur ^ 1) * S_max;
        const T* lt = lg + (long)t * ld_t;
        const float l0 = ls[t];
        for (int s = tid; s < S; s += 256) {
            const double a0 = a[s], a1 = s >= 1 ? a[s - 1] : -INFINITY, a2 = skip[s] ? a[s - 2] : -INFINITY;
            an[s] = lse3(a0, a1, a2) + (double)((float)lt[ext[s]] - l0);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid == 0) {
        const double* a = alpha + cur * S_max;
        const double a0 = a[S - 1], a1 = S >= 2 ? a[S - 2] : -INFINITY;
        const double ll = lse3(a0, a1, -INFINITY);
        nll[b] = (float)(-ll);
    }
}

// Fast form for 2U+1 <= 128 states: the block first gathers every needed log-prob  lp[t][s] = logit[t][ext[s]] - lse[t]
// into LDS in parallel (the only HBM traffic of the loss), then ONE wave runs the alpha recursion with two states per
// lane in registers; neighbours come from whole-wave DPP shifts, so a time step costs no barrier and no memory access
// besides two LDS reads.  fp32 log-space math (as torch's GPU ctc_loss).
__device__ __forceinline__ float wave_shr1(float v, float fill) {   // lane i <- lane i-1, lane 0 <- fill
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
    return __int_as_float(r);
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {   // lane i <- lane i+1, lane 63 <- fill
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, false);
    return __int_as_float(r);
}
__device__ __forceinline__ float lse3f(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

template <typename T>
__global__ __launch_bounds__(256) void ctc_alpha_wave_kernel(const T* __restrict__ logits, long ld_b, long ld_t,
                                                              const float* __restrict__ lse, int Tmax,
                                                              const long* __restrict__ labels, int U,
                                                              const int* __restrict__ in_len, int blank, int tchunk,
                                                              float* __restrict__ nll, int* __restrict__ tgt_len_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* hdr = reinterpret_cast<int*>(smem);                 // [4]
    int* ext = hdr + 4;                                      // [128 + 2]: two blank sentinels behind the last state (the beta recursion looks at s + 2)
    float* gam = reinterpret_cast<float*>(ext + 132);        // [128]: what the backward half hands the forward half
    float* lp = gam + 128;                                   // [tchunk][128]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int n = 0;
        for (int u = 0; u < U; ++u) {
            const long v = labels[(long)b * U + u];
            if (v >= 0) { ext[2 * n + 1] = (int)v; ++n; }
        }
        for (int s = 0; s <= 2 * n; s += 2) ext[s] = blank;
        for (int s = 2 * n + 1; s < 130; ++s) ext[s] = blank;
        hdr[0] = n;
    }
    __syncthreads();
    const int tl = hdr[0], S = 2 * tl + 1;
    const int Tb = min(in_len[b], Tmax);
    if (tid == 0) tgt_len_out[b] = tl;
    if (Tb <= 0) { if (tid == 0) nll[b] = (tl == 0) ? 0.f : INFINITY; return; }
    const T* lg = logits + (long)b * ld_b;
    const float* ls = lse + (long)b * Tmax;
    const int s0 = lane, s1 = lane + 64;
    const bool skip0 = s0 >= 2 && s0 < S && ext[s0] != blank && ext[s0] != ext[s0 - 2];
    const bool skip1 = s1 < S && ext[s1] != blank && ext[s1] != ext[s1 - 2];
    float a0 = -INFINITY, a1 = -INFINITY;
    // Tb <= tchunk (every 10 s clip: 250 frames): the serial chain is halved — wave 0 runs alpha over t = 0..mid while wave 1 runs beta over t = Tb-1..mid+1, and
    //   -nll = logsumexp_s( alpha_mid(s) + gamma(s) ),   gamma(s) = logsumexp over the transitions s -> {s, s+1, s+2 if allowed} of beta_{mid+1}(.)
    // (beta_t includes the emission at t; the virtual beta_Tb is log 1 on the last state only).  Longer inputs keep the single forward chain over chunks.
    const bool bidir = Tb <= tchunk;
    const int mid = (Tb - 1) / 2;
    for (int tc = 0; tc < Tb; tc += tchunk) {
        const int nt = min(tchunk, Tb - tc);
        {   // gather log p(t, ext[s]): a thread keeps its state s (column ext[s]) and walks the time steps, 16 gathers in flight at a time —
            // with one block per utterance the memory round trips of a one-at-a-time loop were most of this kernel's time
            constexpr int UN = 16;
            const int s = tid & 127, tsub = tid >> 7;
            const bool sv = s < S;
            const long col = sv ? ext[s] : blank;
            for (int t = tsub; t < nt; t += 2 * UN) {
                float v[UN], l[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int tt = min(t + 2 * u, nt - 1);                 // clamped: branch-free loads
                    v[u] = (float)lg[(long)(tc + tt) * ld_t + col];
                    l[u] = ls[tc + tt];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int tt = t + 2 * u;
                    if (tt < nt) lp[tt * 128 + s] = sv ? v[u] - l[u] : -INFINITY;
                }
            }
        }
        __syncthreads();
        if (bidir && wave == 1) {
            const bool sk0 = s0 + 2 < S && ext[s0 + 2] != blank && ext[s0 + 2] != ext[s0];
            const bool sk1 = s1 + 2 < S && ext[s1 + 2] != blank && ext[s1 + 2] != ext[s1];
            // virtual beta at time Tb: probability 1 on the last state only — the transitions S-1 -> S-1 and S-2 -> S-1 then make both final states end with weight 1
            float b0 = (s0 == S - 1) ? 0.f : -INFINITY, b1 = (s1 == S - 1) ? 0.f : -INFINITY;
            auto trans = [&](float& n0, float& n1) {             // logsumexp over the successors of every state
                const float bot = __shfl(b1, 0, 64), bot2 = __shfl(b1, 1, 64);     // states 64, 65 feed states 63 / 62, 63
                const float p0 = wave_shl1(b0, bot), q0 = wave_shl1(p0, bot2);
                const float p1 = wave_shl1(b1, -INFINITY), q1 = wave_shl1(p1, -INFINITY);
                n0 = lse3f(b0, p0, sk0 ? q0 : -INFINITY);
                n1 = lse3f(b1, p1, sk1 ? q1 : -INFINITY);
            };
            float l0 = lp[(Tb - 1) * 128 + s0], l1 = lp[(Tb - 1) * 128 + s1];
            for (int t = Tb - 1; t > mid; --t) {
                const int tn = max(t - 1, 0);
                const float l0n = lp[tn * 128 + s0], l1n = lp[tn * 128 + s1];      // next step's emissions: in flight during this step's arithmetic
                float n0, n1;
                trans(n0, n1);
                b0 = (s0 < S) ? n0 + l0 : -INFINITY;
                b1 = (s1 < S) ? n1 + l1 : -INFINITY;
                l0 = l0n; l1 = l1n;
            }
            float g0, g1;
            trans(g0, g1);
            gam[s0] = (s0 < S) ? g0 : -INFINITY;
            gam[s1] = (s1 < S) ? g1 : -INFINITY;
        }
        if (wave == 0) {
            const int tend = bidir ? mid + 1 : nt;
            float l0n = lp[s0], l1n = lp[s1];
            for (int t = 0; t < tend; ++t) {
                const float l0 = l0n, l1 = l1n;
                const int tn = min(t + 1, nt - 1);
                l0n = lp[tn * 128 + s0]; l1n = lp[tn * 128 + s1];                  // prefetch: the LDS latency leaves the serial chain
                if (tc + t == 0) {
                    a0 = (s0 < 2 && s0 < S) ? l0 : -INFINITY;
                    a1 = -INFINITY;
                } else {
                    const float top = __shfl(a0, 63, 64);                 // state 63 feeds state 64
                    const float top2 = __shfl(a0, 62, 64);
                    const float p0 = wave_shr1(a0, -INFINITY);            // alpha[s-1]
                    const float q0 = wave_shr1(p0, -INFINITY);            // alpha[s-2]
                    const float p1 = wave_shr1(a1, top);
                    float q1 = wave_shr1(p1, top2);
                    if (lane == 1) q1 = top;
                    const float n0 = lse3f(a0, p0, skip0 ? q0 : -INFINITY) + l0;
                    const float n1 = lse3f(a1, p1, skip1 ? q1 : -INFINITY) + l1;
                    a0 = (s0 < S) ? n0 : -INFINITY;
                    a1 = (s1 < S) ? n1 : -INFINITY;
                }
            }
        }
        __syncthreads();
    }
    if (bidir) {
        if (wave == 0) {
            const float v0 = a0 + gam[s0], v1 = a1 + gam[s1];      // -inf + anything stays -inf (no NaN: gamma is never +inf)
            const float m = wave_max(fmaxf(v0, v1));
            const float sum = (m == -INFINITY) ? 0.f : wave_sum(__expf(v0 - m) + __expf(v1 - m));
            if (lane == 0) nll[b] = (m == -INFINITY) ? INFINITY : -(m + __logf(sum));
        }
        return;
    }
    if (wave == 0) {
        // log-likelihood = logsumexp(alpha[S-1], alpha[S-2])
        const int sa = S - 1, sb = S - 2;
        const float va = (sa >= 64) ? __shfl(a1, sa - 64, 64) : __shfl(a0, sa, 64);
        const float vb = (sb < 0) ? -INFINITY : ((sb >= 64) ? __shfl(a1, sb - 64, 64) : __shfl(a0, sb, 64));
        if (lane == 0) nll[b] = -lse3f(va, vb, -INFINITY);
    }
}

// reduction semantics of torch.nn.functional.ctc_loss: mean = mean_b(nll_b / max(tl_b,1)), sum, zero_infinity
__global__ void ctc_reduce_kernel(const float* nll, const int* tl, int B, int reduction, int zero_inf, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double acc = 0.0;
    for (int b = 0; b < B; ++b) {
        float v = nll[b];
        if (zero_inf && isinf(v)) v = 0.f;
        acc += (reduction == 1) ? (double)v / (double)max(tl[b], 1) : (double)v;
    }
    out[0] = (float)((reduction == 1) ? acc / B : acc);
}

}  // namespace

// the reduction of mi_ctc_loss_fwd alone: loss = mean_b(nll_b / max(tl_b, 1)) (reduction 1) or sum_b nll_b (0), infinite entries zeroed with zero_infinity
extern "C" int mi_ctc_reduce(const float* nll, const int* tgt_len, int B, int reduction, int zero_infinity, float* loss, hipStream_t stream) {
    MI_ENTER();
    if (!nll || !tgt_len || !loss || B <= 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(64), 0, stream, nll, tgt_len, B, reduction, zero_infinity, loss);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// logits (M, ld) f32 (dtype 0) or bf16 (dtype 1) -> lse (M) f32
extern "C" int mi_row_lse(const void* x, long ld, int dtype, int V, float* lse, int M, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || V <= 0) return MI_ERR_ARG;
    dim3 grid(cdiv(M, 4)), block(256);
    if (dtype == 0) hipLaunchKernelGGL(row_lse_kernel<float>, grid, block, 0, stream, (const float*)x, ld, V, lse, M);
    else if (dtype == 1) hipLaunchKernelGGL(row_lse_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)x, ld, V, lse, M);
    else return MI_ERR_ARG;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// logits (B, T, V+1) with strides (ld_b, ld_t) elements; lse (B*T); labels (B,U) int64, negatives = padding;
// in_len (B) int32; out: nll (B) f32, tgt_len (B) int32, loss (1) f32 with reduction 0 sum / 1 mean.
extern "C" int mi_ctc_loss_fwd(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T,
                               const long* labels, int U, const int* in_len, int blank, int B,
                               int reduction, int zero_infinity, float* nll, int* tgt_len, float* loss,
                               hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || T <= 0 || U < 0) return MI_ERR_ARG;
    const int S_max = 2 * U + 1;
    if (S_max <= 128) {
        const int tchunk = T < 256 ? T : 256;                      // 256 steps x 128 states x 4 B = 128 KiB
        const size_t ldsw = (4 + 132) * sizeof(int) + 128 * sizeof(float) + (size_t)tchunk * 128 * sizeof(float);
        if (dtype == 0)
            hipLaunchKernelGGL(ctc_alpha_wave_kernel<float>, dim3(B), dim3(256), ldsw, stream, (const float*)logits, ld_b, ld_t, lse, T,
                               labels, U, in_len, blank, tchunk, nll, tgt_len);
        else if (dtype == 1)
            hipLaunchKernelGGL(ctc_alpha_wave_kernel<bf16_t>, dim3(B), dim3(256), ldsw, stream, (const bf16_t*)logits, ld_b, ld_t, lse, T,
                               labels, U, in_len, blank, tchunk, nll, tgt_len);
        else return MI_ERR_ARG;
        MI_CHECK_LAUNCH();
        if (loss) {
            hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(64), 0, stream, nll, tgt_len, B, reduction, zero_infinity, loss);
            MI_CHECK_LAUNCH();
        }
        return MI_OK;
    }
    const size_t lds = (((4 + 2 * S_max) * sizeof(int) + 15) / 16) * 16 + 2 * S_max * sizeof(double);
    if (lds > 150 * 1024) return MI_ERR_UNSUPPORTED;
    if (dtype == 0)
        hipLaunchKernelGGL(ctc_alpha_kernel<float>, dim3(B), dim3(256), lds, stream, (const float*)logits, ld_b, ld_t, lse, T,
                           labels, U, in_len, blank, nll, tgt_len);
    else if (dtype == 1)
        hipLaunchKernelGGL(ctc_alpha_kernel<bf16_t>, dim3(B), dim3(256), lds, stream, (const bf16_t*)logits, ld_b, ld_t, lse, T,
                           labels, U, in_len, blank, nll, tgt_len);
    else return MI_ERR_ARG;
    MI_CHECK_LAUNCH();
    if (loss) {
        hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(64), 0, stream, nll, tgt_len, B, reduction, zero_infinity, loss);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}
