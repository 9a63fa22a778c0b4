// Loss gradients on gfx950 (training step):
//   * CTC (e_branchformer.py:472-488, F.ctc_loss): alpha/beta recursions per utterance, then
//       dlogit[t,c] = scale_b * ( softmax(logits)[t,c] - exp(logsum_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll_b - lp[t,c]) )
//     (the gradient torch's ctc_loss_backward composes with log_softmax), zero for t >= input length,
//   * label-smoothed cross entropy of the decoder heads (multi_head_gpt2.py:138-158, CrossEntropyLoss(label_smoothing)),
//   * token / position embedding scatter (embeddings.py:33-62, GPT-2 wte / wpe).
#include "common.hpp"

namespace {

__device__ __forceinline__ float lse3f(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// one block per utterance: alpha forward (stored), beta backward, per-state contributions
//   contrib[b][t][s] = -scale_b * exp(alpha_t(s) + beta_t(s) + nll_b - lp_t(s));   meta[b] = {tl, Tb, scale_b as float bits}
template <typename T>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(const T* __restrict__ logits, long ld_b, long ld_t, const float* __restrict__ lse,
                                                              int Tmax, const long* __restrict__ labels, int U, const int* __restrict__ in_len,
                                                              int blank, const float* __restrict__ nll, int reduction, int B, float gscale,
                                                              float* __restrict__ alpha_ws, float* __restrict__ contrib,
                                                              int* __restrict__ ext_ws, int* __restrict__ meta, int* __restrict__ chain_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S_max = 2 * U + 1;
    int* hdr = reinterpret_cast<int*>(smem);          // [4]
    int* ext = hdr + 4;                               // [S_max]
    float* buf = reinterpret_cast<float*>(ext + S_max);   // [2][S_max]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int n = 0;
        for (int u = 0; u < U; ++u) {
            const long v = labels[(long)b * U + u];
            if (v >= 0) { ext[2 * n + 1] = (int)v; ++n; }
        }
        for (int s = 0; s <= 2 * n; s += 2) ext[s] = blank;
        hdr[0] = n;
    }
    __syncthreads();
    const int tl = hdr[0], S = 2 * tl + 1;
    const int Tb = min(in_len[b], Tmax);
    const float nl = nll[b];
    float scale = (reduction == 1) ? gscale / ((float)max(tl, 1) * (float)B) : gscale;
    if (!isfinite(nl) || Tb <= 0) scale = 0.f;        // infeasible alignment: zero gradient (zero_infinity semantics)
    for (int s = tid; s < S; s += 256) ext_ws[(long)b * S_max + s] = ext[s];
    // chain of the states that emit the same label (repeated labels in the target): chain[s] = next odd s' > s with ext[s'] == ext[s] (0: none), bit 30 set when an
    // earlier state has the label — ctc_grad_rows_kernel lets the FIRST state of a label add the whole chain in order (no LDS float atomics: bit-reproducible rows)
    for (int s = tid; s < S; s += 256) {
        int code = 0;
        if (s & 1) {
            const int v = ext[s];
            for (int q = s + 2; q < S; q += 2) if (ext[q] == v) { code = q; break; }
            for (int q = 1; q < s; q += 2) if (ext[q] == v) { code |= 0x40000000; break; }
        }
        chain_ws[(long)b * S_max + s] = code;
    }
    if (tid == 0) { meta[3 * b] = tl; meta[3 * b + 1] = Tb; meta[3 * b + 2] = __float_as_int(scale); }
    if (scale == 0.f) return;
    const T* lg = logits + (long)b * ld_b;
    const float* ls = lse + (long)b * Tmax;
    float* aw = alpha_ws + (long)b * Tmax * S_max;
    float* cw = contrib + (long)b * Tmax * S_max;
    // ---- alpha
    for (int s = tid; s < S; s += 256) {
        const float v = (s < 2) ? (float)lg[ext[s]] - ls[0] : -INFINITY;
        buf[s] = v; aw[s] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < Tb; ++t) {
        const float* a = buf + cur * S_max;
        float* an = buf + (cur ^ 1) * S_max;
        const T* lt = lg + (long)t * ld_t;
        const float l0 = ls[t];
        for (int s = tid; s < S; s += 256) {
            const bool skip = s >= 2 && ext[s] != blank && ext[s] != ext[s - 2];
            const float v = lse3f(a[s], s >= 1 ? a[s - 1] : -INFINITY, skip ? a[s - 2] : -INFINITY) + ((float)lt[ext[s]] - l0);
            an[s] = v; aw[(long)t * S_max + s] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- beta (buf reused)
    {
        const int t = Tb - 1;
        const T* lt = lg + (long)t * ld_t;
        for (int s = tid; s < S; s += 256) {
            const float lp = (float)lt[ext[s]] - ls[t];
            const float v = (s >= S - 2) ? lp : -INFINITY;
            buf[s] = v;
            cw[(long)t * S_max + s] = -scale * __expf(aw[(long)t * S_max + s] + v + nl - lp);
        }
        __syncthreads();
    }
    cur = 0;
    for (int t = Tb - 2; t >= 0; --t) {
        const float* bt = buf + cur * S_max;
        float* bn = buf + (cur ^ 1) * S_max;
        const T* lt = lg + (long)t * ld_t;
        const float l0 = ls[t];
        for (int s = tid; s < S; s += 256) {
            const bool skip = s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s];
            const float lp = (float)lt[ext[s]] - l0;
            const float v = lse3f(bt[s], s + 1 < S ? bt[s + 1] : -INFINITY, skip ? bt[s + 2] : -INFINITY) + lp;
            bn[s] = v;
            cw[(long)t * S_max + s] = -scale * __expf(aw[(long)t * S_max + s] + v + nl - lp);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Wave form of ctc_alpha_beta_kernel for S_max <= 128 and T <= 256 (every 10 s clip: 250 frames): the emissions lp[t][s] are gathered into LDS by the whole block first
// (the only HBM traffic), then wave 0 runs alpha forward and wave 1 runs beta backward AT THE SAME TIME, two states per lane in registers, neighbours by whole-wave DPP
// shifts, no barrier and no global load inside a time step (the block form above pays a dependent gather + a barrier per step, twice over the sequence).  alpha_t goes to
// `alpha_ws`, beta_t into `contrib` as scratch; the block then turns contrib into -scale * exp(alpha + beta + nll - lp) in place.  Same recursions, same outputs.
__device__ __forceinline__ float w_shr1(float v, float fill) {   // lane i <- lane i-1, lane 0 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ float w_shl1(float v, float fill) {   // lane i <- lane i+1, lane 63 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, false));
}

// LPG (T > 256: the emissions of 500 frames x 128 states are 256 KB, more than the LDS): the table lives in a global scratch (`lp_ws`, L2-resident: the block that wrote it reads
// it) and the two recursions keep the emissions of the next PF steps in registers — the loads do not depend on the recursion, so eight steps (~2 us) ahead covers the L2
// round trip.  The block form above, the only one for such inputs until round 4, pays a dependent gather and a barrier per step, alpha THEN beta: 653 us at config 3.
template <typename T, bool LPG>
__global__ __launch_bounds__(256) void ctc_alpha_beta_wave_kernel(const T* __restrict__ logits, long ld_b, long ld_t, const float* __restrict__ lse,
                                                                   int Tmax, const long* __restrict__ labels, int U, const int* __restrict__ in_len,
                                                                   int blank, const float* __restrict__ nll, int reduction, int B, float gscale,
                                                                   float* __restrict__ alpha_ws, float* __restrict__ contrib,
                                                                   int* __restrict__ ext_ws, int* __restrict__ meta, int* __restrict__ chain_ws, float* __restrict__ lp_ws,
                                                                   float* __restrict__ nll_out, int* __restrict__ tl_out) {
    // nll == NULL (own-nll form, mi_ctc_loss_bwd_nll): the utterance's negative log-likelihood is taken from THIS kernel's alpha recursion (-logsumexp of the two final
    // states at Tb - 1) and written to nll_out / tl_out — the training step then needs no forward alpha kernel at all (it ran the same recursion a second time)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S_max = 2 * U + 1;
    int* hdr = reinterpret_cast<int*>(smem);          // [4]
    int* ext = hdr + 4;                               // [132]: blank sentinels behind the last state
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* lp = LPG ? lp_ws + (long)b * Tmax * 128 : reinterpret_cast<float*>(ext + 132);  // [Tmax][128]
    constexpr int PF = LPG ? 8 : 1;                   // steps of emissions held ahead of the recursion
    if (tid == 0) {
        int n = 0;
        for (int u = 0; u < U; ++u) {
            const long v = labels[(long)b * U + u];
            if (v >= 0) { ext[2 * n + 1] = (int)v; ++n; }
        }
        for (int s = 0; s <= 2 * n; s += 2) ext[s] = blank;
        for (int s = 2 * n + 1; s < 132; ++s) ext[s] = blank;
        hdr[0] = n;
    }
    __syncthreads();
    const int tl = hdr[0], S = 2 * tl + 1;
    const int Tb = min(in_len[b], Tmax);
    const bool own = nll == nullptr;
    float nl = own ? 0.f : nll[b];
    float scale = (reduction == 1) ? gscale / ((float)max(tl, 1) * (float)B) : gscale;
    if (!isfinite(nl) || Tb <= 0) scale = 0.f;        // infeasible alignment: zero gradient (zero_infinity semantics)
    if (own && tid == 0) { tl_out[b] = tl; if (Tb <= 0) nll_out[b] = (tl == 0) ? 0.f : INFINITY; }      // (as mi_ctc_loss_fwd for an empty input)
    for (int s = tid; s < S; s += 256) ext_ws[(long)b * S_max + s] = ext[s];
    // chain of the states that emit the same label (repeated labels in the target): chain[s] = next odd s' > s with ext[s'] == ext[s] (0: none), bit 30 set when an
    // earlier state has the label — ctc_grad_rows_kernel lets the FIRST state of a label add the whole chain in order (no LDS float atomics: bit-reproducible rows)
    for (int s = tid; s < S; s += 256) {
        int code = 0;
        if (s & 1) {
            const int v = ext[s];
            for (int q = s + 2; q < S; q += 2) if (ext[q] == v) { code = q; break; }
            for (int q = 1; q < s; q += 2) if (ext[q] == v) { code |= 0x40000000; break; }
        }
        chain_ws[(long)b * S_max + s] = code;
    }
    if (tid == 0) { meta[3 * b] = tl; meta[3 * b + 1] = Tb; meta[3 * b + 2] = __float_as_int(scale); }      // (own-nll form: the scale is confirmed after the recursion)
    if (scale == 0.f && (!own || Tb <= 0)) return;    // (own-nll form with a zero upstream gradient: the recursion still runs, for the loss)
    const T* lg = logits + (long)b * ld_b;
    const float* ls = lse + (long)b * Tmax;
    float* aw = alpha_ws + (long)b * Tmax * S_max;
    float* cw = contrib + (long)b * Tmax * S_max;
    {   // gather log p(t, ext[s]), 16 loads in flight per thread
        constexpr int UN = 16;
        const int s = tid & 127, tsub = tid >> 7;
        const bool sv = s < S;
        const long col = sv ? ext[s] : blank;
        for (int t = tsub; t < Tb; t += 2 * UN) {
            float v[UN], l[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int tt = min(t + 2 * u, Tb - 1);
                v[u] = (float)lg[(long)tt * ld_t + col];
                l[u] = ls[tt];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int tt = t + 2 * u;
                if (tt < Tb) lp[tt * 128 + s] = sv ? v[u] - l[u] : -INFINITY;
            }
        }
    }
    __syncthreads();
    const int s0 = lane, s1 = lane + 64;
    if (wave == 0) {                                   // ---- alpha, t = 0 .. Tb-1
        const bool skip0 = s0 >= 2 && s0 < S && ext[s0] != blank && ext[s0] != ext[s0 - 2];
        const bool skip1 = s1 < S && ext[s1] != blank && ext[s1] != ext[s1 - 2];
        float a0 = (s0 < 2 && s0 < S) ? lp[s0] : -INFINITY, a1 = -INFINITY;
        if (s0 < S) aw[s0] = a0;
        if (s1 < S) aw[s1] = a1;
        float r0[PF], r1[PF];                          // emissions of steps tb .. tb + PF - 1 (slot k = step tb + k), refilled for step + PF as they are used
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int tt = min(1 + k, Tb - 1); r0[k] = lp[tt * 128 + s0]; r1[k] = lp[tt * 128 + s1]; }
        for (int tb = 1; tb < Tb; tb += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = tb + k;
                if (t >= Tb) break;
                const float l0 = r0[k], l1 = r1[k];
                const int tn = min(t + PF, Tb - 1);
                r0[k] = lp[tn * 128 + s0]; r1[k] = lp[tn * 128 + s1];
                const float top = __shfl(a0, 63, 64), top2 = __shfl(a0, 62, 64);
                const float p0 = w_shr1(a0, -INFINITY), q0 = w_shr1(p0, -INFINITY);
                const float p1 = w_shr1(a1, top);
                float q1 = w_shr1(p1, top2);
                if (lane == 1) q1 = top;
                const float n0 = lse3f(a0, p0, skip0 ? q0 : -INFINITY) + l0;
                const float n1 = lse3f(a1, p1, skip1 ? q1 : -INFINITY) + l1;
                a0 = (s0 < S) ? n0 : -INFINITY;
                a1 = (s1 < S) ? n1 : -INFINITY;
                if (s0 < S) aw[(long)t * S_max + s0] = a0;
                if (s1 < S) aw[(long)t * S_max + s1] = a1;
            }
        }
    } else if (wave == 1) {                            // ---- beta, t = Tb-1 .. 0 (stored in contrib)
        const bool sk0 = s0 + 2 < S && ext[s0 + 2] != blank && ext[s0 + 2] != ext[s0];
        const bool sk1 = s1 + 2 < S && ext[s1 + 2] != blank && ext[s1 + 2] != ext[s1];
        float l0 = lp[(Tb - 1) * 128 + s0], l1 = lp[(Tb - 1) * 128 + s1];
        float b0 = (s0 < S && s0 >= S - 2) ? l0 : -INFINITY, b1 = (s1 < S && s1 >= S - 2) ? l1 : -INFINITY;
        if (s0 < S) cw[(long)(Tb - 1) * S_max + s0] = b0;
        if (s1 < S) cw[(long)(Tb - 1) * S_max + s1] = b1;
        float r0[PF], r1[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int tt = max(Tb - 2 - k, 0); r0[k] = lp[tt * 128 + s0]; r1[k] = lp[tt * 128 + s1]; }
        for (int tb = Tb - 2; tb >= 0; tb -= PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = tb - k;
                if (t < 0) break;
                l0 = r0[k]; l1 = r1[k];
                const int tn = max(t - PF, 0);
                r0[k] = lp[tn * 128 + s0]; r1[k] = lp[tn * 128 + s1];
                const float bot = __shfl(b1, 0, 64), bot2 = __shfl(b1, 1, 64);
                const float p0 = w_shl1(b0, bot), q0 = w_shl1(p0, bot2);
                const float p1 = w_shl1(b1, -INFINITY), q1 = w_shl1(p1, -INFINITY);
                const float n0 = lse3f(b0, p0, sk0 ? q0 : -INFINITY) + l0;
                const float n1 = lse3f(b1, p1, sk1 ? q1 : -INFINITY) + l1;
                b0 = (s0 < S) ? n0 : -INFINITY;
                b1 = (s1 < S) ? n1 : -INFINITY;
                if (s0 < S) cw[(long)t * S_max + s0] = b0;
                if (s1 < S) cw[(long)t * S_max + s1] = b1;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (own) {                                         // log-likelihood = logsumexp(alpha_{Tb-1}[S-1], alpha_{Tb-1}[S-2]) from the rows wave 0 has just stored
        const float va = aw[(long)(Tb - 1) * S_max + S - 1];
        const float vb = S >= 2 ? aw[(long)(Tb - 1) * S_max + S - 2] : -INFINITY;
        nl = -lse3f(va, vb, -INFINITY);
        if (tid == 0) nll_out[b] = nl;
        if (!isfinite(nl)) {                           // infeasible alignment: zero gradient (ctc_grad_rows_kernel does not read `contrib` for a zero scale)
            if (tid == 0) meta[3 * b + 2] = __float_as_int(0.f);
            return;
        }
    }
    // ---- contributions: every (t, s) of this utterance, whole block
    for (int i = tid; i < Tb * S; i += 256) {
        const int t = i / S, s2 = i - t * S;
        const long o = (long)t * S_max + s2;
        cw[o] = -scale * __expf(aw[o] + cw[o] + nl - lp[t * 128 + s2]);
    }
}

// one block per (b, t) row: dense softmax part + scatter of the per-state contributions -> bf16 gradient row (pad columns zero)
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_rows_kernel(const T* __restrict__ logits, long ld_b, long ld_t, const float* __restrict__ lse,
                                                             int Tmax, int U, int V1, const float* __restrict__ contrib,
                                                             const int* __restrict__ ext_ws, const int* __restrict__ meta, const int* __restrict__ chain_ws,
                                                             bf16_t* __restrict__ out, long ldo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);      // [V1]
    const int b = blockIdx.x / Tmax, t = blockIdx.x % Tmax, tid = threadIdx.x;
    const int S_max = 2 * U + 1;
    const int tl = meta[3 * b], Tb = meta[3 * b + 1];
    const float scale = __int_as_float(meta[3 * b + 2]);
    bf16_t* o = out + ((long)b * Tmax + t) * ldo;
    if (t >= Tb || scale == 0.f) {
        for (int c = tid; c < ldo; c += 256) o[c] = (bf16_t)0.f;
        return;
    }
    const T* lt = logits + (long)b * ld_b + (long)t * ld_t;
    const float l0 = lse[(long)b * Tmax + t];
    for (int c = tid; c < V1; c += 256) row[c] = scale * __expf((float)lt[c] - l0);
    __syncthreads();
    const int S = 2 * tl + 1;
    const float* cw = contrib + ((long)b * Tmax + t) * S_max;
    const int* ext = ext_ws + (long)b * S_max;
    const int* chain = chain_ws + (long)b * S_max;
    // labels (odd states): the first state of each label adds its chain in state order — one writer per vocabulary entry
    for (int s = 2 * tid + 1; s < S; s += 512) {
        const int code = chain[s];
        if (code & 0x40000000) continue;
        float acc = cw[s];
        for (int q = code; q != 0; q = chain[q] & 0x3FFFFFFF) acc += cw[q];
        row[ext[s]] += acc;
    }
    // blanks (even states): strided partial sums, wave sums by DPP, the four waves in order
    {
        __shared__ float bred[4];
        float bs = 0.f;
        for (int s = 2 * tid; s < S; s += 512) bs += cw[s];
        bs = wave_sum(bs);
        if ((tid & 63) == 0) bred[tid >> 6] = bs;
        __syncthreads();
        if (tid == 0) row[ext[0]] += (bred[0] + bred[1]) + (bred[2] + bred[3]);
    }
    __syncthreads();
    for (int c = tid; c < ldo; c += 256) o[c] = (c < V1) ? f2bf(row[c]) : (bf16_t)0.f;
}

// d/dlogits of  weight * mean_valid( (1-eps) nll + eps * (-mean_c logp) ),  target of row (b,u) = labels[b, u+shift]
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels, int B, int U,
                                                      int shift, int V, float eps, float weight, const float* __restrict__ acc /* [sum, count] */,
                                                      bf16_t* __restrict__ out, long ldo) {
    __shared__ float red[8];
    const int rowi = blockIdx.x, b = rowi / U, u = rowi % U, tid = threadIdx.x;
    bf16_t* o = out + (long)rowi * ldo;
    const long tgt = (u + shift < U) ? labels[(long)b * U + u + shift] : -100;
    const float cnt = acc[1];
    if (tgt < 0 || cnt <= 0.f) {
        for (int c = tid; c < ldo; c += 256) o[c] = (bf16_t)0.f;
        return;
    }
    const float* x = logits + (long)rowi * ld;
    float mx = -INFINITY;
    for (int c = tid; c < V; c += 256) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = tid; c < V; c += 256) s += __expf(x[c] - mx);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
    __syncthreads();
    s = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.f / s, k = weight / cnt, sm = eps / V;
    for (int c = tid; c < ldo; c += 256) {
        float g = 0.f;
        if (c < V) g = k * (__expf(x[c] - mx) * inv - sm - ((c == tgt) ? (1.f - eps) : 0.f));
        o[c] = f2bf(g);
    }
}

// dwte[ids[m]] += scale * dx[m];  dwpe[pos_offset + m % U] += dx[m]  (learned positions only) — as GATHERS, so that no two threads ever add to one element (the scatter
// with float atomics was the last unordered sum of the decoder's backward): block = one vocabulary entry (or one position); it scans the M token ids, and every thread
// adds the matching rows of its columns in row order.
// block = EW_IDS consecutive vocabulary entries: it walks the M token ids ONCE (chunks of 256, ordered compaction of the chunk's matches by ballot + wave counts), and
// thread t adds the matching rows' columns t, t + 256, ... to the entry's accumulator in row order.  (A block per entry scanning all ids itself: 0.79 ms at config 3.)
constexpr int EW_IDS = 16;
__global__ __launch_bounds__(256) void embed_bwd_wte_kernel(const long* __restrict__ ids, const float* __restrict__ dx, float scale, int d, int M, int V,
                                                             float* __restrict__ dwte, const unsigned char* __restrict__ used, int heavy_id) {
    const int v0 = blockIdx.x * EW_IDS;
    __shared__ int hit_m[256];
    __shared__ unsigned char hit_v[256];
    __shared__ int wcount[4];
    __shared__ int any_used;
    if (threadIdx.x == 0) {
        int a = used ? 0 : 1;
        for (int j = 0; used && j < EW_IDS && v0 + j < V; ++j) a |= used[v0 + j];
        any_used = a;
    }
    __syncthreads();
    if (!any_used) return;
    float acc[EW_IDS][4];                                     // [local entry][column t + 256 j] (d <= 1024)
#pragma unroll
    for (int e = 0; e < EW_IDS; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[e][j] = 0.f;
    for (int m0 = 0; m0 < M; m0 += 256) {
        const int m = m0 + threadIdx.x;
        const long id = m < M ? ids[m] : -1;
        const bool match = id >= v0 && id < v0 + EW_IDS && id < V && id != heavy_id;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(match);
        __syncthreads();                                      // the previous chunk's hits have been consumed
        if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = __builtin_popcountll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wcount[w];
        if (match) {
            const int slot = base + __builtin_popcountll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
            hit_m[slot] = m; hit_v[slot] = (unsigned char)(id - v0);
        }
        const int n = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
        for (int h = 0; h < n; ++h) {                         // in row order: the sum of an entry's rows is a fixed sequence of adds
            const float* src = dx + (long)hit_m[h] * d;
            const int e = hit_v[h];
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int c = threadIdx.x + 256 * j; v[j] = c < d ? src[c] : 0.f; }
#pragma unroll
            for (int ee = 0; ee < EW_IDS; ++ee)
                if (ee == e) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ee][j] += v[j];
                }
        }
    }
#pragma unroll
    for (int e = 0; e < EW_IDS; ++e) {
        if (v0 + e >= V) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = threadIdx.x + 256 * j; if (c < d && acc[e][j] != 0.f) dwte[(long)(v0 + e) * d + c] += scale * acc[e][j]; }
    }
}
// The ONE entry that takes a large share of the rows — the padding token the shifted decoder input is filled with (thousands of rows at config 3) — is not gathered by one
// block (a sequential walk over its rows: 1 ms) but summed as a masked column sum: block (64 columns, 128-row chunk) leaves a partial row, rows_reduce_kernel adds the chunks
// in order.
__global__ __launch_bounds__(256) void embed_bwd_heavy_kernel(const long* __restrict__ ids, const float* __restrict__ dx, int d, int M, long heavy_id, float* __restrict__ partial) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * 128, m1 = min(M, m0 + 128);
    float s = 0.f;
    if (c < d)
        for (int m = m0 + ty; m < m1; m += 4)
            if (ids[m] == heavy_id) s += dx[(long)m * d + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < d) partial[(long)blockIdx.y * d + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}
struct EmitScaled { float* out; float scale; __device__ void operator()(int i, float v) const { out[i] += scale * v; } };
__global__ __launch_bounds__(256) void embed_bwd_mark_kernel(const long* __restrict__ ids, int M, int V, unsigned char* __restrict__ used) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < M) { const long id = ids[m]; if (id >= 0 && id < V) used[id] = 1; }      // same value from every writer: no race to lose
}
__global__ __launch_bounds__(256) void embed_bwd_wpe_kernel(const float* __restrict__ dx, int pos_offset, int U, int d, int M, float* __restrict__ dwpe) {
    const int u = blockIdx.x;                                 // rows u, u + U, u + 2U, ... (one per utterance), added in order
    for (int c = threadIdx.x; c < d; c += 256) {
        float acc = 0.f;
        for (int m = u; m < M; m += U) acc += dx[(long)m * d + c];
        dwpe[(long)(pos_offset + u) * d + c] += acc;
    }
}

}  // namespace

extern "C" size_t mi_ctc_bwd_workspace_bytes(int B, int T, int U) {
    const size_t S = 2 * (size_t)U + 1;
    const size_t lpg = (S <= 128 && T > 256) ? (size_t)B * T * 128 * sizeof(float) : 0;        // the long-input wave form's emission table
    return 2 * (size_t)B * T * S * sizeof(float) + 2 * (size_t)B * S * sizeof(int) + (size_t)3 * B * sizeof(int) + 1024 + lpg;
}

// logits (B,T,V1) f32 (dtype 0) | bf16 (1) with strides; lse (B*T) and nll (B) from mi_row_lse / mi_ctc_loss_fwd;
// dlogits (B*T, ldo) bf16, ldo >= V1 (pad columns zeroed); gscale = upstream gradient of the reduced loss (e.g. ctc_weight)
static int ctc_loss_bwd_impl(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T, const long* labels, int U,
                             const int* in_len, int blank, int B, int reduction, const float* nll, float gscale, void* workspace,
                             size_t workspace_bytes, void* dlogits, long ldo, float* nll_out, int* tl_out, hipStream_t st) {
    const int V1 = blank + 1;
    if (B <= 0 || T <= 0 || U < 0 || ldo < V1 || dtype < 0 || dtype > 1) return MI_ERR_ARG;
    if (workspace_bytes < mi_ctc_bwd_workspace_bytes(B, T, U)) return MI_ERR_ARG;
    const size_t S = 2 * (size_t)U + 1;
    float* alpha_ws = (float*)workspace;
    float* contrib = alpha_ws + (size_t)B * T * S;
    int* ext_ws = (int*)(contrib + (size_t)B * T * S);
    int* meta = ext_ws + (size_t)B * S;
    int* chain_ws = meta + (size_t)3 * B;
    const size_t lds = (4 + S) * sizeof(int) + 2 * S * sizeof(float);
    const size_t lds_rows = (size_t)V1 * sizeof(float);
    if (lds > 150 * 1024 || lds_rows > 150 * 1024) return MI_ERR_UNSUPPORTED;
    const bool wavef = S <= 128 && T <= 256;          // the wave form: states in one wave's registers, all emissions in LDS
    const bool wavel = S <= 128 && T > 256;           // ... and in an L2-resident global table for longer inputs
    if (!nll && !(wavef || wavel)) return MI_ERR_UNSUPPORTED;        // the own-nll form exists in the wave kernels only
    const size_t ldsw = (4 + 132) * sizeof(int) + (size_t)T * 128 * sizeof(float);
    float* lp_ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(chain_ws + (size_t)B * S) + 255) & ~(uintptr_t)255);      // inside the 1024 bytes of slack
#define CTC_AB(TY) do { \
        if (wavef) hipLaunchKernelGGL((ctc_alpha_beta_wave_kernel<TY, false>), dim3(B), dim3(256), ldsw, st, (const TY*)logits, ld_b, ld_t, lse, T, labels, U, in_len, \
                                      blank, nll, reduction, B, gscale, alpha_ws, contrib, ext_ws, meta, chain_ws, (float*)nullptr, nll_out, tl_out); \
        else if (wavel) hipLaunchKernelGGL((ctc_alpha_beta_wave_kernel<TY, true>), dim3(B), dim3(256), (4 + 132) * sizeof(int), st, (const TY*)logits, ld_b, ld_t, lse, T, labels, U, in_len, \
                                      blank, nll, reduction, B, gscale, alpha_ws, contrib, ext_ws, meta, chain_ws, lp_ws, nll_out, tl_out); \
        else hipLaunchKernelGGL(ctc_alpha_beta_kernel<TY>, dim3(B), dim3(256), lds, st, (const TY*)logits, ld_b, ld_t, lse, T, labels, U, in_len, \
                                blank, nll, reduction, B, gscale, alpha_ws, contrib, ext_ws, meta, chain_ws); \
        hipLaunchKernelGGL(ctc_grad_rows_kernel<TY>, dim3(B * T), dim3(256), lds_rows, st, (const TY*)logits, ld_b, ld_t, lse, T, U, V1, \
                           contrib, ext_ws, meta, chain_ws, (bf16_t*)dlogits, ldo); } while (0)
    if (dtype == 0) CTC_AB(float); else CTC_AB(bf16_t);
#undef CTC_AB
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_ctc_loss_bwd(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T, const long* labels, int U,
                               const int* in_len, int blank, int B, int reduction, const float* nll, float gscale, void* workspace,
                               size_t workspace_bytes, void* dlogits, long ldo, hipStream_t st) {
    MI_ENTER();
    if (!nll) return MI_ERR_ARG;
    return ctc_loss_bwd_impl(logits, ld_b, ld_t, dtype, lse, T, labels, U, in_len, blank, B, reduction, nll, gscale, workspace, workspace_bytes, dlogits, ldo, nullptr, nullptr, st);
}

// Loss AND gradient from one pair of recursions (the training step): as mi_ctc_loss_bwd, but the per-utterance negative log-likelihood comes out of this call's own alpha
// recursion — nll_out (B), tgt_len_out (B), loss_out (1: the reduced loss of mi_ctc_loss_fwd, same reduction / zero_infinity semantics) — instead of going in: the forward
// loss kernel (a second run of the alpha recursion) is not needed.  MI_ERR_UNSUPPORTED when the target has more than 63 labels (the wave kernels hold 128 states): the
// caller runs mi_ctc_loss_fwd + mi_ctc_loss_bwd.
extern "C" int mi_ctc_reduce(const float* nll, const int* tgt_len, int B, int reduction, int zero_infinity, float* loss, hipStream_t stream);
extern "C" int mi_ctc_loss_bwd_nll(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T, const long* labels, int U,
                                   const int* in_len, int blank, int B, int reduction, int zero_infinity, float gscale, void* workspace,
                                   size_t workspace_bytes, void* dlogits, long ldo, float* nll_out, int* tgt_len_out, float* loss_out, hipStream_t st) {
    MI_ENTER();
    if (!nll_out || !tgt_len_out) return MI_ERR_ARG;
    const int rc = ctc_loss_bwd_impl(logits, ld_b, ld_t, dtype, lse, T, labels, U, in_len, blank, B, reduction, nullptr, gscale, workspace, workspace_bytes, dlogits, ldo,
                                     nll_out, tgt_len_out, st);
    if (rc != MI_OK) return rc;
    return loss_out ? mi_ctc_reduce(nll_out, tgt_len_out, B, reduction, zero_infinity, loss_out, st) : MI_OK;
}

// logits (B,U,V) f32 rows of stride ld; acc = [sum, count] written by mi_ce_label_smoothing; dlogits (B*U, ldo) bf16
extern "C" int mi_ce_label_smoothing_bwd(const float* logits, long ld, const long* labels, int B, int U, int shift, int V, float eps,
                                         float weight, const float* acc, void* dlogits, long ldo, hipStream_t st) {
    MI_ENTER();
    if (B <= 0 || U <= 0 || V <= 0 || ldo < V) return MI_ERR_ARG;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(B * U), dim3(256), 0, st, logits, ld, labels, B, U, shift, V, eps, weight, acc, (bf16_t*)dlogits, ldo);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// workspace: mi_embed_tokens_bwd_workspace_bytes(M, d, V) bytes (which vocabulary entries occur + the heavy entry's per-chunk partial rows).
// heavy_id >= 0: a vocabulary entry expected on a large share of the rows (the decoder input's padding token); -1: none.
extern "C" size_t mi_embed_tokens_bwd_workspace_bytes(int M, int d, int V) { return (((size_t)V + 255) / 256) * 256 + (size_t)cdiv(M, 128) * (size_t)d * sizeof(float); }
extern "C" int mi_embed_tokens_bwd(const long* ids, const float* dx, float scale, int pos_offset, int U, int d, int M, int V, float* dwte,
                                   float* dwpe, int heavy_id, void* workspace, hipStream_t st) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || U <= 0 || V <= 0 || d > 1024 || !workspace || heavy_id >= V) return MI_ERR_ARG;
    unsigned char* used = (unsigned char*)workspace;
    float* partial = reinterpret_cast<float*>(used + (((size_t)V + 255) / 256) * 256);
    if (hipMemsetAsync(used, 0, (size_t)V, st) != hipSuccess) return MI_ERR_LAUNCH;
    hipLaunchKernelGGL(embed_bwd_mark_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, ids, M, V, used);
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_bwd_wte_kernel, dim3((unsigned)cdiv(V, EW_IDS)), dim3(256), 0, st, ids, dx, scale, d, M, V, dwte, used, heavy_id);
    MI_CHECK_LAUNCH();
    if (heavy_id >= 0) {
        hipLaunchKernelGGL(embed_bwd_heavy_kernel, dim3(cdiv(d, 64), cdiv(M, 128)), dim3(256), 0, st, ids, dx, d, M, (long)heavy_id, partial);
        MI_CHECK_LAUNCH();
        rows_reduce_launch(partial, cdiv(M, 128), d, EmitScaled{dwte + (long)heavy_id * d, scale}, st);
        MI_CHECK_LAUNCH();
    }
    if (dwpe) {
        hipLaunchKernelGGL(embed_bwd_wpe_kernel, dim3((unsigned)(U < M ? U : M)), dim3(256), 0, st, dx, pos_offset, U, d, M, dwpe);
        MI_CHECK_LAUNCH();
    }
    return MI_OK;
}
