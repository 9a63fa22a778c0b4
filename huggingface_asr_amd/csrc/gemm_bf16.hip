// bf16 MFMA GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// This one kernel carries every nn.Linear of the E-Branchformer path (reference
// e_branchformer.py:96-98,139,212-216,247 / tf wav2vec2_conformer FFN :350-357 / extractors.py:108,131 /
// lm_head ⊕ blank_projection e_branchformer.py:456-457) and, with the implicit-im2col A loader,
// the second Conv2d of the sub-sampling front end (extractors.py:71-96).
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16): block 128x128x64, 4 waves as 2x2, each wave 64x64 =
// 2x2 MFMA tiles; A and W tiles are K-contiguous in HBM (nn.Linear layout), staged HBM->VGPR->LDS
// with the next tile's loads in flight under the current tile's MFMAs, LDS double-buffered
// (one barrier per K tile) and XOR-swizzled so that every ds_read_b128 lane group hits 16 distinct
// 16-B slots.  blockIdx is remapped so that the blocks of one XCD walk neighbouring N tiles of the
// same A row panel (A panel + W stay in that XCD's L2).
#include "gemm_args.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <bool CONV>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                 // [2][BM*BK]
    bf16_t* sB = sA + 2 * BM * BK;                                // [2][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective block remap (8 XCDs, round-robin dispatch): logical id walks N fastest.
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging assignment: 4 chunks (16 B) of A and of W per thread; rows (tid>>3)+32*i, chunk tid&7
    const int srow = tid >> 3, schunk = tid & 7;
    const bf16_t* aptr[4];
    bool aval[4];
    long abase[4];      // CONV: element offset of (b, to*stride - pad, fo*stride - pad, 0)
    int ati[4], afi[4];
    const bf16_t* wptr[4];
    bool wval[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        aval[i] = m < p.M;
        if (CONV) {
            const int mm = aval[i] ? m : 0;
            const int fo = mm % p.Fout, to = (mm / p.Fout) % p.Tout, b = mm / (p.Fout * p.Tout);
            ati[i] = to * p.stride - p.pad_t;
            afi[i] = fo * p.stride_f - p.pad_f;
            abase[i] = (long)b * p.Tin * p.Fin * p.Cin;
            aptr[i] = p.A;
        } else {
            aptr[i] = p.A + (long)(aval[i] ? m : 0) * p.lda + schunk * 8;
        }
        const int n = n0 + srow + 32 * i;
        wval[i] = n < p.N;
        wptr[i] = p.W + (long)(wval[i] ? n : 0) * p.ldw + schunk * 8;
    }

    bf16x8 ra[4], rb[4];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto gload = [&](int kt) {
        const int k = kt * BK + schunk * 8;          // this thread's K offset (same for its 4 rows)
        const bool kok = k < p.K;                    // K % 8 == 0: a 16-B chunk never straddles K
        int kh = 0, kw = 0, c0 = 0;
        if (CONV) {
            const int tap = k / p.Cin;               // Cin % 8 == 0: a chunk never straddles a tap
            c0 = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (CONV) {
                const int ti = ati[i] + kh, fi = afi[i] + kw;
                const bool ok = kok && aval[i] && ti >= 0 && ti < p.Tin && fi >= 0 && fi < p.Fin;
                ra[i] = ok ? *reinterpret_cast<const bf16x8*>(p.A + abase[i] + ((long)ti * p.Fin + fi) * p.Cin + c0)
                           : zero8;
            } else {
                ra[i] = (kok && aval[i]) ? *reinterpret_cast<const bf16x8*>(aptr[i] + kt * BK) : zero8;
            }
            rb[i] = (kok && wval[i]) ? *reinterpret_cast<const bf16x8*>(wptr[i] + kt * BK) : zero8;
        }
    };
    auto sstore = [&](int buf) {
        bf16_t* a = sA + buf * BM * BK;
        bf16_t* b = sB + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int off = row * BK + swz(row, schunk) * 8;
            *reinterpret_cast<bf16x8*>(a + off) = ra[i];
            *reinterpret_cast<bf16x8*>(b + off) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const bf16_t* a = sA + buf * BM * BK;
        const bf16_t* b = sB + buf * BN * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + lr;
                fa[i] = *reinterpret_cast<const bf16x8*>(a + row * BK + swz(row, ks * 2 + lh) * 8);
                const int col = wn * 64 + i * 32 + lr;
                fb[i] = *reinterpret_cast<const bf16x8*>(b + col * BK + swz(col, ks * 2 + lh) * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + lr;
        if (n >= p.N) continue;
        const float bcol = (p.bias_mode == 1) ? p.bias[n] : 0.f;
        const long nc = p.col_T ? (long)(n / p.col_T) * p.col_Tp + (n % p.col_T) : n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bcol;
                if (p.bias_mode == 2) v += p.bias[m];
                if (p.act == 1) v = gelu_erf(v);
                else if (p.act == 2) v = gelu_tanh(v);
                if (p.resid) v = p.resid[(long)m * p.ldr + n] + p.alpha * v;
                if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + nc] = v;
                else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + nc] = f2bf(v);
            }
        }
    }
}

}  // namespace
// optional per-launch HIP-event timing of the dense GEMM kernel (bench.py's roofline leg); see encoder.hip
extern "C" int mi_profile_hook_begin(hipStream_t stream, double flops);
extern "C" void mi_profile_hook_end(int slot, hipStream_t stream);
namespace {

int launch(const GemmArgs& a, bool conv, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 8) != 0) return MI_ERR_ARG;
    if (!conv && ((a.lda % 8) != 0)) return MI_ERR_ARG;
    if ((a.ldw % 8) != 0) return MI_ERR_ARG;
    if (conv && (a.Cin % 8) != 0) return MI_ERR_ARG;
    if (a.gated) {                        // fused gated epilogue: the 256-wide phase kernel or nothing (gemm_glds_supported routes eligible shapes to it)
        if (!conv || a.act != 1 || (a.N % 256) != 0 || !gemm_8p_supported(a, true)) return MI_ERR_UNSUPPORTED;
        const int slot = mi_profile_hook_begin(stream, 2.0 * a.M * a.N * a.K);
        const int rc = gemm_8p_launch(a, true, stream);
        if (slot >= 0) mi_profile_hook_end(slot, stream);
        if (rc != MI_OK) return rc;
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    if (gemm_glds_supported(a, conv)) {   // LDS-DMA pipelined fast path (gemm_glds.hip)
        const int slot = mi_profile_hook_begin(stream, 2.0 * a.M * a.N * a.K);     // the implicit-GEMM conv is a dense contraction too
        const int rc = gemm_glds_launch(a, conv, stream);
        if (slot >= 0) mi_profile_hook_end(slot, stream);
        if (rc != MI_OK) return rc;
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    const size_t lds = 2 * (BM + BN) * BK * sizeof(bf16_t);
    const int slot = mi_profile_hook_begin(stream, 2.0 * a.M * a.N * a.K);
    if (conv) launch_dense(PF_GENERIC, gemm_bf16_kernel<true>, dim3(grid), dim3(NT), lds, stream, a);
    else launch_dense(PF_GENERIC, gemm_bf16_kernel<false>, dim3(grid), dim3(NT), lds, stream, a);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace

extern "C" int mi_gemm_bf16_v(const void* A, long lda, const void* W, long ldw, const float* bias, int bias_mode,
                              void* C, long ldc, int out_f32, const float* resid, long ldr, float alpha, int act,
                              int M, int N, int K, int col_T, int col_Tp, int variant, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.variant = variant;
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw;
    a.bias = bias; a.bias_mode = bias ? bias_mode : 0;
    a.C = C; a.ldc = ldc; a.out_f32 = out_f32; a.resid = resid; a.ldr = ldr; a.alpha = alpha; a.act = act;
    a.M = M; a.N = N; a.K = K; a.col_T = col_T; a.col_Tp = col_Tp;
    return launch(a, false, stream);
}
extern "C" int mi_gemm_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, int bias_mode,
                            void* C, long ldc, int out_f32, const float* resid, long ldr, float alpha, int act,
                            int M, int N, int K, int col_T, int col_Tp, hipStream_t stream) {
    return mi_gemm_bf16_v(A, lda, W, ldw, bias, bias_mode, C, ldc, out_f32, resid, ldr, alpha, act, M, N, K, col_T, col_Tp, 0, stream);
}

// Conv2d (KHxKW, stride s, zero padding) over a channels-last bf16 activation as an implicit GEMM:
//   in  (B, Tin, Fin, Cin) bf16, weight (Cout, KH*KW*Cin) bf16 with k = (kh*KW + kw)*Cin + c,
//   out (B, Tout, Fout, Cout) bf16 = act(conv + bias).
extern "C" int mi_conv2d_cl_bf16_v(const void* in, const void* weight, const float* bias, void* out,
                                   int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW, int stride,
                                   int pad_t, int pad_f, int Tout, int Fout, int act, int variant, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.variant = variant;
    a.A = (const bf16_t*)in; a.lda = 0; a.W = (const bf16_t*)weight; a.ldw = (long)KH * KW * Cin;
    a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = out; a.ldc = Cout; a.out_f32 = 0; a.resid = nullptr; a.ldr = 0; a.alpha = 1.f; a.act = act;
    a.M = B * Tout * Fout; a.N = Cout; a.K = KH * KW * Cin;
    a.Tin = Tin; a.Fin = Fin; a.Cin = Cin; a.Tout = Tout; a.Fout = Fout; a.KW = KW; a.stride = stride; a.stride_f = stride;
    a.pad_t = pad_t; a.pad_f = pad_f;
    return launch(a, true, stream);
}
extern "C" int mi_conv2d_cl_bf16(const void* in, const void* weight, const float* bias, void* out,
                                 int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW, int stride,
                                 int pad_t, int pad_f, int Tout, int Fout, int act, hipStream_t stream) {
    return mi_conv2d_cl_bf16_v(in, weight, bias, out, B, Tin, Fin, Cin, Cout, KH, KW, stride, pad_t, pad_f, Tout, Fout, act, 0, stream);
}

// mi_conv2d_cl_bf16 with separate time / frequency strides (the (12,3) / stride (8,2) / padding (4,1) gate conv of GatedConv2dShared, extractors.py:41-47) and,
// with gated != 0, GatedConv2d as ONE implicit GEMM: `weight` (2*Cout, KH*KW*Cin) and `bias` (2*Cout) hold conv and gate interleaved in blocks of 32 channels
// ([conv c0..c0+31 ; gate c0..c0+31]); out (B,Tout,Fout,Cout) = act((conv + b) * sigmoid(gate + bg)).  gated: act must be 1 (GELU), Cout % 128 == 0, Cin % 64 == 0;
// anything else returns MI_ERR_UNSUPPORTED (run it with gated = 0, act = 0 on the same packing and mi_gated_act_bf16(blk = 32) behind it).
extern "C" int mi_conv2d_cl_geo_bf16(const void* in, const void* weight, const float* bias, void* out,
                                     int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW, int stride_t, int stride_f,
                                     int pad_t, int pad_f, int Tout, int Fout, int act, int gated, hipStream_t stream) {
    MI_ENTER();
    if (stride_t <= 0 || stride_f <= 0 || KH <= 0 || KW <= 0 || Tout <= 0 || Fout <= 0) return MI_ERR_ARG;
    if ((long)(Tout - 1) * stride_t - pad_t >= Tin || (long)(Fout - 1) * stride_f - pad_f >= Fin) return MI_ERR_ARG;       // every window starts inside the input
    GemmArgs a{};
    const int Nw = gated ? 2 * Cout : Cout;
    a.A = (const bf16_t*)in; a.lda = 0; a.W = (const bf16_t*)weight; a.ldw = (long)KH * KW * Cin;
    a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = out; a.ldc = Cout; a.out_f32 = 0; a.resid = nullptr; a.ldr = 0; a.alpha = 1.f; a.act = act;
    a.M = B * Tout * Fout; a.N = Nw; a.K = KH * KW * Cin;
    a.Tin = Tin; a.Fin = Fin; a.Cin = Cin; a.Tout = Tout; a.Fout = Fout; a.KW = KW; a.stride = stride_t; a.stride_f = stride_f;
    a.pad_t = pad_t; a.pad_f = pad_f; a.gated = gated ? 1 : 0;
    return launch(a, true, stream);
}

// ---- LayerNorm folded into the GEMMs around it (encoder.hip `ln_fold` path; algebra in gemm_args.hpp).
// Consumer: C (M,N) bf16 = act(LN(x) W^T + b) computed as act(rstd * (xb Wf^T) - rstd * mu * colsum + cbias), xb = bf16(x) (M,K), Wf = bf16(W diag(gamma)) (N,K),
// colsum_n = sum_k Wf[n,k], cbias = W beta + b; `stats`: per-row partial (sum, sumsq) pairs of x, row stride 32 floats, `npart` pairs (1 | 4..16 in steps of 4).
// Runs on the 256x256 phase kernel only: N % 256 == 0, K % 64 == 0, K >= 128; MI_ERR_UNSUPPORTED otherwise (the caller keeps the LayerNorm kernel + plain GEMM).
extern "C" int mi_gemm_lnfold_bf16(const void* xb, long lda, const void* Wf, long ldw, const float* colsum, const float* cbias, const float* stats, int npart, float eps,
                                   void* C, long ldc, int act, int M, int N, int K, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.A = (const bf16_t*)xb; a.lda = lda; a.W = (const bf16_t*)Wf; a.ldw = ldw; a.bias = cbias; a.bias_mode = 1;
    a.C = C; a.ldc = ldc; a.out_f32 = 0; a.alpha = 1.f; a.act = act; a.M = M; a.N = N; a.K = K;
    a.ln_stats = stats; a.ln_npart = npart; a.ln_colsum = colsum; a.ln_eps = eps;
    if (!xb || !Wf || !colsum || !cbias || !stats || !C || act < 0 || act > 1) return MI_ERR_ARG;
    if (!gemm_8p_supported(a, false)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = gemm_8p_launch(a, false, stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- Training epilogues of the 256x256 phase kernel (gemm_args.hpp, act 3 / 4): the FFN's activation passes ride the GEMMs next to them.
// Forward: pre (M,N) bf16 = A W^T + b and h (M,N) bf16 = dropout(act(pre)) from one launch (kind 1 erf-GELU / 2 tanh-GELU; drop_p = 0: no dropout; the mask is
// mi_dropout's for the (M,N) matrix, so the result is bit-identical to mi_gemm_bf16 + mi_act[_dropout]_fwd_bf16).  N % 256 == 0, K % 64 == 0, K >= 128, else MI_ERR_UNSUPPORTED.
// replaces: Wav2Vec2ConformerFeedForward.forward's intermediate_dense + intermediate_act_fn + intermediate_dropout (tf wav2vec2_conformer :350-354) under autograd.
extern "C" int mi_gemm_act_fwd_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, void* pre, long ldp, void* h, long ldh, int kind,
                                    float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = pre; a.ldc = ldp; a.out_f32 = 0; a.alpha = 1.f; a.act = 4; a.M = M; a.N = N; a.K = K;
    a.C2 = (bf16_t*)h; a.ldc2 = ldh; a.aux_kind = kind; a.drop_p = drop_p; a.drop_key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    if (!A || !W || !pre || !h) return MI_ERR_ARG;
    if (!gemm_8p_supported(a, false)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = gemm_8p_launch(a, false, stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}
// Backward: dX (M,N) bf16 = dropout(bf16(dY Wt^T)) * act'(pre) — Wt (N,K) is the transposed weight copy the trainer keeps (dh = dy W2 as a GEMM over its rows),
// pre (M,N) bf16 the saved pre-activation; bit-identical to mi_gemm_bf16 + mi_act[_dropout]_bwd_bf16.  Same shape limits.
extern "C" int mi_gemm_act_bwd_bf16(const void* dY, long ldy, const void* Wt, long ldw, const void* pre, long ldp, void* dX, long ldx, int kind,
                                    float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.A = (const bf16_t*)dY; a.lda = ldy; a.W = (const bf16_t*)Wt; a.ldw = ldw; a.bias = nullptr; a.bias_mode = 0;
    a.C = dX; a.ldc = ldx; a.out_f32 = 0; a.alpha = 1.f; a.act = 3; a.M = M; a.N = N; a.K = K;
    a.aux = (const bf16_t*)pre; a.ldaux = ldp; a.aux_kind = kind; a.drop_p = drop_p; a.drop_key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    if (!dY || !Wt || !pre || !dX) return MI_ERR_ARG;
    if (!gemm_8p_supported(a, false)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = gemm_8p_launch(a, false, stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// C (M,N) = [resid +] alpha * dropout(A W^T + b): the hidden / final / attention-output dropout of the training forward (and of gradients flowing back through such a site)
// in the epilogue of the 128 x 128 kernel — fp32 out with optional fp32 residual (mask of mi_dropout_add_f32), or bf16 out (mask and rounding of mi_dropout's bf16 form).
// N % 128 == 0, K % 128 == 0, K >= 320, else MI_ERR_UNSUPPORTED (the caller runs the GEMM and the dropout kernel).
extern "C" int mi_gemm_dropout_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, void* C, long ldc, int out_f32, const float* resid, long ldr,
                                    float alpha, float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = C; a.ldc = ldc; a.out_f32 = out_f32; a.resid = resid; a.ldr = ldr; a.alpha = alpha; a.act = 0; a.M = M; a.N = N; a.K = K;
    a.drop_p = drop_p; a.drop_key = ((unsigned long long)stream_id << 32) ^ (unsigned long long)seed;
    if (!A || !W || !C || drop_p < 0.f || drop_p >= 1.f || (!out_f32 && (resid || alpha != 1.f))) return MI_ERR_ARG;
    if ((K % 128) != 0 || !gemm_8p128_supported(a)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = gemm_8p128_launch(a, 0, stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// Producer: C (M,N) fp32 = resid + alpha * (A W^T + b) (resid may be C), and in the same epilogue C2 (M,N) bf16 = the stored rows, stats_out = their per-row partial
// (sum, sumsq) pairs (slot = 32-column block of the row, N / 32 <= 16 pairs, row stride 32 floats).  128x128 phase kernel only: N % 128 == 0, N <= 512, K % 128 == 0, K >= 320.
extern "C" int mi_gemm_resid_stats_f32_v(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, const float* resid, long ldr, float alpha,
                                         void* C2, long ldc2, float* stats_out, int M, int N, int K, int variant, hipStream_t stream);
extern "C" int mi_gemm_resid_stats_f32(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, const float* resid, long ldr, float alpha,
                                       void* C2, long ldc2, float* stats_out, int M, int N, int K, hipStream_t stream) {
    return mi_gemm_resid_stats_f32_v(A, lda, W, ldw, bias, C, ldc, resid, ldr, alpha, C2, ldc2, stats_out, M, N, K, 0, stream);
}
extern "C" int mi_gemm_resid_stats_f32_v(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, const float* resid, long ldr, float alpha,
                                         void* C2, long ldc2, float* stats_out, int M, int N, int K, int variant, hipStream_t stream) {
    MI_ENTER();
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = C; a.ldc = ldc; a.out_f32 = 1; a.resid = resid; a.ldr = ldr; a.alpha = alpha; a.act = 0; a.M = M; a.N = N; a.K = K;
    a.C2 = (bf16_t*)C2; a.ldc2 = ldc2; a.stats_out = stats_out;
    if (!A || !W || !C || !resid || (variant != 0 && variant != 40 && variant != 42 && variant != 43)) return MI_ERR_ARG;       // 42 / 43: the 128 x 128 kernel's pipelined / loader-consumer form (A/B, tests)
    const bool wide = variant == 40;
    if (wide ? !gemm_8p_supported(a, false) : !gemm_8p128_supported(a)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = wide ? gemm_8p_launch(a, false, stream) : gemm_8p128_launch(a, variant == 42 ? 3 : (variant == 43 ? 2 : 0), stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- CTC head with the row log-sum-exp out of the GEMM's epilogue: C (M,N) fp32 = A W^T + b on the 256x256 kernel, whose epilogue also leaves one
// (max, sum exp(x - max)) pair per row and 64-column block in `workspace`; a second, small launch merges a row's pairs into lse[m] = log sum_n exp(C[m][n]) — what
// mi_row_lse computes with a pass over the (M, N) logits (160 MB at 8000 x 5001).  The natural-log value; differs from mi_row_lse's in the order of the sums (<= 2 ulp).
namespace {
__global__ __launch_bounds__(256) void lse_merge_kernel(const float* __restrict__ part, int ld, int npair, float* __restrict__ lse, int M) {
    const int g = threadIdx.x & 15;                                   // 16 lanes per row (one DPP row), 16 rows per block
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const float* pr = part + (long)min(row, M - 1) * ld;
    float mx = -INFINITY;
    for (int t = g; t < npair; t += 16) mx = fmaxf(mx, pr[2 * t]);
    mx = fmaxf(mx, dpp_f32<0xB1, 0xF>(mx, mx)); mx = fmaxf(mx, dpp_f32<0x4E, 0xF>(mx, mx));
    mx = fmaxf(mx, dpp_f32<0x141, 0xF>(mx, mx)); mx = fmaxf(mx, dpp_f32<0x140, 0xF>(mx, mx));
    float s = 0.f;
    for (int t = g; t < npair; t += 16) {
        const f32x2 q = *reinterpret_cast<const f32x2*>(pr + 2 * t);
        if (q.x > -INFINITY) s += q.y * __expf(q.x - mx);
    }
    s += dpp_f32<0xB1, 0xF>(0.f, s); s += dpp_f32<0x4E, 0xF>(0.f, s);
    s += dpp_f32<0x141, 0xF>(0.f, s); s += dpp_f32<0x140, 0xF>(0.f, s);
    if (g == 0 && row < M) lse[row] = mx + __logf(s);
}
}  // namespace

extern "C" size_t mi_gemm_lse_workspace_floats(int M, int N) { return (size_t)(M > 0 ? M : 0) * 8 * (size_t)((N + 255) / 256); }

// MI_ERR_UNSUPPORTED: the shape is outside the 256x256 kernel (the caller runs mi_gemm_bf16 + mi_row_lse).
extern "C" int mi_gemm_lse_f32(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, float* lse, float* workspace,
                               int M, int N, int K, hipStream_t stream) {
    MI_ENTER();
    if (!A || !W || !C || !lse || !workspace) return MI_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias; a.bias_mode = bias ? 1 : 0;
    a.C = C; a.ldc = ldc; a.out_f32 = 1; a.alpha = 1.f; a.act = 0; a.M = M; a.N = N; a.K = K;
    a.lse_part = workspace; a.lse_ld = 8 * ((N + 255) / 256);
    if (!gemm_8p_supported(a, false)) return MI_ERR_UNSUPPORTED;
    const int slot = mi_profile_hook_begin(stream, 2.0 * M * N * K);
    const int rc = gemm_8p_launch(a, false, stream);
    if (slot >= 0) mi_profile_hook_end(slot, stream);
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(lse_merge_kernel, dim3((M + 15) / 16), dim3(256), 0, stream, workspace, a.lse_ld, a.lse_ld / 2, lse, M);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
