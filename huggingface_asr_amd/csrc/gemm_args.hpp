// Argument block shared by the two bf16 GEMM kernels (gemm_bf16.hip: generic register-staged kernel;
// gemm_glds.hip: LDS-DMA pipelined kernel for K % 64 == 0).
#pragma once
#include "common.hpp"
#include <hip/hip_ext.h>

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const float* bias; int bias_mode;      // 0 none, 1 per output column, 2 per output row
    void* C; long ldc; int out_f32;
    const float* resid; long ldr; float alpha;   // if resid: out = resid + alpha * (acc + bias)
    int act;                                // 0 none, 1 gelu(erf)
    int col_T, col_Tp;                      // != 0: output column n -> (n / col_T) * col_Tp + n % col_T
    int M, N, K;
    int variant;                            // kernel selection for A/B runs and tests (mi_gemm_bf16_v); 0 = the product's dispatch.  Per call: no process state.
    // implicit im2col (CONV): A is a channels-last activation (B, Tin, Fin, Cin); row m = (b, to, fo);
    // k = (kh*KW + kw)*Cin + c
    int Tin, Fin, Cin, Tout, Fout, KW, stride, pad_t, pad_f;
    int stride_f;                           // frequency-axis stride (`stride` is the time-axis one); the gate conv of GatedConv2dShared is (12,3) / (8,2) (extractors.py:41-47)
    // GatedConv2d as ONE implicit GEMM (extractors.py:23-32): W holds the conv and gate filters interleaved in blocks of 32 output channels
    // ([conv c0..c0+31 ; gate c0..c0+31] per 64 rows), N = 2 * Cout, and the epilogue writes act((conv + b) * sigmoid(gate + bg)) as (M, N / 2).
    // Only the 256-wide phase kernel has that epilogue: unsupported shapes return MI_ERR_UNSUPPORTED (the caller runs the GEMM raw + mi_gated_act_bf16).
    int gated;
    // LayerNorm folded into the GEMM (encoder.hip `ln_fold` path).  LN(x) W^T = rstd (x W'^T) - rstd mu s + (W beta + b) with W' = W diag(gamma), s_n = sum_k W'[n,k]:
    //  consumer (256x256 phase kernel): A = bf16(x) un-normalised, W = bf16(W'), bias = W beta + b (fp32), ln_colsum = s (fp32, summed over the bf16 W' values),
    //    ln_stats = per-row partial (sum, sum of squares) pairs of x over its K columns — row stride 32 floats, ln_npart pairs (1, or N_producer / 32 <= 16) —
    //    reduced in the kernel's prologue in a fixed order; out = act(rstd * acc - rstd * mu * s_n + bias_n)
    //  producer (128x128 fp32-out kernel): C2 = bf16 copy of the fp32 rows it stores, stats_out = their per-row partial (sum, sumsq) per 32-column wave block:
    //    slot (n0 / 128) * 4 + wave column of the row's 16 (N <= 512)
    const float* ln_stats; int ln_npart; const float* ln_colsum; float ln_eps;
    bf16_t* C2; long ldc2; float* stats_out;
    // Training epilogues of the 256x256 phase kernel (act = 3 / 4; gemm_8p.hip).  `aux_kind`: 1 erf-GELU, 2 tanh-GELU.  drop_p > 0: the FFN's activation dropout
    // with the counter-based mask of dropout.hip for logical element m * N + n (key = stream_id << 32 ^ seed), same rounding points as the separate kernels.
    //  act 3 (backward): C = dropout(bf16(acc)) * act'(aux), aux (M, N) bf16 = the saved pre-activation — the dX GEMM dh = dy W2 and mi_act_bwd in one launch
    //  act 4 (forward):  C = bf16(acc + bias) (the pre-activation, kept for the backward), C2 = dropout(act(C)) — the FFN-in GEMM and mi_act_fwd in one launch
    const bf16_t* aux; long ldaux; int aux_kind; float drop_p; unsigned long long drop_key;
    // CTC-head form of the 256x256 kernel's fp32 epilogue (mi_gemm_lse_f32): per row and 64-column wave block one (max, sum of exp(x - max)) pair over the columns < N
    // at lse_part[m * lse_ld + 2 * (n0 / 64 + wave column)] — the row log-sum-exp without a second pass over the (M, N) logits.  lse_ld >= 8 * ceil(N / 256); every pair
    // of a row is written ((-inf, 0) for blocks entirely beyond N).
    float* lse_part; int lse_ld;
};
constexpr int LN_STATS_STRIDE = 32;       // floats per row of a partial-statistics buffer (16 (sum, sumsq) pairs)

// gemm_glds.hip; returns MI_ERR_UNSUPPORTED when the shape/alignment does not fit the fast path
bool gemm_glds_supported(const GemmArgs& a, bool conv);
int gemm_glds_launch(const GemmArgs& a, bool conv, hipStream_t stream);


// gemm_8p.hip: 256x256x64 tiles, 8 waves, phase-interleaved schedule on a two-deep LDS ring (bf16 out, N % 256 == 0, K % 64 == 0, K >= 128)
bool gemm_8p_supported(const GemmArgs& a, bool conv);
int gemm_8p_launch(const GemmArgs& a, bool conv, hipStream_t stream);
bool gemm_8p128_supported(const GemmArgs& a);
int gemm_8p128_launch(const GemmArgs& a, int ring, hipStream_t stream);


// Profiling hand-off (diagnostics; encoder.hip `mi_profile_*`): while a profiling slot is open the dense-contraction launchers start their kernel with
// hipExtLaunchKernelGGL and the slot's (start, stop) events, so the pair carries the DISPATCH's own begin / end timestamps — the quantity rocprofv3 --kernel-trace
// reports — instead of bracketing the launch with two hipEventRecord markers, which also times the gap between the markers and the kernel.
// `family`: 0 gemm8p 256x256 (bf16 out, no activation), 1 gemm8p + GELU epilogue, 2 gemm8p implicit-GEMM conv, 3 gemm8p fp32 out (CTC head), 4 gemm8p128 (N = 512 class),
//           5 gemm_glds, 6 gemm_bf16 (generic)
extern "C" int mi_profile_take_events(hipEvent_t* start, hipEvent_t* stop, int family);
enum { PF_8P = 0, PF_8P_GELU = 1, PF_8P_CONV = 2, PF_8P_OUT32 = 3, PF_8P128 = 4, PF_GLDS = 5, PF_GENERIC = 6, PF_COUNT = 7 };

template <typename F, typename... Args>
inline void launch_dense(int family, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    hipEvent_t e0, e1;
    if (mi_profile_take_events(&e0, &e1, family)) hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}
