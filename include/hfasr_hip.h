/* libhfasr_hip.so — C ABI of the MI355X (gfx950) hot path behind the BUTSpeechFIT/huggingface_asr model surface.
 *
 * The reference is 100 % Python and has no FFI of its own: its "plug-in" boundary is
 * HuggingFace's Auto-class registry (reference src/utilities/bind.py:36-58).  This header is the C boundary
 * that sits directly below that Python surface: one entry point per tensor op of the hot path, plus one
 * whole-encoder driver.  Every function
 *   - takes raw device pointers, element strides and sizes (no torch types), and a hipStream_t,
 *   - enqueues kernels on that stream and returns immediately (no allocation, no synchronisation,
 *     no global state; workspaces are passed in, so calls are hipGraph-capturable),
 *   - returns 0 (MI_OK) or a negative error code (MI_ERR_ARG -1, MI_ERR_LAUNCH -2, MI_ERR_UNSUPPORTED -3).
 * bf16 buffers are raw 2-byte bfloat16; "ld" arguments are row strides in ELEMENTS.
 * The reference-side binding is the ctypes loader in huggingface_asr_amd/_lib.py.
 */
#ifndef HFASR_HIP_H
#define HFASR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mi_stream_t; /* = hipStream_t */

/* diagnostics: text of the last failed kernel launch on the calling thread ("" if none). */
const char* mi_last_error(void);

/* profiling facility (off by default; process-global, not thread-safe, not for production): while enabled, every launch of a dense contraction (GEMM /
 * implicit-GEMM conv) is started with hipExtLaunchKernelGGL and a start / stop event pair, i.e. its duration is the dispatch's own begin -> end timestamps —
 * what rocprofv3 --kernel-trace reports for the same launch — for bench.py's roofline.achieved. */
int mi_profile_create(int capacity);
void mi_profile_enable(int on);
void mi_profile_reset(void);
int mi_profile_count(void);
int mi_profile_summary(double* total_ms, double* total_flops);
int mi_profile_calibrate(mi_stream_t stream, int n, double* median_ms);   /* event pair around an EMPTY kernel, median of n: what the bracket adds + the empty kernel's ~1.3 us */
int mi_profile_summary_family(int family, double* total_ms, double* total_flops, int* launches);   /* family < 0: all; 0 gemm8p, 1 gemm8p+GELU, 2 gemm8p conv, 3 gemm8p fp32 out, 4 gemm8p128, 5 gemm_glds, 6 generic */

/* ---- nn.Linear / lm_head / projections: C[M,N] = epi(A[M,K] * W[N,K]^T), bf16 in, fp32 accumulate (MFMA).
 * replaces: every nn.Linear on the path — reference src/models/encoders/e_branchformer.py:96-98,139,212-216,247,456-457;
 *           src/models/extractors.py:108,131; transformers wav2vec2_conformer FFN (modeling :350-357).
 * bias_mode 0 none / 1 per column / 2 per row; act 0 none / 1 gelu(erf);
 * resid != NULL: out = resid[m*ldr+n] + alpha*(acc+bias); out_f32 selects fp32 or bf16 C.
 * col_T/col_Tp != 0: output column n is remapped to (n / col_T) * col_Tp + n % col_T (time-padded V^T). */
int mi_gemm_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, int bias_mode,
                 void* C, long ldc, int out_f32, const float* resid, long ldr, float alpha, int act,
                 int M, int N, int K, int col_T, int col_Tp, mi_stream_t stream);
/* the same with an explicit kernel selection (A/B runs, tests that must reach a kernel below the size its default dispatch picks it at): 0 = the product's
 * dispatch (= mi_gemm_bf16), 40 / 41 = phase-interleaved kernels wherever supported / never, 42 / 47 = the 128x128 phase kernel (pipelined / two-segment form),
 * 30 / 31 = the older 128x128 LDS-DMA tiles (persistent / one block per tile); 48 = the product's dispatch (as 0) without the tail-round split: with variant 0 a GEMM
 * whose 256x256 tiles leave a last round of the 256 CUs less than half full runs that round's rows on 128x128 tiles (same bits; callers with several batches in flight pass 48).
 * Per call — the library keeps no kernel-selection state. */
int mi_gemm_bf16_v(const void* A, long lda, const void* W, long ldw, const float* bias, int bias_mode,
                   void* C, long ldc, int out_f32, const float* resid, long ldr, float alpha, int act,
                   int M, int N, int K, int col_T, int col_Tp, int variant, mi_stream_t stream);

/* ---- Conv2d sub-sampling, second layer (C->C, KxK, stride s) as implicit GEMM over a channels-last activation.
 * replaces: src/models/extractors.py:71-96 (layers >= 1) incl. the causal left-padded form src/models/streaming_modules.py:31-55. */
int mi_conv2d_cl_bf16(const void* in, const void* weight, const float* bias, void* out,
                      int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW, int stride,
                      int pad_t, int pad_f, int Tout, int Fout, int act, mi_stream_t stream);
int mi_conv2d_cl_bf16_v(const void* in, const void* weight, const float* bias, void* out,
                        int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW, int stride,
                        int pad_t, int pad_f, int Tout, int Fout, int act, int variant, mi_stream_t stream);   /* variant: as mi_gemm_bf16_v */

/* ---- Conv2d sub-sampling, first layer (1->C) + GELU over the padded (B,T,F) fp32 log-mel layout; output channels-last bf16.
 * replaces: src/models/extractors.py:71-96 (layer 0) and :111. */
int mi_conv2d_first_gelu(const float* x, const float* w, const float* bias, void* out_cl_bf16,
                         int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f,
                         int T1, int F1, mi_stream_t stream);

/* ---- context-aware Conv2d sub-sampling: `context_awareness_type` = "gated" / "gated_shared" (src/models/extractors.py:23-65; selected by
 *      recipes_v0.0.1/librispeech_aed/train_gated_baseline.sh:94, train_small_baseline_fp32_gated.sh:96).
 * GatedConv2d (:23-32): conv(x) * sigmoid(gate(x)), same geometry; GatedConv2dShared (:35-54): the gate is a (4k, k) / stride (4s, s) / padding (4p, p) conv, one
 * gate row per four output time steps (the conv's time axis must be divisible by 4: the reference's `view` raises otherwise).  The layer's GELU follows the product. */
/* Conv2d(1 -> C) of general geometry over the (B,T,F) fp32 features -> channels-last bf16; act 1: GELU, 0: raw pre-activation (operand of mi_gated_act_bf16) */
int mi_conv2d_first_geo(const float* x, const float* w, const float* bias, void* out_cl_bf16, int B, int T, int F, int C,
                        int KH, int KW, int stride_t, int stride_f, int pad_t, int pad_f, int T1, int F1, int act, mi_stream_t stream);
/* GatedConv2d as the first layer, fused (3x3): out = GELU((conv + b) * sigmoid(gate + bg)); MI_ERR_UNSUPPORTED for other kernel sizes (run the two raw convs + mi_gated_act_bf16) */
int mi_conv2d_first_gated_gelu(const float* x, const float* w, const float* bias, const float* gw, const float* gbias, void* out_cl_bf16,
                               int B, int T, int F, int C, int K, int stride, int pad_t, int pad_f, int T1, int F1, mi_stream_t stream);
/* mi_conv2d_cl_bf16 with separate time / frequency strides; gated != 0: GatedConv2d as ONE implicit GEMM — weight (2*Cout, KH*KW*Cin) / bias (2*Cout) hold conv and gate
 * filters interleaved in blocks of 32 output channels ([conv c0..c0+31 ; gate c0..c0+31]), out (B,Tout,Fout,Cout) = GELU((conv + b) * sigmoid(gate + bg)) in the epilogue
 * (act must be 1; Cout % 128 == 0, Cin % 64 == 0, else MI_ERR_UNSUPPORTED: run it with gated = 0, act = 0 into a (M, 2*Cout) buffer + mi_gated_act_bf16(blk = 32)). */
int mi_conv2d_cl_geo_bf16(const void* in, const void* weight, const float* bias, void* out, int B, int Tin, int Fin, int Cin, int Cout, int KH, int KW,
                          int stride_t, int stride_f, int pad_t, int pad_f, int Tout, int Fout, int act, int gated, mi_stream_t stream);
/* out (B*T*Fq, C) bf16 = GELU(z * sigmoid(g)), gate row of (b,t,f) = (b, t / share, f).  blk = 0: z (B*T*Fq, C), g (B*(T/share)*Fq, C) plain columns;
 * blk > 0: z == g is the output of one stacked GEMM whose columns are interleaved [conv blk | gate blk] per 2*blk (share must be 1). */
int mi_gated_act_bf16(const void* z, long ldz, const void* g, long ldg, void* out, long ldo, int B, int T, int Fq, int C, int share, int blk, mi_stream_t stream);

/* C (M,N) = [resid +] alpha * dropout(A W^T + b) from one launch of the 128 x 128 GEMM: fp32 out with optional fp32 residual (mask of mi_dropout_add_f32 for element m*N + n)
 * or bf16 out (resid NULL, alpha 1; mask and rounding of mi_dropout's bf16 form).  N % 128 == 0, K % 128 == 0, K >= 320, else MI_ERR_UNSUPPORTED.
 * replaces: a Linear followed by nn.Dropout under autograd (tf wav2vec2_conformer :355-357 output_dense + output_dropout; e_branchformer.py:288, 301). */
int mi_gemm_dropout_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, void* C, long ldc, int out_f32, const float* resid, long ldr,
                         float alpha, float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, mi_stream_t stream);
/* Training epilogues of the 256 x 256 GEMM: the FFN's activation passes ride the GEMMs next to them (bit-identical to the GEMM + mi_act[_dropout]_{fwd,bwd}_bf16 pair).
 * forward: pre (M,N) bf16 = A W^T + b, h (M,N) bf16 = dropout(act(pre)); kind 1 erf-GELU / 2 tanh-GELU; drop_p = 0: none; mask of mi_dropout for (seed, stream_id).
 * backward: dX (M,N) bf16 = dropout(bf16(dY Wt^T)) * act'(pre), Wt (N,K) the transposed weight.  N % 256 == 0, K % 64 == 0, K >= 128, else MI_ERR_UNSUPPORTED.
 * replaces: Wav2Vec2ConformerFeedForward's intermediate_dense + intermediate_act_fn + intermediate_dropout (tf wav2vec2_conformer :350-354) and their autograd backward. */
int mi_gemm_act_fwd_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, void* pre, long ldp, void* h, long ldh, int kind,
                         float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, mi_stream_t stream);
int mi_gemm_act_bwd_bf16(const void* dY, long ldy, const void* Wt, long ldw, const void* pre, long ldp, void* dX, long ldx, int kind,
                         float drop_p, unsigned seed, unsigned stream_id, int M, int N, int K, mi_stream_t stream);
/* ---- LayerNorm folded into the GEMMs around it (the engine's `ln_fold` path).  replaces: the same nn.LayerNorm + nn.Linear pairs as mi_layernorm_chain + mi_gemm_bf16
 *      (e_branchformer.py:233,236,242,261 in front of tf wav2vec2_conformer :350-357, e_branchformer.py:96-98,212), evaluated as
 *      LN(x) W^T + b = rstd (bf16(x) W'^T) - rstd mu s + (W beta + b),  W' = bf16(W diag(gamma)),  s_n = sum_k W'[n,k].
 * mi_gemm_lnfold_bf16: the consumer GEMM (256x256 phase kernel; MI_ERR_UNSUPPORTED for shapes it does not take); stats = per-row partial (sum, sumsq) pairs of x, row stride
 *   32 floats, npart pairs.  mi_gemm_resid_stats_f32: the producer — C fp32 = resid + alpha (A W^T + b), plus C2 = bf16(C) and the partial statistics of the rows it stores
 *   (one pair per 32 columns; 128x128 phase kernel).  mi_layernorm_fold: y = [LN](mask(x)) as fp32 + bf16 + statistics pair 0 (npart = 1). */
int mi_gemm_lnfold_bf16(const void* xb, long lda, const void* Wf, long ldw, const float* colsum, const float* cbias, const float* stats, int npart, float eps,
                        void* C, long ldc, int act, int M, int N, int K, mi_stream_t stream);
int mi_gemm_resid_stats_f32(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, const float* resid, long ldr, float alpha,
                            void* C2, long ldc2, float* stats_out, int M, int N, int K, mi_stream_t stream);
/* the same with the kernel chosen by the caller: variant 40 = the 256 x 256 tile (N % 256 == 0; partial statistics then come one pair per 64 columns), 0 = as above */
int mi_gemm_resid_stats_f32_v(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, const float* resid, long ldr, float alpha,
                              void* C2, long ldc2, float* stats_out, int M, int N, int K, int variant, mi_stream_t stream);
int mi_layernorm_fold(const float* x, long ldx, const int* lengths, int T, const float* g1, const float* b1, float eps1, float* y32, long ldy,
                      void* yb_bf16, long ldyb, float* stats, int M, int d, mi_stream_t stream);

/* ---- LayerNorm chain on the fp32 residual stream (see csrc/norm.hip).
 * replaces: nn.LayerNorm at e_branchformer.py:233,236,242,257,261; feature projection LN (extractors.py:130);
 *           encoder.layer_norm (wav2vec2_conformer :707); zeroing of padded frames (:662-665) via `lengths`. */
int mi_layernorm_chain(const float* x, long ldx, const int* lengths, int T,
                       const float* g1, const float* b1, float eps1, float* y32, long ldy,
                       const float* ga, const float* ba, float eps2, void* outa_bf16, long lda,
                       float* outa_f32, long lda32,
                       const float* gb, const float* bb, void* outb_bf16, long ldb,
                       int M, int d, mi_stream_t stream);

/* ---- fp32 -> bf16 cast of a (M,d) state (A operand of a following GEMM). */
int mi_cast_f32_bf16(const float* x, long ldx, void* out, long ldo, int M, int d, mi_stream_t stream);

/* ---- rotary embedding of the Q/K projection input. replaces: wav2vec2_conformer _apply_rotary_embedding (:509-526). */
int mi_rotary_bf16(const void* x, long ldx, void* out, long ldo, const float* cos_t, const float* sin_t,
                   int M, int T, int H, int hd, mi_stream_t stream);

/* ---- fused self-attention (relative-position / rotary / plain), key-padding + optional causal mask.
 * replaces: Wav2Vec2EBranchformerSelfAttention.forward e_branchformer.py:100-138 and
 *           _apply_relative_embeddings wav2vec2_conformer :528-565.  pos == NULL: no relative term. */
int mi_attention_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, int Tp,
                      const void* pos, long ldp, const float* bias_u, const float* bias_v,
                      const int* lengths, void* out, long ldo, int B, int T, int H, int hd,
                      float scale, int causal, mi_stream_t stream);

/* LDS-staged form for head sizes 64 / 128: q, k, v are columns of ONE fused (B*T, 3d) projection (all [row][hd]),
 * no transposed V copy; the position rows / K / V tiles are shared by the 4 waves of a 128-query block. */
int mi_attention_qkv_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                          const void* pos, long ldp, const float* bias_u, const float* bias_v,
                          const int* lengths, void* out, long ldo, int B, int T, int Tk, long kv_bstride, int H, int hd,
                          float scale, int causal, mi_stream_t stream);
/* A/B form of mi_attention_qkv_bf16: variant 0 = the library's choice (the eight-wave kernel: two waves per SIMD, the wave pair of a query group splits the keys),
 * except with relative positions at head size 64), 1 = the four-wave kernel of rounds 1-3, 2 = the eight-wave kernel.  Measurement only (tools/attn_ab.py); no product call
 * site passes a non-zero variant. */
int mi_attention_qkv_bf16_v(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                            const void* pos, long ldp, const float* bias_u, const float* bias_v,
                            const int* lengths, void* out, long ldo, int B, int T, int Tk, long kv_bstride, int H, int hd,
                            float scale, int causal, int variant, mi_stream_t stream);
/* (kv_bstride = elements between batches of k/v, 0 = Tk*ld (KV caches are (B, Lmax, d)); T = queries per batch, Tk = keys per batch, 0 = T: cross-attention over encoder frames and KV-cache decoding use Tk != T;
 *  with causal != 0 query i sees keys <= i + (Tk - T).) */

/* Training form of the above (self-attention only, Tk = T): also leaves each row's log-sum-exp of the scaled scores (log2 domain) in lse (B, H, T) fp32,
 * which mi_attention_qkv_bwd_probs needs.  replaces: the forward half of Wav2Vec2EBranchformerSelfAttention under autograd, e_branchformer.py:105-138. */
int mi_attention_qkv_lse_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                              const void* pos, long ldp, const float* bias_u, const float* bias_v,
                              const int* lengths, void* out, long ldo, float* lse, int B, int T, int H, int hd,
                              float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
/* (drop_p > 0: attention-probability dropout, e_branchformer.py:132 — the normaliser keeps every key, the context sums the survivors / (1 - p); mask of mi_dropout for the
 *  logical element ((h*B + b)*T + i)*T + j of the (H,B,T,T) probabilities, the one mi_attn_softmax_fwd draws) */
/* First half of the attention backward (what autograd derives from e_branchformer.py:105-138 and the rel-shift of tf wav2vec2_conformer :528-565): one walk over
 * the keys recomputes the scores, P = 2^(S - lse), dP = dctx V^T, dS = P (dP - dctx·ctx) scale, and leaves bf16
 *   prob, ds (H, B, T, ldsr)   and   dbd (H, B, T, ldbd), dbd[i][T-1-i+j + pad] = ds[i][j]  (the gradient of the un-shifted position scores; null without pos).
 * It also accumulates the query gradient over the walk: dq (B*T, lddq) bf16 = dS K + dBD P, and the per-wave column sums of the two terms in
 * dsum_u / dsum_v (B, 4 ceil(T/128), H*hd) fp32 (summed over their first two axes: the pos_bias_u / pos_bias_v gradients; unused without pos).
 * dsum_v == dsum_u + H*hd selects the interleaved layout: ONE (B * 4 ceil(T/128), 2 H*hd) buffer whose rows are [u | v] — the shape mi_ln_partial_reduce_many sums
 * (desc kind 0 with d = H*hd, dgamma = the pos_bias_u gradient, dbeta = the pos_bias_v gradient), so the reduction can leave with a later launch.
 * ldsr, ldbd multiples of 32 with ldsr >= T rounded up to 32 and ldbd >= pad + 2T - 1; 0 <= pad < 32 with (T - 32 + pad) % 32 == 0.  Every element is written. */
int mi_attention_qkv_bwd_probs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                               const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                               const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                               void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                               void* dq, long lddq, float* dsum_u, float* dsum_v,
                               int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
/* the same, also leaving qu_out / qv_out (B*T, ldqb) bf16 = q + pos_bias_u / q + pos_bias_v (both or neither, with pos only): the walk's own A fragments — the operands
 * of the dK = dS^T (q + u) and d(positions) = dBD^T (q + v) products that follow, which a pass of their own used to make */
int mi_attention_qkv_bwd_probs_qb(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                  const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                  const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                  void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                  void* dq, long lddq, float* dsum_u, float* dsum_v, void* qu_out, void* qv_out, long ldqb,
                                  int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
/* ... with flags.  bit 0, sparse writes: the zeros nobody reads are not written — `dbd` must be a buffer the caller zero-filled ONCE and that only this entry writes (a row's
 * relative positions outside its maximal band are the same for every launch), and columns of prob / ds from the key length rounded up to 128 on must never be read
 * (mi_bgemm_sparse_bf16 with m_valid = lengths does not read them). */
int mi_attention_qkv_bwd_probs_f(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                  const void* pos, long ldp, const float* bias_u, const float* bias_v, const int* lengths,
                                  const void* ctx, long ldo, const void* dctx, long ldd, const float* lse,
                                  void* prob, void* ds, long ldsr, void* dbd, long ldbd, int pad,
                                  void* dq, long lddq, float* dsum_u, float* dsum_v, void* qu_out, void* qv_out, long ldqb,
                                  int B, int T, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, int flags, mi_stream_t stream);
/* the two entries above for Tq != Tk and separate q / k / v operands, no relative positions: the GPT-2 decoder's causal self-attention and its cross-attention over the
 * encoder frames in training (multi_head_gpt2.py:80-170).  lse (B, H, Tq); prob / ds (H, B, Tq, ldsr) bf16, ldsr % 32 == 0, ldsr >= Tk rounded up to 32; dq = dS K. */
int mi_attention_x_lse_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const int* lengths, void* out, long ldo, float* lse,
                            int B, int Tq, int Tk, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
int mi_attention_x_bwd_probs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const int* lengths, const void* ctx, long ldo,
                             const void* dctx, long ldd, const float* lse, void* prob, void* ds, long ldsr, void* dq, long lddq,
                             int B, int Tq, int Tk, int H, int hd, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
/* (drop_p / seed / stream_id as given to the forward: prob then holds the DROPPED probabilities, ds the gradient through the un-dropped softmax) */

/* ---- cgMLP gate: per-row LN statistics + fused LN -> depthwise conv(time) -> gate.
 * replaces: ConvolutionalSpatialGatingUnit.forward e_branchformer.py:184-204. */
int mi_row_stats_bf16(const void* x, long ldx, int d, float eps, float* stats, int M, mi_stream_t stream);
int mi_csgu_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta,
                 const float* w, const float* bias, void* out, long ldo,
                 int B, int T, int C, int K, int pad_left, int dilation, int act, mi_stream_t stream);

/* ---- merge block depthwise conv + residual. replaces: e_branchformer.py:296-299. */
int mi_dwconv_residual_bf16(const void* m, long ldm, const float* w, const float* bias, void* out, long ldo,
                            int B, int T, int C, int K, int pad_left, mi_stream_t stream);
/* the split CSGU form — `csgu_use_linear_after_conv` (e_branchformer.py:172-175,196-201: conv -> Linear -> act -> gate) and, on the training path, any
   csgu_activation other than identity:  conv alone (out = dwconv(LN(x_g)) + b), then [a GEMM], then out = x_r * act(g)  (act: 0 identity, 1 gelu, 2 relu, 3 silu) */
int mi_csgu_conv_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                      void* out, long ldo, int B, int T, int C, int K, int pad_left, int dilation, mi_stream_t stream);
int mi_gate_act_mul_bf16(const void* r, long ldr, const void* g, long ldg, void* out, long ldo, long M, int C, int act, mi_stream_t stream);
/* dr = ds * act(g),  dg = ds * x_r * act'(g) */
int mi_gate_act_mul_bwd_bf16(const void* r, long ldr, const void* g, long ldg, const void* ds, long ldds, void* dr, long lddr, void* dg, long lddg,
                             long M, int C, int act, mi_stream_t stream);

/* ---- log-mel front end. replaces: CustomFeatureExtractor.__call__ (src/utilities/feature_extractors.py:51-61) =
 *      Speech2TextFeatureExtractor._extract_fbank_features / utterance_cmvn (transformers) / global_normalize (:47-49). */
int mi_fbank_f64(const float* wave, long ldw, const int* num_samples, int N, const double* window,
                 const double* twiddle, const double* mel_t, const int* mel_lo, const int* mel_hi,
                 float* out, int T_out, int B, int nmel, double mel_floor, double preemph, mi_stream_t stream);
/* replaces: DataPreprocessingManagerCallback.default_transform's array handling (src/utilities/callbacks.py:108-118: audio_object_stripper =
   np.trim_zeros, src/utilities/data_utils.py:173-177, then zero-pad to >= min_len = 8000 samples) for clips that are already on the device. */
int mi_trim_zeros_pad_f32(const float* wave, long ldw, const int* num_samples, int N, int B, int min_len, float* out, long ldo, int N_out,
                          int* first, int* valid, int* eff_len, mi_stream_t stream);
int mi_cmvn_utterance(float* x, const int* frames, int B, int T, int nmel, int norm_means, int norm_vars,
                      float pad, mi_stream_t stream);
int mi_cmvn_global(float* x, long total, int nmel, const float* means, const float* stds, mi_stream_t stream);

/* ---- CTC tail. replaces: log_softmax + F.ctc_loss at e_branchformer.py:472-488. */
int mi_row_lse(const void* x, long ld, int dtype, int V, float* lse, int M, mi_stream_t stream);
/* the CTC head (reference e_branchformer.py:456-457, 472-488: lm_head + blank projection, then log_softmax) as ONE pass over the logits: C (M,N) fp32 = A W^T + b with
 * the rows' log-sum-exp lse (M) — the GEMM's epilogue leaves a (max, sum exp) pair per row and 64 columns in `workspace` (mi_gemm_lse_workspace_floats(M, N) floats), a small
 * second launch merges them.  lse equals mi_row_lse's up to the order of the sums.  MI_ERR_UNSUPPORTED outside the 256 x 256 kernel's shapes (K % 64, K >= 128, 16-B aligned
 * operands, ldc % 4 == 0): the caller runs mi_gemm_bf16 + mi_row_lse. */
size_t mi_gemm_lse_workspace_floats(int M, int N);
int mi_gemm_lse_f32(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, float* lse, float* workspace,
                    int M, int N, int K, mi_stream_t stream);
int mi_ctc_loss_fwd(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T,
                    const long* labels, int U, const int* in_len, int blank, int B,
                    int reduction, int zero_infinity, float* nll, int* tgt_len, float* loss, mi_stream_t stream);

/* ---- joint CTC/attention decoding: batched CTC prefix scores on device.
 * replaces: CTCPrefixScoreTH.__call__ / index_select_state (src/decoding/ctc_scorer.py:58-207) as driven by
 *           CTCRescorerLogitsProcessor.__call__ (:324-354).  x: (B,T,O) padded log-posteriors; r: (T,2,B*W) forward variables. */
int mi_ctc_prefix_prepare(const void* logits, long ld_b, long ld_t, int dtype, const int* lens, int B, int T, int O,
                          int blank, int W, float* lse_scratch, float* x_out, float* r0_out, mi_stream_t stream);
int mi_ctc_prefix_score(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, const long* last_ids,
                        long ld_last, int out_len, const float* s_prev, float* psi_out, float* scores_out, mi_stream_t stream);
int mi_ctc_prefix_select(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, const long* last_ids,
                         long ld_last, int out_len, const int* hyp, const long* tok, long ld_tok, int K, float* r_out,
                         mi_stream_t stream);
/* one processor call after the first = select along (beam 0 of every utterance, last token of every hypothesis) + score with s_prev = psi_old[beam 0 row, last token],
   from one call (no index tensors in between; ctc_scorer.py:327-330 -> :58-207).  r_prev / psi are the state the next call passes as r_old / psi_old. */
int mi_ctc_prefix_advance(const float* x, int B, int T, int O, int blank, int W, const float* r_old, const long* last_old, long ld_last_old, int out_len_old,
                          const float* psi_old, const long* last, long ld_last, int out_len, float* r_prev, float* psi, float* scores, mi_stream_t stream);
/* the same step in ONE scan when the chains of every (hypothesis, token) may be kept between calls: r_all (T, 2, B*W, O) is written by this call and passed as r_prev
   (rp_full = 1, with psi_old) to the next; rp_full = 0: r_prev is mi_ctc_prefix_prepare's (T, 2, B*W) and psi_old is null (first token).  ctc_scorer.py:58-207. */
int mi_ctc_prefix_score_full(const float* x, int B, int T, int O, int blank, int W, const float* r_prev, int rp_full, const float* psi_old, const long* last,
                             long ld_last, int out_len, float* r_all, float* psi, float* scores, mi_stream_t stream);

/* ---- GPT-2 cross-attention decoder helpers (the rest of the decoder runs on the shared LN / GEMM / attention entry points).
 * replaces: GPT2Model embeddings (wte + wpe) and the fixed-position variant src/models/embeddings.py:33-86;
 *           the shifted, label-smoothed CE of src/models/decoders/multi_head_gpt2.py:138-158. */
int mi_embed_tokens(const long* ids, const float* wte, float scale, const float* pos, int pos_offset, int U, int d,
                    int M, int V, float* out, mi_stream_t stream);
/* acc[0] += sum of the valid rows' losses, acc[1] += their count; row_loss: B * (U - shift) floats of workspace (the rows' losses, NaN where the target is ignored),
 * summed by one block in a fixed order: no float atomics, the loss is bit-reproducible. */
int mi_ce_label_smoothing(const float* logits, long ld, const long* labels, int B, int U, int shift, int V, float eps,
                          float* acc, float* row_loss, mi_stream_t stream);

/* ---- training step: backward + optimizer building blocks.
 * replaces: torch autograd of the modules above under HF Trainer's bf16 autocast + torch.optim.AdamW + clip_grad_norm_
 *           (src/utilities/training_utils.py:93-115 GradAwareTrainer.training_step; recipes .../train_small_baseline.sh:43,53-58).
 * Activations bf16, residual stream / parameter gradients fp32; parameter gradients ACCUMULATE (+=). */
int mi_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int M, int N, int Mp, mi_stream_t stream);
int mi_transpose_many_bf16(const void* descs, int count, mi_stream_t stream);   /* descs: device array of {const void* in; void* out; int M, N, Mp, pad;} */
/* out[n] += sum_m x[m][n] (dtype 0 f32, 1 bf16).  workspace: mi_colsum_workspace_floats(M, N) floats — per-chunk column sums, added in chunk order by a second kernel.
 * (Round 4: every parameter-gradient reduction of the training step is ordered — per-block partial rows + a fixed-order sum, csrc/common.hpp rows_reduce_kernel — so the
 *  backward is bit-reproducible run to run; rounds 1-3 ended these sums in float atomics.) */
size_t mi_colsum_workspace_floats(int M, int N);
int mi_colsum(const void* x, long ld, int dtype, int M, int N, float* out, float* workspace, mi_stream_t stream);
/* out (N) bf16 = column sums of x (M, N) f32, rows added in order (M small: the per-group partials of d(posp), train.py `_attention_bwd`) */
int mi_colsum_cast_bf16(const float* x, long ld, int M, int N, void* out, mi_stream_t stream);
/* out_a (N) += column sums of a (M, N), out_b (N) += column sums of b (M, N): fp32, rows added in a fixed order (the pos_bias_u / pos_bias_v gradients from the attention
   backward's per-wave partials; tf wav2vec2_conformer :466-470) */
int mi_colsum2_acc_f32(const float* a, const float* b, long ld, int M, int N, float* out_a, float* out_b, mi_stream_t stream);
int mi_act_fwd_bf16(const void* pre, long ldp, void* out, long ldo, int M, int N, int kind, mi_stream_t stream);
int mi_act_bwd_bf16(const void* dy, long lddy, const void* pre, long ldp, void* dx, long lddx, int M, int N, int kind,
                    mi_stream_t stream);
/* the same with the FFN's activation dropout (tf wav2vec2_conformer :353 `intermediate_dropout`) applied in the pass: out = dropout(act(pre)),
 * dx = dropout(dy) * act'(pre); mask and rounding points of mi_dropout on the (M, N) matrix, i.e. bit-identical to the two-pass form. */
int mi_act_dropout_fwd_bf16(const void* pre, long ldp, void* out, long ldo, int M, int N, int kind, float p, unsigned seed, unsigned stream_id,
                            mi_stream_t stream);
int mi_act_dropout_bwd_bf16(const void* dy, long lddy, const void* pre, long ldp, void* dx, long lddx, int M, int N, int kind, float p,
                            unsigned seed, unsigned stream_id, mi_stream_t stream);
size_t mi_layernorm_bwd_workspace_floats(int d);
int mi_layernorm_bwd(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                     void* dx, long lddx, int dx_bf16, int accumulate, float* dgamma, float* dbeta, float* workspace, int M, int d,
                     mi_stream_t stream);
/* deferred form: the per-block (dgamma | dbeta) partial rows stay in `partial` (*nblk rows of 2 d floats; size as mi_layernorm_bwd_workspace_floats) and are reduced
   later, up to 24 entries per launch, by mi_ln_partial_reduce_many (dgamma[c] += sum, dbeta[c] += sum; fixed summation order, no atomics) */
typedef struct { const float* partial; int nblk, d; float* dgamma; float* dbeta; int kind; } mi_lnred_desc;
/* kind 0: a LayerNorm's rows as above.  kind = K (1..31): the tap-gradient partials a depthwise-conv backward left in its workspace (mi_csgu_bwd_bf16 /
 * mi_dwconv_residual_bwd_bf16 called with dw == NULL): nblk rows of d * 32 floats, d = channels; dgamma = the (d, K) tap gradient, dbeta = the (d) bias gradient or NULL;
 * dgamma[c * K + k] += sum_rows partial[row][c * 32 + k] (k < K), dbeta[c] += sum_rows partial[row][c * 32 + 31] */
int mi_layernorm_bwd_partial(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                             void* dx, long lddx, int dx_bf16, int accumulate, float* partial, int* nblk, int M, int d, mi_stream_t stream);
/* as mi_layernorm_bwd_partial, and in the same pass cast (M,d) bf16 = alpha * dropout(dx) of the finished rows — the bf16 operand of the linear backward that follows
 * (drop_p = 0: the scaled cast; mask of mi_dropout for (seed, stream_id), element m*d + c) */
int mi_layernorm_bwd_partial_cast(const void* x, long ldx, int x_bf16, const float* gamma, float eps, const void* dy, long lddy, int dy_f32,
                                  void* dx, long lddx, int dx_bf16, int accumulate, float* partial, int* nblk, void* cast, long ldcast, float alpha,
                                  float drop_p, unsigned seed, unsigned stream_id, int M, int d, mi_stream_t stream);
/* two LayerNorms of the SAME rows x with the same eps (the layer's two branch norms on x1, reference e_branchformer.py:273,292) in one pass:
 * dx (+)= dLN1/dx . dy + dLN2/dx . dy2 (linear in dy * gamma: x, its statistics and the old dx are read once); the two (dgamma | dbeta) partial sets go to
 * `partial` / `partial2` (*nblk rows each) for mi_ln_partial_reduce_many.  d <= 512.  cast may be NULL; otherwise as in mi_layernorm_bwd_partial_cast. */
int mi_layernorm_bwd_dual_partial(const void* x, long ldx, int x_bf16, float eps, const float* gamma, const void* dy, long lddy, int dy_f32,
                                  const float* gamma2, const void* dy2, long lddy2, int dy2_f32, void* dx, long lddx, int dx_bf16, int accumulate,
                                  float* partial, float* partial2, int* nblk, void* cast, long ldcast, float alpha, float drop_p, unsigned seed,
                                  unsigned stream_id, int M, int d, mi_stream_t stream);
int mi_ln_partial_reduce_many(const mi_lnred_desc* descs, int n, mi_stream_t stream);
int mi_ln_apply_bf16(const void* x, long ldx, const float* stats, const float* gamma, const float* beta, void* y, long ldy,
                     int M, int N, mi_stream_t stream);
int mi_axpy_f32(float* a, const float* b, long n, float alpha, mi_stream_t stream);
int mi_scale_f32(float* a, long n, float alpha, mi_stream_t stream);
/* a *= *alpha_dev (a device scalar; a 16-B aligned); a no-op launch when it is exactly 1: the autograd bridge's d(loss) factor (autograd_bridge.py) */
int mi_scale_dev_f32(float* a, long n, const float* alpha_dev, mi_stream_t stream);
int mi_add2_cast_bf16(const float* a, long lda, const float* b, long ldb, void* out, long ldo, int M, int N, float alpha,
                      mi_stream_t stream);
int mi_add_rowvec_bf16(const void* x, long ldx, const float* vec, void* out, long ldo, int M, int N, mi_stream_t stream);
/* out_u = bf16(x + u), out_v = bf16(x + v) from one read of x (q + pos_bias_u / q + pos_bias_v: e_branchformer's relative-position attention, tf wav2vec2_conformer :466-470) */
int mi_add_rowvec2_bf16(const void* x, long ldx, const float* u, const float* v, void* out_u, void* out_v, long ldo, int M, int N, mi_stream_t stream);
int mi_gate_bwd_bf16(const void* ds, long ldds, const void* c, long ldc, const void* r, long ldr, void* dr, long lddr,
                     void* dc, long lddc, int M, int N, mi_stream_t stream);
/* valid frame counts behind `layers` Conv2d sub-sampling layers (kernel, stride, pad on the time axis) for B utterances: inner[b] = min(count with padding, tmax) — the
 * encoder's masks —, outer[b] = count without padding — the CTC loss's input lengths (reference: e_branchformer.py _get_feat_extract_output_lengths; floor division) */
int mi_subsampled_lengths_i32(const int* lengths, int B, int kernel, int stride, int pad, int layers, int tmax, int* inner, int* outer, mi_stream_t stream);
int mi_mask_rows_f32(float* x, long ld, const int* lengths, int T, int M, int N, mi_stream_t stream);
/* in-model SpecAugment of the encoder input (tf wav2vec2_conformer _mask_hidden_states :1086-1130): time_mask (M) / feat_mask (B, N) bytes */
int mi_spec_mask_apply(float* x, long ld, const unsigned char* time_mask, const float* embed, const unsigned char* feat_mask, int T, int M, int N,
                       mi_stream_t stream);
int mi_spec_mask_bwd(float* dx, long ld, const unsigned char* time_mask, float* dembed, const unsigned char* feat_mask, int T, int M, int N,
                     float* workspace /* ceil(M / 128) * N floats when dembed is given */, mi_stream_t stream);
/* ---- layer mixing of the CTC fine-tuning head — replaces src/models/bestrq.py:239-245
 *      (`(torch.stack(hidden_states) * softmax(per_layer_weights)[:, None, None, None]).sum(0)`) and its autograd backward.
 *  mi_softmax_vec_f32:      s = softmax(w), n <= 1024.        mi_softmax_vec_bwd_f32:  dw += s * (g - <s, g>)  (g_l = <d mixed, hidden_l>).
 *  mi_axpy_dev_f32:         a = (overwrite ? 0 : a) + alpha[0] * b  with the coefficient read from DEVICE memory (no host round trip).
 *  mi_dot_f32:              out[0] += <a, b>  (caller zeroes out; bit-reproducible; workspace: 1024 floats). */
int mi_softmax_vec_f32(const float* w, int n, float* s, mi_stream_t stream);
int mi_softmax_vec_bwd_f32(const float* s, const float* g, int n, float* dw, mi_stream_t stream);
int mi_axpy_dev_f32(float* a, const float* b, long n, const float* alpha, int overwrite, mi_stream_t stream);
int mi_dot_f32(const float* a, const float* b, long n, float* out, float* workspace /* 1024 floats */, mi_stream_t stream);
int mi_sumsq_f32(const float* x, long n, float* sumsq, float* workspace /* 1024 floats */, mi_stream_t stream);
/* norm_coef_skip (3 floats): [sqrt(sumsq), clip coefficient min(1, max_norm / (norm + 1e-6)), skip flag].  skip = 1 (and coef 0) when the norm is not
 * finite or exceeds skip_above > 0 — GradAwareTrainer.training_step drops such a step (src/utilities/training_utils.py:81,101-115). mi_adamw_step
 * given this array leaves parameters and moments untouched when skip is set. */
int mi_clip_coef(const float* sumsq, float max_norm, float skip_above, float* norm_coef_skip, mi_stream_t stream);
int mi_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* decay, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* norm_coef, void* mirror_bf16,
                  mi_stream_t stream);
/* weight-gradient GEMM dW (n_store, K) fp32 += dY[:, :N]^T · X (contraction over the rows of both operands, split over M) */
size_t mi_gemm_tn_workspace_bytes(int M, int N, int K);
int mi_gemm_tn_bf16(const void* dY, long ldy, const void* X, long ldx, float* dW, long ldo, float* db /* optional bias gradient */, int M, int N,
                    int K, int n_store, void* workspace, size_t workspace_bytes, int variant /* 0 = product tile (128 x 128); 1 = 128 x 64 for A/B */, mi_stream_t stream);
/* Conv2d weight gradient without an im2col buffer: dW (n_store, KH*KW*Cin) fp32 += dY[:, :N]^T · im2col(x); x (B,Tin,Fin,Cin) bf16 channels-last, dY rows (b,to,fo),
 * k = (kh*KW + kw)*Cin + c, square stride, leading pads (pad_t, pad_f); Cin % 128 == 0 else MI_ERR_UNSUPPORTED; workspace: mi_gemm_tn_workspace_bytes(B*Tout*Fout, N, KH*KW*Cin).
 * replaces: the weight gradient autograd derives for the second Conv2d of extractors.py:82-89. */
int mi_conv2d_wgrad_cl_bf16(const void* dY, long ldy, const void* x, float* dW, long ldo, float* db, int B, int Tin, int Fin, int Cin, int KH, int KW,
                            int stride, int pad_t, int pad_f, int Tout, int Fout, int N, int n_store, void* workspace, size_t workspace_bytes, mi_stream_t stream);
/* grouped form: the weight-gradient GEMMs of one encoder layer (n <= 48 problems, host arrays of length n) as ONE launch — together their 128 x 128 output tiles fill the
 * chip without splitting M, so every block adds its tile into dW in place: no slabs, no reduce pass, no workspace.  Use it for >= ~256 tiles in total. */
int mi_gemm_tn_group_bf16(int n, const void* const* dY, const long* ldy, const void* const* X, const long* ldx, float* const* dW, const long* ldo,
                          float* const* db, const int* M, const int* N, const int* K, const int* n_store, int tile_k, mi_stream_t stream);
/* the same with overwrite != 0: every dW_i / db_i is written (= instead of +=) — for targets known to hold zeros (a training step's first backward after the gradients
 * were cleared; each target the output of exactly one problem): the kernel's epilogue then has no dependent read of its output tile */
int mi_gemm_tn_group_ow_bf16(int n, const void* const* dY, const long* ldy, const void* const* X, const long* ldx, float* const* dW, const long* ldo,
                             float* const* db, const int* M, const int* N, const int* K, const int* n_store, int tile_k, int overwrite, mi_stream_t stream);
/* (tile_k: k extent of the 256-row output tiles: 128, 256, or 0 = 256 when the problems then still have >= 200 tiles between them, else 128) */
/* strided batched GEMM C[z1,z2] = alpha * A[z1,z2] · B[z1,z2]^T (+ C): the per-(utterance, head) products of attention backward */
int mi_bgemm_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k, const void* B, long b_z1, long b_z2, long b_n, long b_k,
                  void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha, int Z1, int Z2, int M, int N,
                  int K, mi_stream_t stream);
/* mi_bgemm_bf16 for an A operand banded along K: K = band_cg * band_T rows, row (u, i) of A non-zero only in columns [band_a - i, band_a - i + band_T) — the
 * un-shifted relative-position gradient dBD of mi_attention_qkv_bwd_probs in the d(positions) product.  All-zero k tiles are skipped (same result); band_T = 0: no band. */
int mi_bgemm_band_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k, const void* B, long b_z1, long b_z2, long b_n, long b_k,
                       void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha, int Z1, int Z2, int M, int N, int K,
                       int band_T, int band_a, int band_cg, mi_stream_t stream);
/* ... and with m_valid (Z2 ints on the device, or NULL): rows m >= m_valid[z2] of A are zero for batch entry z2 (keys beyond an utterance's length in the dV = P^T dctx and
 * dK = dS^T q products): those M tiles of C are stored as zeros (left alone when accumulating) without being read or multiplied. */
int mi_bgemm_sparse_bf16(const void* A, long a_z1, long a_z2, long a_m, long a_k, const void* B, long b_z1, long b_z2, long b_n, long b_k,
                         void* C, long c_z1, long c_z2, long c_m, int out_f32, int accumulate, float alpha, int Z1, int Z2, int M, int N, int K,
                         int band_T, int band_a, int band_cg, const int* m_valid, mi_stream_t stream);
/* softmax stage of attention (e_branchformer.py:100-135, tf wav2vec2_conformer:528-565 relative shift), head-major (H,B,Tq,Tk) */
int mi_attn_softmax_fwd(const float* ac, const float* bd, const int* lengths, void* prob, void* prob_drop, int H, int B, int Tq, int Tk,
                        long ld_s, long ld_p, float scale, int causal, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
int mi_attn_softmax_bwd(const void* prob, const float* dp, void* ds, void* dbd, int H, int B, int Tq, int Tk, long ld_s, long ld_p,
                        float scale, float drop_p, unsigned seed, unsigned stream_id, mi_stream_t stream);
/* dropout with counter-based masks (torch.nn.Dropout sites of e_branchformer.py:132,203,288,301,451, tf wav2vec2_conformer :353,356,674,
 * GPT-2 embd/attn/resid): keep(idx) is a pure function of (seed, stream_id, logical element index) — see csrc/dropout.hip */
int mi_dropout(const void* x, long ldx, int in_dtype, void* out, long ldo, int out_dtype, int M, int N, float alpha, float p, unsigned seed,
               unsigned stream_id, mi_stream_t stream);
int mi_dropout_add_f32(float* y, long ldy, const float* resid, long ldr, const float* t, long ldt, int M, int N, float alpha, float p,
                       unsigned seed, unsigned stream_id, mi_stream_t stream);
/* depthwise-conv / conv front-end gradients (e_branchformer.py:184-204,296-304; extractors.py:71-113) */
/* the two depthwise-conv backward entries below: dw == NULL leaves the per-utterance tap-gradient partials in `workspace` un-reduced (rows = B, or B * ceil(T / 64) for the
 * dilated form) for a later mi_ln_partial_reduce_many (desc kind = K) — several layers' reductions in one launch */
int mi_csgu_bwd_bf16(const void* u, long ldu, const float* stats, const float* gamma, const float* beta, const float* w,
                     const float* bias, const void* ds, long ldds, void* dr, long lddr, void* dgn, long lddgn, float* dw,
                     float* db, int B, int T, int C, int K, int pad_left, int dilation /* 1, or the causal form's (K-1)/2 */,
                     float* workspace /* B*C*32 floats or NULL */, mi_stream_t stream);
int mi_dwconv_residual_bwd_bf16(const void* m, long ldm, const float* w, const void* dy, long lddy, void* dm, long lddm,
                                float* dw, float* db, int B, int T, int C, int K, int pad_left, int dilation, float* workspace, mi_stream_t stream);
int mi_im2col_cl_bf16(const void* in, void* col, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride, int pad_t,
                      int pad_f, int Tout, int Fout, mi_stream_t stream);
int mi_conv2d_first_bwd(const float* x, const float* w, const float* bias, const void* dcol, float* dw, float* db, int B, int T,
                        int F, int C, int K, int stride, int pad_t, int pad_f, int T1, int F1, int K2, int stride2, int pad2_t,
                        int pad2_f, int T2, int F2, float* workspace /* mi_conv2d_first_bwd_workspace_floats: one partial row per block */, mi_stream_t stream);
size_t mi_conv2d_first_bwd_workspace_floats(int B, int C, int T1, int F1);
/* The second Conv2d's INPUT gradient (3x3, stride 2; what autograd derives for extractors.py:82-89's second layer) as four stride-1 implicit-GEMM convolutions, one per
 * parity class of (t1 + pad_t, f1 + pad_f), instead of one GEMM into the gradient of the im2col operand (9 C1 columns per output position of conv2) and a col2im gather:
 *   mi_conv2d_s2k3_dgrad_elems      bf16 elements of the four phase buffers, back to back; 0 = geometry not supported (keep mi_gemm_bf16 + mi_conv2d_first_bwd)
 *   mi_conv2d_s2k3_dgrad_pack_bf16  wT (9*C1 rows (kh,kw,c), C2 columns, row stride ldwt) -> the four phase weight matrices (9*C1*C2 elements), once per weight update
 *   mi_conv2d_s2k3_dgrad_bf16       dY2 (B,T2,F2,C2) contiguous -> phases
 *   mi_conv2d_first_bwd_phases      mi_conv2d_first_bwd reading the phase buffers (one 16-B load per position and 8 channels instead of up to four) */
size_t mi_conv2d_s2k3_dgrad_elems(int B, int T1, int F1, int C1, int T2, int F2, int pad_t, int pad_f);
int mi_conv2d_s2k3_dgrad_pack_bf16(const void* wT, long ldwt, void* packed, int C1, int C2, mi_stream_t stream);
int mi_conv2d_s2k3_dgrad_bf16(const void* dY2, const void* packed, void* phases, int B, int T1, int F1, int C1, int T2, int F2, int C2, int pad_t, int pad_f,
                              mi_stream_t stream);
int mi_conv2d_first_bwd_phases(const float* x, const float* w, const float* bias, const void* phases, float* dw, float* db, int B, int T, int F, int C, int K,
                               int stride, int pad_t, int pad_f, int T1, int F1, int pad2_t, int pad2_f, int T2, int F2,
                               float* workspace /* mi_conv2d_first_bwd_workspace_floats */, mi_stream_t stream);
/* backward pieces of the context-aware front ends (extractors.py:23-65), un-fused: im2col / col2im with separate strides, the backward of mi_gated_act_bf16
 * (dz = dout GELU'(y) sigmoid(g), dg = sum over the shared rows of dout GELU'(y) z sigmoid(g)(1 - sigmoid(g)), y = z sigmoid(g)), and the weight / bias gradient of a
 * Conv2d(1 -> C) of geometry (3,3) or (12,3) from the gradient of its raw output. */
int mi_im2col_cl_geo_bf16(const void* in, void* col, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride_t, int stride_f, int pad_t,
                          int pad_f, int Tout, int Fout, mi_stream_t stream);
int mi_col2im_cl_bf16(const void* dcol, void* din, int B, int Tin, int Fin, int Cin, int KH, int KW, int stride_t, int stride_f, int pad_t, int pad_f,
                      int Tout, int Fout, int accumulate, mi_stream_t stream);
int mi_gated_act_bwd_bf16(const void* dout, long lddo, const void* z, long ldz, const void* g, long ldg, void* dz, long lddz, void* dg, long lddg,
                          int B, int T, int Fq, int C, int share, mi_stream_t stream);
int mi_conv2d_first_wgrad(const float* x, const void* dy, float* dw, float* db, int B, int T, int F, int C, int KH, int KW, int stride_t, int stride_f,
                          int pad_t, int pad_f, int T1, int F1, float* workspace /* mi_conv2d_first_wgrad_workspace_floats */, mi_stream_t stream);
size_t mi_conv2d_first_wgrad_workspace_floats(int B, int C, int KH, int KW, int T1, int F1);
/* loss gradients: F.ctc_loss backward composed with log_softmax (e_branchformer.py:472-488); label-smoothed CE of the decoder
 * heads (multi_head_gpt2.py:138-158); embedding scatter (embeddings.py:33-62) */
size_t mi_ctc_bwd_workspace_bytes(int B, int T, int U);
int mi_ctc_loss_bwd(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T, const long* labels, int U,
                    const int* in_len, int blank, int B, int reduction, const float* nll, float gscale, void* workspace,
                    size_t workspace_bytes, void* dlogits, long ldo, mi_stream_t stream);
/* loss AND gradient from one pair of recursions (the training step): as mi_ctc_loss_bwd, but the per-utterance negative log-likelihood comes OUT of the call (its own alpha
 * recursion: nll_out (B), tgt_len_out (B), loss_out (1, nullable) = the reduced loss of mi_ctc_loss_fwd with the same reduction / zero_infinity semantics) instead of
 * going in — no forward loss kernel.  MI_ERR_UNSUPPORTED for targets of more than 63 labels (the caller runs mi_ctc_loss_fwd + mi_ctc_loss_bwd). */
int mi_ctc_loss_bwd_nll(const void* logits, long ld_b, long ld_t, int dtype, const float* lse, int T, const long* labels, int U,
                        const int* in_len, int blank, int B, int reduction, int zero_infinity, float gscale, void* workspace,
                        size_t workspace_bytes, void* dlogits, long ldo, float* nll_out, int* tgt_len_out, float* loss_out, mi_stream_t stream);
/* the reduction of mi_ctc_loss_fwd alone */
int mi_ctc_reduce(const float* nll, const int* tgt_len, int B, int reduction, int zero_infinity, float* loss, mi_stream_t stream);
int mi_ce_label_smoothing_bwd(const float* logits, long ld, const long* labels, int B, int U, int shift, int V, float eps,
                              float weight, const float* acc, void* dlogits, long ldo, mi_stream_t stream);
/* as gathers (a block per 16 vocabulary entries / per position adds its rows in order: no atomics).  heavy_id: an entry expected on a large share of the rows (the padding
 * token of the shifted decoder input) is summed as a masked column sum instead, or -1; workspace: mi_embed_tokens_bwd_workspace_bytes(M, d, V) bytes; d <= 1024 */
size_t mi_embed_tokens_bwd_workspace_bytes(int M, int d, int V);
int mi_embed_tokens_bwd(const long* ids, const float* dx, float scale, int pos_offset, int U, int d, int M, int V, float* dwte,
                        float* dwpe, int heavy_id, void* workspace, mi_stream_t stream);

/* ---- feature-level SpecAugment on the device (src/augmentations/spec_aug.py:40-137: bicubic time warp + frequency / time masks) */
int mi_specaug_f32(const float* x, float* out, int B, int T, int F, const int* params, int nf, int nt, float pad_value, mi_stream_t stream);

/* Speed perturbation = polyphase sinc resampling of a (B, N) fp32 waveform batch by orig/nw (both divided by their gcd), the arithmetic of
 * torchaudio.transforms.SpeedPerturbation which the reference's training pre-processing names first (configs/default_data_preprocessing2d.json:3-19;
 * torchaudio functional.py `speed` / `_apply_sinc_resample_kernel`).  kernel: (nw, 2*width + orig) fp32 table built by the host;
 * N_out must equal ceil(nw * N / orig); lengths / out_lengths (B) int32 or NULL (out = ceil(len * nw / orig)). */
int mi_speed_resample_f32(const float* wave, long ld, const int* lengths, int B, int N, int orig, int nw, const float* kernel, int width,
                          float* out, long ld_out, int N_out, int* out_lengths, mi_stream_t stream);

/* ---- BEST-RQ pre-training (src/models/bestrq.py:66-97): random-projection quantizer targets, noise masking of the encoder input */
int mi_rpq_targets(const float* x, long ldx, const float* P, const float* CB, long* targets, int M, int in_dim, int cd, int C, int books,
                   mi_stream_t stream);
int mi_mask_noise_f32(float* x, long ld, const unsigned char* time_mask, int M, int N, float std, unsigned seed, unsigned stream_id,
                      mi_stream_t stream);

/* ---- GPT-2 decoder token step as one call (KV cache, cross-attention over cached encoder K/V) + beam re-ordering of the caches.
 * replaces: GPT2LMMultiHeadModel.forward with past_key_values (multi_head_gpt2.py:80-170; tf gpt2 :262-310) and `_reorder_cache`. */
typedef struct {
    int d, H, L, V;
    float eps;
    int step_form;      /* token step with one new token per row and <= 8 rows: 0 = the fused form (three launches per layer, csrc/decoder_fused.hip) where it applies
                           (head size 64, d <= 512), 1 = always one launch per op (the cross-check; what every other shape runs) */
} mi_gpt2_config;
size_t mi_gpt2_step_workspace_bytes(const mi_gpt2_config* cfg, int B, int U);
int mi_gpt2_step(const mi_gpt2_config* cfg, const void* const* weights, const long* ids_new, int B, int U, int past, int Lmax,
                 void* const* kcache, void* const* vcache, const void* const* cross_kv, int T_enc, const int* enc_len, float emb_scale,
                 void* workspace, size_t workspace_bytes, float* logits, long ld_logits, mi_stream_t stream);
int mi_kv_cache_reorder(const void* const* src_k, const void* const* src_v, void* const* dst_k, void* const* dst_v, const long* beam_idx,
                        int L, int BW, int rows, int Lmax, int d, mi_stream_t stream);

/* One beam-search step of joint CTC / attention decoding on the device (csrc/beam_step.hip): what the reference gets per token from transformers' beam loop on the host
   (src/models/ctc_encoder_plus_autoregressive_decoder.py:360-482; processors src/decoding/ctc_scorer.py:259-365; the loop itself: transformers/generation/utils.py
   `_beam_search` of the installed 5.x, the rules tests/golden/gen_*.npz pin from the reference's own generate()).
   cand = ((1 - w)(logits - lse, pad -> logzero when mask_pad) + w ctc) + beam_scores  [ctc null: no mix];  top 2W per utterance, best first, ties by lower index;
   a candidate stops when its token is EOS or cur_len + 1 >= max_length; the first W that did not stop become the next beams: ids (B*W, Lmax) re-ordered in place + the
   new token at column cur_len, new_tok / beam_idx / beam_scores (B*W) written; stopped candidates among the first W ranks enter the kept hypotheses with
   score / denom (denom = (cur_len + 1 - prompt) ** length_penalty as fp32, computed by the caller): fin_score / fin_len (B, W), fin_tok (B, W, Lmax), nfin (B) hold the
   best W, best first.  done[b] is set when W are kept and beam_scores[b, 0] / heur_denom cannot beat the worst of them (heur_denom = (hypothetical length) **
   length_penalty: cur_len, or max_length - prompt for early_stopping 2 = "never" with a positive penalty), when early_stopping is 1 (True) and W are kept, or at
   max_length; a done utterance only emits pad tokens from beam b*W.
   top_s / top_i (B, 2W) optional; done_out (B) optional: the flags after the step, written for the host (pinned, device-mapped memory).  W <= 16, W * V < 2^24. */
int mi_beam_step(const float* logits, long ldl, const float* lse, const float* ctc, float w_att, float w_ctc, int mask_pad, int pad, int eos, int B, int W, int V,
                 int cur_len, int max_length, int Lmax, float denom, float heur_denom, int early_stopping, long* ids, float* beam_scores, long* new_tok, long* beam_idx,
                 int* done, int* nfin, float* fin_score, int* fin_len, long* fin_tok, float* top_s, int* top_i, int* done_out, mi_stream_t stream);

/* ---- Whisper-style front end + glue (BASELINE config 4).  replaces: transformers WhisperFeatureExtractor numpy path
 *      (selected by configs/default_data_preprocessing_whisper.json:20-29) and the conv/position prologue of WhisperEncoder. */
/* scratch: B * frames * nmel + B floats (frames = n_samples / 160): the log-mel before the clamp, then the per-clip maxima. */
int mi_whisper_logmel(const float* wave, long ldw, const int* num_samples, int n_samples, const double* window,
                      const double* twiddle, const double* mel_t, int nmel, int B, float* scratch,
                      float* out_features, void* out_cl_bf16, mi_stream_t stream);
int mi_transpose_cast_bct_btc(const float* x, void* out_bf16, int B, int C, int T, mi_stream_t stream);
int mi_add_positions(const void* a_bf16, const float* pos, float* x, int M, int T, int d, mi_stream_t stream);

/* ---- whole encoder + CTC head: Wav2Vec2EBranchformerForCTC.forward (e_branchformer.py:422-496), eval mode. */
typedef struct {
    int B, T, F;                 /* padded log-mel input (B,T,F) fp32 */
    int d, H, I, L, V;           /* hidden, heads, intermediate, layers, vocab (blank excluded) */
    int C1, C2, K, stride, pad;  /* Conv2d sub-sampling: 1->C1->C2, KxK, stride, padding */
    int is_causal;
    int pos_type;                /* 0 none, 1 relative, 2 rotary */
    int csgu_kernel, merge_kernel, csgu_act, use_macaron;
    float ln_eps;                /* feature-projection / encoder LayerNorm eps (layer LNs use 1e-5) */
    int logits_f32;              /* 1: fp32 logits, 0: bf16 logits */
    int logits_ld;               /* row stride of the logits buffer in elements (0 = V+1); a multiple of 8 keeps the stores 16-B wide */
    int branch_overlap;          /* experimental, default 0: run each layer's local (cgMLP) branch on a library-owned side stream beside the
                                    attention branch (one forward in flight at a time); see DESIGN.md 'Concurrent kernels' before enabling */
    /* CTC fine-tuning head of a BEST-RQ encoder — BestRQEBranchformerForCTC.forward, src/models/bestrq.py:239-274 */
    int extra_layers;            /* 0 | 1: `finetune_with_additional_layer` — one more layer (weight-table layer index L) between the encoder's
                                    final LayerNorm and the head: padded frames zeroed, same key mask and position table, no LayerNorm after it */
    int layer_mixing;            /* `finetune_with_layer_mixing`: softmax(per_layer_weights, global slot 14)-weighted sum of the L+1 hidden states
                                    (the input of every layer + the final LayerNorm's output) replaces the last hidden state */
    int csgu_linear;             /* `csgu_use_linear_after_conv` (e_branchformer.py:172-175,198-199): a Linear (layer slots CSGU_LIN_W bf16 (I/2, I/2) / CSGU_LIN_B) between
                                    the CSGU conv and its activation */
    int context_mode;            /* `context_awareness_type` (extractors.py:57-65) of a non-causal encoder: 0 plain conv (None and every unknown string, as the reference's dict
                                    lookup resolves them), 1 "gated", 2 "gated_shared".  Global slots: GATE1_W (15) f32 (C1, KH*KW), GATE1_B (16), and for mode 2 GATE2_W (17) bf16
                                    (C2, 12*3*C1), GATE2_B (18).  Mode 1: CONV2_W / CONV2_B hold conv AND gate, (2*C2, 9*C1) / (2*C2), interleaved in blocks of `gate_blk` channels */
    int gate_blk;                /* mode 1: 32 (the fused epilogue's packing; needs C2 % 32 == 0) or C2 (conv rows, then gate rows) */
    int ln_fold;                 /* 1: the LayerNorms in front of the FFN-in / QKV / cgMLP-in GEMMs are folded into those GEMMs (LN(x) W^T = rstd (x W'^T) - rstd mu s + c, W' = W diag(gamma)):
                                    the N = d GEMMs that produce the residual stream also emit its bf16 copy and per-row partial statistics, the consumers take those instead of
                                    a LayerNorm kernel's output — one LayerNorm launch per layer instead of three.  Needs the layer slots *_WF / *_SF / *_CF (engine.py LS), relative
                                    or no positions, macaron FFNs, d in {256, 512}, I % 256 == 0, no fine-tuning head.  0: LayerNorm kernels + plain GEMMs. */
    int wide_tiles;              /* throughput mode (with ln_fold): the N = d GEMMs of a layer (FFN out x2, attention out, cgMLP out, merge) run on 256 x 256 tiles too — a quarter
                                    of the blocks of the 128 x 128 kernel (64 per launch at M = 8000, d = 512), 2x the FLOP per ingested byte.  Alone such a launch leaves most
                                    of the chip idle; it is meant for several independent steps in flight on their own streams (huggingface_asr_amd/pipeline.py), whose kernels
                                    then share the chip by CU instead of taking turns on all of it.  Same results as 0 up to the summation order of the row statistics. */
} mi_ebf_config;

/* weight-table slot indices: see huggingface_asr_amd/engine.py (SLOTS) — the table is an array of device pointers. */
enum { MI_EBF_GLOBAL_SLOTS = 24, MI_EBF_LAYER_SLOTS = 64 };

size_t mi_ebf_workspace_bytes(const mi_ebf_config* cfg);

/* feats (B,T,F) fp32; feat_lengths (B) int32 valid frames (attention_mask.sum(-1)) or NULL;
 * pos_table: relative: (2*T2-1, d) bf16 sinusoid table; rotary: fp32 [cos (T2,hd) | sin (T2,hd)]; none: NULL;
 * posp: (L, 2*T2-1, d) bf16 scratch for the projected relative positions, recomputed when compute_posp != 0
 *       (it depends only on the weights and T2, so callers cache it across calls at inference);
 * workspace: mi_ebf_workspace_bytes(cfg) bytes, ZERO-INITIALISED ONCE by the caller (time padding of V^T);
 * outputs: last_hidden (B*T2, d) fp32 (nullable), logits (B*T2, V+1) fp32|bf16 (nullable),
 *          inner_len/outer_len (B) int32 (mask lengths / CTC input lengths, nullable).
 * With cfg->extra_layers the weight table holds L+1 layers and posp (L+1, 2*T2-1, d); last_hidden stays the ENCODER's last hidden state. */
int mi_ebf_forward(const mi_ebf_config* cfg, const void* const* weights, const float* feats, const int* feat_lengths,
                   const void* pos_table, void* posp, int compute_posp, void* workspace, size_t workspace_bytes,
                   float* last_hidden, void* logits, int* inner_len, int* outer_len, mi_stream_t stream);
/* the same, additionally filling hidden_states (L+1, B*T2, d) fp32 = HuggingFace's `output_hidden_states` tuple (transformers wav2vec2_conformer :685-713): the
 * input of every encoder layer, then the last hidden state; needs last_hidden != NULL. */
int mi_ebf_forward_hs(const mi_ebf_config* cfg, const void* const* weights, const float* feats, const int* feat_lengths,
                      const void* pos_table, void* posp, int compute_posp, void* workspace, size_t workspace_bytes,
                      float* last_hidden, void* logits, int* inner_len, int* outer_len, float* hidden_states, mi_stream_t stream);
/* the same, additionally leaving lse (B*T2) fp32 = the row log-sum-exp of the fp32 logits (the log_softmax of reference e_branchformer.py:472-488 is never materialised: the
 * CTC loss takes the logits and this vector) out of the head GEMM's epilogue — mi_gemm_lse_f32; lse_workspace: mi_gemm_lse_workspace_floats(B*T2, V+1) floats.
 * lse and lse_workspace both or neither; needs fp32 logits. */
int mi_ebf_forward_lse(const mi_ebf_config* cfg, const void* const* weights, const float* feats, const int* feat_lengths,
                       const void* pos_table, void* posp, int compute_posp, void* workspace, size_t workspace_bytes,
                       float* last_hidden, void* logits, int* inner_len, int* outer_len, float* hidden_states,
                       float* lse, float* lse_workspace, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
