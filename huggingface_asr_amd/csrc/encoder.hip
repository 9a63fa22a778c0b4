// Whole-encoder driver: Wav2Vec2EBranchformerForCTC.forward in eval mode as one C call that enqueues
// every kernel of the path on the caller's stream (no allocation, no sync -> hipGraph-capturable).
//
// Reference call stack: e_branchformer.py:422-457 -> wav2vec2_conformer model forward
// (:1133-1195: conv sub-sampling, mask, feature projection) -> encoder loop (:651-717) with
// Wav2Vec2EBranchformerEncoderLayer.forward (e_branchformer.py:263-313) -> lm_head ⊕ blank_projection.
//
// Data layout in HBM (all row-major, rows = b*T2 + t):
//   x     fp32 (M, d)     residual stream (the reference's stream is fp32 under bf16 autocast too)
//   a0/a1/a2 bf16 (M, d)  LayerNorm outputs = GEMM A operands
//   h     bf16 (M, I)     FFN hidden / cgMLP channel_proj1 output [x_r | x_g]
//   qk    bf16 (M, 2d)    [Q | K] projections; vt bf16 (d, B*Tp) = V^T, key-contiguous, time padded to 32
//   cat   bf16 (M, 2d)    [attention branch | cgMLP branch]  (torch.cat at e_branchformer.py:296 is free)
//   act1  bf16 (B,T1,F1,C1), act2 bf16 (B,T2,F2,C2) channels-last conv activations
#include "common.hpp"
#include <hip/hip_ext.h>
#include <stdio.h>
#include "../../include/hfasr_hip.h"

namespace {

enum G { G_CONV1_W, G_CONV1_B, G_CONV2_W, G_CONV2_B, G_FEOUT_W, G_FEOUT_B, G_FP_LN_G, G_FP_LN_B, G_FP_W, G_FP_B,
         G_ENC_LN_G, G_ENC_LN_B, G_HEAD_W, G_HEAD_B, G_MIX_W /* per_layer_weights (L+1) f32, bestrq.py:202-205 */,
         G_GATE1_W, G_GATE1_B, G_GATE2_W, G_GATE2_B /* gate filters of the context-aware front ends (extractors.py:23-54) */ };
enum LS { FF1_LN_G, FF1_LN_B, FF1_W1, FF1_B1, FF1_W2, FF1_B2,
          ATT_LN_G, ATT_LN_B, ATT_WQK, ATT_BQK, ATT_WV, ATT_BV, ATT_WO, ATT_BO, ATT_WPOS, ATT_U, ATT_V,
          MLP_LN_G, MLP_LN_B, MLP_W1, MLP_B1, CSGU_LN_G, CSGU_LN_B, CSGU_W, CSGU_B, MLP_W2, MLP_B2,
          MRG_DW_W, MRG_DW_B, MRG_W, MRG_B, FIN_LN_G, FIN_LN_B,
          FF2_LN_G, FF2_LN_B, FF2_W1, FF2_B1, FF2_W2, FF2_B2, CSGU_LIN_W, CSGU_LIN_B,
          // ln_fold: W' = bf16(W diag(gamma)), its fp32 column sums, and W beta + b for the four LayerNorm -> Linear pairs of a layer
          FF1_WF, FF1_SF, FF1_CF, QKV_WF, QKV_SF, QKV_CF, MLP_WF, MLP_SF, MLP_CF, FF2_WF, FF2_SF, FF2_CF };

struct Dims { int T1, F1, T2, F2, M, Tp, hd; };

int conv_out(int n, int k, int s, int pad_total) { return (n + pad_total - k) / s + 1; }

Dims dims(const mi_ebf_config& c) {
    Dims d;
    const int pt = c.is_causal ? 2 * c.pad : 2 * c.pad;     // causal: all padding on the left (streaming_modules.py:31-55)
    d.T1 = conv_out(c.T, c.K, c.stride, pt); d.F1 = conv_out(c.F, c.K, c.stride, pt);
    d.T2 = conv_out(d.T1, c.K, c.stride, pt); d.F2 = conv_out(d.F1, c.K, c.stride, pt);
    d.M = c.B * d.T2; d.Tp = (d.T2 + 31) / 32 * 32; d.hd = c.d / c.H;
    return d;
}

struct Carver {
    char* base; size_t off;
    void* take(size_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; }
};

struct Ws {
    bf16_t *act1, *act2, *a0, *a1, *a2, *a1r, *h, *qk, *vt, *ctx, *cat, *m2, *s, *hid;
    bf16_t *cv, *lin;         // csgu_use_linear_after_conv: the CSGU conv output and the Linear's output
    bf16_t *z1, *g1, *z2, *g2;   // context-aware front ends: raw conv / gate outputs of the un-fused forms
    float *feo, *x, *stats, *lnst;   // lnst: ln_fold's per-row partial (sum, sumsq) records, 32 floats per row
    float *mixed, *lh, *sw;   // fine-tuning head: weighted sum of the hidden states, fp32 copy of the last one, softmax(per_layer_weights)
    int* lens;   // [inner(B) | outer(B)]
    size_t bytes;
};

Ws carve(const mi_ebf_config& c, void* base) {
    const Dims d = dims(c);
    Carver k{(char*)base, 0};
    Ws w;
    const size_t M = d.M;
    w.act1 = (bf16_t*)k.take((size_t)c.B * d.T1 * d.F1 * c.C1 * 2);
    w.act2 = (bf16_t*)k.take((size_t)c.B * d.T2 * d.F2 * c.C2 * 2);
    w.z1 = w.g1 = w.z2 = w.g2 = nullptr;
    if (c.context_mode == 1) {            // the un-fused fallbacks of GatedConv2d: layer 1 [conv | gate] as two tensors, layer 2 as one stacked (M2, 2*C2) GEMM output
        const bool fused1 = c.K == 3 && (c.C1 % 4) == 0 && c.C1 / 4 <= 256 && (long)c.B * d.T1 * d.F1 < (1L << 31) - 256L * 8192;      // = mi_conv2d_first_gated_gelu's own condition
        if (!fused1) {
            w.z1 = (bf16_t*)k.take((size_t)c.B * d.T1 * d.F1 * c.C1 * 2);
            w.g1 = (bf16_t*)k.take((size_t)c.B * d.T1 * d.F1 * c.C1 * 2);
        }
        w.z2 = (bf16_t*)k.take((size_t)c.B * d.T2 * d.F2 * 2 * c.C2 * 2);
    } else if (c.context_mode == 2) {     // GatedConv2dShared: the gate has a quarter of the conv's time steps
        w.z1 = (bf16_t*)k.take((size_t)c.B * d.T1 * d.F1 * c.C1 * 2);
        w.g1 = (bf16_t*)k.take((size_t)c.B * (d.T1 / 4 + 1) * d.F1 * c.C1 * 2);
        w.z2 = (bf16_t*)k.take((size_t)c.B * d.T2 * d.F2 * c.C2 * 2);
        w.g2 = (bf16_t*)k.take((size_t)c.B * (d.T2 / 4 + 1) * d.F2 * c.C2 * 2);
    }
    w.feo = (float*)k.take(M * c.d * 4);
    w.x = (float*)k.take(M * c.d * 4);
    w.a0 = (bf16_t*)k.take(M * c.d * 2);
    w.a1 = (bf16_t*)k.take(M * c.d * 2);
    w.a2 = (bf16_t*)k.take(M * c.d * 2);
    w.a1r = (bf16_t*)k.take(M * c.d * 2);
    w.h = (bf16_t*)k.take(M * c.I * 2);
    w.qk = (bf16_t*)k.take(M * 3 * c.d * 2);      // [Q | K | V] (V columns used by the LDS-staged attention only)
    w.vt = (bf16_t*)k.take((size_t)c.d * c.B * d.Tp * 2);
    w.ctx = (bf16_t*)k.take(M * c.d * 2);
    w.cat = (bf16_t*)k.take(M * 2 * c.d * 2);
    w.m2 = (bf16_t*)k.take(M * 2 * c.d * 2);
    w.s = (bf16_t*)k.take(M * (c.I / 2) * 2);
    w.cv = w.lin = nullptr;
    if (c.csgu_linear) { w.cv = (bf16_t*)k.take(M * (c.I / 2) * 2); w.lin = (bf16_t*)k.take(M * (c.I / 2) * 2); }
    w.hid = (bf16_t*)k.take(M * c.d * 2);
    w.stats = (float*)k.take(M * 2 * 4);
    w.lnst = c.ln_fold ? (float*)k.take(M * 32 * 4) : nullptr;
    w.lens = (int*)k.take((size_t)2 * c.B * 4);
    w.mixed = w.lh = w.sw = nullptr;
    if (c.layer_mixing) {
        w.mixed = (float*)k.take(M * c.d * 4);
        w.sw = (float*)k.take((size_t)(c.L + 1) * 4);
    }
    if (c.layer_mixing || c.extra_layers) w.lh = (float*)k.take(M * c.d * 4);
    w.bytes = k.off;
    return w;
}

// lengths: inner = padded conv formula (extractors.py:133-162), outer = un-padded formula
// (Wav2Vec2ForCTC._get_feat_extract_output_lengths; quirk of SURVEY.md §8a row 8')
__global__ void lengths_kernel(const int* feat_len, int T, int B, int K, int stride, int pad, int causal, int nconv,
                               int T2, int* inner, int* outer) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int li = feat_len ? feat_len[b] : T, lo = li;
    for (int i = 0; i < nconv; ++i) {
        const int num_i = li + (causal ? K - 1 : 2 * pad) - K;
        li = (num_i >= 0 ? num_i / stride : -((-num_i + stride - 1) / stride)) + 1;
        const int num_o = lo - K;
        lo = (num_o >= 0 ? num_o / stride : -((-num_o + stride - 1) / stride)) + 1;
    }
    inner[b] = li < T2 ? li : T2;
    outer[b] = lo;
}

#define RUN(expr) do { int rc__ = (expr); if (rc__ != MI_OK) return rc__; } while (0)

}  // namespace

// diagnostics: text of the last failed launch on this thread ("" if none); not part of the data path
static thread_local char g_last_error[256] = "";
extern "C" void mi_record_hip_error(int code, const char* file, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s (hipError %d) at %s:%d", hipGetErrorString((hipError_t)code), code, file, line);
}
extern "C" const char* mi_last_error(void) { return g_last_error; }

// ---- profiling facility (diagnostics, off by default): the duration of every dense-contraction launch, taken from the dispatch's own begin / end
// timestamps (hipExtLaunchKernelGGL with a start / stop event pair: gemm_args.hpp `launch_dense`) — the figure rocprofv3 --kernel-trace reports for the
// same launch.  Used by bench.py for roofline.achieved.  Process-global, one profiled stream at a time.
namespace {
struct Prof { hipEvent_t* ev = nullptr; double* flops = nullptr; int* family = nullptr; int cap = 0, n = 0, stride = 1, open = -1; long seen = 0; bool on = false; };
Prof g_prof;
}
extern "C" int mi_profile_create(int capacity) {
    if (g_prof.ev) return MI_ERR_ARG;
    g_prof.ev = new hipEvent_t[2 * (size_t)capacity];
    g_prof.flops = new double[capacity];
    g_prof.family = new int[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return MI_ERR_LAUNCH;
    g_prof.cap = capacity; g_prof.n = 0; g_prof.on = false; g_prof.open = -1;
    return MI_OK;
}
// on = 0: off; on = s > 0: time every s-th launch (s = 1: all)
extern "C" void mi_profile_enable(int on) { g_prof.on = on != 0; g_prof.stride = on > 0 ? on : 1; g_prof.seen = 0; g_prof.open = -1; }
extern "C" void mi_profile_reset(void) { g_prof.n = 0; g_prof.open = -1; }
extern "C" int mi_profile_count(void) { return g_prof.n; }
extern "C" int mi_profile_hook_begin(hipStream_t stream, double flops) {
    (void)stream;
    if (!g_prof.on || g_prof.n >= g_prof.cap) return -1;
    if ((g_prof.seen++ % g_prof.stride) != 0) return -1;
    const int slot = g_prof.n++;
    g_prof.flops[slot] = flops;
    g_prof.family[slot] = -1;              // set by the launcher that takes the events; a slot nobody took counts nothing
    g_prof.open = slot;
    return slot;
}
extern "C" int mi_profile_take_events(hipEvent_t* start, hipEvent_t* stop, int family) {
    const int slot = g_prof.open;
    if (slot < 0) return 0;
    g_prof.open = -1;
    g_prof.family[slot] = family;
    *start = g_prof.ev[2 * slot]; *stop = g_prof.ev[2 * slot + 1];
    return 1;
}
extern "C" void mi_profile_hook_end(int slot, hipStream_t stream) { (void)slot; (void)stream; g_prof.open = -1; }
// after the stream is synchronised: sum of the kernel durations (ms), of the algorithmic flops and the number of recorded launches — all families (family < 0)
// or one (gemm_args.hpp PF_*)
extern "C" int mi_profile_summary_family(int family, double* total_ms, double* total_flops, int* launches) {
    double ms = 0.0, fl = 0.0;
    int n = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (g_prof.family[i] < 0 || (family >= 0 && g_prof.family[i] != family)) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MI_ERR_LAUNCH;
        ms += t; fl += g_prof.flops[i]; ++n;
    }
    *total_ms = ms; *total_flops = fl;
    if (launches) *launches = n;
    return MI_OK;
}
extern "C" int mi_profile_summary(double* total_ms, double* total_flops) { return mi_profile_summary_family(-1, total_ms, total_flops, nullptr); }

// What the (start, stop) event pair of hipExtLaunchKernelGGL adds to a kernel's own dispatch begin -> end time: the pair around an EMPTY one-wave kernel, median of n (ms).
// rocprofv3 --kernel-trace reports the empty kernel itself (`profile_null_kernel`) at ~1.3 us on MI355X; the rest of the figure returned here is the bracket's
// (the runtime's barrier packet in front of a profiled dispatch: 4.1-4.5 us, the same for every kernel family — profiles/r03_*).  bench.py's fall-back roofline leg
// subtracts (this - 1.3 us) per launch; its primary leg reads the dispatch timestamps from a rocprofv3 child run instead.
namespace { __global__ void profile_null_kernel() {} }
extern "C" int mi_profile_calibrate(hipStream_t stream, int n, double* median_ms) {
    if (n <= 0 || n > 256) return MI_ERR_ARG;
    hipEvent_t a[256], b[256];
    float t[256];
    for (int i = 0; i < n; ++i) {
        if (hipEventCreate(&a[i]) != hipSuccess || hipEventCreate(&b[i]) != hipSuccess) return MI_ERR_LAUNCH;
        hipExtLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, stream, a[i], b[i], 0);
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return MI_ERR_LAUNCH;
    for (int i = 0; i < n; ++i) {
        t[i] = 0.f;
        (void)hipEventElapsedTime(&t[i], a[i], b[i]);
        (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]);
    }
    for (int i = 1; i < n; ++i) { const float v = t[i]; int j = i - 1; while (j >= 0 && t[j] > v) { t[j + 1] = t[j]; --j; } t[j + 1] = v; }
    *median_ms = t[n / 2];
    return MI_OK;
}

// side stream + fork/join events of the experimental branch overlap (created on first use; one forward in flight at a time)
namespace {
constexpr int MAX_OVL_LAYERS = 64;
hipStream_t g_side = nullptr;
hipEvent_t g_fork[MAX_OVL_LAYERS], g_join[MAX_OVL_LAYERS];      // one pair per layer: an event is never re-recorded while a wait on it may be pending
bool side_ready() {
    if (g_side) return true;
    if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) { g_side = nullptr; return false; }
    for (int i = 0; i < MAX_OVL_LAYERS; ++i)
        if (hipEventCreateWithFlags(&g_fork[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g_join[i], hipEventDisableTiming) != hipSuccess) {
            g_side = nullptr;
            return false;
        }
    return true;
}
}  // namespace

extern "C" size_t mi_ebf_workspace_bytes(const mi_ebf_config* cfg) { return carve(*cfg, nullptr).bytes; }

// posp: (L, 2*T2-1, d) bf16 projected relative positions, (re)computed from pos_table when compute_posp != 0
// hidden_states (nullable): (L+1, B*T2, d) fp32 = the tuple HF returns for output_hidden_states (tf:685-713): the input of every encoder layer, then the
// encoder's last hidden state (after encoder.layer_norm); needs last_hidden.  Device-to-device copies on the same stream, only when asked for.
// lse / lse_workspace (both or neither; fp32 logits only): lse (B*T2) fp32 = the row log-sum-exp of the logits out of the head GEMM's epilogue (mi_gemm_lse_f32;
// lse_workspace >= mi_gemm_lse_workspace_floats(B*T2, V+1) floats) — what the CTC loss needs beside the logits, without a second pass over them.
extern "C" int mi_gemm_lse_f32(const void* A, long lda, const void* W, long ldw, const float* bias, float* C, long ldc, float* lse, float* workspace,
                               int M, int N, int K, hipStream_t stream);
extern "C" int mi_row_lse(const void* x, long ld, int dtype, int V, float* lse, int M, hipStream_t stream);
extern "C" int mi_ebf_forward_lse(const mi_ebf_config* cfg, const void* const* weights, const float* feats,
                                  const int* feat_lengths, const void* pos_table, void* posp, int compute_posp,
                                  void* workspace, size_t workspace_bytes, float* last_hidden, void* logits,
                                  int* inner_len, int* outer_len, float* hidden_states, float* lse, float* lse_workspace, hipStream_t st) {
    MI_ENTER();
    if ((lse != nullptr) != (lse_workspace != nullptr) || (lse && (!logits || !cfg->logits_f32))) return MI_ERR_ARG;
    const mi_ebf_config& c = *cfg;
    if (hidden_states && !last_hidden) return MI_ERR_ARG;
    if (c.B <= 0 || c.T <= 0 || c.L <= 0 || c.d % c.H || c.I % 2) return MI_ERR_ARG;
    if (c.extra_layers < 0 || c.extra_layers > 1 || (c.layer_mixing && c.L + 1 > 1024)) return MI_ERR_ARG;
    const int Lt = c.L + c.extra_layers;      // the fine-tuning head's additional layer (bestrq.py:247-274) is layer L of the weight table
    const Dims D = dims(c);
    if (D.T2 <= 0) return MI_ERR_ARG;
    Ws w = carve(c, workspace);
    if (w.bytes > workspace_bytes) return MI_ERR_ARG;
    auto Gw = [&](int s) { return weights[s]; };
    auto Lw = [&](int l, int s) { return weights[MI_EBF_GLOBAL_SLOTS + l * MI_EBF_LAYER_SLOTS + s]; };
    auto Gf = [&](int s) { return (const float*)weights[s]; };
    auto Lf = [&](int l, int s) { return (const float*)weights[MI_EBF_GLOBAL_SLOTS + l * MI_EBF_LAYER_SLOTS + s]; };
    const int M = D.M, d = c.d, I = c.I, T2 = D.T2;
    const float leps = 1e-5f;
    int* inner = inner_len ? inner_len : w.lens;
    int* outer = outer_len ? outer_len : w.lens + c.B;

    hipLaunchKernelGGL(lengths_kernel, dim3(cdiv(c.B, 64)), dim3(64), 0, st, feat_lengths, c.T, c.B, c.K, c.stride, c.pad,
                       c.is_causal, 2, T2, inner, outer);
    const int* mask_len = feat_lengths ? inner : nullptr;

    // --- Conv2d sub-sampling (extractors.py:110-113)
    const int pl = c.is_causal ? 2 * c.pad : c.pad;
    if (c.context_mode < 0 || c.context_mode > 2 || (c.context_mode && c.is_causal)) return MI_ERR_ARG;      // the causal stack is CausalConv2d whatever the field says (extractors.py:74-81)
    if (c.context_mode == 0) {
        RUN(mi_conv2d_first_gelu(feats, Gf(G_CONV1_W), Gf(G_CONV1_B), w.act1, c.B, c.T, c.F, c.C1, c.K, c.stride, pl, pl, D.T1, D.F1, st));
        RUN(mi_conv2d_cl_bf16(w.act1, Gw(G_CONV2_W), Gf(G_CONV2_B), w.act2, c.B, D.T1, D.F1, c.C1, c.C2, c.K, c.K, c.stride, pl, pl,
                              D.T2, D.F2, 1, st));
    } else if (c.context_mode == 1) {
        // GatedConv2d (extractors.py:23-32): GELU(conv * sigmoid(gate)), both layers.  Layer 1: two filter banks in one VALU kernel; layer 2: conv and gate rows in ONE
        // implicit GEMM with the product in its epilogue.  Shapes the fused kernels do not take run raw convs + mi_gated_act_bf16.
        if (c.gate_blk <= 0 || (c.C2 % c.gate_blk) != 0) return MI_ERR_ARG;
        int rc = mi_conv2d_first_gated_gelu(feats, Gf(G_CONV1_W), Gf(G_CONV1_B), Gf(G_GATE1_W), Gf(G_GATE1_B), w.act1, c.B, c.T, c.F, c.C1, c.K, c.stride, pl, pl, D.T1, D.F1, st);
        if (rc == MI_ERR_UNSUPPORTED) {
            RUN(mi_conv2d_first_geo(feats, Gf(G_CONV1_W), Gf(G_CONV1_B), w.z1, c.B, c.T, c.F, c.C1, c.K, c.K, c.stride, c.stride, pl, pl, D.T1, D.F1, 0, st));
            RUN(mi_conv2d_first_geo(feats, Gf(G_GATE1_W), Gf(G_GATE1_B), w.g1, c.B, c.T, c.F, c.C1, c.K, c.K, c.stride, c.stride, pl, pl, D.T1, D.F1, 0, st));
            RUN(mi_gated_act_bf16(w.z1, c.C1, w.g1, c.C1, w.act1, c.C1, c.B, D.T1, D.F1, c.C1, 1, 0, st));
        } else if (rc != MI_OK) return rc;
        rc = c.gate_blk == 32 ? mi_conv2d_cl_geo_bf16(w.act1, Gw(G_CONV2_W), Gf(G_CONV2_B), w.act2, c.B, D.T1, D.F1, c.C1, c.C2, c.K, c.K, c.stride, c.stride, pl, pl, D.T2, D.F2, 1, 1, st)
                              : MI_ERR_UNSUPPORTED;
        if (rc == MI_ERR_UNSUPPORTED) {
            RUN(mi_conv2d_cl_geo_bf16(w.act1, Gw(G_CONV2_W), Gf(G_CONV2_B), w.z2, c.B, D.T1, D.F1, c.C1, 2 * c.C2, c.K, c.K, c.stride, c.stride, pl, pl, D.T2, D.F2, 0, 0, st));
            RUN(mi_gated_act_bf16(w.z2, 2 * c.C2, w.z2, 2 * c.C2, w.act2, c.C2, c.B, D.T2, D.F2, c.C2, 1, c.gate_blk, st));
        } else if (rc != MI_OK) return rc;
    } else {
        // GatedConv2dShared (extractors.py:35-54): the gate conv is (4K, K) / stride (4s, s) / padding (4p, p); conv_out.view(B, C, -1, 4, F) * gate.unsqueeze(3) needs the
        // conv's time axis divisible by 4 and a quarter of it to be the gate's — the reference raises otherwise, so does this
        const int KH = 4 * c.K, sg = 4 * c.stride, pg = 4 * c.pad;
        const int Tg1 = conv_out(c.T, KH, sg, 2 * pg), Tg2 = conv_out(D.T1, KH, sg, 2 * pg);
        if ((D.T1 % 4) != 0 || Tg1 != D.T1 / 4 || (D.T2 % 4) != 0 || Tg2 != D.T2 / 4) return MI_ERR_ARG;
        RUN(mi_conv2d_first_geo(feats, Gf(G_CONV1_W), Gf(G_CONV1_B), w.z1, c.B, c.T, c.F, c.C1, c.K, c.K, c.stride, c.stride, pl, pl, D.T1, D.F1, 0, st));
        RUN(mi_conv2d_first_geo(feats, Gf(G_GATE1_W), Gf(G_GATE1_B), w.g1, c.B, c.T, c.F, c.C1, KH, c.K, sg, c.stride, pg, pl, Tg1, D.F1, 0, st));
        RUN(mi_gated_act_bf16(w.z1, c.C1, w.g1, c.C1, w.act1, c.C1, c.B, D.T1, D.F1, c.C1, 4, 0, st));
        RUN(mi_conv2d_cl_bf16(w.act1, Gw(G_CONV2_W), Gf(G_CONV2_B), w.z2, c.B, D.T1, D.F1, c.C1, c.C2, c.K, c.K, c.stride, pl, pl, D.T2, D.F2, 0, st));
        RUN(mi_conv2d_cl_geo_bf16(w.act1, Gw(G_GATE2_W), Gf(G_GATE2_B), w.g2, c.B, D.T1, D.F1, c.C1, c.C2, KH, c.K, sg, c.stride, pg, pl, Tg2, D.F2, 0, 0, st));
        RUN(mi_gated_act_bf16(w.z2, c.C2, w.g2, c.C2, w.act2, c.C2, c.B, D.T2, D.F2, c.C2, 4, 0, st));
    }
    // (B,C,T',F') -> transpose -> flatten -> Linear: act2 is already (B*T', F'*C) with the weight columns permuted to match
    RUN(mi_gemm_bf16(w.act2, (long)D.F2 * c.C2, Gw(G_FEOUT_W), (long)D.F2 * c.C2, Gf(G_FEOUT_B), 1, w.feo, d, 1, nullptr, 0, 1.f, 0,
                     M, d, D.F2 * c.C2, 0, 0, st));
    // --- feature projection: LN -> Linear (extractors.py:130-131; tf:328-333)
    RUN(mi_layernorm_chain(w.feo, d, nullptr, T2, nullptr, nullptr, 0.f, nullptr, 0, Gf(G_FP_LN_G), Gf(G_FP_LN_B), c.ln_eps,
                           w.a0, d, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, st));
    RUN(mi_gemm_bf16(w.a0, d, Gw(G_FP_W), d, Gf(G_FP_B), 1, w.x, d, 1, nullptr, 0, 1.f, 0, M, d, d, 0, 0, st));
    // --- zero padded frames once (tf:662-665) + first LayerNorm(s) of layer 0
    auto enter_layer = [&](const float* src, int l) -> int {      // x = src with padded frames zeroed; first LayerNorm(s) of layer l
        if (c.use_macaron)
            return mi_layernorm_chain(src, d, mask_len, T2, nullptr, nullptr, 0.f, w.x, d, Lf(l, FF1_LN_G), Lf(l, FF1_LN_B), leps,
                                      w.a0, d, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, st);
        return mi_layernorm_chain(src, d, mask_len, T2, nullptr, nullptr, 0.f, w.x, d, Lf(l, ATT_LN_G), Lf(l, ATT_LN_B), leps,
                                  w.a1, d, nullptr, 0, Lf(l, MLP_LN_G), Lf(l, MLP_LN_B), w.a2, d, M, d, st);
    };
    if (!c.ln_fold) RUN(enter_layer(w.x, 0));
    if (c.layer_mixing) RUN(mi_softmax_vec_f32(Gf(G_MIX_W), c.L + 1, w.sw, st));
    // --- relative positions: p_l = linear_pos_l(table) for every layer (batch independent; tf:531-536)
    const int P = 2 * T2 - 1;
    if (c.pos_type == 1 && compute_posp)
        for (int l = 0; l < Lt; ++l)
            RUN(mi_gemm_bf16(pos_table, d, Lw(l, ATT_WPOS), d, nullptr, 0, (bf16_t*)posp + (size_t)l * P * d, d, 0, nullptr, 0, 1.f, 0,
                             P, d, d, 0, 0, st));
    const float* rot_cos = (const float*)pos_table;
    const float* rot_sin = rot_cos ? rot_cos + (size_t)T2 * D.hd : nullptr;
    const float scale = 1.0f / sqrtf((float)D.hd);
    const int kc = c.csgu_kernel, km = c.merge_kernel;

    if (c.ln_fold) {
        // ---- LayerNorm-folded layers (header: mi_ebf_config.ln_fold).  a0 holds bf16(x) of the CURRENT residual stream, lnst its per-row partial statistics (np pairs):
        // written by the kernel that produced x (mi_layernorm_fold at layer entry and after final_layer_norm; the FFN-out / merge GEMMs' epilogues in between).
        if (c.pos_type == 2 || !c.use_macaron || c.extra_layers || c.layer_mixing || c.csgu_linear || (D.hd != 64 && D.hd != 128)) return MI_ERR_ARG;
        const int gv = c.wide_tiles ? 40 : 0;                     // kernel of the N = d GEMMs: 0 the product's 128 x 128, 40 the 256 x 256 tile (throughput mode)
        const int npg = c.wide_tiles ? d / 64 : d / 32;           // pairs a producer GEMM writes (one per 32 / 64 columns)
        // entry: zero padded frames once (tf:662-665); x, bf16(x), statistics
        RUN(mi_layernorm_fold(w.x, d, mask_len, T2, nullptr, nullptr, 0.f, w.x, d, w.a0, d, w.lnst, M, d, st));
        int np = 1;
        for (int l = 0; l < c.L; ++l) {
            if (hidden_states && hipMemcpyAsync(hidden_states + (size_t)l * M * d, w.x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return MI_ERR_LAUNCH;
            // x += 0.5 * FFN(LN(x))   e_branchformer.py:271-273
            RUN(mi_gemm_lnfold_bf16(w.a0, d, Lw(l, FF1_WF), d, Lf(l, FF1_SF), Lf(l, FF1_CF), w.lnst, np, leps, w.h, I, 1, M, I, d, st));
            RUN(mi_gemm_resid_stats_f32_v(w.h, I, Lw(l, FF1_W2), I, Lf(l, FF1_B2), w.x, d, w.x, d, 0.5f, w.a0, d, w.lnst, M, d, I, gv, st));
            np = npg;
            // global branch: self_attn_layer_norm folded into [Q|K|V]   (e_branchformer.py:281-288)
            RUN(mi_gemm_lnfold_bf16(w.a0, d, Lw(l, QKV_WF), d, Lf(l, QKV_SF), Lf(l, QKV_CF), w.lnst, np, leps, w.qk, 3 * d, 0, M, 3 * d, d, st));
            RUN(mi_attention_qkv_bf16(w.qk, 3 * d, w.qk + d, 3 * d, w.qk + 2 * d, 3 * d,
                                      c.pos_type == 1 ? (const bf16_t*)posp + (size_t)l * P * d : nullptr, d,
                                      c.pos_type == 1 ? Lf(l, ATT_U) : nullptr, c.pos_type == 1 ? Lf(l, ATT_V) : nullptr,
                                      mask_len, w.ctx, d, c.B, T2, 0, 0, c.H, D.hd, scale, c.is_causal, st));
            RUN(mi_gemm_bf16_v(w.ctx, d, Lw(l, ATT_WO), d, Lf(l, ATT_BO), 1, w.cat, 2 * d, 0, nullptr, 0, 1.f, 0, M, d, d, 0, 0, gv, st));
            // local branch: cgMLP_layer_norm folded into channel_proj1   (e_branchformer.py:291-292, 184-222)
            RUN(mi_gemm_lnfold_bf16(w.a0, d, Lw(l, MLP_WF), d, Lf(l, MLP_SF), Lf(l, MLP_CF), w.lnst, np, leps, w.h, I, 1, M, I, d, st));
            RUN(mi_row_stats_bf16(w.h + I / 2, I, I / 2, leps, w.stats, M, st));
            const int dil = c.is_causal ? (kc - 1) / 2 : 1;
            const int cpad = c.is_causal ? (kc - 1) * dil : (kc - 1) / 2;
            RUN(mi_csgu_bf16(w.h, I, w.stats, Lf(l, CSGU_LN_G), Lf(l, CSGU_LN_B), Lf(l, CSGU_W), Lf(l, CSGU_B), w.s, I / 2, c.B, T2, I / 2, kc, cpad, dil, c.csgu_act, st));
            RUN(mi_gemm_bf16_v(w.s, I / 2, Lw(l, MLP_W2), I / 2, Lf(l, MLP_B2), 1, w.cat + d, 2 * d, 0, nullptr, 0, 1.f, 0, M, d, I / 2, 0, 0, gv, st));
            // merge (e_branchformer.py:296-304): x += merge_proj(m + dwconv(m)); the epilogue leaves bf16(x) and its statistics for ff2's folded LayerNorm
            RUN(mi_dwconv_residual_bf16(w.cat, 2 * d, Lf(l, MRG_DW_W), Lf(l, MRG_DW_B), w.m2, 2 * d, c.B, T2, 2 * d, km, (km - 1) / 2, st));
            RUN(mi_gemm_resid_stats_f32_v(w.m2, 2 * d, Lw(l, MRG_W), 2 * d, Lf(l, MRG_B), w.x, d, w.x, d, 1.0f, w.a0, d, w.lnst, M, d, 2 * d, gv, st));
            // x += 0.5 * FFN(LN(x))   e_branchformer.py:307-309
            RUN(mi_gemm_lnfold_bf16(w.a0, d, Lw(l, FF2_WF), d, Lf(l, FF2_SF), Lf(l, FF2_CF), w.lnst, np, leps, w.h, I, 1, M, I, d, st));
            RUN(mi_gemm_bf16_v(w.h, I, Lw(l, FF2_W2), I, Lf(l, FF2_B2), 1, w.x, d, 1, w.x, d, 0.5f, 0, M, d, I, 0, 0, gv, st));
            // final_layer_norm (:312): the new residual stream, its bf16 copy and statistics for the next layer — or, after the last layer, chained with encoder.layer_norm
            if (l + 1 == c.L) {
                RUN(mi_layernorm_chain(w.x, d, nullptr, T2, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps, nullptr, 0,
                                       Gf(G_ENC_LN_G), Gf(G_ENC_LN_B), c.ln_eps, w.hid, d, last_hidden, d, nullptr, nullptr, nullptr, 0, M, d, st));
            } else {
                RUN(mi_layernorm_fold(w.x, d, nullptr, T2, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps, w.x, d, w.a0, d, w.lnst, M, d, st));
                np = 1;
            }
        }
    } else
    for (int l = 0; l < Lt; ++l) {
        if (hidden_states && l < c.L && hipMemcpyAsync(hidden_states + (size_t)l * M * d, w.x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return MI_ERR_LAUNCH;
        // layer mixing (bestrq.py:239-245): hidden_states[l] is this layer's input; the weight is read on the device
        if (c.layer_mixing && l < c.L) RUN(mi_axpy_dev_f32(w.mixed, w.x, (long)M * d, w.sw + l, l == 0, st));
        if (c.use_macaron) {   // x += 0.5 * FFN(LN(x))   e_branchformer.py:271-273
            RUN(mi_gemm_bf16(w.a0, d, Lw(l, FF1_W1), d, Lf(l, FF1_B1), 1, w.h, I, 0, nullptr, 0, 1.f, 1, M, I, d, 0, 0, st));
            RUN(mi_gemm_bf16(w.h, I, Lw(l, FF1_W2), I, Lf(l, FF1_B2), 1, w.x, d, 1, w.x, d, 0.5f, 0, M, d, I, 0, 0, st));
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, nullptr, nullptr, 0.f, nullptr, 0, Lf(l, ATT_LN_G), Lf(l, ATT_LN_B), leps,
                                   w.a1, d, nullptr, 0, Lf(l, MLP_LN_G), Lf(l, MLP_LN_B), w.a2, d, M, d, st));
        }
        // the two branches read a1 / a2 and write disjoint buffers (qk, ctx, cat[:, :d] | h, stats, s, cat[:, d:]): with branch_overlap the
        // local one is enqueued on the side stream between a fork and a join event
        const bool ovl = c.branch_overlap && l < MAX_OVL_LAYERS && side_ready();
        hipStream_t sl = ovl ? g_side : st;
        if (ovl) { (void)hipEventRecord(g_fork[l], st); (void)hipStreamWaitEvent(g_side, g_fork[l], 0); }
        // global branch (e_branchformer.py:281-288)
        const bf16_t* qk_in = w.a1;
        if (c.pos_type == 2) {
            RUN(mi_rotary_bf16(w.a1, d, w.a1r, d, rot_cos, rot_sin, M, T2, c.H, D.hd, st));
            qk_in = w.a1r;
        }
        const bool lds_attn = (D.hd == 64 || D.hd == 128);
        if (lds_attn) {
            // fused [Q|K|V] projection (weights are packed [Wq;Wk;Wv]); rotary feeds Q,K from the rotated input only
            if (c.pos_type == 2) {
                RUN(mi_gemm_bf16(qk_in, d, Lw(l, ATT_WQK), d, Lf(l, ATT_BQK), 1, w.qk, 3 * d, 0, nullptr, 0, 1.f, 0, M, 2 * d, d, 0, 0, st));
                RUN(mi_gemm_bf16(w.a1, d, Lw(l, ATT_WV), d, Lf(l, ATT_BV), 1, w.qk + 2 * d, 3 * d, 0, nullptr, 0, 1.f, 0, M, d, d, 0, 0, st));
            } else {
                RUN(mi_gemm_bf16(w.a1, d, Lw(l, ATT_WQK), d, Lf(l, ATT_BQK), 1, w.qk, 3 * d, 0, nullptr, 0, 1.f, 0, M, 3 * d, d, 0, 0, st));
            }
            RUN(mi_attention_qkv_bf16(w.qk, 3 * d, w.qk + d, 3 * d, w.qk + 2 * d, 3 * d,
                                      c.pos_type == 1 ? (const bf16_t*)posp + (size_t)l * P * d : nullptr, d,
                                      c.pos_type == 1 ? Lf(l, ATT_U) : nullptr, c.pos_type == 1 ? Lf(l, ATT_V) : nullptr,
                                      mask_len, w.ctx, d, c.B, T2, 0, 0, c.H, D.hd, scale, c.is_causal, st));
        } else {
            RUN(mi_gemm_bf16(qk_in, d, Lw(l, ATT_WQK), d, Lf(l, ATT_BQK), 1, w.qk, 3 * d, 0, nullptr, 0, 1.f, 0, M, 2 * d, d, 0, 0, st));
            // V^T = Wv · a1^T (+ bv per row), columns remapped to the time-padded (b*Tp + t) layout
            RUN(mi_gemm_bf16(Lw(l, ATT_WV), d, w.a1, d, Lf(l, ATT_BV), 2, w.vt, (long)c.B * D.Tp, 0, nullptr, 0, 1.f, 0, d, M, d,
                             T2, D.Tp, st));
            RUN(mi_attention_bf16(w.qk, 3 * d, w.qk + d, 3 * d, w.vt, (long)c.B * D.Tp, D.Tp,
                                  c.pos_type == 1 ? (const bf16_t*)posp + (size_t)l * P * d : nullptr, d,
                                  c.pos_type == 1 ? Lf(l, ATT_U) : nullptr, c.pos_type == 1 ? Lf(l, ATT_V) : nullptr,
                                  mask_len, w.ctx, d, c.B, T2, c.H, D.hd, scale, c.is_causal, st));
        }
        RUN(mi_gemm_bf16(w.ctx, d, Lw(l, ATT_WO), d, Lf(l, ATT_BO), 1, w.cat, 2 * d, 0, nullptr, 0, 1.f, 0, M, d, d, 0, 0, st));
        // local branch: cgMLP (e_branchformer.py:291-292, 184-222)
        RUN(mi_gemm_bf16(w.a2, d, Lw(l, MLP_W1), d, Lf(l, MLP_B1), 1, w.h, I, 0, nullptr, 0, 1.f, 1, M, I, d, 0, 0, sl));
        RUN(mi_row_stats_bf16(w.h + I / 2, I, I / 2, leps, w.stats, M, sl));
        // quirk: the causal CSGU conv is dilated by (K-1)/2 (e_branchformer.py:153-160 passes it in the dilation slot)
        const int dil = c.is_causal ? (kc - 1) / 2 : 1;
        const int cpad = c.is_causal ? (kc - 1) * dil : (kc - 1) / 2;
        if (c.csgu_linear) {   // conv -> Linear -> act -> gate (e_branchformer.py:196-201)
            RUN(mi_csgu_conv_bf16(w.h, I, w.stats, Lf(l, CSGU_LN_G), Lf(l, CSGU_LN_B), Lf(l, CSGU_W), Lf(l, CSGU_B), w.cv, I / 2, c.B, T2, I / 2, kc, cpad, dil, sl));
            RUN(mi_gemm_bf16(w.cv, I / 2, Lw(l, CSGU_LIN_W), I / 2, Lf(l, CSGU_LIN_B), 1, w.lin, I / 2, 0, nullptr, 0, 1.f, 0, M, I / 2, I / 2, 0, 0, sl));
            RUN(mi_gate_act_mul_bf16(w.h, I, w.lin, I / 2, w.s, I / 2, M, I / 2, c.csgu_act, sl));
        } else
        RUN(mi_csgu_bf16(w.h, I, w.stats, Lf(l, CSGU_LN_G), Lf(l, CSGU_LN_B), Lf(l, CSGU_W), Lf(l, CSGU_B), w.s, I / 2,
                         c.B, T2, I / 2, kc, cpad, dil, c.csgu_act, sl));
        RUN(mi_gemm_bf16(w.s, I / 2, Lw(l, MLP_W2), I / 2, Lf(l, MLP_B2), 1, w.cat + d, 2 * d, 0, nullptr, 0, 1.f, 0, M, d, I / 2, 0, 0, sl));
        if (ovl) { (void)hipEventRecord(g_join[l], g_side); (void)hipStreamWaitEvent(st, g_join[l], 0); }
        // merge (e_branchformer.py:296-304)
        RUN(mi_dwconv_residual_bf16(w.cat, 2 * d, Lf(l, MRG_DW_W), Lf(l, MRG_DW_B), w.m2, 2 * d, c.B, T2, 2 * d, km, (km - 1) / 2, st));
        RUN(mi_gemm_bf16(w.m2, 2 * d, Lw(l, MRG_W), 2 * d, Lf(l, MRG_B), 1, w.x, d, 1, w.x, d, 1.0f, 0, M, d, 2 * d, 0, 0, st));
        if (c.use_macaron) {   // e_branchformer.py:307-309
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, nullptr, nullptr, 0.f, nullptr, 0, Lf(l, FF2_LN_G), Lf(l, FF2_LN_B), leps,
                                   w.a0, d, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, st));
            RUN(mi_gemm_bf16(w.a0, d, Lw(l, FF2_W1), d, Lf(l, FF2_B1), 1, w.h, I, 0, nullptr, 0, 1.f, 1, M, I, d, 0, 0, st));
            RUN(mi_gemm_bf16(w.h, I, Lw(l, FF2_W2), I, Lf(l, FF2_B2), 1, w.x, d, 1, w.x, d, 0.5f, 0, M, d, I, 0, 0, st));
        }
        // final_layer_norm (:312) chained with the next consumer's LayerNorm(s)
        if (l + 1 == c.L) {
            float* lh = (last_hidden || !(c.layer_mixing || c.extra_layers)) ? last_hidden : w.lh;     // the head below needs the fp32 rows even when the caller does not
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps, nullptr, 0,
                                   Gf(G_ENC_LN_G), Gf(G_ENC_LN_B), c.ln_eps, w.hid, d, lh, d, nullptr, nullptr, nullptr, 0, M, d, st));
            if (c.layer_mixing || c.extra_layers) {       // the CTC fine-tuning head of a BEST-RQ encoder (bestrq.py:229-279)
                const float* top = lh;
                if (c.layer_mixing) {
                    RUN(mi_axpy_dev_f32(w.mixed, lh, (long)M * d, w.sw + c.L, 0, st));
                    top = w.mixed;
                }
                if (c.extra_layers) RUN(enter_layer(top, c.L));
                else RUN(mi_add2_cast_bf16(top, d, nullptr, 0, w.hid, d, M, d, 1.f, st));
            }
        } else if (l + 1 == Lt) {                         // end of the additional layer: its final_layer_norm feeds the head directly
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, nullptr, nullptr, 0.f, nullptr, 0, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps,
                                   w.hid, d, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, st));
        } else if (c.use_macaron)
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps, w.x, d,
                                   Lf(l + 1, FF1_LN_G), Lf(l + 1, FF1_LN_B), leps, w.a0, d, nullptr, 0, nullptr, nullptr, nullptr, 0, M, d, st));
        else
            RUN(mi_layernorm_chain(w.x, d, nullptr, T2, Lf(l, FIN_LN_G), Lf(l, FIN_LN_B), leps, w.x, d,
                                   Lf(l + 1, ATT_LN_G), Lf(l + 1, ATT_LN_B), leps, w.a1, d, nullptr, 0,
                                   Lf(l + 1, MLP_LN_G), Lf(l + 1, MLP_LN_B), w.a2, d, M, d, st));
    }
    // CTC head: lm_head ⊕ blank_projection, blank LAST (e_branchformer.py:456-457)
    if (logits) {
        const long ldl = c.logits_ld > 0 ? c.logits_ld : c.V + 1;
        int rc_l = MI_ERR_UNSUPPORTED;
        if (lse) {
            rc_l = mi_gemm_lse_f32(w.hid, d, Gw(G_HEAD_W), d, Gf(G_HEAD_B), (float*)logits, ldl, lse, lse_workspace, M, c.V + 1, d, st);
            if (rc_l != MI_OK && rc_l != MI_ERR_UNSUPPORTED) return rc_l;
        }
        if (rc_l == MI_ERR_UNSUPPORTED) {
            RUN(mi_gemm_bf16(w.hid, d, Gw(G_HEAD_W), d, Gf(G_HEAD_B), 1, logits, ldl, c.logits_f32, nullptr, 0, 1.f, 0, M, c.V + 1, d, 0, 0, st));
            if (lse) RUN(mi_row_lse(logits, ldl, 0, c.V + 1, lse, M, st));
        }
    }
    if (hidden_states && hipMemcpyAsync(hidden_states + (size_t)c.L * M * d, last_hidden, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return MI_ERR_LAUNCH;
    MI_CHECK_LAUNCH();
    return MI_OK;
}

extern "C" int mi_ebf_forward_hs(const mi_ebf_config* cfg, const void* const* weights, const float* feats,
                                 const int* feat_lengths, const void* pos_table, void* posp, int compute_posp,
                                 void* workspace, size_t workspace_bytes, float* last_hidden, void* logits,
                                 int* inner_len, int* outer_len, float* hidden_states, hipStream_t st) {
    return mi_ebf_forward_lse(cfg, weights, feats, feat_lengths, pos_table, posp, compute_posp, workspace, workspace_bytes, last_hidden, logits, inner_len, outer_len,
                              hidden_states, nullptr, nullptr, st);
}

extern "C" int mi_ebf_forward(const mi_ebf_config* cfg, const void* const* weights, const float* feats,
                              const int* feat_lengths, const void* pos_table, void* posp, int compute_posp,
                              void* workspace, size_t workspace_bytes, float* last_hidden, void* logits,
                              int* inner_len, int* outer_len, hipStream_t st) {
    return mi_ebf_forward_hs(cfg, weights, feats, feat_lengths, pos_table, posp, compute_posp, workspace, workspace_bytes, last_hidden, logits, inner_len, outer_len,
                             nullptr, st);
}
