// Small kernels of the GPT-2 cross-attention decoder path (reference src/models/decoders/multi_head_gpt2.py,
// src/models/embeddings.py; transformers GPT2Model embeddings): token+position embedding and the label-smoothed,
// shifted cross-entropy of GPT2LMMultiHeadModel.forward (:138-158).  Everything else of the decoder (LayerNorms,
// Conv1D projections, gelu_new MLP, causal self-attention, cross-attention over encoder frames, lm heads) runs on the
// shared kernels: mi_layernorm_chain, mi_gemm_bf16 (act = 2), mi_attention_qkv_bf16 (Tk != T).
#include "common.hpp"

namespace {

// out[m, :] = wte[ids[m], :] * scale + pos[pos_offset + m % U, :]          (fp32, row-major)
__global__ __launch_bounds__(256) void embed_kernel(const long* __restrict__ ids, const float* __restrict__ wte, float scale,
                                                     const float* __restrict__ pos, int pos_offset, int U, int d, int M, int V,
                                                     float* __restrict__ out) {
    const int d4 = d >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)M * d4; i += (long)gridDim.x * 256) {
        const int m = (int)(i / d4), c = (int)(i % d4);
        long id = ids[m];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        const f32x4 e = reinterpret_cast<const f32x4*>(wte + id * d)[c];
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos + (long)(pos_offset + m % U) * d)[c];
        reinterpret_cast<f32x4*>(out + (long)m * d)[c] = e * scale + pe;
    }
}

// one wave per (b, u) row, u < U - shift: loss_row = (1-eps) * (lse - z[target]) + eps * (lse - mean(z)), target = labels[b, u+shift]
// (torch CrossEntropyLoss(label_smoothing=eps), ignore_index = -100); accumulates sum and valid count.
__global__ __launch_bounds__(256) void ce_smooth_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                         int B, int U, int shift, int V, float eps, float* __restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows_per_b = U - shift;
    if (row >= B * rows_per_b) return;
    const int b = row / rows_per_b, u = row - b * rows_per_b;
    const long tgt = labels[(long)b * U + u + shift];
    if (tgt < 0) return;                                    // wave-uniform
    const float* z = logits + ((long)b * U + u) * ld;
    float mx = -INFINITY, sm = 0.f;
    for (int c = lane; c < V; c += 64) { const float v = z[c]; mx = fmaxf(mx, v); sm += v; }
    mx = wave_max(mx);
    sm = wave_sum(sm);
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += expf(z[c] - mx);
    se = wave_sum(se);
    if (lane == 0) {
        const float lse = mx + logf(se);
        const float loss = (1.f - eps) * (lse - z[tgt]) + eps * (lse - sm / V);
        atomicAdd(acc, loss);
        atomicAdd(acc + 1, 1.f);
    }
}

}  // namespace

extern "C" int mi_embed_tokens(const long* ids, const float* wte, float scale, const float* pos, int pos_offset, int U, int d,
                               int M, int V, float* out, hipStream_t stream) {
    MI_ENTER();
    if (M <= 0 || d <= 0 || (d % 4) || U <= 0 || V <= 0) return MI_ERR_ARG;
    const long total = (long)M * (d / 4);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, stream,
                       ids, wte, scale, pos, pos_offset, U, d, M, V, out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// acc[0] += sum of per-row losses, acc[1] += number of valid targets (caller zeroes acc and divides)
extern "C" int mi_ce_label_smoothing(const float* logits, long ld, const long* labels, int B, int U, int shift, int V, float eps,
                                     float* acc, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || U <= shift || V <= 0 || shift < 0) return MI_ERR_ARG;
    hipLaunchKernelGGL(ce_smooth_kernel, dim3(cdiv((long)B * (U - shift), 4)), dim3(256), 0, stream, logits, ld, labels, B, U, shift, V, eps, acc);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
