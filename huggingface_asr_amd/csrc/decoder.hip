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
                                                         int B, int U, int shift, int V, float eps, float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows_per_b = U - shift;
    if (row >= B * rows_per_b) return;
    const int b = row / rows_per_b, u = row - b * rows_per_b;
    const long tgt = labels[(long)b * U + u + shift];
    if (tgt < 0) { if (lane == 0) row_loss[row] = __builtin_nanf(""); return; }       // wave-uniform; NaN marks an ignored row
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
        row_loss[row] = loss;
    }
}
// acc = [sum of the valid rows' losses, their count]: one block, rows strided over the threads, wave sums by DPP, the four waves in order (no float atomics: the loss and
// everything derived from it are bit-reproducible)
__global__ __launch_bounds__(256) void ce_sum_kernel(const float* __restrict__ row_loss, int n, float* __restrict__ acc) {
    __shared__ float red[8];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = row_loss[i]; if (v == v) { s += v; c += 1.f; } }
    s = wave_sum(s); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { acc[0] += (red[0] + red[1]) + (red[2] + red[3]); acc[1] += (red[4] + red[5]) + (red[6] + red[7]); }
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
                                     float* acc, float* row_loss, hipStream_t stream) {
    MI_ENTER();
    if (B <= 0 || U <= shift || V <= 0 || shift < 0 || !row_loss) return MI_ERR_ARG;          // row_loss: B * (U - shift) floats of workspace
    const int rows = B * (U - shift);
    hipLaunchKernelGGL(ce_smooth_kernel, dim3(cdiv((long)rows, 4)), dim3(256), 0, stream, logits, ld, labels, B, U, shift, V, eps, row_loss);
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(256), 0, stream, row_loss, rows, acc);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
