// VALU issue rate of v_dot2_f32_bf16 against v_fma_f32 on gfx950 (one wave per SIMD and eight; independent chains): is a 2-MAC dot2 worth two FMAs?
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops (the library's flags: no v_pk_fma_f32) tools/experiments/dot2_rate.hip -o /tmp/dot2_rate && /tmp/dot2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, unsigned seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
    const bf16x2_t w = __builtin_bit_cast(bf16x2_t, 0x3F803F80u ^ seed), x = __builtin_bit_cast(bf16x2_t, 0x3F003F00u ^ seed);
    const float wf = 1.0001f + seed, xf = 0.5f + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(wf, xf, a[i]);
                else a[i] = __builtin_amdgcn_fdot2_f32_bf16(w, x, a[i], false);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10, 0u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 0u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    printf("%s, %d threads/block: %.3f ms -> %.2f ns per wave-instruction per SIMD-resident-wave-set (%.1f cycles at 2.4 GHz / waves per SIMD %d)\n", name, threads, ms,
           ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave * 2.4 / (threads / 256.0 > 1 ? threads / 256.0 : 1), threads / 256 > 0 ? threads / 256 : 1);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32      ", 256); run<1>("v_dot2_f32_bf16", 256);
    run<0>("v_fma_f32      ", 512); run<1>("v_dot2_f32_bf16", 512);
    return 0;
}
