#include "common.hpp"
__global__ void k(const float* x, float* o) {
    const float v = x[blockIdx.x * 64 + threadIdx.x];
    const float s = wave_sum(v), m = wave_max(v);
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = s; o[2 * blockIdx.x + 1] = m; }
    o[1024 + blockIdx.x * 64 + threadIdx.x] = s;       // every lane must see the total
}
extern "C" void launch(const float* x, float* o, int n, hipStream_t st) { hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, st, x, o); }
