// How fast can one CU pull L2-resident bytes, per load path?  512 blocks x 256 threads, 64 KiB LDS each (two per CU, like the GEMM).
// Every iteration a block moves PIECES KiB (4 x 1 KiB per wave = 16 KiB default) from a region that fits the L2s / MALL:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)          mode 1: global_load_dwordx4 to VGPRs (consumed by an add)
//   mode 2: global_load_dwordx4 + ds_write_b128        mode 3: half the pieces by LDS-DMA, half to VGPRs
// `depth` iterations stay in flight (counted vmcnt).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* src, long region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long stride = (long)gridDim.x * 16384;
    long off = (long)blockIdx.x * 16384;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH][4];
    for (int it = 0; it < iters + DEPTH; ++it) {
        const int slot = it % DEPTH;
        if (it >= DEPTH) {                    // consume the iteration issued DEPTH ago: everything but the younger DEPTH-1 iterations has landed
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (MODE == 1) { for (int q = 0; q < 4; ++q) acc += r[slot][q]; }
            if (MODE == 3) { for (int q = 2; q < 4; ++q) acc += r[slot][q]; }
            if (MODE == 2) { for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(smem + (slot & 3) * 16384 + wave * 4096 + q * 1024 + lane * 16) = r[slot][q]; }
        }
        if (it < iters) {
            const char* g = src + (off % region) + wave * 4096 + lane * 16;
            char* sbase = smem + (slot & 3) * 16384 + wave * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 0 || (MODE == 3 && q < 2)) __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 1024), (lptr_t)(sbase + q * 1024), 16, 0, 0);
                else r[slot][q] = *reinterpret_cast<const u32x4*>(g + q * 1024);
            }
            off += stride;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x += *reinterpret_cast<unsigned*>(smem + tid * 4);
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = acc.x;
}

template <int MODE>
static void launch(int depth, const char* src, long region, int blocks, int iters, unsigned* sink, hipStream_t st) {
    if (depth == 1) hipLaunchKernelGGL((stream_kernel<MODE, 1>), dim3(blocks), dim3(256), 65536, st, src, region, iters, sink);
    else if (depth == 2) hipLaunchKernelGGL((stream_kernel<MODE, 2>), dim3(blocks), dim3(256), 65536, st, src, region, iters, sink);
    else if (depth == 3) hipLaunchKernelGGL((stream_kernel<MODE, 3>), dim3(blocks), dim3(256), 65536, st, src, region, iters, sink);
    else hipLaunchKernelGGL((stream_kernel<MODE, 4>), dim3(blocks), dim3(256), 65536, st, src, region, iters, sink);
}
extern "C" int stream_launch(int mode, int depth, const void* src, long region, int blocks, int iters, void* sink, hipStream_t st) {
    if (mode == 0) launch<0>(depth, (const char*)src, region, blocks, iters, (unsigned*)sink, st);
    else if (mode == 1) launch<1>(depth, (const char*)src, region, blocks, iters, (unsigned*)sink, st);
    else if (mode == 2) launch<2>(depth, (const char*)src, region, blocks, iters, (unsigned*)sink, st);
    else launch<3>(depth, (const char*)src, region, blocks, iters, (unsigned*)sink, st);
    return (int)hipGetLastError();
}

// GEMM-shaped access: per K step a block pulls 256 rows x 128 B (rows `ld` bytes apart) by LDS-DMA, 8 rows per wave-instruction as the
// GEMM does (lane -> row lane>>3, 16-B chunk lane&7), two K steps in flight; after K/64 steps it moves to another 256-row block.
__global__ __launch_bounds__(256) void rows_kernel(const char* src, long ld, int nrowblocks, int ksteps, int tiles, int kb, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int step = 0;
    for (int t = 0; t < tiles; ++t) {
        const long rb = ((long)blockIdx.x + (long)t * gridDim.x) % nrowblocks;
        const char* base = src + rb * 256 * ld + (long)(wave * 64 + (lane >> 3)) * ld + (lane & 7) * 16;
        for (int k = 0; k < ksteps; ++k, ++step) {
            char* sbase = smem + (step & 1) * 32768 + wave * 8192;
            if (step >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(base + (long)q * 8 * ld + (long)k * kb), (lptr_t)(sbase + q * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*reinterpret_cast<unsigned*>(smem + tid * 4) == 0x12345678u) sink[0] = 1;
}
extern "C" int rows_launch(const void* src, long ld, int nrowblocks, int ksteps, int tiles, int kb, int blocks, void* sink, hipStream_t st) {
    hipLaunchKernelGGL(rows_kernel, dim3(blocks), dim3(256), 65536, st, (const char*)src, ld, nrowblocks, ksteps, tiles, kb, (unsigned*)sink);
    return (int)hipGetLastError();
}
