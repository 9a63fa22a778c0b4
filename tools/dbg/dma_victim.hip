// Micro-reproducer: does LDS-DMA traffic of one kernel disturb ordinary vector loads of a co-resident kernel (another stream)?
//   aggressor<mode>: 256 threads, 64 KiB LDS, streams a big buffer into LDS.  mode 0: global_load_lds_dwordx4, 1: raw_buffer_load_lds 16 B,
//                    2: plain global_load_dwordx4 + ds_write_b128
//   victim: 256 threads, 63 KiB LDS; every thread repeatedly loads 16 B of a small constant table (all 1.0f) next to a streaming 16-B load
//           and counts words that are not 1.0f.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE>
__global__ __launch_bounds__(256) void aggressor(const char* src, long bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long stride = (long)gridDim.x * 16384;
    long off = (long)blockIdx.x * 16384;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        char* sbase = smem + (it & 3) * 16384 + wave * 4096;
        const char* g = src + (off % (bytes - 16384)) + wave * 4096 + lane * 16;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 1024), (lptr_t)(sbase + q * 1024), 16, 0, 0);
        } else if (MODE == 1) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
            const int voff = (int)((off % (bytes - 16384)) + wave * 4096 + lane * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(sbase + q * 1024), 16, voff + q * 1024, 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(sbase + q * 1024 + lane * 16) = *reinterpret_cast<const u32x4*>(g + q * 1024);
        }
        if ((it & 1) == 1) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            acc += *reinterpret_cast<volatile float*>(smem + ((it - 1) & 3) * 16384 + tid * 4);
        }
        off += stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (acc == 12345.678f) sink[0] = acc;
}

struct VRec { int block, tid, iter, word; unsigned got; };
__global__ __launch_bounds__(256) void victim(const float* table, const char* stream_src, long bytes, int iters, int* counter, VRec* recs, int max_recs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    float* l = reinterpret_cast<float*>(smem);
    long off = ((long)blockIdx.x * 256 + tid) * 16;
    const long stride = (long)gridDim.x * 256 * 16;
    float sum = 0.f;
    for (int it = 0; it < iters; ++it) {
        const u32x4 s = *reinterpret_cast<const volatile u32x4*>(stream_src + (off % (bytes - 16)));
        const f32x4 a = *reinterpret_cast<const volatile f32x4*>(table + (tid & 7) * 8);
        const f32x4 b = *reinterpret_cast<const volatile f32x4*>(table + (tid & 7) * 8 + 4);
        sum += __uint_as_float(s.x & 0x3f800000u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (__float_as_uint(a[j]) != 0x3f800000u) { const int slot = atomicAdd(counter, 1); if (slot < max_recs) recs[slot] = VRec{(int)blockIdx.x, tid, it, j, __float_as_uint(a[j])}; }
            if (__float_as_uint(b[j]) != 0x3f800000u) { const int slot = atomicAdd(counter, 1); if (slot < max_recs) recs[slot] = VRec{(int)blockIdx.x, tid, it, 4 + j, __float_as_uint(b[j])}; }
        }
        l[(it * 256 + tid) & 8191] = a.x + b.z;
        off += stride;
    }
    if (sum == 12345.678f) counter[1] = 1;
}

extern "C" int aggressor_launch(int mode, const void* src, long bytes, int blocks, int iters, float* sink, hipStream_t st) {
    if (mode == 0) hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 65536, st, (const char*)src, bytes, iters, sink);
    else if (mode == 1) hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 65536, st, (const char*)src, bytes, iters, sink);
    else hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(256), 65536, st, (const char*)src, bytes, iters, sink);
    return (int)hipGetLastError();
}
extern "C" int victim_launch(const float* table, const void* src, long bytes, int blocks, int iters, int* counter, void* recs, int max_recs, hipStream_t st) {
    hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 64512, st, table, (const char*)src, bytes, iters, counter, (VRec*)recs, max_recs);
    return (int)hipGetLastError();
}
