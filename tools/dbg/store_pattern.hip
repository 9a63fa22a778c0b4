// How much does the GEMM epilogue's store pattern cost?  Writes an (M, N) bf16 matrix tile by tile (128 x 128 tiles, 4 waves of 64 x 64) from registers:
//   mode 0: the current epilogue's pattern — a lane owns a ROW of a 32 x 32 sub-tile, each store instruction writes 16 B per lane at a row stride
//           (32 rows x 2 halves touched per instruction, 32 B per row);
//   mode 1: row-major — 8 consecutive lanes cover 128 contiguous bytes of one row (8 full rows of the wave's 64-column tile per instruction).
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(unsigned short* C, long ldc, int M, int N, int tiles_n, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const u32x4 v = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = (tile / tiles_n) * 128 + wm * 64, n0 = (tile % tiles_n) * 128 + wn * 64;
        if (MODE == 0) {
            const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int m = m0 + i * 32 + lr;
                        if (m < M) *reinterpret_cast<u32x4*>(C + (long)m * ldc + n0 + j * 32 + 16 * k + 8 * lh) = v;
                    }
        } else {
            const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = m0 + q * 8 + r8;
                if (m < M) *reinterpret_cast<u32x4*>(C + (long)m * ldc + n0 + c8 * 8) = v;
            }
        }
    }
}
extern "C" int store_launch(int mode, void* C, long ldc, int M, int N, hipStream_t st) {
    const int tn = N / 128, nt = ((M + 127) / 128) * tn;
    const int grid = nt < 512 ? nt : 512;
    if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, st, (unsigned short*)C, ldc, M, N, tn, nt);
    else hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, st, (unsigned short*)C, ldc, M, N, tn, nt);
    return (int)hipGetLastError();
}
