// Cost of an in-kernel grid barrier on gfx950, for pricing a persistent decoder token step.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/bin/barrier_bench tools/barrier_bench.hip  (tools/bin/ is git-ignored; the binary travels with gpurun)
//
//   all: G blocks anywhere on the chip, agent-scope release / acquire around an atomic counter
//   xcd: only the blocks that landed on XCC 0 take part (8 G launched)
// Each barrier also hands one float per block to the next phase (written before, read after), like a real phase boundary.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, float* buf, int nbar, int members, int only_xcc0, unsigned* err, int* seen) {
    if (threadIdx.x == 0) atomicAdd(&seen[xcc_id()], 1);
    if (only_xcc0 && xcc_id() != 0) return;
    __shared__ unsigned slot;
    if (threadIdx.x == 0) slot = atomicAdd(&ctr[1], 1u);          // rank among the members
    __syncthreads();
    const unsigned me = slot;
    if (me >= (unsigned)members) return;
    float acc = 0.f;
    for (int it = 0; it < nbar; ++it) {
        if (threadIdx.x == 0) __hip_atomic_store(&buf[(it & 1) * 1024 + me], acc + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * members;
            long spins = 0;
            while (__hip_atomic_load(&ctr[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > 20000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        acc = __hip_atomic_load(&buf[(it & 1) * 1024 + (me + 1) % members], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) buf[2048 + me] = acc;
}

// flag form: every block publishes its epoch in its own word, wave 0 polls all the words with one load per round (no read-modify-write contention)
__global__ __launch_bounds__(256) void flag_kernel(unsigned* flags, float* buf, int nbar, unsigned* err) {
    const unsigned me = blockIdx.x, members = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < nbar; ++it) {
        if (threadIdx.x == 0) buf[(it & 1) * 1024 + me] = acc + 1.f;
        __syncthreads();
        if (threadIdx.x < 64) {
            if (threadIdx.x == 0) __hip_atomic_store(&flags[me], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            for (;;) {
                const unsigned v = threadIdx.x < members ? __hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                if (__all(v >= (unsigned)(it + 1))) break;
                if (++spins > 20000000) { *err = 1; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        acc = buf[(it & 1) * 1024 + (me + 1) % members];
    }
    if (threadIdx.x == 0) buf[2048 + me] = acc;
}

int main() {
    unsigned *ctr, *err; float* buf; int* seen;
    hipMalloc(&ctr, 64); hipMalloc(&err, 4); hipMalloc(&buf, 4096 * 4); hipMalloc(&seen, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nbar = 2000;
    for (int only = 0; only < 2; ++only)
        for (int members : {8, 16, 32, 64, 128, 256}) {
            if (only && members > 32) continue;
            float best = 1e9f; unsigned herr = 0; float last = 0; int hseen[16];
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(ctr, 0, 64); hipMemset(err, 0, 4); hipMemset(buf, 0, 4096 * 4); hipMemset(seen, 0, 64);
                hipEventRecord(e0);
                hipLaunchKernelGGL(bar_kernel, dim3(only ? 8 * members : members), dim3(256), 0, 0, ctr, buf, nbar, members, only, err, seen);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&last, buf + 2048, 4, hipMemcpyDeviceToHost); hipMemcpy(hseen, seen, 64, hipMemcpyDeviceToHost);
            }
            printf("%s members %3d: %.2f us per barrier (err %u, value %.0f, blocks on xcc0..7: %d %d %d %d %d %d %d %d)\n", only ? "xcc0" : "all ", members, best * 1e3 / nbar, herr, last,
                   hseen[0], hseen[1], hseen[2], hseen[3], hseen[4], hseen[5], hseen[6], hseen[7]);
        }
    for (int members : {8, 16, 32, 64}) {
        float best = 1e9f; unsigned herr = 0; float last = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(ctr, 0, 64); hipMemset(err, 0, 4); hipMemset(buf, 0, 4096 * 4);
            unsigned* flags; hipMalloc(&flags, 1024); hipMemset(flags, 0, 1024);
            hipEventRecord(e0);
            hipLaunchKernelGGL(flag_kernel, dim3(members), dim3(256), 0, 0, flags, buf, nbar, err);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&last, buf + 2048, 4, hipMemcpyDeviceToHost);
            hipFree(flags);
        }
        printf("flags members %3d: %.2f us per barrier (err %u, value %.0f)\n", members, best * 1e3 / nbar, herr, last);
    }
    return 0;
}
