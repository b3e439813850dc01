// Lab: cost of a software barrier across co-resident workgroups (the building block a register-resident Sinkhorn
// would need: 64 CUs hold one 2048 x 2048 score matrix, two barriers per round).  Groups of G blocks synchronise
// through one L2 counter each; every wait is bounded (a stuck barrier sets a flag and the kernel drains), so the
// program cannot hang the GPU.  512 threads per block, one block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define MAX_SPINS 400000

__global__ __launch_bounds__(512) void barrier_kernel(unsigned* cnt, unsigned* failed, int group, int rounds, float* sink) {
    const int g = blockIdx.x / group;
    unsigned* c = cnt + g * 64;  // one counter per group, 256 bytes apart
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        acc += __sinf(acc + threadIdx.x);  // a little work between barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * (unsigned)group;
            int spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > MAX_SPINS || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    unsigned *cnt, *failed;
    float* sink;
    hipMalloc(&cnt, 64 * 64 * sizeof(unsigned));
    hipMalloc(&failed, sizeof(unsigned));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int cfgs[][2] = {{8, 8}, {64, 16}, {64, 64}, {256, 64}, {256, 256}, {128, 32}};  // {blocks, group size}
    for (auto& cf : cfgs) {
        const int blocks = cf[0], group = cf[1];
        for (int rounds : {1, 101}) {
            hipMemset(cnt, 0, 64 * 64 * sizeof(unsigned));
            hipMemset(failed, 0, sizeof(unsigned));
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(512), 0, 0, cnt, failed, group, rounds, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned f;
            hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
            printf("blocks=%3d group=%3d rounds=%3d : %8.1f us total%s\n", blocks, group, rounds, ms * 1000.f, f ? "  (BARRIER TIMED OUT)" : "");
        }
    }
    printf("per-barrier cost = (t[101 rounds] - t[1 round]) / 100\n");
    return 0;
}
