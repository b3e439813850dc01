// Lab: workgroup turn-around cost as a function of the LDS / VGPR allocation.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS_BYTES, int BIGV>
__global__ __launch_bounds__(256) void empty_kernel(float* out, int n) {
    extern __shared__ float dyn[];
    if (BIGV) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    if (n == -1) {  // never true; keeps the allocation alive
        dyn[threadIdx.x] = 1.0f;
        __syncthreads();
        out[threadIdx.x] = dyn[255 - threadIdx.x];
    }
}
template <int BIGV>
static void run(int lds, int blocks, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)empty_kernel<0, BIGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((empty_kernel<0, BIGV>), dim3(blocks), dim3(256), lds, 0, d, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((empty_kernel<0, BIGV>), dim3(blocks), dim3(256), lds, 0, d, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  bigv=%d lds=%6d blocks=%5d : %7.1f us/launch  (%.1f ns/block)\n", BIGV, lds, blocks, ms * 1000.f / 20, ms * 1e6f / 20 / blocks);
}
int main() {
    float* d;
    hipMalloc(&d, 1 << 20);
    const int ldss[] = {0, 1024, 16384, 32768, 65536, 67584, 81920, 131072};
    for (int lds : ldss) run<0>(lds, 2048, d);
    for (int lds : ldss) run<1>(lds, 2048, d);
    const int bl[] = {256, 512, 1024, 4096, 8192};
    for (int b : bl) run<0>(67584, b, d);
    for (int b : bl) run<1>(0, b, d);
    return 0;
}
