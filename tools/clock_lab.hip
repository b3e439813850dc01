// Lab: sustained MFMA rate and shader clock under load (s_memtime = core clock, s_memrealtime = 100 MHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clk, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) {
        x[j] = (_Float16)(((threadIdx.x * 37 + j * 11 + seed) % 255) / 128.0f - 1.0f);
        y[j] = (_Float16)(((threadIdx.x * 53 + j * 29 + seed) % 255) / 128.0f - 1.0f);
    }
    if (seed == 0) {
        for (int j = 0; j < 8; ++j) x[j] = y[j] = 0;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 1.2345f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}
int main() {
    float* d;
    unsigned long long* c;
    hipMalloc(&d, 1 << 20);
    hipMalloc(&c, 64);
    for (int nacc = 1; nacc <= 4; nacc *= 2)
    for (int wavesPerSimd = 1; wavesPerSimd <= 2; ++wavesPerSimd)
        for (unsigned seed = 0; seed < 2; ++seed)
            for (int rep = 0; rep < 2; ++rep) {
                const int blocks = 256 * wavesPerSimd, iters = 4000;
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                if (nacc == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, d, c, iters * 4, seed);
                else if (nacc == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, d, c, iters * 2, seed);
                else hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, d, c, iters, seed);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2];
                hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
                const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
                printf("chains=%d waves/SIMD=%d data=%s  %.1f us  %.0f TF/s   core clk %.0f MHz (memtime %llu / realtime %llu)\n", nacc, wavesPerSimd,
                       seed ? "random" : "zero", ms * 1e3, flops / (ms * 1e-3) * 1e-12, (double)h[0] / h[1] * 100.0, h[0], h[1]);
            }
    return 0;
}
