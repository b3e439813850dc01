// Lab: do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap?
// block = 512 threads (8 waves, 2 per SIMD: wave w and w+4 share a SIMD); role by wave group.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// mode bit0: group0 does MFMA; bit1: group1 does VALU; bit2: group1 does MFMA too; bit3: group0 does VALU too (same wave interleaved)
template <int mode>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    const int grp = threadIdx.x >> 8;
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(1.0f - j * 0.1f); }
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 0.5f, v5 = 0.25f, v6 = 4.f, v7 = 5.f;
    const bool do_mfma = (grp == 0 && (mode & 1)) || (grp == 1 && (mode & 4));
    const bool do_valu = (grp == 1 && (mode & 2)) || (grp == 0 && (mode & 8));
    for (int i = 0; i < iters; ++i) {
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // 64 VALU ops per iteration (8 MFMAs = 256 cycles of matrix pipe)
                if (mode & 16) {  // eight independent chains: throughput-bound VALU stream
                    v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 1.5f);
                    v4 = fmaf(v4, 1.0003f, 2.5f); v5 = fmaf(v5, 0.9997f, 3.5f); v6 = fmaf(v6, 1.0004f, 4.5f); v7 = fmaf(v7, 0.9996f, 5.5f);
                } else {
                    v0 = fmaf(v0, 1.0001f, v1); v1 = fmaf(v1, 0.9999f, v2); v2 = fmaf(v2, 1.0002f, v3); v3 = fmaf(v3, 0.9998f, v4);
                    v4 = fmaf(v4, 1.0003f, v5); v5 = fmaf(v5, 0.9997f, v6); v6 = fmaf(v6, 1.0004f, v7); v7 = fmaf(v7, 0.9996f, v0);
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 1.2345f) out[threadIdx.x] = s;
}
int main() {
    float* d;
    hipMalloc(&d, 1 << 20);
    const char* names[32] = {"", "MFMA(g0) alone", "VALU(g1) alone", "MFMA(g0) + VALU(g1) same SIMD", "", "MFMA both groups", "", "", "", "MFMA+VALU same wave (g0)"};
    names[18] = "VALU(g1) alone, independent chains";
    names[19] = "MFMA(g0) + independent VALU(g1)";
    names[25] = "MFMA + independent VALU, same wave";
    const int modes[] = {1, 2, 3, 5, 9, 18, 19, 25};
    for (int m : modes)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            switch (m) {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(256), dim3(512), 0, 0, d, 4000); break;
                L(1) L(2) L(3) L(5) L(9) L(18) L(19) L(25)
#undef L
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %2d  %-34s %8.1f us\n", m, names[m], ms * 1e3);
        }
    return 0;
}
