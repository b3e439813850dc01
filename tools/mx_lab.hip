// Lab for VERDICT round 5, item 1 (step B): the block-scaled fp6 matrix instruction as the carrier of the CORRECTION products of attention's
// P.V (the one class of product the CPU probe accepts: profiles/r06_lab_mx_corrections.txt).  Three questions, answered on the device:
//   1. `v_cvt_scalef32_2xpk16_fp6_f32`: in which order do the 2 x 16 inputs land in the 32 packed fields, and what does the scale operand do?
//   2. `v_mfma_scale_f32_32x32x64_f8f6f4` with fp6 (e2m3) operands: which (row, k) does field j of lane l hold, which byte carries the e8m0 scale?
//      (checked against a host product under the hypothesis lane l = row l % 32, k = 32 (l / 32) + j; a probe prints the map if it fails)
//   3. issue rate of the mixes the attention kernel would run per 64-key tile: 48 f16 MFMAs (three products in both contractions) against
//      40 (two-product P.V) against 32 f16 + 4 fp6-scaled (f16 main term + two fp6 corrections of P.V).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mx_lab.hip -o tools/mx_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// ---- fp6 e2m3 on the host: 1 sign, 2 exponent (bias 1), 3 mantissa bits; largest 7.5, sub-normal step 0.125
static double fp6_decode(unsigned c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const double v = e == 0 ? m / 8.0 : ldexp(1.0 + m / 8.0, e - 1);
    return s ? -v : v;
}
static unsigned fp6_encode_rne(double x) {  // nearest representable (ties to even code), saturating
    unsigned best = 0;
    double bd = 1e300;
    for (unsigned c = 0; c < 64; ++c) {
        const double d = fabs(fp6_decode(c) - x);
        if (d < bd || (d == bd && (c & 1) == 0 && (best & 1) == 1)) {
            bd = d;
            best = c;
        }
    }
    return best;
}
static unsigned field(const unsigned* w, int j) {  // j-th 6-bit field of a little-endian packed run
    const int bit = 6 * j, wd = bit >> 5, sh = bit & 31;
    unsigned long long v = w[wd];
    if (sh > 26) v |= (unsigned long long)w[wd + 1] << 32;
    return (unsigned)((v >> sh) & 63);
}
static void set_field(unsigned* w, int j, unsigned c) {
    const int bit = 6 * j, wd = bit >> 5, sh = bit & 31;
    w[wd] |= c << sh;
    if (sh > 26) w[wd + 1] |= c >> (32 - sh);
}

// ---- 1. the conversion
__global__ void cvt_kernel(const float* in, unsigned* out, float scale) {
    f32x16 a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = in[threadIdx.x * 32 + i];
        b[i] = in[threadIdx.x * 32 + 16 + i];
    }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
#pragma unroll
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}

// ---- 2. the matrix instruction
__global__ void mfma_kernel(const unsigned* A, const unsigned* B, const int* sa, const int* sb, float* C) {
    i32x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = i < 6 ? (int)A[threadIdx.x * 6 + i] : 0;
        b[i] = i < 6 ? (int)B[threadIdx.x * 6 + i] : 0;
    }
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
#pragma unroll
    for (int i = 0; i < 16; ++i) C[threadIdx.x * 16 + i] = c[i];
}
// single-field probe: A has ONE non-zero field (lane la, field ja); B's only non-zero field walks over every (lane, field); out[lb * 32 + jb] =
// sum |C| of the wave (non-zero where the two fields meet in k), rowcol[...] = which accumulator (lane, register) lit up
__global__ void probe_kernel(int la, int ja, float* out, int* where) {
    const int lane = threadIdx.x;
    const int one = 0x08;  // e2m3 code of 1.0: e = 1, m = 0
    const int sc = 127;
    for (int lb = 0; lb < 64; ++lb)
        for (int jb = 0; jb < 32; ++jb) {
            unsigned wa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (lane == la) {
                const int bit = 6 * ja;
                wa[bit >> 5] |= (unsigned)one << (bit & 31);
                if ((bit & 31) > 26) wa[(bit >> 5) + 1] |= (unsigned)one >> (32 - (bit & 31));
            }
            if (lane == lb) {
                const int bit = 6 * jb;
                wb[bit >> 5] |= (unsigned)one << (bit & 31);
                if ((bit & 31) > 26) wb[(bit >> 5) + 1] |= (unsigned)one >> (32 - (bit & 31));
            }
            i32x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = (int)wa[i];
                b[i] = (int)wb[i];
            }
            f32x16 c;
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = 0.0f;
            int s = sc;
            asm volatile("" : "+v"(s));
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, s, 0, s);
            float t = 0.0f;
            int w = -1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                t += fabsf(c[i]);
                if (c[i] != 0.0f) w = lane * 16 + i;
            }
            for (int o = 32; o > 0; o >>= 1) {
                t += __shfl_xor(t, o, 64);
                w = max(w, __shfl_xor(w, o, 64));
            }
            if (lane == 0) {
                out[lb * 32 + jb] = t;
                where[lb * 32 + jb] = w;
            }
        }
}

// ---- 3. issue rates (register-resident operands, independent accumulators, no memory in the loop)
template <int NF16, int NFP6>
__global__ __launch_bounds__(256, 2) void rate_kernel(float* out, int iters, int seed) {
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * ((threadIdx.x * 7 + i + seed) % 13));
        b[i] = (_Float16)(0.002f * ((threadIdx.x * 5 + i + seed) % 11));
    }
    i32x8 a6, b6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a6[i] = i < 6 ? (int)(0x08208208u ^ (threadIdx.x * 2654435761u + i)) & 0x1f7df7df : 0;
        b6[i] = i < 6 ? (int)(0x08208208u ^ (threadIdx.x * 40503u + 3 * i)) & 0x1f7df7df : 0;
    }
    int sa = 120 + (threadIdx.x & 3), sb = 121;
    asm volatile("" : "+v"(sa), "+v"(sb));
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NF16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFP6; ++j) acc[j & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc[j & 3], 2, 2, 0, sa, 0, sb);
    }
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[k][i];
    if (t == 1234.5f) out[0] = t;
}
template <int NF16, int NFP6>
static void rate(const char* name, float* d) {
    const int blocks = 256 * 2 * 4, iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<NF16, NFP6>), dim3(blocks), dim3(256), 0, 0, d, iters, 1);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rate_kernel<NF16, NFP6>), dim3(blocks), dim3(256), 0, 0, d, iters, r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double tiles = (double)blocks * 4 * iters;  // one loop body per wave = one "key tile"
    printf("  %-58s %7.3f ms  -> %6.1f ns per tile body per wave-slot (%d f16 + %d fp6-scaled MFMAs)\n", name, ms, ms * 1e6 / (tiles / (256.0 * 4 * 2)), NF16, NFP6);
}

int main() {
    // ---------------------------------------------------------------- 1
    printf("## 1. v_cvt_scalef32_2xpk16_fp6_f32\n");
    {
        std::vector<float> in(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) in[l * 32 + i] = 0.0f;
        // lane 0: a[i] = 0.125 (i + 1) (sub-normal and first normal codes), b[i] = -(1 + 0.25 i)
        for (int i = 0; i < 16; ++i) {
            in[i] = 0.125f * (i + 1);
            in[16 + i] = -(1.0f + 0.25f * i);
        }
        // lane 1: rounding / saturation cases
        const float cases[16] = {0.0624f, 0.0626f, 0.1875f, 0.3125f, 1.0625f, 1.1875f, 3.75f, 4.25f, 7.25f, 7.75f, 100.0f, -100.0f, 2.125f, 2.375f, 6.5f, 5.75f};
        for (int i = 0; i < 16; ++i) in[32 + i] = in[32 + 16 + i] = cases[i];
        float *din;
        unsigned* dout;
        CK(hipMalloc(&din, in.size() * 4));
        CK(hipMalloc(&dout, 64 * 6 * 4));
        CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        const float scales[4] = {1.0f, 4.0f, 0.25f, 3.0f};
        for (float sc : scales) {
            hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dout, sc);
            std::vector<unsigned> out(64 * 6 + 2, 0);
            CK(hipMemcpy(out.data(), dout, 64 * 6 * 4, hipMemcpyDeviceToHost));
            printf("scale operand %.2f\n  lane 0 fields:", sc);
            for (int j = 0; j < 32; ++j) printf(" %g", fp6_decode(field(&out[0], j)));
            printf("\n  lane 1 in :");
            for (int i = 0; i < 16; ++i) printf(" %g", cases[i]);
            printf("\n  lane 1 out (fields 0..31):");
            for (int j = 0; j < 32; ++j) printf(" %g", fp6_decode(field(&out[6], j)));
            printf("\n");
        }
    }
    // ---------------------------------------------------------------- 2
    printf("\n## 2. v_mfma_scale_f32_32x32x64_f8f6f4, fp6 e2m3 x fp6 e2m3, hypothesis: lane l = row (col) l %% 32, k = 32 (l / 32) + field, scale = byte 0 of the lane's scale register (e8m0)\n");
    {
        std::vector<unsigned> A(64 * 6 + 2, 0), B(64 * 6 + 2, 0);
        std::vector<int> sa(64), sb(64);
        static double Am[32][64], Bm[64][32];
        unsigned x = 777u;
        for (int l = 0; l < 64; ++l) {
            const int row = l & 31, kb = l >> 5;
            sa[l] = 127 + (int)((x = x * 1664525u + 1013904223u) >> 29) - 4;  // 2^-4 .. 2^3
            sb[l] = 127 + (int)((x = x * 1664525u + 1013904223u) >> 29) - 3;
            sa[l] |= 0x55443300;  // garbage in the upper bytes: only byte 0 may matter
            sb[l] |= 0x11aa2200;
            for (int j = 0; j < 32; ++j) {
                const unsigned ca = ((x = x * 1664525u + 1013904223u) >> 20) & 63, cb = ((x = x * 1664525u + 1013904223u) >> 20) & 63;
                set_field(&A[l * 6], j, ca);
                set_field(&B[l * 6], j, cb);
                Am[row][32 * kb + j] = fp6_decode(ca) * ldexp(1.0, (sa[l] & 255) - 127);
                Bm[32 * kb + j][row] = fp6_decode(cb) * ldexp(1.0, (sb[l] & 255) - 127);
            }
        }
        unsigned *dA, *dB;
        int *dsa, *dsb;
        float* dC;
        CK(hipMalloc(&dA, 64 * 6 * 4));
        CK(hipMalloc(&dB, 64 * 6 * 4));
        CK(hipMalloc(&dsa, 256));
        CK(hipMalloc(&dsb, 256));
        CK(hipMalloc(&dC, 64 * 16 * 4));
        CK(hipMemcpy(dA, A.data(), 64 * 6 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 64 * 6 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
        std::vector<float> C(64 * 16);
        CK(hipMemcpy(C.data(), dC, 64 * 16 * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += Am[row][k] * Bm[k][col];
                worst = fmax(worst, fabs(ref - C[l * 16 + r]));
                big = fmax(big, fabs(ref));
            }
        printf("max |C - host| = %.3e (max |C| %.3e): %s\n", worst, big, worst <= 1e-6 * big ? "layout and scale hypothesis CONFIRMED (exact products, f32 accumulation)" : "MISMATCH -- see the probe below");
        float* dout;
        int* dwh;
        CK(hipMalloc(&dout, 2048 * 4));
        CK(hipMalloc(&dwh, 2048 * 4));
        const int probes[6][2] = {{0, 0}, {0, 1}, {0, 31}, {32, 0}, {5, 7}, {37, 30}};
        for (auto& pr : probes) {
            hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, pr[0], pr[1], dout, dwh);
            std::vector<float> o(2048);
            std::vector<int> w(2048);
            CK(hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(w.data(), dwh, 2048 * 4, hipMemcpyDeviceToHost));
            printf("A field (lane %2d, %2d) meets B fields:", pr[0], pr[1]);
            int n = 0;
            for (int i = 0; i < 2048; ++i)
                if (o[i] != 0.0f && n++ < 40) printf(" (l%d,f%d)->acc(l%d,r%d)", i >> 5, i & 31, w[i] >> 4, w[i] & 15);
            printf("  [%d hits]\n", n);
        }
    }
    // ---------------------------------------------------------------- 3
    printf("\n## 3. issue rate of the per-tile MFMA mixes (8192 waves of 4-wave workgroups, two waves per SIMD, registers only)\n");
    {
        float* d;
        CK(hipMalloc(&d, 64));
        rate<48, 0>("48 f16 (three products in K.Q^T and in P.V: variant 8)", d);
        rate<40, 0>("40 f16 (two-product P.V: variant 7)", d);
        rate<32, 4>("32 f16 + 4 fp6-scaled (P.V = f16 main + two fp6 corrections)", d);
        rate<32, 0>("32 f16 (the f16 part alone)", d);
        rate<0, 4>("4 fp6-scaled alone", d);
        rate<24, 0>("24 f16", d);
        rate<0, 16>("16 fp6-scaled", d);
    }
    return 0;
}
