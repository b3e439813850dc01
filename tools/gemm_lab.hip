// Kernel lab (not part of the product): knock-out variants of the split GEMM main loop to
// locate its bottleneck.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../image-matching-webui_amd/csrc -I../include gemm_lab.hip -o gemm_lab
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm.hip"

void imcui_prof_begin(imcui_hip_s*, int, hipStream_t) {}
void imcui_prof_end(imcui_hip_s*, int, hipStream_t) {}
int imcui_set_err(imcui_hip_s*, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    return code;
}

// KO bits: 1 = no MFMA, 2 = no global loads inside the loop, 4 = no split / LDS stores inside the loop, 8 = no epilogue
template <int KO>
__global__ __launch_bounds__(256, 2) void lab_kernel(GemmP p) {
    __shared__ uint4 smem[STAGE_C_BYTES / 16];
    uint4* Ah = smem;
    uint4* Al = smem + (BK64 / 8) * LDS_ROWS;
    uint4* Bh = smem + 2 * (BK64 / 8) * LDS_ROWS;
    uint4* Bl = smem + 3 * (BK64 / 8) * LDS_ROWS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    TileCtx c;
    if (!gemm_tile_setup(p, c)) return;
    const float* A = p.A;
    const float* A2 = nullptr;
    const unsigned short* Wh = p.Wh;
    const unsigned short* Wl = p.Wl;
    const float wsc = p.wscale[0];
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const int s_ko = tid & 7;
    const int s_r = tid >> 3;
    float4 raX0a, raX0b, raX1a, raX1b, raX2a, raX2b, raX3a, raX3b, raY0a, raY0b, raY1a, raY1b, raY2a, raY2b, raY3a, raY3b;
    uint4 rb0a, rb0b, rb1a, rb1b, rb2a, rb2b, rb3a, rb3b;
    const int nkt = p.K / BK64;
    constexpr bool PRESPLIT = true;
    const float* W = nullptr;
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < BK64 / 16; ++s) {
            const int ko = 2 * s + hi;
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = Ah[ko * LDS_ROWS + wm * 64 + m * 32 + lo];
                al[m] = Al[ko * LDS_ROWS + wm * 64 + m * 32 + lo];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = Bh[ko * LDS_ROWS + wn * 64 + n * 32 + lo];
                bl[n] = Bl[ko * LDS_ROWS + wn * 64 + n * 32 + lo];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (KO & 1) {
                        acc[m][n][0] += __builtin_bit_cast(float, bh[n].x ^ al[m].y ^ bl[n].z ^ ah[m].w);
                        acc[m][n][1] += __builtin_bit_cast(float, bh[n].y ^ al[m].z ^ bl[n].w ^ ah[m].x);
                    } else {
                        acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                        acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                        acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                    }
                }
        }
    };
    unsigned sink = 0;
#define SINK4(v) sink ^= __builtin_bit_cast(unsigned, (v).x) ^ __builtin_bit_cast(unsigned, (v).y) ^ __builtin_bit_cast(unsigned, (v).z) ^ __builtin_bit_cast(unsigned, (v).w);
#define CONSUME(S) { SINK4(ra##S##0a) SINK4(ra##S##0b) SINK4(ra##S##1a) SINK4(ra##S##1b) SINK4(ra##S##2a) SINK4(ra##S##2b) SINK4(ra##S##3a) SINK4(ra##S##3b) \
                     SINK4(rb0a) SINK4(rb0b) SINK4(rb1a) SINK4(rb1b) SINK4(rb2a) SINK4(rb2b) SINK4(rb3a) SINK4(rb3b) }
    if (KO & 512) {
        raX0a = raX0b = raX1a = raX1b = raX2a = raX2b = raX3a = raX3b = make_float4(1.f, 2.f, 3.f, 4.f);
        raY0a = raY0b = raY1a = raY1b = raY2a = raY2b = raY3a = raY3b = make_float4(1.f, 2.f, 3.f, 4.f);
        rb0a = rb0b = rb1a = rb1b = rb2a = rb2b = rb3a = rb3b = make_uint4(1, 2, 3, 4);
    } else {
        LOADA(X, 0)
        LOADB(0)
        if (nkt > 1) LOADA(Y, 1)
    }
    if (KO & 1024) return;
    if (KO & 4) {
        STORESET(X)
        __syncthreads();
    }
    for (int kt = 0; kt < nkt; kt += 2) {
        if (KO & 32) CONSUME(X) else if (!(KO & 4)) STORESET(X)
        if (!(KO & 64)) __syncthreads();
        if (!(KO & 2)) {
            if (!(KO & 256)) if (kt + 1 < nkt) LOADB(kt + 1)
            if (!(KO & 128)) if (kt + 2 < nkt) LOADA(X, kt + 2)
        }
        if (!(KO & 16)) compute();
        if (!(KO & 64)) __syncthreads();
        if (kt + 1 < nkt) {
            if (KO & 32) CONSUME(Y) else if (!(KO & 4)) STORESET(Y)
            if (!(KO & 64)) __syncthreads();
            if (!(KO & 2)) {
                if (!(KO & 256)) if (kt + 2 < nkt) LOADB(kt + 2)
                if (!(KO & 128)) if (kt + 3 < nkt) LOADA(Y, kt + 3)
            }
            if (!(KO & 16)) compute();
            if (!(KO & 64)) __syncthreads();
        }
    }
    if (KO & 8) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[m][n][r];
        if (s == 1.2345f || sink == 0x12345u) p.C[tid] = s;
        return;
    }
    gemm_epilogue<EPI_BIAS>(p, c, acc, wsc, wm, wn, lo, hi, smem);
}

template <int KO>
static float run(const GemmP& p, int iters) {
    const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_kernel<KO>, dim3(ntiles), dim3(256), 0, 0, p);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(lab_kernel<KO>, dim3(ntiles), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}
static float run_prod(const GemmP& p, int iters) {
    const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, true>), dim3(ntiles), dim3(256), 0, 0, p);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, true>), dim3(ntiles), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const bool prod_only = argc > 1;
    const int shapes[][3] = {{65536, 512, 512}, {65536, 256, 256}, {65536, 768, 256}, {65536, 256, 512}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N, 0.1f);
        for (size_t i = 0; i < hA.size(); ++i) hA[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        for (size_t i = 0; i < hW.size(); ++i) hW[i] = (float)((i * 40503u) % 2001) / 20000.f - 0.05f;
        std::vector<unsigned short> hh(hW.size()), hl(hW.size());
        const float sc = split_weights_host(hW.data(), hW.size(), hh.data(), hl.data());
        float *dA, *dC, *db, *dsc;
        unsigned short *dh, *dl;
        hipMalloc(&dA, hA.size() * 4);
        hipMalloc(&dC, (size_t)M * N * 4);
        hipMalloc(&db, N * 4);
        hipMalloc(&dsc, 4);
        hipMalloc(&dh, hh.size() * 2);
        hipMalloc(&dl, hl.size() * 2);
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsc, &sc, 4, hipMemcpyHostToDevice);
        hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
        GemmP p;
        p.A = dA;
        p.lda = K;
        p.Wh = dh;
        p.Wl = dl;
        p.wscale = dsc;
        p.ldw = K;
        p.bias = db;
        p.C = dC;
        p.ldc = N;
        p.M = M;
        p.N = N;
        p.K = K;
        const double gf = 3.0 * 2.0 * M * N * K * 1e-9;
        printf("M=%d N=%d K=%d executed %.1f GF\n", M, N, K, gf);
        const float t = run_prod(p, prod_only ? 2 : 20);
        printf("  production            %8.1f us  %6.1f TF/s\n", t, gf / t);
        if (prod_only) continue;
#define RUN(KO, name)                                                              \
    {                                                                              \
        const float t_ = run<KO>(p, 20);                                           \
        printf("  %-22s%8.1f us  %6.1f TF/s\n", name, t_, gf / t_);   \
    }
        RUN(0, "lab copy")
        RUN(1, "no mfma")
        RUN(2, "no loop gloads")
        RUN(6, "no gloads/split/ldsst")
        RUN(8, "no epilogue")
        RUN(14, "mfma+ldsread only")
        RUN(15, "ldsread only")
        RUN(9, "no mfma, no epi")
        RUN(15 | 512, "ldsread, no prologue ld")
        RUN(15 | 512 | 64, "ldsread, no prol, nosync")
        RUN(14 | 512, "mfma+ldsrd, no prol")
        RUN(14 | 512 | 64, "mfma+ldsrd,noprol,nosync")
        RUN(512 | 1024, "empty kernel")
        RUN(8 | 16 | 32 | 64, "loads only")
        RUN(8 | 16 | 32 | 64 | 128, "B loads only")
        RUN(8 | 16 | 32 | 64 | 256, "A loads only")
        RUN(8 | 16 | 32, "loads + syncs")
        RUN(8 | 16, "loads+split+ldsst+sync")
        RUN(8 | 16 | 2, "split+ldsst+sync")
        RUN(8 | 2, "all but gloads, epi")
        hipFree(dA);
        hipFree(dC);
        hipFree(db);
        hipFree(dsc);
        hipFree(dh);
        hipFree(dl);
    }
    return 0;
}
