// Lab: what LightGlue's K = 256 projection launches wait for (gemm_wreg_kernel<3> / <4> at 0.35 of the matrix
// pipe, 2.9 TB/s -- neither bound).  Includes the product kernel (csrc/gemm_wreg.hip) compiled with WR_LAB, whose device word of switches
// knocks out parts of it: the epilogue's global stores, the K loop, the epilogue, the weight stream, the activation stream.  Synthetic
// operands at the headline's shape: 64 pairs x 2 images x 2048 tokens, K = 256, N = 768 (q | k | v) and N = 512 (cross: qk | v).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWR_LAB -fno-slp-vectorize -I include -I image-matching-webui_amd/csrc tools/wreg_lab.hip -o tools/wreg_lab
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../image-matching-webui_amd/csrc/gemm_wreg.hip"

int imcui_set_err(imcui_hip_s*, int code, const char*, ...) { return code; }

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static float run(const GemmP& p, int flags, int tile, int reps = 10) {
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wr_lab_flags), &flags, sizeof(int)));
    imcui_hip_s h = {};
    h.opt[OPT_GEMM_WREG] = 2;
    h.opt[OPT_WREG_PIPE] = 1;
    h.opt[OPT_WREG_TILE] = tile;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) gemm_wreg_launch(&h, p, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) gemm_wreg_launch(&h, p, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

// ---- how fast can 805 MB be WRITTEN at all?  Pure store kernels in the epilogue's three patterns.
//  0: every wave instruction writes 1 KB contiguous (the q / k planes: 8 token rows x 128 B, consecutive)
//  1: every wave instruction writes 16 segments of 64 B, 4 KB apart (the V^T planes: 16 feature rows x 4 lanes x 16 B, row pitch R halves)
//  2: every wave instruction writes 8 segments of 128 B, 4 KB apart (V^T with whole lines)
template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(uint4* out, size_t n16, int iters) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4 v = make_uint4(lane, wave, 3u, 4u);
    for (int it = 0; it < iters; ++it) {
        size_t idx;
        if (PAT == 0) idx = (wave * iters + it) * 64 + lane;
        else if (PAT == 1) idx = ((wave * iters + it) >> 6) * 16384 + (size_t)(lane >> 2) * 256 + (((wave * iters + it) & 63) * 4) + (lane & 3);  // 16 rows of a 64-row block, 4 KB pitch
        else idx = ((wave * iters + it) >> 5) * 8192 + (size_t)(lane >> 3) * 256 + (((wave * iters + it) & 31) * 8) + (lane & 7);
        if (idx < n16) out[idx] = v;
    }
}
template <int PAT>
static void store_rate(const char* name, uint4* buf, size_t bytes) {
    const size_t n16 = bytes / 16;
    const int iters = 64;
    const int blocks = (int)(n16 / 64 / iters / 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((store_kernel<PAT>), dim3(blocks), dim3(256), 0, 0, buf, n16, iters);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((store_kernel<PAT>), dim3(blocks), dim3(256), 0, 0, buf, n16, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("store pattern %-58s %7.1f us per %zu MB  = %.2f TB/s\n", name, ms / 5 * 1000.f, bytes >> 20, (double)bytes / (ms / 5 * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 64;
    const int R = 2048, S = 2 * pairs, K = 256;
    const long M = (long)S * R;
    float *A, *bias, *cs, *sn, *wsc;
    unsigned short *Wh, *Wl;
    float *Q, *Kt, *V;
    int* cnt;
    CK(hipMalloc(&A, M * K * 4));
    CK(hipMalloc(&bias, 768 * 4));
    CK(hipMalloc(&cs, M * 32 * 4));
    CK(hipMalloc(&sn, M * 32 * 4));
    CK(hipMalloc(&wsc, 4));
    CK(hipMalloc(&Wh, 768 * K * 2));
    CK(hipMalloc(&Wl, 768 * K * 2));
    const size_t plane = (size_t)M * 256;  // halves per plane
    CK(hipMalloc(&Q, plane * 4));
    CK(hipMalloc(&Kt, plane * 4));
    CK(hipMalloc(&V, plane * 4));
    CK(hipMalloc(&cnt, S * 4));
    {
        std::vector<float> a((size_t)M * K);
        unsigned x = 12345u;
        for (auto& v : a) {
            x = x * 1664525u + 1013904223u;
            v = ((x >> 8) & 0xffff) / 65536.0f - 0.5f;
        }
        CK(hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> w(768 * K);
        for (auto& v : w) {
            x = x * 1664525u + 1013904223u;
            v = (unsigned short)(0x3000 | ((x >> 12) & 0x83ff));  // f16 values of magnitude ~0.1 .. 0.25, random signs
        }
        CK(hipMemcpy(Wh, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        for (auto& v : w) v = (unsigned short)(v & 0x8fff);
        CK(hipMemcpy(Wl, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> t((size_t)M * 32, 0.7f);
        CK(hipMemcpy(cs, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(sn, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> b(768, 0.01f);
        CK(hipMemcpy(bias, b.data(), 768 * 4, hipMemcpyHostToDevice));
        const float one = 1.0f;
        CK(hipMemcpy(wsc, &one, 4, hipMemcpyHostToDevice));
        std::vector<int> c(S, R);
        CK(hipMemcpy(cnt, c.data(), S * 4, hipMemcpyHostToDevice));
    }
    store_rate<0>("1 KB contiguous per wave instruction (q / k planes)", reinterpret_cast<uint4*>(Q), plane * 4);
    store_rate<1>("16 x 64 B, 4 KB apart (V^T planes today)", reinterpret_cast<uint4*>(Q), plane * 4);
    store_rate<2>("8 x 128 B, 4 KB apart (V^T planes, whole lines)", reinterpret_cast<uint4*>(Q), plane * 4);
    GemmP g;
    g.A = A;
    g.lda = K;
    g.M = (int)M;
    g.K = K;
    g.Wh = Wh;
    g.Wl = Wl;
    g.wscale = wsc;
    g.bias = bias;
    g.cnt = cnt;
    g.rows_per_seq = R;
    g.v_transposed = 1;
    g.split_out = 1;
    g.plane_halves = plane;
    g.Q = Q;
    g.Kt = Kt;
    g.V = V;
    g.rope_cos = cs;
    g.rope_sin = sn;
    g.alpha = 0.18f;
    g.heads = 4;
    g.C = Q;  // (WR_F_NOEPI's keep-alive store)
    g.st_nct = 256;  // (lab: the CU count for the stagger switch)
    struct { const char* name; int f; } rows[] = {
        {"full kernel", 0},
        {"no global stores in the epilogue", WR_F_NOSTORE},
        {"no epilogue at all", WR_F_NOEPI},
        {"one K tile (of 8), full epilogue", WR_F_ONETILE},
        {"one K tile, no stores", WR_F_ONETILE | WR_F_NOSTORE},
        {"weights loaded once (registers re-used), full epilogue", WR_F_NOWLOAD},
        {"activations loaded once, full epilogue", WR_F_NOXLOAD},
        {"no weight stream, no activation stream, full epilogue", WR_F_NOWLOAD | WR_F_NOXLOAD},
        {"no streams, no epilogue (the bare matrix loop)", WR_F_NOWLOAD | WR_F_NOXLOAD | WR_F_NOEPI},
        {"weights + activations streamed, no epilogue", WR_F_NOEPI},
        {"no stores in the V^T panels (64-byte segments)", WR_F_NOSTORE_V},
        {"no stores in the q / k panels (128-byte rows)", WR_F_NOSTORE_QK},
    };
    for (int pass = 0; pass < 2; ++pass) {
        g.epi = pass == 0 ? EPI_QKV : EPI_CROSS;
        g.N = pass == 0 ? 768 : 512;
        printf("## %s: M = %ld tokens (%d pairs), K = 256, N = %d; %ld workgroups of 128 tokens x 256 features\n", pass == 0 ? "EPI_QKV  (gemm_wreg_kernel<3>)" : "EPI_CROSS (gemm_wreg_kernel<4>)",
               M, pairs, g.N, M / 128 * (g.N / 256));
        const double gf = 2.0 * M * g.N * K * 3 / 1e9, mbw = (double)M * g.N * 4 / 1e6, mbr = (double)M * K * 4 / 1e6;
        printf("## executed %.0f GF (three f16 products); written %.0f MB, read %.0f MB (activations once)\n", gf, mbw, mbr);
        for (auto& r : rows) {
            const float us = run(g, r.f, 128);
            printf("%-62s %8.1f us   (%.0f TF/s executed, %.2f TB/s of compulsory traffic)\n", r.name, us, gf / us * 1e3, (mbw + mbr) / us);
        }
        printf("token tile 64 (twice the workgroups), full kernel:%*s %8.1f us\n", 13, "", run(g, 0, 64));
        for (int tile : {128, 64, 32})
            for (int d : {0, 4, 16, 40}) printf("token tile %3d, later dispatch rounds start %2d x ~0.5 us x round late:%*s %8.1f us\n", tile, d, 3, "", run(g, d << 8, tile));
        printf("\n");
    }
    return 0;
}
