// Mutual nearest-neighbour matcher (imcui/hloc/matchers/nearest_neighbor.py:6-66) on MI355X:
// sim = D0 . D1^T on the f32 matrix cores, then fused row / column best-two + mutual check.
#include <math.h>

#include "gemm.h"
#include "imcui_hip.h"

struct NnWs {
    float* sim;
    int *m0, *m1;
    float* s0;
    size_t total;
    bool ok;
};
static NnWs nn_carve(void* ws, size_t bytes, int B, int N, int M) {
    WsAlloc a(ws, bytes);
    NnWs w;
    w.sim = a.get<float>((size_t)B * N * M);
    w.m0 = a.get<int>((size_t)B * N);
    w.m1 = a.get<int>((size_t)B * M);
    w.s0 = a.get<float>((size_t)B * N);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}
extern "C" size_t imcui_hip_mutual_nn_workspace_bytes(int B, int N, int M) {
    return nn_carve(nullptr, 0, B, N > 0 ? N : 1, M > 0 ? M : 1).total;
}

// best / second best of a strided vector, one wave; ties resolve to the lowest index
__device__ __forceinline__ void nn_best2(const float* v, long stride, int n, int lane, float& b1, int& i1, float& b2) {
    b1 = -INFINITY;
    b2 = -INFINITY;
    i1 = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float x = v[(long)j * stride];
        if (x > b1) {
            b2 = b1;
            b1 = x;
            i1 = j;
        } else if (x > b2) {
            b2 = x;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob1 = __shfl_xor(b1, o, 64), ob2 = __shfl_xor(b2, o, 64);
        const int oi1 = __shfl_xor(i1, o, 64);
        if (ob1 > b1 || (ob1 == b1 && oi1 < i1)) {
            b2 = fmaxf(b1, ob2);
            b1 = ob1;
            i1 = oi1;
        } else {
            b2 = fmaxf(b2, ob1);
        }
    }
}

// find_nn along rows (dir 0: for each n over m) or columns (dir 1)
__global__ __launch_bounds__(256) void nn_find_kernel(const float* __restrict__ sim, int N, int M, int dir, float ratio2,
                                                      float dist2, int use_ratio, int use_dist, int* __restrict__ match,
                                                      float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_out = dir ? M : N, n_in = dir ? N : M;
    if (i >= n_out) return;
    const float* base = sim + (size_t)b * N * M;
    float b1, b2;
    int i1;
    if (dir == 0)
        nn_best2(base + (size_t)i * M, 1, n_in, lane, b1, i1, b2);
    else
        nn_best2(base + i, M, n_in, lane, b1, i1, b2);
    if (lane == 0) {
        // dist_nn = 2 * (1 - sim_nn)
        const float d1 = 2.0f * (1.0f - b1);
        bool ok = true;
        if (use_ratio) ok = ok && (d1 <= ratio2 * (2.0f * (1.0f - b2)));
        if (use_dist) ok = ok && (d1 <= dist2);
        match[(size_t)b * n_out + i] = ok ? i1 : -1;
        if (score) score[(size_t)b * n_out + i] = ok ? (b1 + 1.0f) / 2.0f : 0.0f;
    }
}

__global__ void nn_mutual_kernel(const int* __restrict__ m0, const int* __restrict__ m1, int N, int M, int do_mutual,
                                 int* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int m = m0[(size_t)b * N + i];
    if (do_mutual && m > -1) {
        if (m1[(size_t)b * M + m] != i) m = -1;
    }
    out[(size_t)b * N + i] = m;
}

extern "C" int imcui_hip_mutual_nn(imcui_hip_t* h, const float* desc0, const float* desc1, int B, int N, int M, int D,
                                   double ratio_threshold, double distance_threshold, int do_mutual_check, int* matches0,
                                   float* scores0, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (B <= 0 || N <= 0) return IMCUI_OK;
    if (!matches0 || !scores0) return imcui_set_err(h, IMCUI_ERR_ARG, "mutual_nn: null output");
    if (M <= 0) {  // empty second set: everything unmatched (nearest_neighbor.py:39-48)
        hipMemsetAsync(matches0, 0xFF, (size_t)B * N * sizeof(int), stream);
        hipMemsetAsync(scores0, 0, (size_t)B * N * sizeof(float), stream);
        return IMCUI_OK;
    }
    if (D % 32 != 0 || !desc0 || !desc1) return imcui_set_err(h, IMCUI_ERR_ARG, "mutual_nn: D=%d must be a multiple of 32", D);
    NnWs w = nn_carve(ws, ws_bytes, B, N, M);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "mutual_nn: workspace too small (%zu < %zu)", ws_bytes, w.total);
    GemmP g;
    g.epi = EPI_BIAS;
    g.batch = B;
    g.A = desc0;
    g.lda = D;
    g.a_bs = (long)N * D;
    g.W = desc1;
    g.ldw = D;
    g.w_bs = (long)M * D;
    g.C = w.sim;
    g.ldc = M;
    g.c_bs = (long)N * M;
    g.M = N;
    g.N = M;
    g.K = D;
    int rc = gemm_launch(h, g, stream);
    if (rc != IMCUI_OK) return rc;
    // a single neighbour cannot pass a ratio test (nearest_neighbor.py:50-51)
    const int use_ratio = (ratio_threshold > 0.0) && N > 1 && M > 1;
    const int use_dist = distance_threshold > 0.0;
    // thresholds are squared in double (Python floats) before meeting the fp32 tensors
    const float r2 = (float)(ratio_threshold * ratio_threshold), d2 = (float)(distance_threshold * distance_threshold);
    hipLaunchKernelGGL(nn_find_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, stream, w.sim, N, M, 0, r2, d2, use_ratio, use_dist,
                       w.m0, scores0);
    hipLaunchKernelGGL(nn_find_kernel, dim3(cdiv(M, 4), B), dim3(256), 0, stream, w.sim, N, M, 1, r2, d2, use_ratio, use_dist,
                       w.m1, (float*)nullptr);
    hipLaunchKernelGGL(nn_mutual_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, stream, w.m0, w.m1, N, M, do_mutual_check,
                       matches0);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
