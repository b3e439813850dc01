// Dual-softmax matcher (imcui/hloc/matchers/dual_softmax.py:8-41) on MI355X: L2-normalise the descriptors,
// sim = D0^T D1 * inv_temperature on the matrix cores, P = softmax over rows x softmax over columns, match (i, j) when P is both its
// row and its column maximum and exceeds the threshold.
// Round 5: descriptors of up to 256 channels (every zoo entry: DISK 128, SuperPoint 256) run on the matrix-free two-pass kernel of
// simred.hip -- sim is never stored.  Wider descriptors keep the round-1 form: sim materialised (16.8 MB per 2048 x 2048 pair), then
// HBM-bound passes over it.
#include <math.h>

#include "gemm.h"
#include "imcui_hip.h"
#include "simred.h"

struct DsWs {
    float *a, *b, *sim, *rmax, *rsum, *cmax, *csum, *rbest, *cbest;
    int* bestj;
    SimDsWs ds;
    size_t total;
    bool ok;
};
// channel count the descriptors are padded to: the widths the persistent kernel takes (64 / 128 / 256), else a multiple of 32
static int ds_pad(int C) { return C <= 64 ? 64 : C <= 128 ? 128 : C <= 256 ? 256 : (int)align_up((size_t)C, 32); }
static DsWs ds_carve(void* ws, size_t bytes, int B, int Cp, int N, int M) {
    WsAlloc al(ws, bytes);
    DsWs w;
    w.a = al.get<float>((size_t)B * N * Cp);
    w.b = al.get<float>((size_t)B * M * Cp);
    w.sim = nullptr;
    w.bestj = nullptr;
    if (simred_ok(Cp)) {
        simred_ds_carve(al, B, N, M, Cp, w.ds);
        w.bestj = al.get<int>((size_t)B * N);
    } else {
        w.sim = al.get<float>((size_t)B * N * M);
    }
    w.rmax = al.get<float>((size_t)B * N);
    w.rsum = al.get<float>((size_t)B * N);
    w.rbest = al.get<float>((size_t)B * N);
    w.cmax = al.get<float>((size_t)B * M);
    w.csum = al.get<float>((size_t)B * M);
    w.cbest = al.get<float>((size_t)B * M);
    w.total = al.off;
    w.ok = al.ok;
    return w;
}
extern "C" size_t imcui_hip_dual_softmax_workspace_bytes(int B, int C, int N, int M) {
    return ds_carve(nullptr, 0, B > 0 ? B : 1, ds_pad(C > 0 ? C : 1), N > 0 ? N : 1, M > 0 ? M : 1).total;
}

// [B, C, n] channels-first -> [B, n, Cp] rows, divided by the L2 norm over C (dual_softmax.py:20-22), zero padded to Cp
__global__ __launch_bounds__(256) void ds_prep_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int Cp, int n,
                                                      int normalize) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* src = in + (size_t)b * C * n + i;
    float ss = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float x = src[(size_t)c * n];
        ss = fmaf(x, x, ss);
    }
    const float nrm = normalize ? sqrtf(ss) : 1.0f;
    float* dst = out + ((size_t)b * n + i) * Cp;
    for (int c = 0; c < C; ++c) dst[c] = src[(size_t)c * n] / nrm;
    for (int c = C; c < Cp; ++c) dst[c] = 0.0f;
}

// row pass: one wave per row i -> max_j, sum_j exp(s - max)
__global__ __launch_bounds__(256) void ds_rowstat_kernel(const float* __restrict__ sim, int N, int M, float* __restrict__ rmax,
                                                         float* __restrict__ rsum) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* row = sim + ((size_t)b * N + i) * M;
    float m = -INFINITY;
    for (int j = lane; j < M; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.0f;
    for (int j = lane; j < M; j += 64) s += expf(row[j] - m);
    s = wave_sum(s);
    if (lane == 0) {
        rmax[(size_t)b * N + i] = m;
        rsum[(size_t)b * N + i] = s;
    }
}
// column pass: block = 64 columns x 4 row groups, eight rows per trip
__global__ __launch_bounds__(256) void ds_colstat_kernel(const float* __restrict__ sim, int N, int M, float* __restrict__ cmax,
                                                         float* __restrict__ csum) {
    __shared__ float sm[4][64], ss[4][64];
    const int b = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const float* base = sim + (size_t)b * N * M;
    float m = -INFINITY;
    if (j < M)
        for (int i = g; i < N; i += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i + 4 * u < N) ? base[(size_t)(i + 4 * u) * M + j] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
        }
    sm[g][c] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c]));
    float s = 0.0f;
    if (j < M)
        for (int i = g; i < N; i += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i + 4 * u < N) ? base[(size_t)(i + 4 * u) * M + j] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + 4 * u < N) s += expf(v[u] - m);
        }
    ss[g][c] = s;
    __syncthreads();
    if (g == 0 && j < M) {
        cmax[(size_t)b * M + j] = m;
        csum[(size_t)b * M + j] = ss[0][c] + ss[1][c] + ss[2][c] + ss[3][c];
    }
}
// P[i, j] = softmax(sim, dim=-2)[i, j] * softmax(sim, dim=-1)[i, j]; always evaluated by this one expression so that
// the equality tests below compare bit-identical values
__device__ __forceinline__ float ds_p(float s, float cm, float cs, float rm, float rs) { return (expf(s - cm) / cs) * (expf(s - rm) / rs); }

__global__ __launch_bounds__(256) void ds_rowbest_kernel(const float* __restrict__ sim, int N, int M, const float* __restrict__ rmax,
                                                         const float* __restrict__ rsum, const float* __restrict__ cmax,
                                                         const float* __restrict__ csum, float* __restrict__ rbest) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* row = sim + ((size_t)b * N + i) * M;
    const float rm = rmax[(size_t)b * N + i], rs = rsum[(size_t)b * N + i];
    float best = -1.0f;
    for (int j = lane; j < M; j += 64) best = fmaxf(best, ds_p(row[j], cmax[(size_t)b * M + j], csum[(size_t)b * M + j], rm, rs));
    best = wave_max(best);
    if (lane == 0) rbest[(size_t)b * N + i] = best;
}
__global__ __launch_bounds__(256) void ds_colbest_kernel(const float* __restrict__ sim, int N, int M, const float* __restrict__ rmax,
                                                         const float* __restrict__ rsum, const float* __restrict__ cmax,
                                                         const float* __restrict__ csum, float* __restrict__ cbest) {
    __shared__ float sv[4][64];
    const int b = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const float* base = sim + (size_t)b * N * M;
    float best = -1.0f;
    if (j < M) {
        const float cm = cmax[(size_t)b * M + j], cs = csum[(size_t)b * M + j];
        for (int i0 = g; i0 < N; i0 += 32) {
            float v[8], rm[8], rs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + 4 * u, N - 1);
                v[u] = base[(size_t)i * M + j];
                rm[u] = rmax[(size_t)b * N + i];
                rs[u] = rsum[(size_t)b * N + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 4 * u < N) best = fmaxf(best, ds_p(v[u], cm, cs, rm[u], rs[u]));
        }
    }
    sv[g][c] = best;
    __syncthreads();
    if (g == 0 && j < M) cbest[(size_t)b * M + j] = fmaxf(fmaxf(sv[0][c], sv[1][c]), fmaxf(sv[2][c], sv[3][c]));
}
// per row: the LAST column that is the row maximum, the column maximum and above the threshold (the reference
// scatters nonzero() results in row-major order, so the last qualifying column wins, dual_softmax.py:24-32)
__global__ __launch_bounds__(256) void ds_decide_kernel(const float* __restrict__ sim, int N, int M, const float* __restrict__ rmax,
                                                        const float* __restrict__ rsum, const float* __restrict__ cmax,
                                                        const float* __restrict__ csum, const float* __restrict__ rbest,
                                                        const float* __restrict__ cbest, float thr, int* __restrict__ matches0,
                                                        float* __restrict__ scores0) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* row = sim + ((size_t)b * N + i) * M;
    const float rm = rmax[(size_t)b * N + i], rs = rsum[(size_t)b * N + i], rb = rbest[(size_t)b * N + i];
    int bj = -1;
    float bp = 0.0f;
    for (int j = lane; j < M; j += 64) {
        const float pv = ds_p(row[j], cmax[(size_t)b * M + j], csum[(size_t)b * M + j], rm, rs);
        if (pv == rb && pv == cbest[(size_t)b * M + j] && pv > thr) {  // j ascends per lane: keeps the largest
            bj = j;
            bp = pv;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int oj = __shfl_xor(bj, o, 64);
        const float op = __shfl_xor(bp, o, 64);
        if (oj > bj) {
            bj = oj;
            bp = op;
        }
    }
    if (lane == 0) {
        matches0[(size_t)b * N + i] = bj;
        scores0[(size_t)b * N + i] = bp;
    }
}

// matrix-free path: the row best (value, column) and the column bests come from simred.hip, which saw the columns in REVERSED order (the
// reference keeps the LAST qualifying column of a row, the kernel reports the first column attaining the row maximum)
__global__ void ds_decide2_kernel(const float* __restrict__ best, const int* __restrict__ bestj, const float* __restrict__ cbest, int N, int M, float thr,
                                  int* __restrict__ matches0, float* __restrict__ scores0) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float v = best[(size_t)b * N + i];
    const int jr = bestj[(size_t)b * N + i];
    const bool ok = jr >= 0 && jr < M && v > thr && v == cbest[(size_t)b * M + jr];
    matches0[(size_t)b * N + i] = ok ? M - 1 - jr : -1;
    scores0[(size_t)b * N + i] = ok ? v : 0.0f;
}

extern "C" int imcui_hip_dual_softmax(imcui_hip_t* h, const float* desc0, const float* desc1, int B, int C, int N, int M,
                                      double threshold, double inv_temperature, int normalize, int* matches0, float* scores0,
                                      void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h || !matches0 || !scores0 || B < 0 || C <= 0 || N < 0 || M < 0) return imcui_set_err(h, IMCUI_ERR_ARG, "dual_softmax: bad argument");
    if (B == 0 || N == 0) return IMCUI_OK;
    if (M == 0) {  // nothing to match against: every row unmatched
        hipMemsetAsync(matches0, 0xFF, (size_t)B * N * sizeof(int), stream);
        hipMemsetAsync(scores0, 0, (size_t)B * N * sizeof(float), stream);
        return IMCUI_OK;
    }
    if (!desc0 || !desc1 || !ws) return imcui_set_err(h, IMCUI_ERR_ARG, "dual_softmax: null argument");
    const int Cp = ds_pad(C);
    const DsWs w = ds_carve(ws, ws_bytes, B, Cp, N, M);
    if (!w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "dual_softmax: workspace too small (%zu < %zu bytes)", ws_bytes, w.total);
    const dim3 blk(256);
    hipLaunchKernelGGL(ds_prep_kernel, dim3(cdiv(N, 256), B), blk, 0, stream, desc0, w.a, C, Cp, N, normalize);
    hipLaunchKernelGGL(ds_prep_kernel, dim3(cdiv(M, 256), B), blk, 0, stream, desc1, w.b, C, Cp, M, normalize);
    if (simred_ok(Cp)) {
        // descriptors1 enter in reversed row order (negative row stride from the last row): first-in-reversed = last-in-original
        const int rc = simred_dual_softmax(h, w.ds, w.a, Cp, (long)N * Cp, w.b + (size_t)(M - 1) * Cp, -(long)Cp, (long)M * Cp, B, N, M, Cp, (float)inv_temperature,
                                           (float)threshold, w.rmax, w.rsum, w.cmax, w.csum, w.rbest, w.bestj, w.cbest, stream);
        if (rc != IMCUI_OK) return rc;
        hipLaunchKernelGGL(ds_decide2_kernel, dim3(cdiv(N, 256), B), blk, 0, stream, w.rbest, w.bestj, w.cbest, N, M, (float)threshold, matches0, scores0);
        IMCUI_CHECK_LAUNCH(h);
        return IMCUI_OK;
    }
    GemmP g;
    g.epi = EPI_BIAS;
    g.batch = B;
    g.A = w.a;
    g.lda = Cp;
    g.a_bs = (long)N * Cp;
    g.W = w.b;
    g.ldw = Cp;
    g.w_bs = (long)M * Cp;
    g.C = w.sim;
    g.ldc = M;
    g.c_bs = (long)N * M;
    g.M = N;
    g.N = M;
    g.K = Cp;
    g.alpha = (float)inv_temperature;
    const int r = gemm_launch(h, g, stream);
    if (r != IMCUI_OK) return r;
    const dim3 rg(cdiv(N, 4), B), cg(cdiv(M, 64), B);
    hipLaunchKernelGGL(ds_rowstat_kernel, rg, blk, 0, stream, w.sim, N, M, w.rmax, w.rsum);
    hipLaunchKernelGGL(ds_colstat_kernel, cg, blk, 0, stream, w.sim, N, M, w.cmax, w.csum);
    hipLaunchKernelGGL(ds_rowbest_kernel, rg, blk, 0, stream, w.sim, N, M, w.rmax, w.rsum, w.cmax, w.csum, w.rbest);
    hipLaunchKernelGGL(ds_colbest_kernel, cg, blk, 0, stream, w.sim, N, M, w.rmax, w.rsum, w.cmax, w.csum, w.cbest);
    hipLaunchKernelGGL(ds_decide_kernel, rg, blk, 0, stream, w.sim, N, M, w.rmax, w.rsum, w.cmax, w.csum, w.rbest, w.cbest,
                       (float)threshold, matches0, scores0);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
