// Mutual nearest-neighbour matcher (imcui/hloc/matchers/nearest_neighbor.py:6-66) on MI355X:
// sim = D0 . D1^T on the matrix cores; split arithmetic: the similarity tiles are reduced to best / second-best partials in the GEMM's
// epilogue (EPI_NNSTAT) and never stored; exact-f32 mode: materialised, then row / column best-two passes.  Then the mutual check.
#include <math.h>

#include "gemm.h"
#include "imcui_hip.h"
#include "simred.h"

struct NnWs {
    uint4 *pk0, *pk1;            // round 5: the descriptors as MFMA fragments (simred_pack), D in {64, 128, 256}
    float* sim;                  // materialised path (exact-f32 mode, other descriptor widths) only
    float *rb1, *rb2, *cb1, *cb2;  // fused path: nearest-neighbour partials of the tiles (EPI_NNSTAT)
    int *ri1, *ci1;
    int *m0, *m1;
    float* s0;
    int nct, nrh;
    size_t total;
    bool ok;
};
// fused = the split arithmetic: the similarity tiles are reduced in the GEMM's epilogue and never stored (12 B per row and 128-column
// tile + 12 B per column and 64-row half instead of 4 N M bytes written once and read twice)
// D > 0: room for the packed descriptors of the persistent kernel (round 5) as well; D = 0: the round-4 layouts only
static NnWs nn_carve(void* ws, size_t bytes, int B, int N, int M, bool fused, int D) {
    WsAlloc a(ws, bytes);
    NnWs w{};
    w.nct = cdiv(M, 128);
    w.nrh = 2 * cdiv(N, 128);
    if (D > 0 && simred_ok(D)) {
        w.pk0 = a.get<uint4>((size_t)B * simred_packed_uint4(N, D));
        w.pk1 = a.get<uint4>((size_t)B * simred_packed_uint4(M, D));
    }
    if (fused) {
        w.rb1 = a.get<float>((size_t)B * w.nct * N);
        w.rb2 = a.get<float>((size_t)B * w.nct * N);
        w.ri1 = a.get<int>((size_t)B * w.nct * N);
        w.cb1 = a.get<float>((size_t)B * w.nrh * M);
        w.cb2 = a.get<float>((size_t)B * w.nrh * M);
        w.ci1 = a.get<int>((size_t)B * w.nrh * M);
    } else {
        w.sim = a.get<float>((size_t)B * N * M);
    }
    w.m0 = a.get<int>((size_t)B * N);
    w.m1 = a.get<int>((size_t)B * M);
    w.s0 = a.get<float>((size_t)B * N);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}
// The entry points without a descriptor width size the workspace for D = 256, the widest the persistent kernel takes.
// Size that serves either arithmetic (the materialised similarity of the exact-f32 mode is the larger one) ...
extern "C" size_t imcui_hip_mutual_nn_workspace_bytes(int B, int N, int M) {
    const size_t a = nn_carve(nullptr, 0, B, N > 0 ? N : 1, M > 0 ? M : 1, false, 256).total;
    const size_t b = nn_carve(nullptr, 0, B, N > 0 ? N : 1, M > 0 ? M : 1, true, 256).total;
    return a > b ? a : b;
}
// does this call run on the persistent kernel?  (either arithmetic; the tile paths keep every other descriptor width)
static bool nn_persistent(const imcui_hip_s* h, int D) { return h != nullptr && h->opt[OPT_SIMRED] != 0 && simred_ok(D); }
// ... and what THIS handle's arithmetic needs (precision 1: no similarity matrix)
extern "C" size_t imcui_hip_mutual_nn_workspace_bytes_for(imcui_hip_t* h, int B, int N, int M) {
    if (h == nullptr || h->precision != 1) return imcui_hip_mutual_nn_workspace_bytes(B, N, M);  // (exact-f32 mode: the width decides the path at call time)
    return nn_carve(nullptr, 0, B, N > 0 ? N : 1, M > 0 ? M : 1, h != nullptr && (h->precision == 1 || h->opt[OPT_SIMRED] != 0), 256).total;
}
// ... for descriptors of width D (round 5)
extern "C" size_t imcui_hip_mutual_nn_workspace_bytes_d(imcui_hip_t* h, int B, int N, int M, int D) {
    return nn_carve(nullptr, 0, B, N > 0 ? N : 1, M > 0 ? M : 1, h != nullptr && (h->precision == 1 || nn_persistent(h, D)), D).total;
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
        match[(size_t)b * n_out + i] = (ok && i1 != 0x7fffffff) ? i1 : -1;  // (an all-NaN row never updates i1: unmatched, not an out-of-range index)
        if (score) score[(size_t)b * n_out + i] = ok ? (b1 + 1.0f) / 2.0f : 0.0f;
    }
}

// fused path: fold the tile partials of one direction (slots in increasing index order; lowest index on ties) and apply find_nn's tests
__global__ __launch_bounds__(256) void nn_merge_kernel(const float* __restrict__ pb1, const int* __restrict__ pi1, const float* __restrict__ pb2,
                                                       int nslots, int n_out, float ratio2, float dist2, int use_ratio, int use_dist,
                                                       int* __restrict__ match, float* __restrict__ score) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    float b1 = -INFINITY, b2 = -INFINITY;
    int i1 = 0x7fffffff;
    for (int s = 0; s < nslots; ++s) {
        const size_t o = ((size_t)b * nslots + s) * n_out + i;
        const float ob1 = pb1[o], ob2 = pb2 ? pb2[o] : -INFINITY;  // (no second best without a ratio test: SR_NN1)
        const int oi1 = pi1[o];
        if (ob1 > b1 || (ob1 == b1 && oi1 < i1)) {
            b2 = fmaxf(b1, ob2);
            b1 = ob1;
            i1 = oi1;
        } else {
            b2 = fmaxf(b2, ob1);
        }
    }
    const float d1 = 2.0f * (1.0f - b1);  // dist_nn = 2 * (1 - sim_nn)
    bool ok = true;
    if (use_ratio) ok = ok && (d1 <= ratio2 * (2.0f * (1.0f - b2)));
    if (use_dist) ok = ok && (d1 <= dist2);
    match[(size_t)b * n_out + i] = (ok && i1 != 0x7fffffff) ? i1 : -1;  // (an all-NaN row never updates i1: unmatched, not an out-of-range index)
    if (score) score[(size_t)b * n_out + i] = ok ? (b1 + 1.0f) / 2.0f : 0.0f;
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

// descriptors given by strides: element (row, k) of batch b at d[b * bs + row * ldr + k * ldk]  ([B, N, D]: ldr = D, ldk = 1; the
// reference's [B, D, N]: ldr = 1, ldk = N -- only the persistent path takes that one)
static int nn_forward(imcui_hip_t* h, const float* desc0, long ldr0, long ldk0, long bs0, const float* desc1, long ldr1, long ldk1, long bs1, int B, int N, int M,
                      int D, double ratio_threshold, double distance_threshold, int do_mutual_check, int* matches0, float* scores0, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
    const bool persistent = nn_persistent(h, D);
    // a single neighbour cannot pass a ratio test (nearest_neighbor.py:50-51)
    const int use_ratio = (ratio_threshold > 0.0) && N > 1 && M > 1;
    const bool fused = h->precision == 1 || persistent;
    NnWs w = nn_carve(ws, ws_bytes, B, N, M, fused, persistent ? D : 0);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "mutual_nn: workspace too small (%zu < %zu)", ws_bytes, w.total);
    int nslot_r = w.nct, nslot_c = w.nrh;
    if (persistent) {
        // round 5: both descriptor sets packed once into MFMA fragments, then ONE launch of the persistent similarity-and-reduce kernel
        // (simred.hip): row results in registers across the column tiles, column partials per 128-row block
        if (h->precision == 1 && h->range_flag && ldk0 == 1 && ldk1 == 1) {
            for (int z = 0; z < B; ++z) {
                imcui_range_check(h, desc0 + (size_t)z * bs0, N, D, ldr0, nullptr, 0, stream);
                imcui_range_check(h, desc1 + (size_t)z * bs1, M, D, ldr1, nullptr, 0, stream);
            }
        }
        simred_pack(h, desc0, ldr0, ldk0, bs0, N, D, B, nullptr, 0, w.pk0, stream);
        simred_pack(h, desc1, ldr1, ldk1, bs1, M, D, B, nullptr, 0, w.pk1, stream);
        SimRedP p;
        p.mode = use_ratio ? SR_NN : SR_NN1;  // the second best is only read by the ratio test
        p.Ap = w.pk0, p.Bp = w.pk1;
        p.ap_bs = (long)simred_packed_uint4(N, D), p.bp_bs = (long)simred_packed_uint4(M, D);
        p.M = N, p.N = M, p.K = D, p.batch = B;
        p.nchunk = simred_chunks(B, N, M);  // (<= nct: the row partial arrays above have a slot per column tile)
        p.r0 = w.rb1, p.r1 = w.rb2, p.ri = w.ri1, p.r_pitch = N;
        p.c0 = w.cb1, p.c1 = w.cb2, p.ci = w.ci1, p.c_pitch = M;
        const int rc = simred_launch(h, p, stream);
        if (rc != IMCUI_OK) return rc;
        nslot_r = p.nchunk;
        nslot_c = cdiv(N, 128);
    } else {
        if (ldk0 != 1 || ldk1 != 1) return imcui_set_err(h, IMCUI_ERR_ARG, "mutual_nn: the tile path needs row-major descriptors");
        GemmP g;
        g.epi = fused ? EPI_NNSTAT : EPI_BIAS;
        g.batch = B;
        g.A = desc0;
        g.lda = ldr0;
        g.a_bs = bs0;
        g.W = desc1;
        g.ldw = ldr1;
        g.w_bs = bs1;
        g.C = w.sim;
        g.ldc = M;
        g.c_bs = (long)N * M;
        if (fused) {
            g.st_rpm = w.rb1, g.st_rps = w.rb2, g.st_rpi = w.ri1, g.st_rpitch = N, g.st_nct = w.nct;
            g.st_cpm = w.cb1, g.st_cps = w.cb2, g.st_cpi = w.ci1, g.st_cpitch = M, g.st_nrh = w.nrh;
        }
        g.M = N;
        g.N = M;
        g.K = D;
        const int rc = gemm_launch(h, g, stream);
        if (rc != IMCUI_OK) return rc;
    }
    const int use_dist = distance_threshold > 0.0;
    // thresholds are squared in double (Python floats) before meeting the fp32 tensors
    const float r2 = (float)(ratio_threshold * ratio_threshold), d2 = (float)(distance_threshold * distance_threshold);
    if (fused) {
        hipLaunchKernelGGL(nn_merge_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, stream, w.rb1, w.ri1, use_ratio || !persistent ? w.rb2 : nullptr, nslot_r, N, r2, d2, use_ratio, use_dist, w.m0,
                           scores0);
        hipLaunchKernelGGL(nn_merge_kernel, dim3(cdiv(M, 256), B), dim3(256), 0, stream, w.cb1, w.ci1, use_ratio || !persistent ? w.cb2 : nullptr, nslot_c, M, r2, d2, use_ratio, use_dist, w.m1,
                           (float*)nullptr);
    } else {
        hipLaunchKernelGGL(nn_find_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, stream, w.sim, N, M, 0, r2, d2, use_ratio, use_dist,
                           w.m0, scores0);
        hipLaunchKernelGGL(nn_find_kernel, dim3(cdiv(M, 4), B), dim3(256), 0, stream, w.sim, N, M, 1, r2, d2, use_ratio, use_dist,
                           w.m1, (float*)nullptr);
    }
    hipLaunchKernelGGL(nn_mutual_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, stream, w.m0, w.m1, N, M, do_mutual_check,
                       matches0);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
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
    return nn_forward(h, desc0, D, 1, (long)N * D, desc1, D, 1, (long)M * D, B, N, M, D, ratio_threshold, distance_threshold, do_mutual_check, matches0, scores0, ws,
                      ws_bytes, stream);
}

// ------------------------------------------------------------------ the same matcher on the reference's own layout
// `NearestNeighbor._forward` receives descriptors0 [B, D, N] / descriptors1 [B, D, M] (one COLUMN per descriptor); the kernels above
// want one row per descriptor.  A tiled transpose through LDS (32 x 32 tiles, 33-float rows: both sides move whole 128-byte runs) into
// the workspace, then the call above -- torch's `permute().contiguous()` copy fetched 8 x the bytes it moved (12 % of a matcher step).
__global__ __launch_bounds__(256) void nn_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int D, int N) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, n0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* src = in + (size_t)b * D * N;
    float* dst = out + (size_t)b * N * D;
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (d0 + r < D && n0 + tx < N) tile[r][tx] = src[(size_t)(d0 + r) * N + n0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < N && d0 + tx < D) dst[(size_t)(n0 + r) * D + d0 + tx] = tile[tx][r];
}

extern "C" size_t imcui_hip_mutual_nn_dn_workspace_bytes_for(imcui_hip_t* h, int B, int N, int M, int D) {
    if (nn_persistent(h, D)) return imcui_hip_mutual_nn_workspace_bytes_d(h, B, N, M, D);  // packed straight from [B, D, N]: no transposed copy
    WsAlloc a(nullptr, 0);
    a.get<float>((size_t)B * (N > 0 ? N : 1) * D);
    a.get<float>((size_t)B * (M > 0 ? M : 1) * D);
    return a.off + imcui_hip_mutual_nn_workspace_bytes_d(h, B, N, M, D);
}

extern "C" int imcui_hip_mutual_nn_dn(imcui_hip_t* h, const float* desc0_dn, const float* desc1_dm, int B, int N, int M, int D, double ratio_threshold,
                                      double distance_threshold, int do_mutual_check, int* matches0, float* scores0, void* ws, size_t ws_bytes,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (B <= 0 || N <= 0) return IMCUI_OK;
    if (M <= 0) return imcui_hip_mutual_nn(h, desc0_dn, desc1_dm, B, N, M, D, ratio_threshold, distance_threshold, do_mutual_check, matches0, scores0, ws, ws_bytes, stream_);
    if (D <= 0 || !desc0_dn || !desc1_dm) return imcui_set_err(h, IMCUI_ERR_ARG, "mutual_nn_dn: null descriptors / D=%d", D);
    if (!matches0 || !scores0) return imcui_set_err(h, IMCUI_ERR_ARG, "mutual_nn_dn: null output");
    if (nn_persistent(h, D))  // the fragment packer reads the [D, N] layout directly (lanes run along N: coalesced)
        return nn_forward(h, desc0_dn, 1, N, (long)N * D, desc1_dm, 1, M, (long)M * D, B, N, M, D, ratio_threshold, distance_threshold, do_mutual_check, matches0,
                          scores0, ws, ws_bytes, stream);
    WsAlloc a(ws, ws_bytes);
    float* t0 = a.get<float>((size_t)B * N * D);
    float* t1 = a.get<float>((size_t)B * M * D);
    if (!ws || !a.ok) return imcui_set_err(h, IMCUI_ERR_WS, "mutual_nn_dn: workspace too small (%zu bytes)", ws_bytes);
    hipLaunchKernelGGL(nn_transpose_kernel, dim3(cdiv(N, 32), cdiv(D, 32), B), dim3(256), 0, stream, desc0_dn, t0, D, N);
    hipLaunchKernelGGL(nn_transpose_kernel, dim3(cdiv(M, 32), cdiv(D, 32), B), dim3(256), 0, stream, desc1_dm, t1, D, M);
    IMCUI_CHECK_LAUNCH(h);
    return imcui_hip_mutual_nn(h, t0, t1, B, N, M, D, ratio_threshold, distance_threshold, do_mutual_check, matches0, scores0, (char*)ws + a.off, ws_bytes - a.off,
                               stream_);
}

// ------------------------------------------------------------------ nearest neighbour by dot product, fused (no similarity matrix)
// idx[q] = first arg-max over n of <queries[q], db[n]>  (what `cdistMatcher(dist='dot').query` / `bruteforce_reciprocal_nns` of
// upstream's mast3r/fast_nn.py returns for the queries -- the primitive `fast_reciprocal_NNs` iterates, imcui/hloc/matchers/
// mast3r.py:68-75; 65 536 queries against the 262 144 descriptors of a 512x512 map would be a 69 GB similarity matrix).
// Exact-f32 MFMA (v_mfma_f32_32x32x2_f32, an fmaf chain over the D components), db rows as the A operand and queries as B: a lane
// then owns ONE query column and 16 db rows per 32x32 tile, so the running (max, arg-max) is lane-local; rows are visited in
// increasing order with a strict compare = first maximum.  Workgroup = 256 queries (4 waves x 2 column tiles) x one range of db rows
// (64-row tiles through LDS, double-buffered); `nsplit` ranges per query block keep the chip full when few queries are left, a second
// kernel folds the per-range results in range order.
template <int D>
__global__ __launch_bounds__(256) void nn_argmax_kernel(const float* __restrict__ q, const float* __restrict__ db, int Q, int N, int chunk,
                                                        float* __restrict__ pbest, int* __restrict__ pidx) {
    constexpr int DH = D / 2, RS = D + 1;  // row stride in LDS (odd: conflict-free column reads)
    __shared__ float tile[2][64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.x * 256 + wid * 64;
    const int split = blockIdx.y;
    const int n0 = split * chunk, n1 = min(N, n0 + chunk);
    // B operand: query column lo of column tile c, components 2 s + hi
    float bq[2][DH];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* qr = q + (size_t)min(q0 + c * 32 + lo, Q - 1) * D + hi;
#pragma unroll
        for (int s = 0; s < DH; ++s) bq[c][s] = qr[2 * s];
    }
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {n0, n0};
    // staging: 64 rows x D floats per tile, D / 4 float4 per row
    constexpr int F4 = D / 4, NF4 = 64 * F4;
    auto stage = [&](int buf, int base) {
        for (int i = tid; i < NF4; i += 256) {
            const int r = i / F4, f = i - r * F4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + r < n1) v = *reinterpret_cast<const float4*>(db + (size_t)(base + r) * D + 4 * f);
            float* d = &tile[buf][r * RS + 4 * f];
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
    };
    if (n0 < n1) stage(0, n0);
    __syncthreads();
    int buf = 0;
    for (int base = n0; base < n1; base += 64, buf ^= 1) {
        if (base + 64 < n1) stage(buf ^ 1, base + 64);
        f32x16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][c][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < DH; ++s) {
            const float a0 = tile[buf][lo * RS + 2 * s + hi], a1 = tile[buf][(32 + lo) * RS + 2 * s + hi];
            acc[0][0] = mfma32(a0, bq[0][s], acc[0][0]);
            acc[0][1] = mfma32(a0, bq[1][s], acc[0][1]);
            acc[1][0] = mfma32(a1, bq[0][s], acc[1][0]);
            acc[1][1] = mfma32(a1, bq[1][s], acc[1][1]);
        }
        // a tile rarely holds a new maximum once a few tiles have been seen: compare the tile's maximum first (a v_max3 tree, 8
        // instructions per column) and walk the 32 candidates of a column, in increasing row order with a strict compare (= first
        // maximum), only where it beats the running one.  Rows past the range are zero in LDS: they can only trigger the walk.
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float tmax = acc[0][c][0];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, acc[rt][c][r]);
            if (tmax > best[c]) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = base + rt * 32 + frag_row(r, hi);  // increasing in (rt, r) for a lane
                        const float v = acc[rt][c][r];
                        if (row < n1 && v > best[c]) {
                            best[c] = v;
                            bidx[c] = row;
                        }
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // the two lanes of a query column (hi = 0 / 1) hold disjoint row sets: larger value wins, equal values -> smaller row
        const float ov = __shfl_xor(best[c], 32, 64);
        const int oi = __shfl_xor(bidx[c], 32, 64);
        if (ov > best[c] || (ov == best[c] && oi < bidx[c])) {
            best[c] = ov;
            bidx[c] = oi;
        }
        const int qi = q0 + c * 32 + lo;
        if (hi == 0 && qi < Q) {
            pbest[(size_t)split * Q + qi] = best[c];
            pidx[(size_t)split * Q + qi] = bidx[c];
        }
    }
}

__global__ __launch_bounds__(256) void nn_argmax_fold_kernel(const float* __restrict__ pbest, const int* __restrict__ pidx, int Q, int nsplit,
                                                             int* __restrict__ idx, float* __restrict__ best) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    float b = pbest[i];
    int j = pidx[i];
    for (int s = 1; s < nsplit; ++s) {  // ranges in increasing row order, strict compare: the first maximum survives
        const float v = pbest[(size_t)s * Q + i];
        if (v > b) {
            b = v;
            j = pidx[(size_t)s * Q + i];
        }
    }
    idx[i] = j;
    if (best) best[i] = b;
}

static int nn_argmax_nsplit(int Q, int N) {
    const int qb = (Q + 255) / 256;
    int ns = (1024 + qb - 1) / qb;  // about four workgroups per CU
    const int maxs = (N + 1023) / 1024;  // at least 16 tiles per range
    if (ns > maxs) ns = maxs;
    if (ns > 256) ns = 256;
    return ns < 1 ? 1 : ns;
}
extern "C" size_t imcui_hip_nn_argmax_workspace_bytes(int Q, int N) {
    if (Q <= 0 || N <= 0) return 256;
    return (size_t)nn_argmax_nsplit(Q, N) * Q * 8 + 512;
}
extern "C" int imcui_hip_nn_argmax_f32(imcui_hip_t* h, const float* queries, const float* db, int Q, int N, int D, int* idx, float* best, void* ws,
                                       size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (Q <= 0) return IMCUI_OK;
    if (N <= 0 || !queries || !db || !idx) return imcui_set_err(h, IMCUI_ERR_ARG, "nn_argmax: empty data base or null argument");
    if (D != 16 && D != 24 && D != 32) return imcui_set_err(h, IMCUI_ERR_ARG, "nn_argmax: D=%d (supported: 16, 24, 32)", D);
    const int ns = nn_argmax_nsplit(Q, N);
    const size_t need = (size_t)ns * Q * 8 + 512;
    if (!ws || ws_bytes < need) return imcui_set_err(h, IMCUI_ERR_WS, "nn_argmax: workspace too small (%zu < %zu)", ws_bytes, need);
    float* pbest = (float*)ws;
    int* pidx = (int*)((char*)ws + align_up((size_t)ns * Q * 4, 256));
    const int chunk = (((N + ns - 1) / ns) + 63) / 64 * 64;
    const dim3 grid((Q + 255) / 256, ns);
    if (D == 16)
        hipLaunchKernelGGL(nn_argmax_kernel<16>, grid, dim3(256), 0, stream, queries, db, Q, N, chunk, pbest, pidx);
    else if (D == 24)
        hipLaunchKernelGGL(nn_argmax_kernel<24>, grid, dim3(256), 0, stream, queries, db, Q, N, chunk, pbest, pidx);
    else
        hipLaunchKernelGGL(nn_argmax_kernel<32>, grid, dim3(256), 0, stream, queries, db, Q, N, chunk, pbest, pidx);
    hipLaunchKernelGGL(nn_argmax_fold_kernel, dim3((Q + 255) / 256), dim3(256), 0, stream, pbest, pidx, Q, ns, idx, best);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ the same search in the 3 x f16 split arithmetic (opt-in)
// Rows are scaled by 2^8 (keeps the low parts of unit-norm descriptors out of the f16 subnormal range; a common factor does not
// move an arg-max), split once into f16 hi / lo planes padded to 32 components ([rows][32] halves per plane), and a dot product is
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16: a quarter of the matrix cycles of the exact-f32 instruction.  The results
// are fp32-grade (dropped lo.lo term 2^-22 relative), NOT the fmaf chain of the f32 instruction: an arg-max between candidates
// closer than ~3e-7 may differ from the exact-f32 kernel's.  Workgroup = 512 queries (4 waves x 4 column tiles) x one row range.
__global__ __launch_bounds__(256) void nn_split_rows_kernel(const float* __restrict__ src, int N, int D, unsigned short* __restrict__ hi,
                                                            unsigned short* __restrict__ lo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (row, chunk of 8 components)
    if (i >= (long)N * 4) return;
    const long row = i >> 2;
    const int c0 = (int)(i & 3) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c0 + j < D) ? src[row * D + c0 + j] * 256.0f : 0.0f;
    uint4 h, l;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, l);
    *reinterpret_cast<uint4*>(hi + row * 32 + c0) = h;
    *reinterpret_cast<uint4*>(lo + row * 32 + c0) = l;
}

// Main pass: the running maximum of every query and the 64-ROW TILE that first reached it (per tile and column group: eight
// v_max3 over the 32 accumulators + compare + two selects, no branch).  Finding the row inside the tile while scanning costs a
// 32-way compare / select ladder whenever ANY of a wave's 64 lanes improves -- 38 % of the tiles of a 32 768-row range -- and
// matrix and vector instructions do not overlap on a SIMD: round 2's kernel ran the data base of a 512 x 512 image against 65 536
// queries in 4.0 ms at 0.35 of the matrix pipe.  The row is recovered afterwards by nn_argmax_fixup_kernel, which multiplies the
// winning tile of every query again with the same instruction sequence (bitwise the same values) and takes the first row that
// equals the maximum: 12 more MFMAs per query against 98 304 in the scan.
__global__ __launch_bounds__(256, 2) void nn_argmax_split_kernel(const unsigned short* __restrict__ qh, const unsigned short* __restrict__ ql,
                                                                const unsigned short* __restrict__ dh, const unsigned short* __restrict__ dl, int Q,
                                                                int N, int chunk, float* __restrict__ pbest, int* __restrict__ ptile) {
    constexpr int RSB = 80;  // bytes per staged row (64 + 16 padding: 16-lane groups of a ds_read_b128 cover all banks once)
    __shared__ uint4 tile4[2][2][64 * RSB / 16];  // [buffer][plane][row]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.x * 512 + wid * 128;
    const int split = blockIdx.y;
    const int n0 = split * chunk, n1 = min(N, n0 + chunk);
    // B operand: query column lo of column tile c: 8 halves k = 16 ks + 8 hi .. + 7 of either plane
    uint4 bh[4][2], bl[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t qi = (size_t)min(q0 + c * 32 + lo, Q - 1) * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bh[c][ks] = *reinterpret_cast<const uint4*>(qh + qi + ks * 16 + hi * 8);
            bl[c][ks] = *reinterpret_cast<const uint4*>(ql + qi + ks * 16 + hi * 8);
        }
    }
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int btile[4] = {n0, n0, n0, n0};
    // staging: 64 rows x 64 bytes per plane = 256 uint4 per plane: one per thread and plane.  The rows of tile t + 2 are requested
    // before the MFMAs of tile t and written to LDS after the MFMAs of tile t + 1 (two register sets, loop unrolled by two): a load
    // whose LDS write sits in front of the tile's own fragment reads exposed the whole L2 latency once per tile (2.9 ms per
    // 65 536 x 262 144 search; the compiler keeps LDS stores and loads in program order).
    const int sr = tid >> 2, sg = tid & 3;
    auto gload = [&](int base, uint4& vh, uint4& vl) __attribute__((always_inline)) {
        vh = make_uint4(0u, 0u, 0u, 0u);
        vl = vh;
        if (base + sr < n1) {
            vh = *reinterpret_cast<const uint4*>(dh + (size_t)(base + sr) * 32 + sg * 8);
            vl = *reinterpret_cast<const uint4*>(dl + (size_t)(base + sr) * 32 + sg * 8);
        }
    };
    auto lwrite = [&](int buf, const uint4& vh, const uint4& vl) __attribute__((always_inline)) {
        tile4[buf][0][(sr * RSB + sg * 16) / 16] = vh;
        tile4[buf][1][(sr * RSB + sg * 16) / 16] = vl;
    };
    auto compute = [&](int base, int buf) __attribute__((always_inline)) {
        f32x16 acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][c][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int o = ((rt * 32 + lo) * RSB + ks * 32 + hi * 16) / 16;
                const uint4 ah = tile4[buf][0][o], al = tile4[buf][1][o];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[rt][c] = mfma16(ah, bl[c][ks], acc[rt][c]);
                    acc[rt][c] = mfma16(al, bh[c][ks], acc[rt][c]);
                    acc[rt][c] = mfma16(ah, bh[c][ks], acc[rt][c]);
                }
            }
        if (base + 64 > n1) {  // the last tile of the range: rows past its end were staged as zeros and must not win
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (base + rt * 32 + frag_row(r, hi) >= n1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[rt][c][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float m[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x16& a = acc[rt][c];
                const float m0 = fmaxf(fmaxf(a[0], a[1]), a[2]), m1 = fmaxf(fmaxf(a[3], a[4]), a[5]), m2 = fmaxf(fmaxf(a[6], a[7]), a[8]);
                const float m3 = fmaxf(fmaxf(a[9], a[10]), a[11]), m4 = fmaxf(fmaxf(a[12], a[13]), a[14]);
                m[rt] = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), a[15]));
            }
            const float tmax = fmaxf(m[0], m[1]);
            const bool up = tmax > best[c];  // strict: the first tile that reaches the maximum keeps it
            best[c] = up ? tmax : best[c];
            btile[c] = up ? base : btile[c];
        }
    };
    if (n0 < n1) {
        uint4 rah, ral, rbh, rbl;
        gload(n0, rah, ral);
        lwrite(0, rah, ral);
        gload(n0 + 64, rah, ral);  // tile 1 waits in registers
        __syncthreads();
        for (int base = n0; base < n1; base += 128) {
            gload(base + 128, rbh, rbl);
            compute(base, 0);
            lwrite(1, rah, ral);
            __syncthreads();
            if (base + 64 >= n1) break;
            gload(base + 192, rah, ral);
            compute(base + 64, 1);
            lwrite(0, rbh, rbl);
            __syncthreads();
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float ov = __shfl_xor(best[c], 32, 64);
        const int ot = __shfl_xor(btile[c], 32, 64);
        if (ov > best[c] || (ov == best[c] && ot < btile[c])) {
            best[c] = ov;
            btile[c] = ot;
        }
        const int qi = q0 + c * 32 + lo;
        if (hi == 0 && qi < Q) {
            pbest[(size_t)split * Q + qi] = best[c];  // raw (both operands carry 2^8): the fix-up pass compares against it bit for bit
            ptile[(size_t)split * Q + qi] = btile[c];
        }
    }
}

// idx[q] holds the first row of the 64-row tile that reached best[q] (fold of the ranges: largest value, first tile): find the first
// row of that tile whose product equals best[q].  One wave = 32 queries; for query j the tile is multiplied against all 32
// queries of the wave with the instruction sequence of the scan and lanes lo == j read their column.  best is rescaled on the way out.
__global__ __launch_bounds__(256) void nn_argmax_fixup_kernel(const unsigned short* __restrict__ qh, const unsigned short* __restrict__ ql,
                                                              const unsigned short* __restrict__ dh, const unsigned short* __restrict__ dl, int Q,
                                                              int N, int* __restrict__ idx, const float* __restrict__ best_raw, float* __restrict__ best_out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wid) * 32;
    if (q0 >= Q) return;
    const int q = min(q0 + lo, Q - 1);
    uint4 bh[2], bl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bh[ks] = *reinterpret_cast<const uint4*>(qh + (size_t)q * 32 + ks * 16 + hi * 8);
        bl[ks] = *reinterpret_cast<const uint4*>(ql + (size_t)q * 32 + ks * 16 + hi * 8);
    }
    const int mytile = idx[q];
    const float mybest = best_raw[q];
    int found = 0x7fffffff;
    for (int j = 0; j < 32 && q0 + j < Q; ++j) {
        const int T = __builtin_amdgcn_readlane(mytile, j);
        f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
            const size_t row = (size_t)min(T + rt * 32 + lo, N - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 ah = *reinterpret_cast<const uint4*>(dh + row * 32 + ks * 16 + hi * 8);
                const uint4 al = *reinterpret_cast<const uint4*>(dl + row * 32 + ks * 16 + hi * 8);
                acc[rt] = mfma16(ah, bl[ks], acc[rt]);
                acc[rt] = mfma16(al, bh[ks], acc[rt]);
                acc[rt] = mfma16(ah, bh[ks], acc[rt]);
            }
        }
        if (lo == j) {
#pragma unroll
            for (int rt = 1; rt >= 0; --rt)
#pragma unroll
                for (int r = 15; r >= 0; --r) {
                    const int row = T + rt * 32 + frag_row(r, hi);
                    if (acc[rt][r] == mybest && row < N) found = row;  // descending scan: the smallest row of this lane's half survives
                }
        }
    }
    const int other = __shfl_xor(found, 32, 64);
    found = min(found, other);
    if (hi == 0 && q0 + lo < Q) {
        idx[q] = found != 0x7fffffff ? found : mytile;  // (a maximum that no row reproduces cannot happen: same data, same instructions)
        if (best_out) best_out[q] = mybest * (1.0f / 65536.0f);
    }
}

static int nn_argmax_split_nsplit(int Q, int N) {
    const int qb = (Q + 511) / 512;
    int ns = (1024 + qb - 1) / qb;
    const int maxs = (N + 1023) / 1024;
    if (ns > maxs) ns = maxs;
    if (ns > 256) ns = 256;
    return ns < 1 ? 1 : ns;
}
// workspace: partial results + the f16 planes of both operands
extern "C" size_t imcui_hip_nn_argmax_split_workspace_bytes(int Q, int N) {
    if (Q <= 0 || N <= 0) return 256;
    return (size_t)nn_argmax_split_nsplit(Q, N) * Q * 8 + (size_t)Q * 4 + ((size_t)Q + N) * 128 + 2560;
}
extern "C" int imcui_hip_nn_argmax_split_f32(imcui_hip_t* h, const float* queries, const float* db, int Q, int N, int D, int* idx, float* best,
                                             void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (Q <= 0) return IMCUI_OK;
    if (N <= 0 || !queries || !db || !idx) return imcui_set_err(h, IMCUI_ERR_ARG, "nn_argmax_split: empty data base or null argument");
    if (D <= 0 || D > 32) return imcui_set_err(h, IMCUI_ERR_ARG, "nn_argmax_split: D=%d (1..32)", D);
    const int ns = nn_argmax_split_nsplit(Q, N);
    WsAlloc a(ws, ws_bytes);
    float* pbest = a.get<float>((size_t)ns * Q);
    int* pidx = a.get<int>((size_t)ns * Q);
    float* fbest = a.get<float>((size_t)Q);  // raw maximum of every query after the fold of the ranges
    unsigned short* qh = a.get<unsigned short>((size_t)Q * 32);
    unsigned short* ql = a.get<unsigned short>((size_t)Q * 32);
    unsigned short* dh = a.get<unsigned short>((size_t)N * 32);
    unsigned short* dl = a.get<unsigned short>((size_t)N * 32);
    if (!ws || !a.ok) return imcui_set_err(h, IMCUI_ERR_WS, "nn_argmax_split: workspace too small (%zu < %zu)", ws_bytes, a.off);
    hipLaunchKernelGGL(nn_split_rows_kernel, dim3((unsigned)(((long)Q * 4 + 255) / 256)), dim3(256), 0, stream, queries, Q, D, qh, ql);
    hipLaunchKernelGGL(nn_split_rows_kernel, dim3((unsigned)(((long)N * 4 + 255) / 256)), dim3(256), 0, stream, db, N, D, dh, dl);
    const int chunk = (((N + ns - 1) / ns) + 63) / 64 * 64;
    hipLaunchKernelGGL(nn_argmax_split_kernel, dim3((Q + 511) / 512, ns), dim3(256), 0, stream, qh, ql, dh, dl, Q, N, chunk, pbest, pidx);
    // fold of the ranges (largest value, first tile), then the row inside the winning tile
    hipLaunchKernelGGL(nn_argmax_fold_kernel, dim3((Q + 255) / 256), dim3(256), 0, stream, pbest, pidx, Q, ns, idx, fbest);
    hipLaunchKernelGGL(nn_argmax_fixup_kernel, dim3((Q + 127) / 128), dim3(256), 0, stream, qh, ql, dh, dl, Q, N, idx, fbest, best);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
