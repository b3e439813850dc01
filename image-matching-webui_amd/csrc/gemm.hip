// f32 MFMA GEMM (v_mfma_f32_32x32x2_f32) for MI355X / gfx950.
//
// Tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 fragments of 32x32 (64 accumulator
// registers).  Operands are staged global -> registers -> LDS in a [k-quad][row][4] image
// (row stride padded to 129 float4) so that both the 16-byte staging writes and the 16-byte
// fragment reads are bank-conflict free; the next K tile is prefetched into registers while the
// current one is being multiplied.  The f32 matrix pipe needs only 4 floats per lane per
// 4 x 64-cycle MFMAs, so this simple structure is matrix-bound.
#include "gemm.h"

#define BM 128
#define BN 128
#define BK 32
#define LDS_ROWS 129  // padded row count per k-quad (float4 units)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    __shared__ float4 As[(BK / 4) * LDS_ROWS];
    __shared__ float4 Bs[(BK / 4) * LDS_ROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;

    const int z = blockIdx.z;
    int M = p.M, N = p.N;
    if (p.mcnt) M = p.mcnt[z * p.cnt_stride];
    if (p.ncnt) N = p.ncnt[z * p.cnt_stride];
    const int ncol = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / ncol) * BM;
    const int col0 = (tile % ncol) * BN;
    if (row0 >= M || col0 >= N) return;

    const float* W = p.W + (size_t)z * p.w_bs;
    const float* bias = p.bias;
    int seq = 0;
    if (p.rows_per_seq > 0) {
        seq = row0 / p.rows_per_seq;
        const int i0 = row0 - seq * p.rows_per_seq;
        if (p.cnt && i0 >= p.cnt[seq]) return;
        if (p.active && p.active[seq >> 1] == 0) return;
        if (p.wsel) {
            const int sel = p.wsel[seq >> 1] + p.wsel_off;
            W += (size_t)sel * p.w_stride;
            if (bias) bias += (size_t)sel * p.b_stride;
        }
    }
    const float* A = p.A + (size_t)z * p.a_bs;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // staging: 128 rows x 8 k-quads per operand = 1024 float4, 4 per thread
    const int s_kq = tid & 7;
    const int s_r = tid >> 3;  // 0..31, +32 per iteration
    float4 ra[4], rb[4];
    const int nkt = p.K / BK;

    auto load_tile = [&](int kt) {
        const int k = kt * BK + s_kq * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = s_r + 32 * it;
            const int ar = min(row0 + r, M - 1);
            const int br = min(col0 + r, N - 1);
            const float* src;
            if (p.A2 != nullptr && k >= p.K1)
                src = p.A2 + (size_t)ar * p.lda2 + (k - p.K1);
            else
                src = A + (size_t)ar * p.lda + k;
            ra[it] = *reinterpret_cast<const float4*>(src);
            rb[it] = *reinterpret_cast<const float4*>(W + (size_t)br * p.ldw + k);
        }
    };

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            As[s_kq * LDS_ROWS + s_r + 32 * it] = ra[it];
            Bs[s_kq * LDS_ROWS + s_r + 32 * it] = rb[it];
        }
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            const int kq = 2 * t + hi;
            float4 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = As[kq * LDS_ROWS + wm * 64 + m * 32 + lo];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = Bs[kq * LDS_ROWS + wn * 64 + n * 32 + lo];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = mfma32(a[m].x, b[n].x, acc[m][n]);
                    acc[m][n] = mfma32(a[m].y, b[n].y, acc[m][n]);
                    acc[m][n] = mfma32(a[m].z, b[n].z, acc[m][n]);
                    acc[m][n] = mfma32(a[m].w, b[n].w, acc[m][n]);
                }
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    float* C = p.C ? p.C + (size_t)z * p.c_bs : nullptr;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = col0 + wn * 64 + n * 32 + lo;
        const bool colok = col < N;
        const float bv = (bias != nullptr && colok) ? bias[col] : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 64 + m * 32 + frag_row(r, hi);
                const bool ok = colok && row < M;
                float v = acc[m][n][r] + bv;
                if (EPI == EPI_BIAS) {
                    if (ok) C[(size_t)row * p.ldc + col] = v * p.alpha;
                } else if (EPI == EPI_RELU) {
                    if (ok) C[(size_t)row * p.ldc + col] = fmaxf(v, 0.0f);
                } else if (EPI == EPI_RESID) {
                    if (ok) C[(size_t)row * p.ldc + col] += v;
                } else if (EPI == EPI_QKV) {
                    // col = t*256 + head*64 + d (weights were de-interleaved at pack time)
                    const int t = col >> 8, hd = (col >> 6) & 3, d = col & 63;
                    const float partner = __shfl_xor(v, 1, 64);  // (d ^ 1) of the same row
                    const int i = row - seq * p.rows_per_seq;
                    if (t < 2) {
                        const float c = p.rope_cos[(size_t)row * 32 + (d >> 1)];
                        const float s = p.rope_sin[(size_t)row * 32 + (d >> 1)];
                        // x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0)
                        v = v * c + ((d & 1) ? partner : -partner) * s;
                        if (t == 0) v *= p.alpha;
                    }
                    float* dst = (t == 0) ? p.Q : (t == 1) ? p.Kt : p.V;
                    if (ok) dst[(((size_t)seq * p.heads + hd) * p.rows_per_seq + i) * 64 + d] = v;
                } else if (EPI == EPI_CROSS) {
                    const int t = col >> 8, hd = (col >> 6) & 3, d = col & 63;
                    const int i = row - seq * p.rows_per_seq;
                    if (t == 0) v *= p.alpha;
                    float* dst = (t == 0) ? p.Q : p.V;
                    if (ok) dst[(((size_t)seq * p.heads + hd) * p.rows_per_seq + i) * 64 + d] = v;
                }
            }
        }
    }
}

int gemm_launch(imcui_hip_s* h, const GemmP& p, hipStream_t stream) {
    if (p.K % BK != 0 || p.K <= 0) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K=%d must be a positive multiple of %d", p.K, BK);
    if (p.A2 && (p.K1 % BK != 0)) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K1=%d must be a multiple of %d", p.K1, BK);
    if (p.rows_per_seq > 0 && p.rows_per_seq % BM != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: rows_per_seq=%d must be a multiple of %d", p.rows_per_seq, BM);
    if (p.M <= 0 || p.N <= 0) return IMCUI_OK;
    const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    dim3 grid(ntiles, 1, p.batch), block(256);
    imcui_prof_begin(h, PROF_GEMM, stream);
    switch (p.epi) {
        case EPI_BIAS: hipLaunchKernelGGL(gemm_kernel<EPI_BIAS>, grid, block, 0, stream, p); break;
        case EPI_RELU: hipLaunchKernelGGL(gemm_kernel<EPI_RELU>, grid, block, 0, stream, p); break;
        case EPI_RESID: hipLaunchKernelGGL(gemm_kernel<EPI_RESID>, grid, block, 0, stream, p); break;
        case EPI_QKV: hipLaunchKernelGGL(gemm_kernel<EPI_QKV>, grid, block, 0, stream, p); break;
        case EPI_CROSS: hipLaunchKernelGGL(gemm_kernel<EPI_CROSS>, grid, block, 0, stream, p); break;
        default: return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: bad epilogue %d", p.epi);
    }
    imcui_prof_end(h, PROF_GEMM, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
