// LightGlue assignment on the materialised similarity sim [B][R][R] in TWO reads (round 2: four).
//
// Reference (upstream lightglue.py, restated in oracle/lightglue.py:50-79):
//   scores = log_softmax(sim, 2) + log_softmax(sim, 1) + logsigmoid(z0)[:, :, None] + logsigmoid(z1)[:, None, :]
//   m0 = scores.max(2), m1 = scores.max(1), mutual check, exp, threshold (filter_matches)
// Round 2 read the matrix four times after the GEMM had written it (row statistics, column statistics, row arg-max, column
// arg-max: 7.5 GB and 2.09 ms per 64-pair step incl. the GEMM).  Now: lg_stats2_kernel reads it ONCE for both sets of
// soft-max partials -- per row over 1024-column chunks, per column over 64-row bands -- a merge kernel folds the partials in
// ascending order, and ONE more pass (lg_best2_kernel) evaluates the log-assignment once per element and keeps, from that
// one value, the row maximum (first column attaining it) and the column maximum (first row attaining it): 2.1 GB and
// 1.29 ms per step.  Partials are merged in a fixed order, so results do not depend on scheduling.
// Alternative (IMCUI_LG_ASSIGN_STATS=epilogue): the similarity GEMM's own epilogue (EPI_SIMSTAT, gemm.hip) reduces every
// tile it stores to the same partial format, which leaves ONE read of the matrix (1.07 GB) -- and is slower: the GEMM is
// bound by instruction issue, the 128 exponentials per thread cost it 320 us where the HBM-bound pass takes 260 us.
#pragma once
#include "common.h"

#define LG2_ROWS 64     // rows per workgroup band (4 waves x 16 rows)
#define LG2_COLS 1024   // columns per workgroup chunk (a lane: 4 x float4 per row)

// log assignment score of (i, j) exactly as the reference associates it:
//   (log_softmax_row + log_softmax_col) + (logsigmoid(z0_i) + logsigmoid(z1_j))
__device__ __forceinline__ float lg_score(float s, float rm, float rl, float cm, float cl, float l0, float l1) {
    return (((s - rm) - rl) + ((s - cm) - cl)) + (l0 + l1);
}

// one read of sim: row partials (max, sum exp) per LG2_COLS chunk -> rpm / rps [B][nrp][R]; column partials per LG2_ROWS band
// -> cpm / cps [B][ncp][R]   (same format as the GEMM epilogue's, with coarser row partials)
__global__ __launch_bounds__(256) void lg_stats2_kernel(const float* __restrict__ sim, const int* __restrict__ cnt, int R, int nrp,
                                                        int ncp, float* __restrict__ rpm, float* __restrict__ rps,
                                                        float* __restrict__ cpm, float* __restrict__ cps) {
    __shared__ float lm[4][LG2_COLS], ls[4][LG2_COLS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = blockIdx.x, bd = blockIdx.y, b = blockIdx.z;
    const int n0 = cnt[2 * b], n1 = cnt[2 * b + 1];
    const int c0 = ch * LG2_COLS, i0 = bd * LG2_ROWS + wv * 16;
    if (c0 >= n1 || bd * LG2_ROWS >= n0) return;
    const float* base = sim + (size_t)b * R * R;
    float cm[16], cs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        cm[e] = -INFINITY;
        cs[e] = 0.0f;
    }
#pragma unroll 2
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + r;
        if (i >= n0) break;
        const float* row = base + (size_t)i * R;
        float x[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = c0 + k * 256 + lane * 4;
            float4 t = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (j < n1) t = *reinterpret_cast<const float4*>(row + j);  // R % 128 == 0: the vector stays inside the row
            x[4 * k] = t.x, x[4 * k + 1] = (j + 1 < n1) ? t.y : -INFINITY, x[4 * k + 2] = (j + 2 < n1) ? t.z : -INFINITY,
                  x[4 * k + 3] = (j + 3 < n1) ? t.w : -INFINITY;
        }
        float m = x[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) m = fmaxf(m, x[e]);
        m = wave_max(m);
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s += (x[e] > -INFINITY) ? __expf(x[e] - m) : 0.0f;
            const float mn = fmaxf(cm[e], x[e]);
            if (mn > -INFINITY) cs[e] = cs[e] * __expf(cm[e] - mn) + ((x[e] > -INFINITY) ? __expf(x[e] - mn) : 0.0f);
            cm[e] = mn;
        }
        s = wave_sum(s);
        if (lane == 0) {
            rpm[((size_t)b * nrp + ch) * R + i] = m;
            rps[((size_t)b * nrp + ch) * R + i] = s;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<float4*>(&lm[wv][k * 256 + lane * 4]) = make_float4(cm[4 * k], cm[4 * k + 1], cm[4 * k + 2], cm[4 * k + 3]);
        *reinterpret_cast<float4*>(&ls[wv][k * 256 + lane * 4]) = make_float4(cs[4 * k], cs[4 * k + 1], cs[4 * k + 2], cs[4 * k + 3]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LG2_COLS; c += 256) {
        const int j = c0 + c;
        if (j >= n1) continue;
        const float m = fmaxf(fmaxf(lm[0][c], lm[1][c]), fmaxf(lm[2][c], lm[3][c]));
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (lm[w][c] > -INFINITY) s += ls[w][c] * __expf(lm[w][c] - m);
        cpm[((size_t)b * ncp + bd) * R + j] = m;
        cps[((size_t)b * ncp + bd) * R + j] = s;
    }
}

// merge the partials in ascending order: rows (blockIdx.z = 0) over the column partials that exist for n1 columns
// (granularity rgran), columns (blockIdx.z = 1) over the row partials that exist for n0 rows (granularity cgran; the GEMM
// epilogue writes both halves of every 128-row tile it computes, hence `ctile`: the tile height the slots come in).
// out: max and log(sum exp(x - max))
__global__ __launch_bounds__(256) void lg_stat_merge_kernel(const float* __restrict__ rpm, const float* __restrict__ rps,
                                                            const float* __restrict__ cpm, const float* __restrict__ cps,
                                                            const int* __restrict__ cnt, int R, int nrp, int ncp, int rgran,
                                                            int cgran, int ctile, float* __restrict__ rmax, float* __restrict__ rls,
                                                            float* __restrict__ cmax, float* __restrict__ cls) {
    const int b = blockIdx.y, cols = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n0 = cnt[2 * b], n1 = cnt[2 * b + 1];
    if (t >= (cols ? n1 : n0) || n0 <= 0 || n1 <= 0) return;
    const float* pm = cols ? cpm : rpm;
    const float* ps = cols ? cps : rps;
    const int np_all = cols ? ncp : nrp;
    // partial slots that were written
    const int np = cols ? ((n0 + ctile - 1) / ctile) * (ctile / cgran) : (n1 + rgran - 1) / rgran;
    float m = -INFINITY;
    for (int q = 0; q < np; ++q) m = fmaxf(m, pm[((size_t)b * np_all + q) * R + t]);
    float s = 0.0f;
    for (int q = 0; q < np; ++q) {
        const float v = pm[((size_t)b * np_all + q) * R + t];
        if (v > -INFINITY) s += ps[((size_t)b * np_all + q) * R + t] * __expf(v - m);
    }
    (cols ? cmax : rmax)[(size_t)b * R + t] = m;
    (cols ? cls : rls)[(size_t)b * R + t] = logf(s);
}

// one read of sim: score of every (i, j) once; row best (value, first column) per LG2_COLS chunk -> rpv / rpj [B][nch][R];
// column best (value, first row) per LG2_ROWS band -> cpv / cpi [B][nbd][R]
__global__ __launch_bounds__(256) void lg_best2_kernel(const float* __restrict__ sim, const int* __restrict__ cnt, int R, int nch,
                                                       int nbd, const float* __restrict__ rmax, const float* __restrict__ rls,
                                                       const float* __restrict__ cmax, const float* __restrict__ cls,
                                                       const float* __restrict__ ls, float* __restrict__ rpv, int* __restrict__ rpj,
                                                       float* __restrict__ cpv, int* __restrict__ cpi) {
    __shared__ float lb[4][LG2_COLS];
    __shared__ int li[4][LG2_COLS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = blockIdx.x, bd = blockIdx.y, b = blockIdx.z;
    const int n0 = cnt[2 * b], n1 = cnt[2 * b + 1];
    const int c0 = ch * LG2_COLS, i0 = bd * LG2_ROWS + wv * 16;
    if (c0 >= n1 || bd * LG2_ROWS >= n0) return;
    const float* base = sim + (size_t)b * R * R;
    const float* l0p = ls + ((size_t)2 * b) * R;
    const float* l1p = ls + ((size_t)2 * b + 1) * R;
    float cmx[16], clx[16], l1x[16], cbv[16];
    int cbi[16];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = min(c0 + k * 256 + lane * 4 + e, n1 - 1);
            cmx[4 * k + e] = cmax[(size_t)b * R + j];
            clx[4 * k + e] = cls[(size_t)b * R + j];
            l1x[4 * k + e] = l1p[j];
            cbv[4 * k + e] = -INFINITY;
            cbi[4 * k + e] = 0x7fffffff;
        }
#pragma unroll 2
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + r;
        if (i >= n0) break;
        const float* row = base + (size_t)i * R;
        const float rm = rmax[(size_t)b * R + i], rl = rls[(size_t)b * R + i], l0 = l0p[i];
        float x[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = c0 + k * 256 + lane * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < n1) t = *reinterpret_cast<const float4*>(row + j);
            x[4 * k] = t.x, x[4 * k + 1] = t.y, x[4 * k + 2] = t.z, x[4 * k + 3] = t.w;
        }
        float bv = -INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = c0 + k * 256 + lane * 4 + e;
                // columns past the pair's edge score -inf: they never win a strict comparison (branch-free selects)
                const float v = (j < n1) ? lg_score(x[4 * k + e], rm, rl, cmx[4 * k + e], clx[4 * k + e], l0, l1x[4 * k + e]) : -INFINITY;
                const bool rw = v > bv;  // j ascends within a lane: the first maximum is kept
                bv = rw ? v : bv;
                bj = rw ? j : bj;
                const bool cw = v > cbv[4 * k + e];  // i ascends within a wave
                cbv[4 * k + e] = cw ? v : cbv[4 * k + e];
                cbi[4 * k + e] = cw ? i : cbi[4 * k + e];
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ov > bv || (ov == bv && oj < bj)) {
                bv = ov;
                bj = oj;
            }
        }
        if (lane == 0) {
            rpv[((size_t)b * nch + ch) * R + i] = bv;
            rpj[((size_t)b * nch + ch) * R + i] = bj;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<float4*>(&lb[wv][k * 256 + lane * 4]) = make_float4(cbv[4 * k], cbv[4 * k + 1], cbv[4 * k + 2], cbv[4 * k + 3]);
        *reinterpret_cast<int4*>(&li[wv][k * 256 + lane * 4]) = make_int4(cbi[4 * k], cbi[4 * k + 1], cbi[4 * k + 2], cbi[4 * k + 3]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LG2_COLS; c += 256) {
        const int j = c0 + c;
        if (j >= n1) continue;
        float v = lb[0][c];
        int i = li[0][c];
#pragma unroll
        for (int w = 1; w < 4; ++w)  // wave w holds rows above wave w - 1's: a later wave wins only with a strictly larger value
            if (lb[w][c] > v) {
                v = lb[w][c];
                i = li[w][c];
            }
        cpv[((size_t)b * nbd + bd) * R + j] = v;
        cpi[((size_t)b * nbd + bd) * R + j] = i;
    }
}

// rows (blockIdx.z = 0): best over the column chunks -> max0, m0; columns (1): best over the row bands -> m1.  Ascending
// order, a later partial wins only with a strictly larger value = the first index attaining the maximum (torch.max)
// rgran / cgran: columns per row-partial slot / rows per column-partial slot (round 5: the slots of simred.hip are chunks of column tiles and
// 128-row blocks)
__global__ __launch_bounds__(256) void lg_best_merge_kernel(const float* __restrict__ rpv, const int* __restrict__ rpj,
                                                            const float* __restrict__ cpv, const int* __restrict__ cpi,
                                                            const int* __restrict__ cnt, int R, int nch, int nbd, int rgran, int cgran,
                                                            float* __restrict__ max0, int* __restrict__ m0, int* __restrict__ m1) {
    const int b = blockIdx.y, cols = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n0 = cnt[2 * b], n1 = cnt[2 * b + 1];
    if (t >= (cols ? n1 : n0) || n0 <= 0 || n1 <= 0) return;
    if (!cols) {
        const int np = (n1 + rgran - 1) / rgran;
        float bv = -INFINITY;
        int bj = 0x7fffffff;
        for (int q = 0; q < np; ++q) {
            const float v = rpv[((size_t)b * nch + q) * R + t];
            if (v > bv) {
                bv = v;
                bj = rpj[((size_t)b * nch + q) * R + t];
            }
        }
        max0[(size_t)b * R + t] = bv;
        m0[(size_t)b * R + t] = bj < n1 ? bj : 0;  // only a row of NaN scores leaves no maximum: keep the index inside the pair
    } else {
        const int np = (n0 + cgran - 1) / cgran;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = 0; q < np; ++q) {
            const float v = cpv[((size_t)b * nbd + q) * R + t];
            if (v > bv) {
                bv = v;
                bi = cpi[((size_t)b * nbd + q) * R + t];
            }
        }
        m1[(size_t)b * R + t] = bi < n0 ? bi : 0;
    }
}
