// LightGlue assignment: the merge kernels around the two launches of the similarity-and-reduce kernel (simred.hip).
//
// Reference (upstream lightglue.py, restated in oracle/lightglue.py:50-79):
//   scores = log_softmax(sim, 2) + log_softmax(sim, 1) + logsigmoid(z0)[:, :, None] + logsigmoid(z1)[:, None, :]
//   m0 = scores.max(2), m1 = scores.max(1), mutual check, exp, threshold (filter_matches)
// Rounds 1-4 materialised sim [B][R][R] (16.8 MB per pair) and read it back -- four times, then twice (lg_stats2_kernel / lg_best2_kernel,
// or the statistics from the GEMM's epilogue, EPI_SIMSTAT): 3.2 GB of HBM traffic per 64-pair step against 0.27 GB of descriptors.
// Round 5: sim is never stored.  simred.hip computes every 128 x 128 tile twice -- SR_LSE: soft-max statistics of both directions (a
// row's (max, sum) lives in registers across the column tiles; a column's per 128-row block), SR_LGBEST: the log assignment once per
// element with the reference's association, row best (first column) and column best (first row) -- and the two kernels below fold the
// per-chunk / per-block partials in ascending order, so results do not depend on scheduling.
#pragma once
#include "common.h"

// merge the partials in ascending order: rows (blockIdx.z = 0) over the column partials that exist for n1 columns
// (granularity rgran), columns (blockIdx.z = 1) over the row partials that exist for n0 rows (granularity cgran; `ctile`: the tile height the slots come in).
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
