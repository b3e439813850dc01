// Tile bookkeeping shared by the MFMA GEMM kernels (gemm.hip, gemm_wreg.hip): which rows / columns a workgroup owns,
// ragged-sequence and per-pair-weight selection.
#pragma once
#include "gemm.h"

#define BM 128
#define BN 128

struct TileCtx {
    int M, N, row0, col0, seq, z;
    const float* bias;
    long wsel;  // selected weight index (per-pair heads)
};

// returns false when the workgroup has nothing to do
__device__ __forceinline__ bool gemm_tile_setup(const GemmP& p, TileCtx& c, int bn = BN, int bm = BM) {
    c.z = blockIdx.z;
    c.M = p.mcnt ? p.mcnt[c.z * p.cnt_stride] : p.M;
    c.N = p.ncnt ? p.ncnt[c.z * p.cnt_stride] : p.N;
    const int ncol = (p.N + bn - 1) / bn;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (p.group_rows > 1) {
        // both operands are large (similarity matrices): walk the tiles group by group of `group_rows` row panels, column
        // panel by column panel inside a group, so a weight-side panel is re-used by the whole group while it is in L2 and
        // the group's row panels stay resident across the columns (row-major order streams the whole second operand once
        // per row panel: 2.1 GB per 16384 x 16384 x 256 product instead of 0.27 GB)
        const int nrow = (p.M + BM - 1) / BM;
        const int per = p.group_rows * ncol;
        const int g = tile / per, t = tile - g * per;
        const int rows = min(p.group_rows, nrow - g * p.group_rows);
        c.row0 = (g * p.group_rows + t % rows) * BM;
        c.col0 = (t / rows) * bn;
    } else {
        c.row0 = (tile / ncol) * bm;  // (only this branch takes a row tile other than BM: gemm_wreg's small-batch tiles)
        c.col0 = (tile % ncol) * bn;
    }
    if (c.row0 >= c.M || c.col0 >= c.N) return false;
    c.bias = p.bias;
    c.seq = 0;
    c.wsel = 0;
    if (p.rows_per_seq > 0) {
        c.seq = c.row0 / p.rows_per_seq;
        const int i0 = c.row0 - c.seq * p.rows_per_seq;
        if (p.cnt && i0 >= p.cnt[c.seq]) return false;
        if (p.active && p.active[c.seq >> 1] == 0) return false;
        if (p.wsel) {
            c.wsel = p.wsel[c.seq >> 1] + p.wsel_off;
            if (c.bias) c.bias += (size_t)c.wsel * p.b_stride;
        }
    }
    return true;
}
