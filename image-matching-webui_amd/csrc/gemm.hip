// MFMA GEMM for MI355X / gfx950 in two arithmetic modes.
//
// f32 mode: tile 128x128x32, v_mfma_f32_32x32x2_f32, operands staged global -> registers -> LDS
// as [k-quad][row][4 floats] (row stride padded to 129 granules: conflict-free).
// Split mode (default): every f32 value is an f16 (hi, lo) pair and each 32x32x16 step issues
// ah*bh + ah*bl + al*bh; see gemm_split_kernel below.  In both modes 4 waves (2x2) each own a 64x64
// output tile (64 accumulators) and the products are issued with the weight fragment as the MFMA A
// operand (C^T accumulation, see gemm_epilogue).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gemm.h"
#include "gemm_tile.h"

#define LDS_ROWS 129  // padded row count per k-granule (16-byte units)

// Epilogue.  The products are issued with the WEIGHT fragment as the MFMA A operand and the
// activation fragment as B, i.e. the accumulators hold C^T: column (lane & 31) = token row,
// fragment row = output feature.  A lane therefore owns 4 CONSECUTIVE features of one token per
// register quad (r & 3), which makes every store a 16-byte store (a 64-dword-store epilogue is
// store-issue bound and was 10x the matrix time), puts RoPE pairs in one lane, and lets V^T be
// written with lanes running along the token axis.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const TileCtx& c, f32x16 (&acc)[2][2], float wsc, int wm,
                                              int wn, int lo, int hi) {
    float* C = p.C ? p.C + (size_t)c.z * p.c_bs : nullptr;
    const bool vec_ok = ((p.ldc & 3) == 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = c.row0 + wm * 64 + m * 32 + lo;  // token
        const bool rowok = row < c.M;
        const int i = row - c.seq * p.rows_per_seq;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = c.col0 + wn * 64 + n * 32 + 8 * q + 4 * hi;  // first of 4 features
                if (!rowok || f0 >= c.N) continue;
                float v[4] = {acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                              acc[m][n][4 * q + 3] * wsc};
                const bool full = (f0 + 3 < c.N);
                if (c.bias != nullptr) {
                    if (full) {
                        const float4 b4 = *reinterpret_cast<const float4*>(c.bias + f0);
                        v[0] += b4.x;
                        v[1] += b4.y;
                        v[2] += b4.z;
                        v[3] += b4.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (f0 + j < c.N) v[j] += c.bias[f0 + j];
                    }
                }
                if (EPI == EPI_BIAS || EPI == EPI_RELU || EPI == EPI_RESID || EPI == EPI_CONV) {
                    float* dst = C + (size_t)row * p.ldc + f0;
                    if (EPI == EPI_CONV) {
                        if (p.resid != nullptr) {
                            const float* rs = p.resid + (size_t)row * p.ldr + f0;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (f0 + j < c.N) v[j] += rs[j];
                        }
                        if (p.act == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                        } else if (p.act == 2) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.01f * v[j];
                        } else if (p.act == 3) {  // GELU with the erf of common.h (1.3e-7 absolute on erf; ATen: 0.5 x (1 + erf(x / sqrt 2)))
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = gelu_poly(v[j]);
                        }
                    } else if (EPI == EPI_BIAS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                    } else if (EPI == EPI_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                    }
                    if (full && vec_ok) {
                        if (EPI == EPI_RESID) {
                            const float4 o4 = *reinterpret_cast<const float4*>(dst);
                            v[0] += o4.x;
                            v[1] += o4.y;
                            v[2] += o4.z;
                            v[3] += o4.w;
                        }
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (f0 + j < c.N) dst[j] = (EPI == EPI_RESID) ? dst[j] + v[j] : v[j];
                    }
                } else {
                    // f0 = t*256 + head*64 + d0 (weights were de-interleaved at pack time); N % 4 == 0
                    const int t = f0 >> 8, hd = (f0 >> 6) & 3, d0 = f0 & 63;
                    float* dst;
                    bool vt;
                    if (EPI == EPI_QKV) {
                        if (t < 2) {
                            // x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0)
                            const float2 cs = *reinterpret_cast<const float2*>(p.rope_cos + (size_t)row * 32 + (d0 >> 1));
                            const float2 sn = *reinterpret_cast<const float2*>(p.rope_sin + (size_t)row * 32 + (d0 >> 1));
                            const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                            v[0] = a0 * cs.x + (-a1) * sn.x;
                            v[1] = a1 * cs.x + a0 * sn.x;
                            v[2] = a2 * cs.y + (-a3) * sn.y;
                            v[3] = a3 * cs.y + a2 * sn.y;
                            if (t == 0) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                            }
                        }
                        dst = (t == 0) ? p.Q : (t == 1) ? p.Kt : p.V;
                        vt = (t == 2);
                    } else {
                        if (t == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                        }
                        dst = (t == 0) ? p.Q : p.V;
                        vt = (t == 1);
                    }
                    if (p.split_out) {
                        // pre-split operands for attn_split_kernel: f16 hi plane, lo plane
                        unsigned h01, l01, h23, l23;
                        split2(v[0], v[1], h01, l01);
                        split2(v[2], v[3], h23, l23);
                        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
                        if (vt) {
                            unsigned short* o = d16 + (((size_t)c.seq * p.heads + hd) * 64 + d0) * p.rows_per_seq + i;
                            const size_t rs = p.rows_per_seq;
                            o[0] = (unsigned short)(h01 & 0xFFFFu);
                            o[rs] = (unsigned short)(h01 >> 16);
                            o[2 * rs] = (unsigned short)(h23 & 0xFFFFu);
                            o[3 * rs] = (unsigned short)(h23 >> 16);
                            o += p.plane_halves;
                            o[0] = (unsigned short)(l01 & 0xFFFFu);
                            o[rs] = (unsigned short)(l01 >> 16);
                            o[2 * rs] = (unsigned short)(l23 & 0xFFFFu);
                            o[3 * rs] = (unsigned short)(l23 >> 16);
                        } else {
                            unsigned short* o = d16 + (((size_t)c.seq * p.heads + hd) * p.rows_per_seq + i) * 64 + d0;
                            *reinterpret_cast<uint2*>(o) = make_uint2(h01, h23);
                            *reinterpret_cast<uint2*>(o + p.plane_halves) = make_uint2(l01, l23);
                        }
                    } else if (vt && p.v_transposed) {
                        // V^T [seq][head][64][rows]: lanes run along the token axis -> coalesced
                        float* o = dst + (((size_t)c.seq * p.heads + hd) * 64 + d0) * p.rows_per_seq + i;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[(size_t)j * p.rows_per_seq] = v[j];
                    } else {
                        *reinterpret_cast<float4*>(dst + (((size_t)c.seq * p.heads + hd) * p.rows_per_seq + i) * 64 + d0) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
    }
}

// Source of `n` consecutive K elements starting at k for GEMM row `ar` (nullptr = zeros):
// plain / concatenated matrix, or implicit im2col of an NHWC convolution.
__device__ __forceinline__ const float* gemm_a_src(const GemmP& p, const float* A, const float* A2, int ar, int k) {
    if (p.conv_k > 0) {
        const int tap = k / p.conv_cin, c0 = k - tap * p.conv_cin;
        const int ky = tap / p.conv_k, kx = tap - ky * p.conv_k;
        const int ox = ar % p.conv_wout;
        const int t = ar / p.conv_wout;
        const int oy = t % p.conv_hout, b = t / p.conv_hout;
        const int iy = oy * p.conv_stride - p.conv_pad + ky, ix = ox * p.conv_stride - p.conv_pad + kx;
        if (iy < 0 || iy >= p.conv_hin || ix < 0 || ix >= p.conv_win) return nullptr;
        return A + (((size_t)b * p.conv_hin + iy) * p.conv_win + ix) * p.conv_cin + c0;
    }
    if (A2 != nullptr && k >= p.K1) return A2 + (size_t)ar * p.lda2 + (k - p.K1);
    return A + (size_t)ar * p.lda + k;
}

// ------------------------------------------------------------------ exact f32 kernel
#define BK32 32
template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    __shared__ float4 As[(BK32 / 4) * LDS_ROWS];
    __shared__ float4 Bs[(BK32 / 4) * LDS_ROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    TileCtx c;
    if (!gemm_tile_setup(p, c)) return;
    const float* W = p.W + (size_t)c.z * p.w_bs + (size_t)c.wsel * p.w_stride;
    const float* A = p.A + (size_t)c.z * p.a_bs;
    const float* A2 = p.A2 ? p.A2 + (size_t)c.z * p.a2_bs : nullptr;

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
    // named registers, not arrays: a register array that is only copied global -> LDS is kept in
    // scratch by the compiler (it is turned into a private-memory copy)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int nkt = p.K / BK32;
#define FOR4(X) X(0) X(1) X(2) X(3)
#define LD32(it)                                                                     \
    {                                                                                \
        const int r = s_r + 32 * it;                                                 \
        const int ar = min(c.row0 + r, c.M - 1);                                     \
        const int br = min(c.col0 + r, c.N - 1);                                     \
        const float* src = gemm_a_src(p, A, A2, ar, k);                                  \
        ra##it = src ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f); \
        rb##it = *reinterpret_cast<const float4*>(W + (size_t)br * p.ldw + k);       \
    }
#define ST32(it)                                    \
    As[s_kq * LDS_ROWS + s_r + 32 * it] = ra##it;   \
    Bs[s_kq * LDS_ROWS + s_r + 32 * it] = rb##it;
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        const int k = kt * BK32 + s_kq * 4;
        FOR4(LD32)
    };

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        FOR4(ST32)
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int t = 0; t < BK32 / 8; ++t) {
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
                    // weight fragment = MFMA A operand (rows = features), activations = B (cols = tokens)
                    acc[m][n] = mfma32(b[n].x, a[m].x, acc[m][n]);
                    acc[m][n] = mfma32(b[n].y, a[m].y, acc[m][n]);
                    acc[m][n] = mfma32(b[n].z, a[m].z, acc[m][n]);
                    acc[m][n] = mfma32(b[n].w, a[m].w, acc[m][n]);
                }
        }
        __syncthreads();
    }
    gemm_epilogue<EPI>(p, c, acc, 1.0f, wm, wn, lo, hi);
}

// ------------------------------------------------------------------ staged epilogues of the split kernel
// Written straight from the C^T fragments, a store instruction touches 32 token rows x 32 B (row-major
// outputs) or is an 8- / 2-byte scatter (Q / K / V^T planes); the memory system serves that at a
// third of the rate of whole lines and the epilogue was a third of the kernel.  The tile therefore
// leaves through LDS, one 64-token half at a time (<= 36.9 KB, so three workgroups fit a CU):
// the waves that own the half park it, then all 256 threads write whole rows.
#define STG_C_ROW 132   // floats per parked [token][128 features] row (conflict-free both ways)
#define STG_H_ROW 144   // bytes per parked row of 64 halves (128 + 16 pad)
#define STG_H_PLANE 18432
// NW = 32-column fragments per wave: 2 -> 128-column tile (both wn waves park a half), 4 -> 256-column tile
// (processed as two 128-column sub-tiles, each owned by the waves of one wn).
template <int EPI, int NW>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmP& p, const TileCtx& cc, f32x16 (&acc)[2][NW], float wsc, int wm,
                                                     int wn, int lo, int hi, char* sb) {
    const int tid = threadIdx.x;
    constexpr int NPASS = NW;  // (token half) x (128-column sub-tile)
#define PASS_H(ps) ((ps) & 1)
#define PASS_C2(ps) ((ps) >> 1)
#define PARKS(ps) (wm == PASS_H(ps) && (NW == 2 || wn == PASS_C2(ps)))
#define FRAG_COL(n) ((NW == 2 ? wn * 64 : 0) + (n) * 32)
    if (EPI == EPI_QKV || EPI == EPI_CROSS) {
        if (!p.split_out) {
            if constexpr (NW == 2) gemm_epilogue<EPI>(p, cc, acc, wsc, wm, wn, lo, hi);
            return;
        }
        // the 128 columns of a sub-tile are two heads of ONE of q / k / v (256 features each)
        const int t = cc.col0 >> 8;
        float* dst;
        bool vt, rope, scale;
        if (EPI == EPI_QKV) {
            dst = (t == 0) ? p.Q : (t == 1) ? p.Kt : p.V;
            vt = (t == 2);
            rope = (t < 2);
            scale = (t == 0);
        } else {
            dst = (t == 0) ? p.Q : p.V;
            vt = (t == 1);
            rope = false;
            scale = (t == 0);
        }
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
        const int i0 = cc.row0 - cc.seq * p.rows_per_seq;
        if (rope) {
            // q / k of the self block.  RoPE from the fragments means 32 scattered 8-byte table loads per
            // lane; parked as f32 [token][128] instead, a thread finishes 8 consecutive features of one
            // token (bias, RoPE with one float4 of cos and of sin, scale, split) and 8 lanes write one whole
            // 128-byte row of each plane.
            float* st = reinterpret_cast<float*>(sb);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int h = PASS_H(ps);
                const int col0 = cc.col0 + PASS_C2(ps) * 128, hd0 = (col0 >> 6) & 3;
                if (PARKS(ps)) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NW; ++n)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int tl = m * 32 + lo, fl = FRAG_COL(n) + 8 * q + 4 * hi;
                                *reinterpret_cast<float4*>(st + tl * STG_C_ROW + fl) =
                                    make_float4(acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                                acc[m][n][4 * q + 3] * wsc);
                            }
                }
                __syncthreads();
                const int ch = tid & 15;  // 8-feature chunk of the 128 columns
                const int fl = ch * 8, d0 = fl & 63, hh = fl >> 6;
                float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
                if (cc.bias != nullptr) {
                    ba = *reinterpret_cast<const float4*>(cc.bias + col0 + fl);
                    bb = *reinterpret_cast<const float4*>(cc.bias + col0 + fl + 4);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tl = (tid >> 4) + 16 * it;
                    const int row = cc.row0 + h * 64 + tl;
                    const float4 va = *reinterpret_cast<const float4*>(st + tl * STG_C_ROW + fl);
                    const float4 vb = *reinterpret_cast<const float4*>(st + tl * STG_C_ROW + fl + 4);
                    const float4 cs = *reinterpret_cast<const float4*>(p.rope_cos + (size_t)row * 32 + (d0 >> 1));
                    const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (size_t)row * 32 + (d0 >> 1));
                    float v[8] = {va.x + ba.x, va.y + ba.y, va.z + ba.z, va.w + ba.w, vb.x + bb.x, vb.y + bb.y, vb.z + bb.z, vb.w + bb.w};
                    const float cw[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0)
                        const float a0 = v[2 * j], a1 = v[2 * j + 1];
                        v[2 * j] = a0 * cw[j] + (-a1) * ss[j];
                        v[2 * j + 1] = a1 * cw[j] + a0 * ss[j];
                    }
                    if (scale) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
                    }
                    uint4 hv, lv;
                    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hv, lv);
                    unsigned short* o = d16 + (((size_t)cc.seq * p.heads + hd0 + hh) * p.rows_per_seq + i0 + h * 64 + tl) * 64 + d0;
                    *reinterpret_cast<uint4*>(o) = hv;
                    *reinterpret_cast<uint4*>(o + p.plane_halves) = lv;
                }
                if (ps + 1 < NPASS) __syncthreads();
            }
            return;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int h = PASS_H(ps);
            const int col0 = cc.col0 + PASS_C2(ps) * 128, hd0 = (col0 >> 6) & 3;
            if (PARKS(ps)) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int tl = m * 32 + lo;  // token within the half
#pragma unroll
                    for (int n = 0; n < NW; ++n) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int fl = FRAG_COL(n) + 8 * q + 4 * hi;  // first of 4 features within the sub-tile
                            const int f0 = col0 + fl;
                            float v[4] = {acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                          acc[m][n][4 * q + 3] * wsc};
                            if (cc.bias != nullptr) {
                                const float4 b4 = *reinterpret_cast<const float4*>(cc.bias + f0);
                                v[0] += b4.x;
                                v[1] += b4.y;
                                v[2] += b4.z;
                                v[3] += b4.w;
                            }
                            if (scale) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                            }
                            unsigned h01, l01, h23, l23;
                            split2(v[0], v[1], h01, l01);
                            split2(v[2], v[3], h23, l23);
                            if (!vt) {
                                char* o = sb + ((fl >> 6) * 64 + tl) * STG_H_ROW + (fl & 63) * 2;
                                *reinterpret_cast<uint2*>(o) = make_uint2(h01, h23);
                                *reinterpret_cast<uint2*>(o + STG_H_PLANE) = make_uint2(l01, l23);
                            } else {
                                // pair lanes (tokens 2u, 2u+1): the even lane ends up with features f, f+1 of
                                // both tokens, the odd lane with f+2, f+3 -> 4-byte [feature][token pair] words
                                const bool odd = (lo & 1) != 0;
                                const unsigned rh = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? h01 : h23), 0xB1, 0xF, 0xF, true);
                                const unsigned rl = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? l01 : l23), 0xB1, 0xF, 0xF, true);
                                unsigned wh0, wh1, wl0, wl1;
                                if (!odd) {
                                    wh0 = (h01 & 0xFFFFu) | (rh << 16);
                                    wh1 = (h01 >> 16) | (rh & 0xFFFF0000u);
                                    wl0 = (l01 & 0xFFFFu) | (rl << 16);
                                    wl1 = (l01 >> 16) | (rl & 0xFFFF0000u);
                                } else {
                                    wh0 = (rh & 0xFFFFu) | (h23 << 16);
                                    wh1 = (rh >> 16) | (h23 & 0xFFFF0000u);
                                    wl0 = (rl & 0xFFFFu) | (l23 << 16);
                                    wl1 = (rl >> 16) | (l23 & 0xFFFF0000u);
                                }
                                char* o = sb + (fl + (odd ? 2 : 0)) * STG_H_ROW + (tl >> 1) * 4;
                                *reinterpret_cast<unsigned*>(o) = wh0;
                                *reinterpret_cast<unsigned*>(o + STG_H_ROW) = wh1;
                                *reinterpret_cast<unsigned*>(o + STG_H_PLANE) = wl0;
                                *reinterpret_cast<unsigned*>(o + STG_H_PLANE + STG_H_ROW) = wl1;
                            }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int g = it * 256 + tid;  // 16-byte granule: 2 planes x 1024
                const int plane = g >> 10, rem = g & 1023;
                if (!vt) {
                    // per (plane, head): 64 tokens x 128 B = one contiguous 8 KiB run
                    const int hh = rem >> 9, tok = (rem >> 3) & 63, gr = rem & 7;
                    const uint4 val = *reinterpret_cast<const uint4*>(sb + plane * STG_H_PLANE + (hh * 64 + tok) * STG_H_ROW + gr * 16);
                    unsigned short* o = d16 + (size_t)plane * p.plane_halves +
                                        (((size_t)cc.seq * p.heads + hd0 + hh) * p.rows_per_seq + i0 + h * 64 + tok) * 64 + gr * 8;
                    *reinterpret_cast<uint4*>(o) = val;
                } else {
                    // V^T [seq][head][64][rows]: one whole 128-byte line per (plane, feature)
                    const int feat = rem >> 3, gr = rem & 7;
                    const uint4 val = *reinterpret_cast<const uint4*>(sb + plane * STG_H_PLANE + feat * STG_H_ROW + gr * 16);
                    unsigned short* o = d16 + (size_t)plane * p.plane_halves +
                                        (((size_t)cc.seq * p.heads + hd0 + (feat >> 6)) * 64 + (feat & 63)) * p.rows_per_seq + i0 + h * 64 +
                                        gr * 8;
                    *reinterpret_cast<uint4*>(o) = val;
                }
            }
            if (ps + 1 < NPASS) __syncthreads();
        }
        return;
    }
    // ---- row-major f32 outputs
    float* C = p.C + (size_t)cc.z * p.c_bs;
    float* st = reinterpret_cast<float*>(sb);
    const bool res_vec = (EPI == EPI_CONV) && p.resid != nullptr && ((p.ldr & 3) == 0);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int h = PASS_H(ps);
        const int f0 = cc.col0 + PASS_C2(ps) * 128 + 4 * (tid & 31);
        const bool colok = f0 < cc.N;
        const bool full = (f0 + 3 < cc.N) && ((p.ldc & 3) == 0);
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (cc.bias != nullptr && colok) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < cc.N) b[j] = cc.bias[f0 + j];
        }
        if (PARKS(ps)) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int tl = m * 32 + lo, fl = FRAG_COL(n) + 8 * q + 4 * hi;
                        *reinterpret_cast<float4*>(st + tl * STG_C_ROW + fl) =
                            make_float4(acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                        acc[m][n][4 * q + 3] * wsc);
                    }
        }
        __syncthreads();
        if constexpr (EPI == EPI_NNSTAT) {
            // similarity tile of the mutual-NN matcher: never stored.  While the half is parked in LDS,
            //  * rows: thread = (row tid & 63, 32-column segment tid >> 6) scans its 32 values in increasing column order (the 16 lanes of
            //    a 16-byte read group sit on 16 different rows: conflict-free), the four segments of a row are folded in order by threads
            //    0..63, which write 64 consecutive entries of each partial array;
            //  * columns: a thread sees 8 rows (increasing) of its 4 columns, the 8 row groups are folded by threads 128..255.
            // Every comparison is `>` or an explicit lowest-index rule, so ties resolve as a sequential first-maximum scan does.
            // (A first version reduced a row with five xor-shuffle steps of (best, index, second) per 8 rows and wrote the row partials
            // from one lane per half-wave: 4.75 ms per 64 pairs of 5000 x 5000 -- the dependent ds_bpermute chains, not the MFMAs.)
            const int ctile = (cc.col0 >> 7) + PASS_C2(ps);
            float* clb1 = st + 64 * STG_C_ROW;  // [8 row groups][128 columns] best, second, index (behind the parked half)
            float* clb2 = clb1 + 8 * 128;
            int* cli1 = reinterpret_cast<int*>(clb2 + 8 * 128);
            float* rlb1 = reinterpret_cast<float*>(cli1 + 8 * 128);  // [4 segments][64 rows] best, second, index
            float* rlb2 = rlb1 + 4 * 64;
            int* rli1 = reinterpret_cast<int*>(rlb2 + 4 * 64);
            {
                const int tl = tid & 63, sg = tid >> 6;
                const int cbase = cc.col0 + PASS_C2(ps) * 128 + sg * 32;
                float b1 = -INFINITY, b2 = -INFINITY;
                int i1 = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 t4 = *reinterpret_cast<const float4*>(st + tl * STG_C_ROW + sg * 32 + 4 * k);
                    const float x[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = cbase + 4 * k + j;
                        const float xv = col < cc.N ? x[j] : -INFINITY;
                        const bool up = xv > b1;
                        b2 = up ? b1 : fmaxf(b2, xv);
                        i1 = up ? col : i1;
                        b1 = up ? xv : b1;
                    }
                }
                rlb1[sg * 64 + tl] = b1;
                rlb2[sg * 64 + tl] = b2;
                rli1[sg * 64 + tl] = i1;
            }
            {
                float cb1[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, cb2[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int ci1[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int tl = (tid >> 5) + 8 * it;
                    const int row = cc.row0 + h * 64 + tl;
                    const float4 t4 = *reinterpret_cast<const float4*>(st + tl * STG_C_ROW + 4 * (tid & 31));
                    const float x[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = (row < cc.M && f0 + j < cc.N) ? x[j] : -INFINITY;
                        const bool up = xv > cb1[j];
                        cb2[j] = up ? cb1[j] : fmaxf(cb2[j], xv);
                        ci1[j] = up ? row : ci1[j];
                        cb1[j] = up ? xv : cb1[j];
                    }
                }
                *reinterpret_cast<float4*>(clb1 + (tid >> 5) * 128 + 4 * (tid & 31)) = make_float4(cb1[0], cb1[1], cb1[2], cb1[3]);
                *reinterpret_cast<float4*>(clb2 + (tid >> 5) * 128 + 4 * (tid & 31)) = make_float4(cb2[0], cb2[1], cb2[2], cb2[3]);
                *reinterpret_cast<int4*>(cli1 + (tid >> 5) * 128 + 4 * (tid & 31)) = make_int4(ci1[0], ci1[1], ci1[2], ci1[3]);
            }
            __syncthreads();
            if (tid < 64) {
                const int row = cc.row0 + h * 64 + tid;
                if (row < cc.M) {
                    float b1 = rlb1[tid], b2 = rlb2[tid];
                    int i1 = rli1[tid];
#pragma unroll
                    for (int sg = 1; sg < 4; ++sg) {  // increasing columns: a later segment wins only with a larger value
                        const float ob1 = rlb1[sg * 64 + tid], ob2 = rlb2[sg * 64 + tid];
                        const bool up = ob1 > b1;
                        b2 = up ? fmaxf(b1, ob2) : fmaxf(b2, ob1);
                        i1 = up ? rli1[sg * 64 + tid] : i1;
                        b1 = up ? ob1 : b1;
                    }
                    const size_t o = ((size_t)cc.z * p.st_nct + ctile) * p.st_rpitch + row;
                    p.st_rpm[o] = b1;
                    p.st_rps[o] = b2;
                    p.st_rpi[o] = i1;
                }
            } else if (tid >= 128) {
                const int c = tid - 128;
                const int col = cc.col0 + PASS_C2(ps) * 128 + c;
                if (col < cc.N) {
                    float b1 = clb1[c], b2 = clb2[c];
                    int i1 = cli1[c];
#pragma unroll
                    for (int gq = 1; gq < 8; ++gq) {  // the row groups interleave (rows gq, gq + 8, ...): lowest index on equal values
                        const float ob1 = clb1[gq * 128 + c], ob2 = clb2[gq * 128 + c];
                        const int oi1 = cli1[gq * 128 + c];
                        const bool up = ob1 > b1 || (ob1 == b1 && oi1 < i1);
                        b2 = up ? fmaxf(b1, ob2) : fmaxf(b2, ob1);
                        i1 = up ? oi1 : i1;
                        b1 = up ? ob1 : b1;
                    }
                    const size_t o = ((size_t)cc.z * p.st_nrh + (cc.row0 >> 6) + h) * p.st_cpitch + col;
                    p.st_cpm[o] = b1;
                    p.st_cps[o] = b2;
                    p.st_cpi[o] = i1;
                }
            }
        } else if (colok) {
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int tl = (tid >> 5) + 8 * it;
                const int row = cc.row0 + h * 64 + tl;
                if (row >= cc.M) break;
                const float4 t4 = *reinterpret_cast<const float4*>(st + tl * STG_C_ROW + 4 * (tid & 31));
                float v[4] = {t4.x + b[0], t4.y + b[1], t4.z + b[2], t4.w + b[3]};
                float* dst = C + (size_t)row * p.ldc + f0;
                if (EPI == EPI_CONV) {
                    if (p.resid != nullptr && p.rup_h > 0) {
                        // residual = bilinear x2 of a half-resolution map, ATen's formula and evaluation order (what the
                        // stand-alone up-sampling kernels compute), N % 4 == 0 checked at launch
                        const int hh = p.rup_h, ww = p.rup_w, Wo = 2 * ww, Ho = 2 * hh;
                        const int ox = row % Wo, t2 = row / Wo, oy = t2 % Ho, bi = t2 / Ho;
                        float fy, fx;
                        int y1, x1;
                        if (p.rup_align) {
                            fy = ((float)(hh - 1) / (float)(Ho - 1)) * (float)oy;
                            fx = ((float)(ww - 1) / (float)(Wo - 1)) * (float)ox;
                        } else {
                            fy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.0f);
                            fx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.0f);
                        }
                        const int y0 = (int)fy, x0 = (int)fx;
                        if (p.rup_align) {
                            y1 = min(y0 + 1, hh - 1);
                            x1 = min(x0 + 1, ww - 1);
                        } else {
                            y1 = y0 + (y0 < hh - 1 ? 1 : 0);
                            x1 = x0 + (x0 < ww - 1 ? 1 : 0);
                        }
                        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
                        const float* base = p.resid + (size_t)bi * hh * ww * p.ldr + f0;
                        const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * ww + x0) * p.ldr);
                        const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * ww + x1) * p.ldr);
                        const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * ww + x0) * p.ldr);
                        const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * ww + x1) * p.ldr);
                        v[0] += hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
                        v[1] += hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
                        v[2] += hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
                        v[3] += hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
                    } else if (p.resid != nullptr) {
                        const float* rs = p.resid + (size_t)row * p.ldr + f0;
                        if (full && res_vec) {
                            const float4 r4 = *reinterpret_cast<const float4*>(rs);
                            v[0] += r4.x;
                            v[1] += r4.y;
                            v[2] += r4.z;
                            v[3] += r4.w;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (f0 + j < cc.N) v[j] += rs[j];
                        }
                    }
                    if (p.act == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.01f * v[j];
                    } else if (p.act == 3) {  // GELU with the erf of common.h (1.3e-7 absolute on erf; ATen: 0.5 x (1 + erf(x / sqrt 2)))
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_poly(v[j]);
                    }
                } else if (EPI == EPI_BIAS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                } else if (EPI == EPI_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                }
                if (full) {
                    if (EPI == EPI_RESID) {
                        const float4 o4 = *reinterpret_cast<const float4*>(dst);
                        v[0] += o4.x;
                        v[1] += o4.y;
                        v[2] += o4.z;
                        v[3] += o4.w;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (f0 + j < cc.N) dst[j] = (EPI == EPI_RESID) ? dst[j] + v[j] : v[j];
                }
            }
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
#undef PASS_H
#undef PASS_C2
#undef PARKS
#undef FRAG_COL
}

// ------------------------------------------------------------------ 3 x f16 split kernel
// Tile 128 x 128 x 32 (three workgroups per CU: 36 KB LDS, <= 168 VGPRs) or 128 x 256 x 32, 4 waves (2 x 2).
//  * Weights (WDMA = pre-split planes): packed at load time as FRAGMENT-MAJOR f16 hi / lo planes
//    [N/32][K/16][64 lanes][8 halves]: one coalesced 16-byte load + ds_write_b128 per lane moves a
//    ready-to-use, conflict-free 1 KiB MFMA fragment (no VALU on the weight side); the fragments of tile kt+1
//    are requested while tile kt is multiplied.  (An LDS-DMA version -- `global_load_lds_dwordx4` from inline
//    asm with a hand-placed vmcnt drain, because a compiler-visible DMA makes hipcc drain vmcnt to 0 at every
//    register-load use -- measured 3-5 % slower on 128-column tiles and equal on 256-column ones, and hiding
//    register loads in asm to count waits by hand is NOT safe: the allocator may copy a destination register
//    before the load has landed.  Everything is ordinary loads now; the compiler counts the waits.)
//  * Activations: fetched one tile ahead into registers, split into (hi, lo) f16 and written to a
//    single LDS buffer in the same fragment order with the granule position XOR-swizzled by 2*(k-octet)
//    (both the ds_write_b128 of 8 lanes = 2 rows x 4 octets and the fragment ds_read_b128 are then
//    conflict-free).
#define BK3 32
// ASRC: 0 = matrix (optionally two K slabs), 1 = implicit im2col of an NHWC image
// WDMA: weights are pre-split fragment-major planes; otherwise B is an f32 matrix [N][K] (activations,
//       e.g. similarity products) staged like A
// NW:   32-column fragments per wave: 2 -> tile 128 x 128 (three workgroups per CU), 4 -> tile 128 x 256 (two per
//       CU, 80 KB LDS): per MFMA half the LDS fragment reads and half the activation staging -- on a SIMD the
//       matrix pipe and everything else serialise, so the wave tile is the efficiency lever
// SINGLE: one f16 product per element pair instead of three (hi planes only, f32 accumulate): the arithmetic class of a
//       bf16 / fp16 autocast run (11-bit operands) for callers that ask for it (GemmP.single; pre-split weights only)
template <int EPI, int ASRC, bool WDMA, int NW, bool SINGLE = false>
__global__ __launch_bounds__(256, (WDMA && NW == 2) ? 3 : 2) void gemm_split_kernel(GemmP p) {
    // 128-column tiles: the four weight fragments of a wave in four registers, single LDS buffer (written between
    // the two barriers like the activations).  256-column tiles: eight fragments, four registers, two halves,
    // double-buffered in LDS (parked while the current tile is multiplied).
    constexpr bool WREGP = WDMA && NW == 2;
    static_assert(NW == 2 || (NW == 4 && WDMA), "256-column tiles need pre-split weight planes");
    static_assert(!SINGLE || WDMA, "the single-product variant reads pre-split weight planes");
    constexpr int BPL = 4 * NW * 1024;  // bytes per B plane per stage: [ks 2][nf 2 NW] fragments of 1 KiB
    // A_hi 8K | A_lo 8K | B stage 0 (hi, lo) | B stage 1 (256-column tiles) ; reused by the epilogue (<= 36864 B)
    __shared__ uint4 smem[((WDMA && NW == 2) ? 36864 : 16384 + 4 * BPL) / 16];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    TileCtx c;
    if (!gemm_tile_setup(p, c, 64 * NW)) return;
    const float* A = p.A + (size_t)c.z * p.a_bs;
    const float* A2 = p.A2 ? p.A2 + (size_t)c.z * p.a2_bs : nullptr;
    const float wsc = (WDMA && p.wscale) ? p.wscale[c.wsel] : 1.0f;
    const int nkt = p.K / BK3;

    f32x16 acc[2][NW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // ---- staging geometry: thread -> k-octet g = tid & 3 (8 consecutive k) of rows tid >> 2 and + 64
    const int g = tid & 3;
    const int rl0 = tid >> 2, rl1 = 64 + (tid >> 2);
    // LDS byte offset of the granule: fragment (ks = g >> 1, rf = row >> 5), half g & 1, position (row & 31) ^ 2g
    const int wo0 = (((g >> 1) * 4 + (rl0 >> 5)) * 64 + (g & 1) * 32 + ((rl0 & 31) ^ (2 * g))) * 16;
    const int wo1 = (((g >> 1) * 4 + (rl1 >> 5)) * 64 + (g & 1) * 32 + ((rl1 & 31) ^ (2 * g))) * 16;
    const int ar0 = min(c.row0 + rl0, c.M - 1), ar1 = min(c.row0 + rl1, c.M - 1);
    const float *pa0, *pa1, *pb0 = nullptr, *pb1 = nullptr;
    int iy0 = 0, ix0 = 0, iy1 = 0, ix1 = 0;  // conv: top-left input pixel of the two rows
    if (ASRC == 1) {
        const int ox0 = ar0 % p.conv_wout, t0 = ar0 / p.conv_wout, ox1 = ar1 % p.conv_wout, t1 = ar1 / p.conv_wout;
        const int oy0 = t0 % p.conv_hout, b0 = t0 / p.conv_hout, oy1 = t1 % p.conv_hout, b1 = t1 / p.conv_hout;
        iy0 = oy0 * p.conv_stride - p.conv_pad;
        ix0 = ox0 * p.conv_stride - p.conv_pad;
        iy1 = oy1 * p.conv_stride - p.conv_pad;
        ix1 = ox1 * p.conv_stride - p.conv_pad;
        pa0 = A + (((long)b0 * p.conv_hin + iy0) * p.conv_win + ix0) * p.conv_cin + g * 8;
        pa1 = A + (((long)b1 * p.conv_hin + iy1) * p.conv_win + ix1) * p.conv_cin + g * 8;
    } else {
        pa0 = A + (size_t)ar0 * p.lda + g * 8;
        pa1 = A + (size_t)ar1 * p.lda + g * 8;
        if (A2 != nullptr) {
            pb0 = A2 + (size_t)ar0 * p.lda2 + g * 8 - p.K1;
            pb1 = A2 + (size_t)ar1 * p.lda2 + g * 8 - p.K1;
        }
    }
    f32x4 xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1;
    bool xv0 = true, xv1 = true, yv0 = true, yv1 = true;  // conv: row inside the image for this tap
    int cv_c0 = 0, cv_ky = 0, cv_kx = 0;                    // conv: running (tap, channel) of the next tile
    // tiles are requested strictly in order kt = 0, 1, 2, ...
    auto issue_a = [&](int kt, f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1, bool& v0, bool& v1) __attribute__((always_inline)) {
        const float *q0, *q1;
        if (ASRC == 1) {
            const int off = (cv_ky * p.conv_win + cv_kx) * p.conv_cin + cv_c0;
            v0 = (unsigned)(iy0 + cv_ky) < (unsigned)p.conv_hin && (unsigned)(ix0 + cv_kx) < (unsigned)p.conv_win;
            v1 = (unsigned)(iy1 + cv_ky) < (unsigned)p.conv_hin && (unsigned)(ix1 + cv_kx) < (unsigned)p.conv_win;
            q0 = v0 ? pa0 + off : A;
            q1 = v1 ? pa1 + off : A;
            cv_c0 += BK3;
            if (cv_c0 == p.conv_cin) {
                cv_c0 = 0;
                if (++cv_kx == p.conv_k) {
                    cv_kx = 0;
                    ++cv_ky;
                }
            }
        } else {
            const int k = kt * BK3;
            if (A2 != nullptr && k >= p.K1) {
                q0 = pb0 + k;
                q1 = pb1 + k;
            } else {
                q0 = pa0 + k;
                q1 = pa1 + k;
            }
        }
        a0 = *reinterpret_cast<const f32x4*>(q0);
        b0 = *reinterpret_cast<const f32x4*>(q0 + 4);
        a1 = *reinterpret_cast<const f32x4*>(q1);
        b1 = *reinterpret_cast<const f32x4*>(q1 + 4);
    };
    auto store_a = [&](f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1, bool v0, bool v1) __attribute__((always_inline)) {
        uint4 h, l;
        float4 fa = __builtin_bit_cast(float4, a0), fb = __builtin_bit_cast(float4, b0);
        if (ASRC == 1 && !v0) fa = fb = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SINGLE) {  // one product: the nearest f16 of either operand (common.h)
            h = half8_rtn(fa, fb);
            *reinterpret_cast<uint4*>(sm + wo0) = h;
            fa = __builtin_bit_cast(float4, a1);
            fb = __builtin_bit_cast(float4, b1);
            if (ASRC == 1 && !v1) fa = fb = make_float4(0.f, 0.f, 0.f, 0.f);
            h = half8_rtn(fa, fb);
            *reinterpret_cast<uint4*>(sm + wo1) = h;
            return;
        }
        split8(fa, fb, h, l);
        *reinterpret_cast<uint4*>(sm + wo0) = h;
        if constexpr (!SINGLE) *reinterpret_cast<uint4*>(sm + 8192 + wo0) = l;
        fa = __builtin_bit_cast(float4, a1);
        fb = __builtin_bit_cast(float4, b1);
        if (ASRC == 1 && !v1) fa = fb = make_float4(0.f, 0.f, 0.f, 0.f);
        split8(fa, fb, h, l);
        *reinterpret_cast<uint4*>(sm + wo1) = h;
        if constexpr (!SINGLE) *reinterpret_cast<uint4*>(sm + 8192 + wo1) = l;
    };

    // ---- B operand
    // WDMA: wave w moves the fragments of column blocks nf = w (+ 4 for 256-column tiles), all planes / k-steps
    const int nks = p.K >> 4;
    const int nfr = (p.N + 31) >> 5;
    const uint4 *wh[NW / 2], *wl[NW / 2];
#pragma unroll
    for (int u = 0; u < NW / 2; ++u) {
        wh[u] = wl[u] = nullptr;
        if (WDMA) {
            const int nfg = min((c.col0 >> 5) + wid + 4 * u, nfr - 1);
            wh[u] = reinterpret_cast<const uint4*>(p.Wh + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
            wl[u] = reinterpret_cast<const uint4*>(p.Wl + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
        }
    }
    // WREGP: the fragment-major planes through registers
    uint4 bq0, bq1, bq2, bq3;
    auto load_bp2 = [&](int kt, uint4& q0, uint4& q1, uint4& q2, uint4& q3) __attribute__((always_inline)) {
        q0 = wh[0][(size_t)(kt * 2 + 0) * 64];
        q1 = wh[0][(size_t)(kt * 2 + 1) * 64];
        if constexpr (!SINGLE) {
            q2 = wl[0][(size_t)(kt * 2 + 0) * 64];
            q3 = wl[0][(size_t)(kt * 2 + 1) * 64];
        }
    };
    auto store_bp2 = [&](int stg, uint4& q0, uint4& q1, uint4& q2, uint4& q3) __attribute__((always_inline)) {
        char* d = sm + 16384 + wid * 1024 + lane * 16;  // single buffer: written between the two barriers, like A
        (void)stg;
        *reinterpret_cast<uint4*>(d) = q0;
        *reinterpret_cast<uint4*>(d + BPL / 2) = q1;
        if constexpr (!SINGLE) {
            *reinterpret_cast<uint4*>(d + BPL) = q2;
            *reinterpret_cast<uint4*>(d + BPL + BPL / 2) = q3;
        }
    };
    // 256-column tiles: the wave's two column blocks (u = 0, 1) go through the same four registers one after the other
    auto load_bu = [&](int kt, int u) __attribute__((always_inline)) {
        bq0 = wh[u][(size_t)(kt * 2 + 0) * 64];
        bq1 = wh[u][(size_t)(kt * 2 + 1) * 64];
        if constexpr (!SINGLE) {
            bq2 = wl[u][(size_t)(kt * 2 + 0) * 64];
            bq3 = wl[u][(size_t)(kt * 2 + 1) * 64];
        }
    };
    auto store_bu = [&](int stg, int u) __attribute__((always_inline)) {
        char* d = sm + 16384 + stg * 2 * BPL + (wid + 4 * u) * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(d) = bq0;
        *reinterpret_cast<uint4*>(d + BPL / 2) = bq1;
        if constexpr (!SINGLE) {
            *reinterpret_cast<uint4*>(d + BPL) = bq2;
            *reinterpret_cast<uint4*>(d + BPL + BPL / 2) = bq3;
        }
    };
    auto load_bp = [&](int kt) __attribute__((always_inline)) { load_bp2(kt, bq0, bq1, bq2, bq3); };
    auto store_bp = [&](int stg) __attribute__((always_inline)) { store_bp2(stg, bq0, bq1, bq2, bq3); };
    // !WDMA: B rows staged like A (single buffer = stage 0)
    f32x4 xc0, xd0, xc1, xd1, yc0, yd0, yc1, yd1;
    const float *pw0 = nullptr, *pw1 = nullptr;
    if (!WDMA) {
        const float* W = p.W + (size_t)c.z * p.w_bs + (size_t)c.wsel * p.w_stride;
        pw0 = W + (size_t)min(c.col0 + rl0, c.N - 1) * p.ldw + g * 8;
        pw1 = W + (size_t)min(c.col0 + rl1, c.N - 1) * p.ldw + g * 8;
    }
    auto issue_b = [&](int kt, f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1) __attribute__((always_inline)) {
        const float *q0 = pw0 + kt * BK3, *q1 = pw1 + kt * BK3;
        a0 = *reinterpret_cast<const f32x4*>(q0);
        b0 = *reinterpret_cast<const f32x4*>(q0 + 4);
        a1 = *reinterpret_cast<const f32x4*>(q1);
        b1 = *reinterpret_cast<const f32x4*>(q1 + 4);
    };
    auto store_b = [&](f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1) __attribute__((always_inline)) {
        uint4 h, l;
        split8(__builtin_bit_cast(float4, a0), __builtin_bit_cast(float4, b0), h, l);
        *reinterpret_cast<uint4*>(sm + 16384 + wo0) = h;
        *reinterpret_cast<uint4*>(sm + 24576 + wo0) = l;
        split8(__builtin_bit_cast(float4, a1), __builtin_bit_cast(float4, b1), h, l);
        *reinterpret_cast<uint4*>(sm + 16384 + wo1) = h;
        *reinterpret_cast<uint4*>(sm + 24576 + wo1) = l;
    };

    auto compute_ks = [&](int stg, int ks) __attribute__((always_inline)) {
        const char* sbt = sm + 16384 + ((WDMA && !WREGP) ? stg * 2 * BPL : 0);
        {
            uint4 ah[2], al[2], bh[NW], bl[NW];
            const int apos = (hi * 32 + (lo ^ (2 * (2 * ks + hi)))) * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int fo = (ks * 4 + wm * 2 + m) * 1024 + apos;
                ah[m] = *reinterpret_cast<const uint4*>(sm + fo);
                if constexpr (!SINGLE) al[m] = *reinterpret_cast<const uint4*>(sm + 8192 + fo);
            }
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int fo = (ks * 2 * NW + wn * NW + n) * 1024 + (WDMA ? lane * 16 : apos);
                bh[n] = *reinterpret_cast<const uint4*>(sbt + fo);
                if constexpr (!SINGLE) bl[n] = *reinterpret_cast<const uint4*>(sbt + BPL + fo);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    // weight fragment = MFMA A operand (rows = features), activations = B (cols = tokens)
                    if constexpr (!SINGLE) {
                        acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                        acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                    }
                    acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                }
        }
    };
    auto compute = [&](int stg) __attribute__((always_inline)) {
        compute_ks(stg, 0);
        compute_ks(stg, 1);
    };
    if constexpr (WREGP) {
        issue_a(0, xa0, xb0, xa1, xb1, xv0, xv1);
        load_bp(0);
        store_a(xa0, xb0, xa1, xb1, xv0, xv1);
        store_bp(0);
        if (nkt > 1) {
            issue_a(1, xa0, xb0, xa1, xb1, xv0, xv1);
            load_bp(1);
        }
        __syncthreads();
        for (int kt = 0; kt < nkt; kt += 2) {
            compute(0);
            __syncthreads();
            if (kt + 1 < nkt) {
                store_a(xa0, xb0, xa1, xb1, xv0, xv1);
                store_bp(1);
                if (kt + 2 < nkt) {
                    issue_a(kt + 2, xa0, xb0, xa1, xb1, xv0, xv1);
                    load_bp(kt + 2);
                }
            }
            __syncthreads();
            if (kt + 1 < nkt) {
                compute(1);
                __syncthreads();
                if (kt + 2 < nkt) {
                    store_a(xa0, xb0, xa1, xb1, xv0, xv1);
                    store_bp(0);
                    if (kt + 3 < nkt) {
                        issue_a(kt + 3, xa0, xb0, xa1, xb1, xv0, xv1);
                        load_bp(kt + 3);
                    }
                }
                __syncthreads();
            }
        }
    } else if constexpr (WDMA) {
        // 256-column tiles: weights through four registers per lane, half a tile at a time -- the first column
        // block of tile kt+1 is requested before the first k-step of tile kt and parked in the other LDS stage
        // after it, the second around the second k-step.  Every load is an ordinary load: the compiler counts
        // the waits (an LDS-DMA version needed a hand-placed vmcnt(0) drain per tile and was 3-5 % slower).
        issue_a(0, xa0, xb0, xa1, xb1, xv0, xv1);
        load_bu(0, 0);
        store_bu(0, 0);
        load_bu(0, 1);
        store_bu(0, 1);
        store_a(xa0, xb0, xa1, xb1, xv0, xv1);
        if (nkt > 1) issue_a(1, xa0, xb0, xa1, xb1, xv0, xv1);
        __syncthreads();
#define WIDE_STEP(kt_, stg_)                                                                  \
    {                                                                                         \
        const bool nxt__ = (kt_) + 1 < nkt;                                                   \
        if (nxt__) load_bu((kt_) + 1, 0);                                                     \
        compute_ks(stg_, 0);                                                                  \
        if (nxt__) {                                                                          \
            store_bu((stg_) ^ 1, 0);                                                          \
            load_bu((kt_) + 1, 1);                                                            \
        }                                                                                     \
        compute_ks(stg_, 1);                                                                  \
        __syncthreads();                                                                      \
        if (nxt__) {                                                                          \
            store_bu((stg_) ^ 1, 1);                                                          \
            store_a(xa0, xb0, xa1, xb1, xv0, xv1);                                            \
            if ((kt_) + 2 < nkt) issue_a((kt_) + 2, xa0, xb0, xa1, xb1, xv0, xv1);            \
        }                                                                                     \
        __syncthreads();                                                                      \
    }
        for (int kt = 0; kt < nkt; kt += 2) {
            WIDE_STEP(kt, 0)
            if (kt + 1 < nkt) WIDE_STEP(kt + 1, 1)
        }
#undef WIDE_STEP
    } else {
    // prologue: tile 0 -> LDS, tile 1 in flight
        issue_a(0, xa0, xb0, xa1, xb1, xv0, xv1);
        issue_b(0, xc0, xd0, xc1, xd1);
        if (nkt > 1) {
            issue_a(1, ya0, yb0, ya1, yb1, yv0, yv1);
            issue_b(1, yc0, yd0, yc1, yd1);
        }
        store_a(xa0, xb0, xa1, xb1, xv0, xv1);
        store_b(xc0, xd0, xc1, xd1);
        __syncthreads();
        for (int kt = 0; kt < nkt; kt += 2) {
            // tile kt is in LDS (weights in stage 0); y holds tile kt+1 (in flight)
            if (kt + 2 < nkt) {
                issue_a(kt + 2, xa0, xb0, xa1, xb1, xv0, xv1);
                issue_b(kt + 2, xc0, xd0, xc1, xd1);
            }
            compute(0);
            __syncthreads();
            if (kt + 1 < nkt) {
                store_a(ya0, yb0, ya1, yb1, yv0, yv1);
                store_b(yc0, yd0, yc1, yd1);
            }
            __syncthreads();
            if (kt + 1 < nkt) {
                if (kt + 3 < nkt) {
                    issue_a(kt + 3, ya0, yb0, ya1, yb1, yv0, yv1);
                    issue_b(kt + 3, yc0, yd0, yc1, yd1);
                }
                compute(1);
                __syncthreads();
                if (kt + 2 < nkt) {
                    store_a(xa0, xb0, xa1, xb1, xv0, xv1);
                    store_b(xc0, xd0, xc1, xd1);
                }
                __syncthreads();
            }
        }
    }
    gemm_epilogue_staged<EPI, NW>(p, c, acc, wsc, wm, wn, lo, hi, sm);
}

// The 128-column kernel runs three workgroups per CU (768 slots, register-limited: a 128-VGPR build for four
// was 40 % slower and mis-compiled); a launch whose tile count leaves the last round mostly empty is faster
// with two (512 slots, less sharing per workgroup).  Extra dynamic LDS is the occupancy knob:
// 36 KB static + 32 KB dynamic -> two workgroups per CU.
static unsigned occupancy_pad(long nblocks) {
    const long r3 = (nblocks + 767) / 768, r2 = (nblocks + 511) / 512;
    return (r2 * 4 <= r3 * 5) ? 32768u : 0u;  // a round at 3/CU costs ~1.25x a round at 2/CU
}
// 256-column tiles whenever the weights are DMA-able and N is a multiple of 256, for the row-major epilogues
// (measured: type BIAS / RESID -4..6 %; the QKV / CROSS plane epilogues lose more in their four single-wave
// parking passes than the main loop gains, so they stay on 128-column tiles).
static bool wide_ok(const GemmP& p) { return p.Wh != nullptr && p.N % 256 == 0 && p.epi != EPI_QKV && p.epi != EPI_CROSS; }
template <int EPI>
static void launch_one(const GemmP& p, bool split, hipStream_t stream) {
    const bool wide = split && wide_ok(p);
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, wide ? 256 : BN), 1, p.batch);
    if (!split)
        hipLaunchKernelGGL(gemm_kernel<EPI>, grid, dim3(256), 0, stream, p);
    else if (wide) {
        if constexpr (EPI == EPI_CONV) {
            if (p.single) {
                hipLaunchKernelGGL((gemm_split_kernel<EPI, 0, true, 4, true>), grid, dim3(256), 0, stream, p);
                return;
            }
        }
        if constexpr (EPI != EPI_QKV && EPI != EPI_CROSS)
            hipLaunchKernelGGL((gemm_split_kernel<EPI, 0, true, 4>), grid, dim3(256), 0, stream, p);
    }
    else if (p.Wh != nullptr) {
        if constexpr (EPI == EPI_CONV) {
            if (p.single) {
                hipLaunchKernelGGL((gemm_split_kernel<EPI, 0, true, 2, true>), grid, dim3(256), occupancy_pad((long)grid.x * grid.z), stream, p);
                return;
            }
        }
        hipLaunchKernelGGL((gemm_split_kernel<EPI, 0, true, 2>), grid, dim3(256), occupancy_pad((long)grid.x * grid.z), stream, p);
    }
    else
        hipLaunchKernelGGL((gemm_split_kernel<EPI, 0, false, 2>), grid, dim3(256), 0, stream, p);
}
static void launch_conv(const GemmP& p, bool split, hipStream_t stream) {
    const bool wide = split && wide_ok(p);
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, wide ? 256 : BN), 1, p.batch);
    if (!split)
        hipLaunchKernelGGL(gemm_kernel<EPI_CONV>, grid, dim3(256), 0, stream, p);
    else if (wide && p.single)
        hipLaunchKernelGGL((gemm_split_kernel<EPI_CONV, 1, true, 4, true>), grid, dim3(256), 0, stream, p);
    else if (wide)
        hipLaunchKernelGGL((gemm_split_kernel<EPI_CONV, 1, true, 4>), grid, dim3(256), 0, stream, p);
    else if (p.Wh != nullptr && p.single)
        hipLaunchKernelGGL((gemm_split_kernel<EPI_CONV, 1, true, 2, true>), grid, dim3(256), occupancy_pad((long)grid.x * grid.z), stream, p);
    else if (p.Wh != nullptr)
        hipLaunchKernelGGL((gemm_split_kernel<EPI_CONV, 1, true, 2>), grid, dim3(256), occupancy_pad((long)grid.x * grid.z), stream, p);
    else
        hipLaunchKernelGGL((gemm_split_kernel<EPI_CONV, 1, false, 2>), grid, dim3(256), 0, stream, p);
}

int gemm_launch(imcui_hip_s* h, const GemmP& p, hipStream_t stream) {
    if (p.K % BK32 != 0 || p.K <= 0) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K=%d must be a positive multiple of %d", p.K, BK32);
    if (p.A2 && (p.K1 % BK32 != 0)) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K1=%d must be a multiple of %d", p.K1, BK32);
    if (p.rows_per_seq > 0 && p.rows_per_seq % BM != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: rows_per_seq=%d must be a multiple of %d", p.rows_per_seq, BM);
    if (p.M <= 0 || p.N <= 0) return IMCUI_OK;
    if (p.conv_k > 0 && (p.conv_cin % BK32 != 0 || p.K != p.conv_k * p.conv_k * p.conv_cin || p.epi != EPI_CONV || p.A2))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm(conv): cin=%d must be a multiple of %d, K=%d == k*k*cin, epilogue EPI_CONV", p.conv_cin,
                             BK32, p.K);
    const bool split = h->precision == 1;
    if (p.rup_h > 0 && (!split || p.epi != EPI_CONV || p.N % 4 != 0 || p.ldr % 4 != 0 || p.M % (4 * p.rup_h * p.rup_w) != 0))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: the up-sampled residual needs the split mode, EPI_CONV, N %% 4 == 0 and M = images x 4 rup_h rup_w");
    if (p.split_out && (!split || p.rows_per_seq <= 0 || p.N % BN != 0 || p.M % BM != 0 || p.batch != 1))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: split_out needs the split mode and whole 128x128 tiles (M=%d N=%d)", p.M, p.N);
    if (!split && p.W == nullptr) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: f32 weights missing");
    if (p.single && (!split || p.Wh == nullptr || p.epi != EPI_CONV))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: the single-product variant needs the split mode, pre-split weight planes and EPI_CONV");
    static const bool dbg = getenv("IMCUI_HIP_GEMM_DEBUG") != nullptr;
    if (dbg) {
        hipStreamSynchronize(stream);
        fprintf(stderr, "[gemm] epi=%d M=%d N=%d K=%d K1=%d batch=%d split=%d wdma=%d conv=%d rows_per_seq=%d split_out=%d\n", p.epi, p.M, p.N,
                p.K, p.K1, p.batch, (int)split, p.Wh != nullptr, p.conv_k, p.rows_per_seq, p.split_out);
        fflush(stderr);
    }
    if (split && h->range_flag && p.conv_k == 0) {  // opt-in debugging aid: scan the f32 operands the kernel is about to split
        for (int z = 0; z < p.batch; ++z) {
            // live rows: ragged sequences (cnt) or the device-side row count of a batched product (mcnt / ncnt)
            const int* ac = p.mcnt ? p.mcnt + (size_t)z * p.cnt_stride : p.cnt;
            const int arps = p.mcnt ? p.M : p.rows_per_seq;
            imcui_range_check(h, p.A + (size_t)z * p.a_bs, p.M, p.A2 ? p.K1 : p.K, p.lda, ac, arps, stream);
            if (p.A2) imcui_range_check(h, p.A2 + (size_t)z * p.a2_bs, p.M, p.K - p.K1, p.lda2, ac, arps, stream);
            if (!p.Wh && p.W) imcui_range_check(h, p.W + (size_t)z * p.w_bs, p.N, p.K, p.ldw, p.ncnt ? p.ncnt + (size_t)z * p.cnt_stride : nullptr, p.N, stream);
        }
    } else if (split && h->range_flag) {
        imcui_range_check(h, p.A, (long)p.M / (p.conv_hout * p.conv_wout) * p.conv_hin * p.conv_win, p.conv_cin, p.conv_cin, nullptr, 0, stream);
    }
    if (p.ln_stats != nullptr && !(split && gemm_wreg_ok(h, p)))
        return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "gemm: a folded LayerNorm (ln_stats) needs the weights-in-registers kernel (epi %d, N=%d, K=%d)", p.epi, p.N, p.K);
    imcui_prof_begin(h, PROF_GEMM, stream);
    if (split && gemm_wreg_ok(h, p)) {
        gemm_wreg_launch(h, p, stream);
        imcui_prof_end(h, PROF_GEMM, stream);
        IMCUI_CHECK_LAUNCH(h);
        return IMCUI_OK;
    }
    switch (p.epi) {
        case EPI_BIAS: launch_one<EPI_BIAS>(p, split, stream); break;
        case EPI_RELU: launch_one<EPI_RELU>(p, split, stream); break;
        case EPI_RESID: launch_one<EPI_RESID>(p, split, stream); break;
        case EPI_QKV: launch_one<EPI_QKV>(p, split, stream); break;
        case EPI_CROSS: launch_one<EPI_CROSS>(p, split, stream); break;
        case EPI_QKV_VIT:
            return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "gemm: EPI_QKV_VIT is implemented by gemm_wreg_kernel only (split mode, pre-split weights, N %% (64 heads) == 0)");
        case EPI_NNSTAT:
            if (!split || p.Wh != nullptr || !p.st_rpm || !p.st_rps || !p.st_rpi || !p.st_cpm || !p.st_cps || !p.st_cpi || p.bias != nullptr || p.st_rpitch < p.M ||
                p.st_cpitch < p.N || p.st_nct < cdiv(p.N, BN) || p.st_nrh < 2 * cdiv(p.M, BM))
                return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: EPI_NNSTAT needs the split mode, an f32 B operand, no bias and the six partial buffers");
            hipLaunchKernelGGL((gemm_split_kernel<EPI_NNSTAT, 0, false, 2>), dim3(cdiv(p.M, BM) * cdiv(p.N, BN), 1, p.batch), dim3(256), 0, stream, p);
            break;
        case EPI_CONV:
            if (p.conv_k > 0)
                launch_conv(p, split, stream);
            else
                launch_one<EPI_CONV>(p, split, stream);
            break;
        default: return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: bad epilogue %d", p.epi);
    }
    imcui_prof_end(h, PROF_GEMM, stream);
    if (dbg) {
        const hipError_t e = hipStreamSynchronize(stream);
        fprintf(stderr, "[gemm] done: %s\n", hipGetErrorString(e));
        fflush(stderr);
    }
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ host-side weight split
// f32 -> f16 on the host, integer arithmetic (exhaustively identical, over all 2^32 inputs, to the float-compare formulation
// they replaced: truncate, then pick the nearer of the two neighbours, ties to even; inf / nan / overflow saturate to the
// largest finite value, like v_cvt_pkrtz_f16_f32).  mag = magnitude truncated toward zero, rem = the dropped bits, half = half a
// unit of the last kept place.
static inline void f16_parts(float f, unsigned& sign, unsigned& mag, unsigned& rem, unsigned& half) {
    unsigned x;
    memcpy(&x, &f, 4);
    sign = (x >> 16) & 0x8000u;
    const unsigned ex = (x >> 23) & 0xFFu;
    const int e = (int)ex - 112;
    unsigned m = x & 0x7FFFFFu;
    rem = 0;
    half = 0;
    if (ex == 0xFFu || e >= 31) {
        mag = 0x7BFFu;
        return;
    }
    if (e <= 0) {
        if (e < -10) {  // below half of the smallest subnormal: rounds to zero either way
            mag = 0;
            rem = (ex == 0 && m == 0) ? 0u : 1u;
            half = 2u;
            return;
        }
        m |= 0x800000u;
        const int sh = 14 - e;  // 14 .. 24
        mag = m >> sh;
        rem = m & ((1u << sh) - 1u);
        half = 1u << (sh - 1);
        return;
    }
    mag = ((unsigned)e << 10) | (m >> 13);
    rem = m & 0x1FFFu;
    half = 0x1000u;
}
static inline unsigned short f32_to_f16_rtn(float f) {
    unsigned s, g, r, h;
    f16_parts(f, s, g, r, h);
    if (g < 0x7BFFu && (r > h || (r == h && (g & 1u)))) g += 1u;
    return (unsigned short)(s | g);
}
static inline float f16_to_f32(unsigned short hv) {
    const unsigned sign = (hv & 0x8000u) << 16;
    int e = (hv >> 10) & 0x1F;
    unsigned m = hv & 0x3FFu;
    unsigned x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            e = 1;
            while (!(m & 0x400u)) {
                m <<= 1;
                --e;
            }
            m &= 0x3FFu;
            x = sign | ((unsigned)(e + 127 - 15) << 23) | (m << 13);
        }
    } else {
        x = sign | ((unsigned)(e + 127 - 15) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}
// scale 2^e that puts max |w| into [4096, 8192] (keeps the low parts out of the f16 subnormal range); e in [-8, 24]
static inline int split_scale_exp(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    int e = 0;
    if (mx > 0.f) {
        e = (int)floorf(log2f(8192.0f / mx));
        if (e > 24) e = 24;
        if (e < -8) e = -8;
    }
    return e;
}
// Weights: hi = the NEAREST f16 (round to nearest even), lo = the nearest f16 of the exact remainder (|lo| <= half an ulp of hi).
// Either rounding of hi gives an fp32-grade three-product split; nearest also makes the hi plane alone -- what the single-product
// arithmetic of the DUSt3R / EfficientLoFTR "fp16" options multiplies -- an UNBIASED 11-bit operand (round 2 truncated it toward
// zero: a coherent shrink that grew through the ~40 sequential layers of the ViT and left that mode slightly farther from the fp32
// result than a bf16-autocast run; with nearest rounding on both operands it is closer, tests/test_gpu_dust3r.py anchor).
static inline void split_one(float x, unsigned short& hi, unsigned short& lo) {
    hi = f32_to_f16_rtn(x);
    lo = f32_to_f16_rtn(x - f16_to_f32(hi));
}

float split_weights_frag_host(const float* w, int N, int K, unsigned short* hi, unsigned short* lo) {
    // planes [ceil(N/32)][K/16][2][32][8]: element (nf, ks, h, r, j) = W[nf*32 + r][ks*16 + h*8 + j]
    const int e = split_scale_exp(w, (size_t)N * K);
    const float sc = ldexpf(1.0f, e);
    const int nfr = (N + 31) / 32, nks = K / 16;
    for (int nf = 0; nf < nfr; ++nf)
        for (int r = 0; r < 32; ++r) {
            const int row = nf * 32 + r;
            for (int ks = 0; ks < nks; ++ks)
                for (int hh = 0; hh < 2; ++hh) {
                    const size_t dst = ((((size_t)nf * nks + ks) * 2 + hh) * 32 + r) * 8;
                    if (row < N) {
                        const float* src = w + (size_t)row * K + ks * 16 + hh * 8;
                        for (int j = 0; j < 8; ++j) split_one(src[j] * sc, hi[dst + j], lo[dst + j]);  // the scaling is exact (power of two)
                    } else {
                        for (int j = 0; j < 8; ++j) hi[dst + j] = lo[dst + j] = 0;
                    }
                }
        }
    return ldexpf(1.0f, -e);
}

void pack_conv_gemm(const float* w, int Cout, int Cin, int ks, int Cin_pad, float* dst) {
    // dst[co][tap][ci] = w[co][ci][tap], channels ci >= Cin are zero
    const int taps = ks * ks;
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < Cin_pad; ++ci)
                dst[((size_t)co * taps + t) * Cin_pad + ci] = (ci < Cin) ? w[((size_t)co * Cin + ci) * taps + t] : 0.0f;
}

float split_weights_host(const float* w, size_t n, unsigned short* hi, unsigned short* lo) {
    const int e = split_scale_exp(w, n);
    const float sc = ldexpf(1.0f, e);
    for (size_t i = 0; i < n; ++i) split_one(w[i] * sc, hi[i], lo[i]);  // exact scaling (power of two)
    return ldexpf(1.0f, -e);
}
