// MFMA GEMM for MI355X / gfx950 in two arithmetic modes.
//
// f32 mode: tile 128x128x32, v_mfma_f32_32x32x2_f32, operands staged global -> registers -> LDS
// as [k-quad][row][4 floats].  Split mode: tile 128x128x64, every f32 operand value is split into
// an f16 (hi, lo) pair while it is staged (3 VALU per element), LDS holds [k-octet][row][8 halves]
// images of hi and lo, and each 32x32x16 step issues ah*bh + ah*bl + al*bh.  Weights of the
// networks are pre-split (and pre-scaled by a power of two) at pack time.  In both modes the row
// stride of the LDS image is padded to 129 granules of 16 B, which makes the staging writes and
// the fragment reads bank-conflict free, and the next K tile is prefetched into registers while
// the current one is multiplied.  4 waves (2x2), each wave a 64x64 output tile (64 accumulators).
#include <math.h>
#include <string.h>

#include "gemm.h"

#define BM 128
#define BN 128
#define LDS_ROWS 129  // padded row count per k-granule (16-byte units)

struct TileCtx {
    int M, N, row0, col0, seq, z;
    const float* bias;
    long wsel;  // selected weight index (per-pair heads)
};

// returns false when the workgroup has nothing to do
__device__ __forceinline__ bool gemm_tile_setup(const GemmP& p, TileCtx& c) {
    c.z = blockIdx.z;
    c.M = p.mcnt ? p.mcnt[c.z * p.cnt_stride] : p.M;
    c.N = p.ncnt ? p.ncnt[c.z * p.cnt_stride] : p.N;
    const int ncol = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    c.row0 = (tile / ncol) * BM;
    c.col0 = (tile % ncol) * BN;
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

// Epilogue.  The products are issued with the WEIGHT fragment as the MFMA A operand and the
// activation fragment as B, i.e. the accumulators hold C^T: column (lane & 31) = token row,
// fragment row = output feature.  A lane therefore owns 4 CONSECUTIVE features of one token per
// register quad (r & 3), which makes every store a 16-byte store (a 64-dword-store epilogue is
// store-issue bound and was 10x the matrix time), puts RoPE pairs in one lane, and lets V^T be
// written with lanes running along the token axis.
//
// split_out projections (Q / K / V^T as f16 hi + lo planes for attn_split_kernel) go through LDS
// (`stage`, >= 73728 B, the GEMM's operand buffers): written straight from the fragments they are
// 8-byte (Q/K) or 2-byte (V^T) scattered stores and the kernel was store-bound at 15 % of the matrix
// rate; staged, the 128x128 tile leaves as 16-byte stores of fully contiguous 16 KiB (Q/K) or
// 256-byte (V^T) runs.
#define STAGE_QK_ROW 144    // bytes per staged [token][64 halves] row (128 + 16 pad)
#define STAGE_QK_PLANE (2 * 128 * STAGE_QK_ROW)
#define STAGE_VT_ROW 272    // bytes per staged [feature][128 token halves] row (256 + 16 pad)
#define STAGE_VT_PLANE (128 * STAGE_VT_ROW)
#define STAGE_BYTES 73728
#define STAGE_C_ROW 132     // floats per staged [token][128 features] row of the f32 outputs
#define STAGE_C_BYTES (128 * STAGE_C_ROW * 4)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const TileCtx& c, f32x16 (&acc)[2][2], float wsc, int wm,
                                              int wn, int lo, int hi, uint4* stage = nullptr) {
    if ((EPI == EPI_QKV || EPI == EPI_CROSS) && stage != nullptr && p.split_out) {
        // the 128 columns of a tile are two heads of ONE of q / k / v (256 features each)
        const int t = c.col0 >> 8, hd0 = (c.col0 >> 6) & 3;
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
        char* sb = reinterpret_cast<char*>(stage);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int tl = wm * 64 + m * 32 + lo;  // token within the tile
            const int row = c.row0 + tl;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int fl = wn * 64 + n * 32 + 8 * q + 4 * hi;  // first of 4 features within the tile
                    const int f0 = c.col0 + fl;
                    float v[4] = {acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                  acc[m][n][4 * q + 3] * wsc};
                    if (c.bias != nullptr) {
                        const float4 b4 = *reinterpret_cast<const float4*>(c.bias + f0);
                        v[0] += b4.x;
                        v[1] += b4.y;
                        v[2] += b4.z;
                        v[3] += b4.w;
                    }
                    if (rope) {
                        const int d0 = fl & 63;
                        const float2 cs = *reinterpret_cast<const float2*>(p.rope_cos + (size_t)row * 32 + (d0 >> 1));
                        const float2 sn = *reinterpret_cast<const float2*>(p.rope_sin + (size_t)row * 32 + (d0 >> 1));
                        const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                        v[0] = a0 * cs.x + (-a1) * sn.x;
                        v[1] = a1 * cs.x + a0 * sn.x;
                        v[2] = a2 * cs.y + (-a3) * sn.y;
                        v[3] = a3 * cs.y + a2 * sn.y;
                    }
                    if (scale) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                    }
                    unsigned h01, l01, h23, l23;
                    split2(v[0], v[1], h01, l01);
                    split2(v[2], v[3], h23, l23);
                    if (!vt) {
                        char* o = sb + ((fl >> 6) * 128 + tl) * STAGE_QK_ROW + (fl & 63) * 2;
                        *reinterpret_cast<uint2*>(o) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(o + STAGE_QK_PLANE) = make_uint2(l01, l23);
                    } else {
                        // pair lanes (tokens 2u, 2u+1): the even lane ends up with features f, f+1 of both
                        // tokens, the odd lane with f+2, f+3 -> 4-byte [feature][token pair] words
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
                        char* o = sb + (fl + (odd ? 2 : 0)) * STAGE_VT_ROW + (tl >> 1) * 4;
                        *reinterpret_cast<unsigned*>(o) = wh0;
                        *reinterpret_cast<unsigned*>(o + STAGE_VT_ROW) = wh1;
                        *reinterpret_cast<unsigned*>(o + STAGE_VT_PLANE) = wl0;
                        *reinterpret_cast<unsigned*>(o + STAGE_VT_PLANE + STAGE_VT_ROW) = wl1;
                    }
                }
            }
        }
        __syncthreads();
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
        const int i0 = c.row0 - c.seq * p.rows_per_seq;
        const int tid = threadIdx.x;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int g = it * 256 + tid;  // 16-byte granule: 2 planes x 2048
            const int plane = g >> 11, rem = g & 2047;
            if (!vt) {
                const int hh = rem >> 10, tok = (rem >> 3) & 127, gr = rem & 7;
                const uint4 val = *reinterpret_cast<const uint4*>(sb + plane * STAGE_QK_PLANE + (hh * 128 + tok) * STAGE_QK_ROW + gr * 16);
                unsigned short* o = d16 + (size_t)plane * p.plane_halves +
                                    (((size_t)c.seq * p.heads + hd0 + hh) * p.rows_per_seq + i0 + tok) * 64 + gr * 8;
                *reinterpret_cast<uint4*>(o) = val;
            } else {
                const int feat = rem >> 4, gr = rem & 15;
                const uint4 val = *reinterpret_cast<const uint4*>(sb + plane * STAGE_VT_PLANE + feat * STAGE_VT_ROW + gr * 16);
                unsigned short* o = d16 + (size_t)plane * p.plane_halves +
                                    (((size_t)c.seq * p.heads + hd0 + (feat >> 6)) * 64 + (feat & 63)) * p.rows_per_seq + i0 + gr * 8;
                *reinterpret_cast<uint4*>(o) = val;
            }
        }
        return;
    }
    float* C = p.C ? p.C + (size_t)c.z * p.c_bs : nullptr;
    const bool vec_ok = ((p.ldc & 3) == 0);
    if ((EPI == EPI_BIAS || EPI == EPI_RELU || EPI == EPI_RESID || EPI == EPI_CONV) && stage != nullptr) {
        // Row-major outputs: a fragment store touches 32 token rows x 32 B, which the memory system
        // serves at a third of the rate of whole lines (the epilogue was a third of the kernel).
        // Transpose the tile through LDS instead ([token][feature], 132-float rows: conflict-free
        // both ways) and let every wave write whole 512-byte rows.
        float* st = reinterpret_cast<float*>(stage);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tl = wm * 64 + m * 32 + lo, fl = wn * 64 + n * 32 + 8 * q + 4 * hi;
                    *reinterpret_cast<float4*>(st + tl * STAGE_C_ROW + fl) =
                        make_float4(acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                    acc[m][n][4 * q + 3] * wsc);
                }
        __syncthreads();
        const int tid = threadIdx.x;
        const int f0 = c.col0 + 4 * (tid & 31);
        if (f0 >= c.N) return;
        const bool full = (f0 + 3 < c.N) && vec_ok;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (c.bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < c.N) b[j] = c.bias[f0 + j];
        }
        const bool res_vec = (EPI == EPI_CONV) && p.resid != nullptr && ((p.ldr & 3) == 0);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int tl = (tid >> 5) + 8 * it;
            const int row = c.row0 + tl;
            if (row >= c.M) break;
            const float4 t4 = *reinterpret_cast<const float4*>(st + tl * STAGE_C_ROW + 4 * (tid & 31));
            float v[4] = {t4.x + b[0], t4.y + b[1], t4.z + b[2], t4.w + b[3]};
            float* dst = C + (size_t)row * p.ldc + f0;
            if (EPI == EPI_CONV) {
                if (p.resid != nullptr) {
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
                            if (f0 + j < c.N) v[j] += rs[j];
                    }
                }
                if (p.act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                } else if (p.act == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.01f * v[j];
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
                    if (f0 + j < c.N) dst[j] = (EPI == EPI_RESID) ? dst[j] + v[j] : v[j];
            }
        }
        return;
    }
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

// ------------------------------------------------------------------ 3 x f16 split kernel
#define BK64 64
// PRESPLIT: B operand comes from pre-split f16 planes (network weights); otherwise it is an
// f32 activation matrix split on the fly like A (similarity products).
template <int EPI, bool PRESPLIT>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmP p) {
    // A_hi | A_lo | B_hi | B_lo, 16512 B each; the split_out epilogues restage the tile in it (73728 B)
    constexpr int SMEM_U4 = (EPI == EPI_QKV || EPI == EPI_CROSS) ? STAGE_BYTES / 16 : STAGE_C_BYTES / 16;
    static_assert(SMEM_U4 >= 4 * (BK64 / 8) * LDS_ROWS, "operand buffers must fit");
    __shared__ uint4 smem[SMEM_U4];
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
    const float* A = p.A + (size_t)c.z * p.a_bs;
    const float* A2 = p.A2 ? p.A2 + (size_t)c.z * p.a2_bs : nullptr;
    const float* W = PRESPLIT ? nullptr : p.W + (size_t)c.z * p.w_bs + (size_t)c.wsel * p.w_stride;
    const unsigned short* Wh = PRESPLIT ? p.Wh + (size_t)c.wsel * p.w_stride : nullptr;
    const unsigned short* Wl = PRESPLIT ? p.Wl + (size_t)c.wsel * p.w_stride : nullptr;
    const float wsc = (PRESPLIT && p.wscale) ? p.wscale[c.wsel] : 1.0f;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // staging: 128 rows x 8 k-octets per operand = 1024 items of 8 values, 4 per thread
    const int s_ko = tid & 7;
    const int s_r = tid >> 3;
    // A: two float4 per item (split while staged); B: hi/lo planes (pre-split weights, pure copy)
    // or two float4 (activations, split while staged) held as raw bits in the same registers.
    // The A operand (activations streamed from HBM / MALL) has TWO register sets (X, Y): tile kt+2
    // is requested before tile kt is multiplied, so its load has a whole stage + multiply phase
    // (> 3k cycles) to land.  The B operand (weights, L2-resident) is fetched one tile ahead.
    float4 raX0a, raX0b, raX1a, raX1b, raX2a, raX2b, raX3a, raX3b, raY0a, raY0b, raY1a, raY1b, raY2a, raY2b, raY3a, raY3b;
    uint4 rb0a, rb0b, rb1a, rb1b, rb2a, rb2b, rb3a, rb3b;
    const int nkt = p.K / BK64;
#define LDA64(S, it)                                                                 \
    {                                                                                \
        const int ar = min(c.row0 + s_r + 32 * it, c.M - 1);                         \
        const float* src = gemm_a_src(p, A, A2, ar, k);                                  \
        ra##S##it##a = src ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);     \
        ra##S##it##b = src ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }
#define LDB64(it)                                                                    \
    {                                                                                \
        const int br = min(c.col0 + s_r + 32 * it, c.N - 1);                         \
        if (PRESPLIT) {                                                              \
            rb##it##a = *reinterpret_cast<const uint4*>(Wh + (size_t)br * p.ldw + k); \
            rb##it##b = *reinterpret_cast<const uint4*>(Wl + (size_t)br * p.ldw + k); \
        } else {                                                                     \
            const float* ws = W + (size_t)br * p.ldw + k;                            \
            rb##it##a = *reinterpret_cast<const uint4*>(ws);                         \
            rb##it##b = *reinterpret_cast<const uint4*>(ws + 4);                     \
        }                                                                            \
    }
#define ST64(S, it)                                                                  \
    {                                                                                \
        const int o = s_ko * LDS_ROWS + s_r + 32 * it;                               \
        uint4 h, l;                                                                  \
        split8(ra##S##it##a, ra##S##it##b, h, l);                                    \
        Ah[o] = h;                                                                   \
        Al[o] = l;                                                                   \
        if (PRESPLIT) {                                                              \
            Bh[o] = rb##it##a;                                                       \
            Bl[o] = rb##it##b;                                                       \
        } else {                                                                     \
            split8(__builtin_bit_cast(float4, rb##it##a), __builtin_bit_cast(float4, rb##it##b), h, l); \
            Bh[o] = h;                                                               \
            Bl[o] = l;                                                               \
        }                                                                            \
    }
#define LOADA(S, kt_)                                    \
    {                                                    \
        const int k = (kt_) * BK64 + s_ko * 8;           \
        LDA64(S, 0) LDA64(S, 1) LDA64(S, 2) LDA64(S, 3)  \
    }
#define LOADB(kt_)                               \
    {                                            \
        const int k = (kt_) * BK64 + s_ko * 8;   \
        LDB64(0) LDB64(1) LDB64(2) LDB64(3)      \
    }
#define STORESET(S) ST64(S, 0) ST64(S, 1) ST64(S, 2) ST64(S, 3)

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
                // weight fragment = MFMA A operand (rows = features), activations = B (cols = tokens)
                acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
            }
    }
    };

    LOADA(X, 0)
    LOADB(0)
    if (nkt > 1) LOADA(Y, 1)
    for (int kt = 0; kt < nkt; kt += 2) {
        STORESET(X)
        __syncthreads();
        if (kt + 1 < nkt) LOADB(kt + 1)
        if (kt + 2 < nkt) LOADA(X, kt + 2)
        compute();
        __syncthreads();
        if (kt + 1 < nkt) {
            STORESET(Y)
            __syncthreads();
            if (kt + 2 < nkt) LOADB(kt + 2)
            if (kt + 3 < nkt) LOADA(Y, kt + 3)
            compute();
            __syncthreads();
        }
    }
    gemm_epilogue<EPI>(p, c, acc, wsc, wm, wn, lo, hi, smem);
}

template <int EPI>
static void launch_one(const GemmP& p, bool split, dim3 grid, hipStream_t stream) {
    if (!split)
        hipLaunchKernelGGL(gemm_kernel<EPI>, grid, dim3(256), 0, stream, p);
    else if (p.Wh != nullptr)
        hipLaunchKernelGGL((gemm_split_kernel<EPI, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_split_kernel<EPI, false>), grid, dim3(256), 0, stream, p);
}

int gemm_launch(imcui_hip_s* h, const GemmP& p, hipStream_t stream) {
    if (p.K % BK32 != 0 || p.K <= 0) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K=%d must be a positive multiple of %d", p.K, BK32);
    if (p.A2 && (p.K1 % BK64 != 0)) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: K1=%d must be a multiple of %d", p.K1, BK64);
    if (p.rows_per_seq > 0 && p.rows_per_seq % BM != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: rows_per_seq=%d must be a multiple of %d", p.rows_per_seq, BM);
    if (p.M <= 0 || p.N <= 0) return IMCUI_OK;
    if (p.conv_k > 0 && (p.conv_cin % BK64 != 0 || p.K != p.conv_k * p.conv_k * p.conv_cin))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm(conv): cin=%d must be a multiple of %d and K=%d == k*k*cin", p.conv_cin, BK64, p.K);
    const bool split = h->precision == 1 && (p.K % BK64 == 0);
    if (p.split_out && (!split || p.rows_per_seq <= 0 || p.N % BN != 0 || p.M % BM != 0 || p.batch != 1))
        return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: split_out needs the split mode and whole 128x128 tiles (M=%d N=%d)", p.M, p.N);
    if (!split && p.W == nullptr) return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: f32 weights missing");
    const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    dim3 grid(ntiles, 1, p.batch);
    imcui_prof_begin(h, PROF_GEMM, stream);
    switch (p.epi) {
        case EPI_BIAS: launch_one<EPI_BIAS>(p, split, grid, stream); break;
        case EPI_RELU: launch_one<EPI_RELU>(p, split, grid, stream); break;
        case EPI_RESID: launch_one<EPI_RESID>(p, split, grid, stream); break;
        case EPI_QKV: launch_one<EPI_QKV>(p, split, grid, stream); break;
        case EPI_CROSS: launch_one<EPI_CROSS>(p, split, grid, stream); break;
        case EPI_CONV: launch_one<EPI_CONV>(p, split, grid, stream); break;
        default: return imcui_set_err(h, IMCUI_ERR_ARG, "gemm: bad epilogue %d", p.epi);
    }
    imcui_prof_end(h, PROF_GEMM, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ host-side weight split
static unsigned short f32_to_f16_rtz(float f) {
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned sign = (x >> 16) & 0x8000u;
    const int e = (int)((x >> 23) & 0xFF) - 127 + 15;
    unsigned m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (unsigned short)(sign | 0x7BFFu);  // inf/nan -> max finite
    if (e >= 31) return (unsigned short)(sign | 0x7BFFu);                      // saturate
    if (e <= 0) {
        if (e < -10) return (unsigned short)sign;
        m |= 0x800000u;
        return (unsigned short)(sign | (m >> (14 - e)));  // subnormal, truncate
    }
    return (unsigned short)(sign | (e << 10) | (m >> 13));
}
static float f16_to_f32(unsigned short hv) {
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
static unsigned short f32_to_f16_rtn(float f) {
    // round to nearest even via the truncated value and its successor
    const unsigned short t = f32_to_f16_rtz(f);
    if ((t & 0x7FFFu) >= 0x7BFFu) return t;
    const unsigned short u = (unsigned short)(t + 1);  // next magnitude, same sign
    const float ft = f16_to_f32(t), fu = f16_to_f32(u);
    const float dt = fabsf(f - ft), du = fabsf(fu - f);
    if (dt < du) return t;
    if (du < dt) return u;
    return (t & 1) ? u : t;
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
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    int e = 0;
    if (mx > 0.f) {
        e = (int)floorf(log2f(8192.0f / mx));  // max |w| * 2^e in [4096, 8192]
        if (e > 24) e = 24;
        if (e < -8) e = -8;
    }
    const float sc = ldexpf(1.0f, e);
    for (size_t i = 0; i < n; ++i) {
        const float x = w[i] * sc;  // exact (power of two)
        const unsigned short hh = f32_to_f16_rtz(x);
        hi[i] = hh;
        lo[i] = f32_to_f16_rtn(x - f16_to_f32(hh));
    }
    return ldexpf(1.0f, -e);
}
