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

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const TileCtx& c, f32x16 (&acc)[2][2], float wsc, int wm,
                                              int wn, int lo, int hi) {
    float* C = p.C ? p.C + (size_t)c.z * p.c_bs : nullptr;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = c.col0 + wn * 64 + n * 32 + lo;
        const bool colok = col < c.N;
        const float bv = (c.bias != nullptr && colok) ? c.bias[col] : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.row0 + wm * 64 + m * 32 + frag_row(r, hi);
                const bool ok = colok && row < c.M;
                float v = acc[m][n][r] * wsc + bv;
                if (EPI == EPI_BIAS) {
                    if (ok) C[(size_t)row * p.ldc + col] = v * p.alpha;
                } else if (EPI == EPI_RELU) {
                    if (ok) C[(size_t)row * p.ldc + col] = fmaxf(v, 0.0f);
                } else if (EPI == EPI_RESID) {
                    if (ok) C[(size_t)row * p.ldc + col] += v;
                } else if (EPI == EPI_QKV || EPI == EPI_CROSS) {
                    // col = t*256 + head*64 + d (weights were de-interleaved at pack time)
                    const int t = col >> 8, hd = (col >> 6) & 3, d = col & 63;
                    const int i = row - c.seq * p.rows_per_seq;
                    float* dst;
                    bool vt = false;
                    if (EPI == EPI_QKV) {
                        const float partner = __shfl_xor(v, 1, 64);  // (d ^ 1) of the same row
                        if (t < 2) {
                            const float cs = p.rope_cos[(size_t)row * 32 + (d >> 1)];
                            const float sn = p.rope_sin[(size_t)row * 32 + (d >> 1)];
                            // x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0)
                            v = v * cs + ((d & 1) ? partner : -partner) * sn;
                            if (t == 0) v *= p.alpha;
                        }
                        dst = (t == 0) ? p.Q : (t == 1) ? p.Kt : p.V;
                        vt = (t == 2);
                    } else {
                        if (t == 0) v *= p.alpha;
                        dst = (t == 0) ? p.Q : p.V;
                        vt = (t == 1);
                    }
                    if (ok) {
                        if (vt && p.v_transposed)
                            dst[(((size_t)c.seq * p.heads + hd) * 64 + d) * p.rows_per_seq + i] = v;
                        else
                            dst[(((size_t)c.seq * p.heads + hd) * p.rows_per_seq + i) * 64 + d] = v;
                    }
                }
            }
        }
    }
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
    const int nkt = p.K / BK32;

    auto load_tile = [&](int kt) {
        const int k = kt * BK32 + s_kq * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = s_r + 32 * it;
            const int ar = min(c.row0 + r, c.M - 1);
            const int br = min(c.col0 + r, c.N - 1);
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
                    acc[m][n] = mfma32(a[m].x, b[n].x, acc[m][n]);
                    acc[m][n] = mfma32(a[m].y, b[n].y, acc[m][n]);
                    acc[m][n] = mfma32(a[m].z, b[n].z, acc[m][n]);
                    acc[m][n] = mfma32(a[m].w, b[n].w, acc[m][n]);
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
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmP p) {
    __shared__ uint4 smem[4 * (BK64 / 8) * LDS_ROWS];  // A_hi | A_lo | B_hi | B_lo, 16512 B each
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
    float4 ra[4][2];
    float4 rbf[PRESPLIT ? 1 : 4][2];
    uint4 rbh[PRESPLIT ? 4 : 1], rbl[PRESPLIT ? 4 : 1];
    const int nkt = p.K / BK64;

    auto load_tile = [&](int kt) {
        const int k = kt * BK64 + s_ko * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = s_r + 32 * it;
            const int ar = min(c.row0 + r, c.M - 1);
            const int br = min(c.col0 + r, c.N - 1);
            const float* src;
            if (p.A2 != nullptr && k >= p.K1)
                src = p.A2 + (size_t)ar * p.lda2 + (k - p.K1);
            else
                src = A + (size_t)ar * p.lda + k;
            ra[it][0] = *reinterpret_cast<const float4*>(src);
            ra[it][1] = *reinterpret_cast<const float4*>(src + 4);
            if (PRESPLIT) {
                rbh[it] = *reinterpret_cast<const uint4*>(Wh + (size_t)br * p.ldw + k);
                rbl[it] = *reinterpret_cast<const uint4*>(Wl + (size_t)br * p.ldw + k);
            } else {
                const float* ws = W + (size_t)br * p.ldw + k;
                rbf[it][0] = *reinterpret_cast<const float4*>(ws);
                rbf[it][1] = *reinterpret_cast<const float4*>(ws + 4);
            }
        }
    };

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int o = s_ko * LDS_ROWS + s_r + 32 * it;
            uint4 h, l;
            split8(ra[it][0], ra[it][1], h, l);
            Ah[o] = h;
            Al[o] = l;
            if (PRESPLIT) {
                Bh[o] = rbh[it];
                Bl[o] = rbl[it];
            } else {
                split8(rbf[it][0], rbf[it][1], h, l);
                Bh[o] = h;
                Bl[o] = l;
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
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
                    acc[m][n] = mfma16(al[m], bh[n], acc[m][n]);
                    acc[m][n] = mfma16(ah[m], bl[n], acc[m][n]);
                    acc[m][n] = mfma16(ah[m], bh[n], acc[m][n]);
                }
        }
        __syncthreads();
    }
    gemm_epilogue<EPI>(p, c, acc, wsc, wm, wn, lo, hi);
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
    const bool split = h->precision == 1 && (p.K % BK64 == 0);
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
