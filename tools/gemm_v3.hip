// Lab: candidate split GEMM v3 -- BK=32, LDS double-buffered stages, weights by LDS-DMA from
// fragment-major planes, activations split in registers and written XOR-swizzled.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm.hip"

void imcui_prof_begin(imcui_hip_s*, int, hipStream_t) {}
void imcui_prof_end(imcui_hip_s*, int, hipStream_t) {}
int imcui_set_err(imcui_hip_s*, int code, const char* fmt, ...) { return code; }

#define BK3 32
// stage layout (bytes): A_hi 8K | A_lo 8K | B_hi 8K | B_lo 8K ; fragment (ks, rf) at ((ks*4+rf) * 1024), lane granule 16 B
#define STG_BYTES 32768
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// KO: 1 no MFMA (fragments xor-consumed), 2 no A loads in loop, 4 no B DMA in loop, 8 no epilogue, 16 no A split/LDS store in loop
template <int EPI, int KO = 0>
__global__ __launch_bounds__(256, 2) void gemm_v3_kernel(GemmP p) {
    __shared__ uint4 smem[STAGE_BYTES / 16];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    TileCtx c;
    if (!gemm_tile_setup(p, c)) return;
    if (KO & 32) {  // desynchronise the co-resident workgroups
        const int dly = (KO & 64) ? (blockIdx.x >> 3) & 3 : (blockIdx.x >> 11) & 1;
        for (int i = 0; i < dly * ((KO & 64) ? 1 : 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const float wsc = p.wscale ? p.wscale[c.wsel] : 1.0f;
    const int nkt = p.K / BK3;
    const int nks = p.K >> 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // ---- A staging: thread -> (granule g = tid & 3, rows (tid >> 2) and +64); one granule = 8 consecutive k
    const int g = tid & 3;
    const int ar0 = min(c.row0 + (tid >> 2), c.M - 1), ar1 = min(c.row0 + 64 + (tid >> 2), c.M - 1);
    const float* pa0 = p.A + (size_t)ar0 * p.lda + g * 8;
    const float* pa1 = p.A + (size_t)ar1 * p.lda + g * 8;
    // LDS byte offset of the granule: fragment (ks = g >> 1, rf), half hi = g & 1, position lo ^ (2 g)
    const int rl0 = tid >> 2, rl1 = 64 + (tid >> 2);
    const int wo0 = (((g >> 1) * 4 + (rl0 >> 5)) * 64 + (g & 1) * 32 + ((rl0 & 31) ^ (2 * g))) * 16;
    const int wo1 = (((g >> 1) * 4 + (rl1 >> 5)) * 64 + (g & 1) * 32 + ((rl1 & 31) ^ (2 * g))) * 16;
    f32x4 xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1;
#define GLD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define LDA3(S, kt_)                                    \
    {                                                   \
        const float* q0 = pa0 + (kt_) * BK3;            \
        const float* q1 = pa1 + (kt_) * BK3;            \
        GLD4(S##a0, q0);                                \
        GLD4(S##b0, q0 + 4);                            \
        GLD4(S##a1, q1);                                \
        GLD4(S##b1, q1 + 4);                            \
    }
#define VMWAIT(n, S) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(S##a0), "+v"(S##b0), "+v"(S##a1), "+v"(S##b1)::"memory")
#define STA3(S, stg)                                                        \
    {                                                                       \
        uint4 h, l;                                                         \
        split8(__builtin_bit_cast(float4, S##a0), __builtin_bit_cast(float4, S##b0), h, l); \
        *reinterpret_cast<uint4*>(sm + (stg) * STG_BYTES + wo0) = h;        \
        *reinterpret_cast<uint4*>(sm + (stg) * STG_BYTES + 8192 + wo0) = l; \
        split8(__builtin_bit_cast(float4, S##a1), __builtin_bit_cast(float4, S##b1), h, l); \
        *reinterpret_cast<uint4*>(sm + (stg) * STG_BYTES + wo1) = h;        \
        *reinterpret_cast<uint4*>(sm + (stg) * STG_BYTES + 8192 + wo1) = l; \
    }
    // ---- B by LDS-DMA: 16 fragments per k-tile (plane, ks, nf); wave w moves fragments w, w+4, w+8, w+12
    // -> plane = j >> 1 ... ordered f = plane*8 + ks*4 + nf ; wave handles nf = wid for all (plane, ks)
    const int nfr = (p.N + 31) >> 5;
    const int nfg = min((c.col0 >> 5) + wid, nfr - 1);
    const uint4* wh = reinterpret_cast<const uint4*>(p.Wh + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
    const uint4* wl = reinterpret_cast<const uint4*>(p.Wl + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
#define GLDS(gptr, ldsaddr)                                                                                       \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep)                                                                                \
                     : "v"(gptr), "s"(ldsaddr)                                                                    \
                     : "memory");                                                                                 \
    }
#define DMAB(kt_, stg)                                                                            \
    {                                                                                             \
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + (stg) * STG_BYTES + 16384 + wid * 1024); \
        GLDS(wh + (size_t)((kt_) * 2 + 0) * 64, d);                                               \
        GLDS(wh + (size_t)((kt_) * 2 + 1) * 64, d + 4096);                                        \
        GLDS(wl + (size_t)((kt_) * 2 + 0) * 64, d + 8192);                                        \
        GLDS(wl + (size_t)((kt_) * 2 + 1) * 64, d + 8192 + 4096);                                 \
    }
    // ---- fragment read offsets
    // A fragment (ks, rf = wm*2 + m): granule position lo ^ (2 * (2 ks + hi))
    // B fragment (ks, nf = wn*2 + n): lane-linear
    auto compute = [&](int stg) __attribute__((always_inline)) {
        const char* s0 = sm + stg * STG_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 ah[2], al[2], bh[2], bl[2];
            const int apos = (hi * 32 + (lo ^ (2 * (2 * ks + hi)))) * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int fo = (ks * 4 + wm * 2 + m) * 1024 + apos;
                ah[m] = *reinterpret_cast<const uint4*>(s0 + fo);
                al[m] = *reinterpret_cast<const uint4*>(s0 + 8192 + fo);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int fo = (ks * 4 + wn * 2 + n) * 1024 + lane * 16;
                bh[n] = *reinterpret_cast<const uint4*>(s0 + 16384 + fo);
                bl[n] = *reinterpret_cast<const uint4*>(s0 + 24576 + fo);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (KO & 1) {
                        const uint4 xx = make_uint4(bh[n].x ^ al[m].x ^ bl[n].x ^ ah[m].x, bh[n].y ^ al[m].y ^ bl[n].y ^ ah[m].y,
                                                    bh[n].z ^ al[m].z ^ bl[n].z ^ ah[m].z, bh[n].w ^ al[m].w ^ bl[n].w ^ ah[m].w);
                        acc[m][n][0] += __builtin_bit_cast(float, xx.x);
                        acc[m][n][1] += __builtin_bit_cast(float, xx.y);
                        acc[m][n][2] += __builtin_bit_cast(float, xx.z);
                        acc[m][n][3] += __builtin_bit_cast(float, xx.w);
                    } else {
                        acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                        acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                        acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                    }
                }
        }
    };

    // prologue: tile 0 -> stage 0, tile 1 in flight.  All loop loads are issued from inline asm so
    // that the waits can be counted by hand (the compiler would drain vmcnt to 0 at every use).
    LDA3(x, 0)
    DMAB(0, 0)
    if (nkt > 1) {
        LDA3(y, 1)
        VMWAIT(4, x);
    } else {
        VMWAIT(0, x);
    }
    STA3(x, 0)
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // even tile in stage 0; y holds tile kt+1 (in flight)
        if (!(KO & 4)) if (kt + 1 < nkt) DMAB(kt + 1, 1)
        if (!(KO & 2)) if (kt + 2 < nkt) LDA3(x, kt + 2)
        compute(0);
        if (kt + 1 < nkt) {
            if (KO & 2) {
                if (KO & 4) { VMWAIT(0, y); } else { VMWAIT(0, y); }
            } else if (KO & 4) {
                if (kt + 2 < nkt) { VMWAIT(4, y); } else { VMWAIT(0, y); }
            } else if (kt + 2 < nkt) {
                VMWAIT(4, y);
            } else {
                VMWAIT(0, y);
            }
            if (!(KO & 16)) STA3(y, 1)
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            if (!(KO & 4)) if (kt + 2 < nkt) DMAB(kt + 2, 0)
            if (!(KO & 2)) if (kt + 3 < nkt) LDA3(y, kt + 3)
            compute(1);
            if (kt + 2 < nkt) {
                if (KO & 2) {
                    VMWAIT(0, x);
                } else if (kt + 3 < nkt) {
                    VMWAIT(4, x);
                } else {
                    VMWAIT(0, x);
                }
                if (!(KO & 16)) STA3(x, 0)
            }
            __syncthreads();
        }
    }
    if (KO & 8) {
        float sacc = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[m][n][r];
        if (sacc == 1.2345f) p.C[tid] = sacc;
        return;
    }
    gemm_epilogue<EPI>(p, c, acc, wsc, wm, wn, lo, hi, smem);
}

// KO: 1 no MFMA (fragments xor-consumed), 2 no A loads in loop, 4 no B DMA in loop, 8 no epilogue, 16 no A split/LDS store in loop
template <int EPI, int KO = 0>
__global__ __launch_bounds__(256, 3) void gemm_v4_kernel(GemmP p) {
    __shared__ uint4 smem[49152 / 16];  // A_hi 8K | A_lo 8K | B stage 0 (hi 8K, lo 8K) | B stage 1
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    TileCtx c;
    if (!gemm_tile_setup(p, c)) return;
    if (KO & 32) {  // desynchronise the co-resident workgroups
        const int dly = (KO & 64) ? (blockIdx.x >> 3) & 3 : (blockIdx.x >> 11) & 1;
        for (int i = 0; i < dly * ((KO & 64) ? 1 : 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const float wsc = p.wscale ? p.wscale[c.wsel] : 1.0f;
    const int nkt = p.K / BK3;
    const int nks = p.K >> 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // ---- A staging: thread -> (granule g = tid & 3, rows (tid >> 2) and +64); one granule = 8 consecutive k
    const int g = tid & 3;
    const int ar0 = min(c.row0 + (tid >> 2), c.M - 1), ar1 = min(c.row0 + 64 + (tid >> 2), c.M - 1);
    const float* pa0 = p.A + (size_t)ar0 * p.lda + g * 8;
    const float* pa1 = p.A + (size_t)ar1 * p.lda + g * 8;
    // LDS byte offset of the granule: fragment (ks = g >> 1, rf), half hi = g & 1, position lo ^ (2 g)
    const int rl0 = tid >> 2, rl1 = 64 + (tid >> 2);
    const int wo0 = (((g >> 1) * 4 + (rl0 >> 5)) * 64 + (g & 1) * 32 + ((rl0 & 31) ^ (2 * g))) * 16;
    const int wo1 = (((g >> 1) * 4 + (rl1 >> 5)) * 64 + (g & 1) * 32 + ((rl1 & 31) ^ (2 * g))) * 16;
    f32x4 xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1;
#define GLD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define LDA3(S, kt_)                                    \
    {                                                   \
        const float* q0 = pa0 + (kt_) * BK3;            \
        const float* q1 = pa1 + (kt_) * BK3;            \
        GLD4(S##a0, q0);                                \
        GLD4(S##b0, q0 + 4);                            \
        GLD4(S##a1, q1);                                \
        GLD4(S##b1, q1 + 4);                            \
    }
#define VMWAIT(n, S) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(S##a0), "+v"(S##b0), "+v"(S##a1), "+v"(S##b1)::"memory")
#define STA3(S, stg)                                                        \
    {                                                                       \
        uint4 h, l;                                                         \
        split8(__builtin_bit_cast(float4, S##a0), __builtin_bit_cast(float4, S##b0), h, l); \
        *reinterpret_cast<uint4*>(sm + wo0) = h;        \
        *reinterpret_cast<uint4*>(sm + 8192 + wo0) = l; \
        split8(__builtin_bit_cast(float4, S##a1), __builtin_bit_cast(float4, S##b1), h, l); \
        *reinterpret_cast<uint4*>(sm + wo1) = h;        \
        *reinterpret_cast<uint4*>(sm + 8192 + wo1) = l; \
    }
    // ---- B by LDS-DMA: 16 fragments per k-tile (plane, ks, nf); wave w moves fragments w, w+4, w+8, w+12
    // -> plane = j >> 1 ... ordered f = plane*8 + ks*4 + nf ; wave handles nf = wid for all (plane, ks)
    const int nfr = (p.N + 31) >> 5;
    const int nfg = min((c.col0 >> 5) + wid, nfr - 1);
    const uint4* wh = reinterpret_cast<const uint4*>(p.Wh + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
    const uint4* wl = reinterpret_cast<const uint4*>(p.Wl + (size_t)c.wsel * p.w_stride) + ((size_t)nfg * nks) * 64 + lane;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
#define GLDS(gptr, ldsaddr)                                                                                       \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep)                                                                                \
                     : "v"(gptr), "s"(ldsaddr)                                                                    \
                     : "memory");                                                                                 \
    }
#define DMAB(kt_, stg)                                                                            \
    {                                                                                             \
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + 16384 + (stg) * 16384 + wid * 1024); \
        GLDS(wh + (size_t)((kt_) * 2 + 0) * 64, d);                                               \
        GLDS(wh + (size_t)((kt_) * 2 + 1) * 64, d + 4096);                                        \
        GLDS(wl + (size_t)((kt_) * 2 + 0) * 64, d + 8192);                                        \
        GLDS(wl + (size_t)((kt_) * 2 + 1) * 64, d + 8192 + 4096);                                 \
    }
    // ---- fragment read offsets
    // A fragment (ks, rf = wm*2 + m): granule position lo ^ (2 * (2 ks + hi))
    // B fragment (ks, nf = wn*2 + n): lane-linear
    auto compute = [&](int stg) __attribute__((always_inline)) {
        const char* s0 = sm;
        const char* sb = sm + 16384 + stg * 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 ah[2], al[2], bh[2], bl[2];
            const int apos = (hi * 32 + (lo ^ (2 * (2 * ks + hi)))) * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int fo = (ks * 4 + wm * 2 + m) * 1024 + apos;
                ah[m] = *reinterpret_cast<const uint4*>(s0 + fo);
                al[m] = *reinterpret_cast<const uint4*>(s0 + 8192 + fo);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int fo = (ks * 4 + wn * 2 + n) * 1024 + lane * 16;
                bh[n] = *reinterpret_cast<const uint4*>(sb + fo);
                bl[n] = *reinterpret_cast<const uint4*>(sb + 8192 + fo);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (KO & 1) {
                        const uint4 xx = make_uint4(bh[n].x ^ al[m].x ^ bl[n].x ^ ah[m].x, bh[n].y ^ al[m].y ^ bl[n].y ^ ah[m].y,
                                                    bh[n].z ^ al[m].z ^ bl[n].z ^ ah[m].z, bh[n].w ^ al[m].w ^ bl[n].w ^ ah[m].w);
                        acc[m][n][0] += __builtin_bit_cast(float, xx.x);
                        acc[m][n][1] += __builtin_bit_cast(float, xx.y);
                        acc[m][n][2] += __builtin_bit_cast(float, xx.z);
                        acc[m][n][3] += __builtin_bit_cast(float, xx.w);
                    } else {
                        acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                        acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                        acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                    }
                }
        }
    };

    // prologue: tile 0 -> stage 0, tile 1 in flight.  All loop loads are issued from inline asm so
    // that the waits can be counted by hand (the compiler would drain vmcnt to 0 at every use).
    LDA3(x, 0)
    DMAB(0, 0)
    if (nkt > 1) {
        LDA3(y, 1)
        VMWAIT(4, x);
    } else {
        VMWAIT(0, x);
    }
    STA3(x, 0)
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // even tile in stage 0; y holds tile kt+1 (in flight)
        if (!(KO & 4)) if (kt + 1 < nkt) DMAB(kt + 1, 1)
        if (!(KO & 2)) if (kt + 2 < nkt) LDA3(x, kt + 2)
        compute(0);
        __syncthreads();
        if (kt + 1 < nkt) {
            if (KO & 2) {
                if (KO & 4) { VMWAIT(0, y); } else { VMWAIT(0, y); }
            } else if (KO & 4) {
                if (kt + 2 < nkt) { VMWAIT(4, y); } else { VMWAIT(0, y); }
            } else if (kt + 2 < nkt) {
                VMWAIT(4, y);
            } else {
                VMWAIT(0, y);
            }
            if (!(KO & 16)) STA3(y, 1)
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            if (!(KO & 4)) if (kt + 2 < nkt) DMAB(kt + 2, 0)
            if (!(KO & 2)) if (kt + 3 < nkt) LDA3(y, kt + 3)
            compute(1);
            __syncthreads();
            if (kt + 2 < nkt) {
                if (KO & 2) {
                    VMWAIT(0, x);
                } else if (kt + 3 < nkt) {
                    VMWAIT(4, x);
                } else {
                    VMWAIT(0, x);
                }
                if (!(KO & 16)) STA3(x, 0)
            }
            __syncthreads();
        }
    }
    if (KO & 8) {
        float sacc = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[m][n][r];
        if (sacc == 1.2345f) p.C[tid] = sacc;
        return;
    }
    {
        float* st = reinterpret_cast<float*>(smem);
        const int f0 = c.col0 + 4 * (tid & 31);
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (c.bias != nullptr)
            for (int j = 0; j < 4; ++j)
                if (f0 + j < c.N) b[j] = c.bias[f0 + j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wm == h) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int tl = m * 32 + lo, fl = wn * 64 + n * 32 + 8 * q + 4 * hi;
                            *reinterpret_cast<float4*>(st + tl * STAGE_C_ROW + fl) =
                                make_float4(acc[m][n][4 * q + 0] * wsc, acc[m][n][4 * q + 1] * wsc, acc[m][n][4 * q + 2] * wsc,
                                            acc[m][n][4 * q + 3] * wsc);
                        }
            }
            __syncthreads();
            if (f0 < c.N) {
#pragma unroll 4
                for (int it = 0; it < 8; ++it) {
                    const int tl = (tid >> 5) + 8 * it;
                    const int row = c.row0 + h * 64 + tl;
                    if (row >= c.M) break;
                    const float4 t4 = *reinterpret_cast<const float4*>(st + tl * STAGE_C_ROW + 4 * (tid & 31));
                    *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + f0) =
                        make_float4((t4.x + b[0]) * p.alpha, (t4.y + b[1]) * p.alpha, (t4.z + b[2]) * p.alpha, (t4.w + b[3]) * p.alpha);
                }
            }
            if (h == 0) __syncthreads();
        }
    }
}

// fragment-major planes [ceil(N/32)][K/16][2][32][8]
static float split_weights_frag(const float* w, int N, int K, unsigned short* hi, unsigned short* lo) {
    const size_t n = (size_t)N * K;
    std::vector<unsigned short> th(n), tl(n);
    const float sc = split_weights_host(w, n, th.data(), tl.data());
    const int nfr = (N + 31) / 32, nks = K / 16;
    for (int nf = 0; nf < nfr; ++nf)
        for (int ks = 0; ks < nks; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                for (int r = 0; r < 32; ++r) {
                    const int row = nf * 32 + r;
                    const size_t dst = ((((size_t)nf * nks + ks) * 2 + hh) * 32 + r) * 8;
                    for (int j = 0; j < 8; ++j) {
                        const size_t src = (size_t)row * K + ks * 16 + hh * 8 + j;
                        hi[dst + j] = row < N ? th[src] : 0;
                        lo[dst + j] = row < N ? tl[src] : 0;
                    }
                }
    return sc;
}

template <typename F>
static float timeit(F f, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const bool only = argc > 1;
    const int shapes[][3] = {{65536, 512, 512}, {65536, 256, 256}, {65536, 768, 256}, {65536, 256, 512}, {12800, 256, 2304}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N, 0.1f);
        for (size_t i = 0; i < hA.size(); ++i) hA[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        for (size_t i = 0; i < hW.size(); ++i) hW[i] = (float)((i * 40503u) % 2001) / 20000.f - 0.05f;
        const int Np = (N + 31) / 32 * 32;
        std::vector<unsigned short> hh(hW.size()), hl(hW.size()), fh((size_t)Np * K), fl((size_t)Np * K);
        const float sc = split_weights_host(hW.data(), hW.size(), hh.data(), hl.data());
        split_weights_frag(hW.data(), N, K, fh.data(), fl.data());
        float *dA, *dC, *dC2, *db, *dsc;
        unsigned short *dh, *dl, *dfh, *dfl;
        hipMalloc(&dA, hA.size() * 4);
        hipMalloc(&dC, (size_t)M * N * 4);
        hipMalloc(&dC2, (size_t)M * N * 4);
        hipMalloc(&db, N * 4);
        hipMalloc(&dsc, 4);
        hipMalloc(&dh, hh.size() * 2);
        hipMalloc(&dl, hl.size() * 2);
        hipMalloc(&dfh, fh.size() * 2);
        hipMalloc(&dfl, fl.size() * 2);
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsc, &sc, 4, hipMemcpyHostToDevice);
        hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dfh, fh.data(), fh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dfl, fl.data(), fl.size() * 2, hipMemcpyHostToDevice);
        GemmP p;
        p.A = dA;
        p.lda = K;
        p.Wh = dh;
        p.Wl = dl;
        p.wscale = dsc;
        p.ldw = K;
        p.bias = db;
        p.C = dC;
        p.ldc = N;
        p.M = M;
        p.N = N;
        p.K = K;
        GemmP q = p;
        q.Wh = dfh;
        q.Wl = dfl;
        q.C = dC2;
        const int ntiles = cdiv(M, BM) * cdiv(N, BN);
        const double gf = 3.0 * 2.0 * M * N * K * 1e-9;
        if (only && !(N == 512 && K == 512)) continue;
        const float t0 = timeit([&] { hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, true>), dim3(ntiles), dim3(256), 0, 0, p); }, 20);
        const float t1 = timeit([&] { hipLaunchKernelGGL((gemm_v3_kernel<EPI_BIAS>), dim3(ntiles), dim3(256), 0, 0, q); }, 20);
        hipMemset(dC2, 0, (size_t)M * N * 4);
        const float t2 = timeit([&] { hipLaunchKernelGGL((gemm_v4_kernel<EPI_BIAS>), dim3(ntiles), dim3(256), 0, 0, q); }, 20);
        {
            std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
            hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(c1.data(), dC2, c1.size() * 4, hipMemcpyDeviceToHost);
            double md = 0;
            for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
            printf("   v4 %7.1f us (%5.0f TF) maxdiff %.3g\n", t2, gf / t2 * 1e3, md);
        }
        hipLaunchKernelGGL((gemm_v3_kernel<EPI_BIAS>), dim3(ntiles), dim3(256), 0, 0, q);
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c1.data(), dC2, c1.size() * 4, hipMemcpyDeviceToHost);
        double md = 0;
        for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
#define KOR(KO, name) { const float t_ = timeit([&] { hipLaunchKernelGGL((gemm_v3_kernel<EPI_BIAS, KO>), dim3(ntiles), dim3(256), 0, 0, q); }, 20); printf("    %-28s %7.1f us\n", name, t_); }
        printf("M=%d N=%d K=%d  prod %7.1f us (%5.0f TF)   v3 %7.1f us (%5.0f TF)   maxdiff %.3g  err=%s\n", M, N, K, t0, gf / t0 * 1e3, t1,
               gf / t1 * 1e3, md, hipGetErrorString(hipGetLastError()));
        if (K == 512 && N == 512 && !only) {
            KOR(32, "desync by bit 11 (3 sleeps)")
            KOR(32 | 64, "desync (b>>3)&3 sleeps")
            KOR(8, "no epilogue")
            KOR(1, "no mfma")
            KOR(2, "no A loads")
            KOR(4, "no B dma")
            KOR(6, "no A loads, no B dma")
            KOR(6 | 16, "no loads, no A split/store")
            KOR(6 | 16 | 8, "mfma + lds reads only")
            KOR(6 | 16 | 8 | 1, "lds reads only")
            KOR(1 | 8, "loads + lds traffic, no mfma/epi")
            KOR(2 | 4 | 8, "no loads, no epi")
        }
        hipFree(dA); hipFree(dC); hipFree(dC2); hipFree(db); hipFree(dsc); hipFree(dh); hipFree(dl); hipFree(dfh); hipFree(dfl);
    }
    return 0;
}
