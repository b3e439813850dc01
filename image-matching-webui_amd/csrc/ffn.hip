// Fused FFN of a LightGlue block on gfx950, 3 x f16 split arithmetic (see ffn.h for the operator).
//
// Why one kernel: as three launches the block moves the 512-wide hidden activations through HBM three times
// (GEMM 1 writes 268 MB, LayerNorm/GELU reads and writes them, GEMM 2 reads them: 1.07 GB of the 1.6 GB the
// three kernels touch at 64 x 2048 tokens) and re-splits them into f16 planes inside GEMM 2's K loop.  Here a
// workgroup owns 128 tokens from the two K slabs to the residual store:
//
//   GEMM 1   8 waves, wave w owns hidden features [64w, 64w + 64) of all 128 tokens: 2 x 4 accumulator
//            fragments (128 VGPRs).  The activations ([x | ctx], f32) are split and staged through LDS in the
//            swizzled fragment order of gemm.hip, two 16 KB stages, one barrier per 32-wide K tile.  No two waves
//            share a weight fragment, so the pre-split fragment-major planes go global -> registers directly
//            (16-byte loads, one k-step ahead): the weights never touch LDS.
//   LN/GELU  on the accumulators: two LDS exchanges of per-token partial sums (mean, then centred sum of squares,
//            the two-pass form of the reference), then normalise, GELU, split -- all in registers.
//   GEMM 2   the MFMA B operand wants, per lane, 8 consecutive k of one token; a lane's accumulator registers
//            8ks .. 8ks+7 ARE 8 hidden features of its token, so the wave writes them as one 16-byte fragment row
//            and W2's K axis is permuted once at pack time to match (ffn_permute_k).  The 128 x 512 hidden tile
//            is 256 KB as two f16 planes, so it passes through LDS in two halves of 128 KB (fragment n = 0, then
//            n = 1, of every wave); wave w accumulates output features [32w, 32w + 32) of all tokens (64 VGPRs),
//            W2 fragments again straight from global memory.
//   store    through LDS so that every global store / residual load is a full 1 KB row.
//
// Occupancy: one workgroup of 512 threads per CU (2 waves per SIMD, 256 VGPRs each), 136 KB of LDS.
//
// Token tile (round 5): MT = 4 token fragments = 128 tokens per workgroup is the throughput geometry (every weight fragment a wave pulls
// from L2 feeds 12 matrix instructions).  With few tokens it leaves the chip empty -- ONE pair of 2048 key-points is 32 workgroups on 256
// CUs, and a workgroup's 128-token pass is 57 us whatever the grid: a third of the one-pair step.  MT = 2 / 1 (64 / 32 tokens per
// workgroup, same thread roles, same LDS addresses with the unused token blocks left empty) spread the same tokens over 2 x / 4 x the
// CUs; ffn_launch picks the largest tile that still gives every CU a workgroup.  A token's arithmetic does not depend on the tile it
// rides in: the three geometries are bitwise equal.
#include "ffn.h"

#include <stdlib.h>
#include <string.h>

#define FFN_G_BYTES 131072               // two f16 planes of one half of the hidden tile (also: 2 A stages, the output tile)
#define FFN_LDS_BYTES (FFN_G_BYTES + 8192)  // + 2 x [8 waves][128 tokens] partial sums
#define FFN_YLD 260                      // row stride (floats) of the output tile in LDS

__device__ __forceinline__ float ffn_gelu(float y) { return gelu_poly(y); }  // common.h

template <int VAR, int ACT, int MT>
__global__ __launch_bounds__(512, 2) void lg_ffn_kernel(FfnP p) {
    constexpr int TOK = 32 * MT;  // tokens per workgroup
    extern __shared__ uint4 ffn_smem[];
    char* sm = reinterpret_cast<char*>(ffn_smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * TOK;
    if (row0 >= p.M) return;
    if (p.rows_per_seq > 0) {
        const int seq = row0 / p.rows_per_seq;
        if (p.cnt && row0 - seq * p.rows_per_seq >= p.cnt[seq]) return;
        if (p.active && p.active[seq >> 1] == 0) return;
    }

#define FFN_STAMP(i)                                                                        \
    if (p.dbg != nullptr && tid == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = (long long)wall_clock64();
    FFN_STAMP(0)
    // ================================================================== GEMM 1: H^T[512][128] = W1 * [x | ctx]^T
    f32x16 acc[2][MT];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.0f;

    // staging: thread -> k-octet g (8 consecutive k of the 32-wide tile) of token srow
    const int srow = tid >> 2, g = tid & 3;
    const bool stager = MT == 4 || srow < TOK;  // (a smaller tile is staged by the first 4 * TOK threads)
    const int grow = min(row0 + min(srow, TOK - 1), p.M - 1);
    const float* px = p.x + (size_t)grow * 256 + g * 8;
    const float* pc = p.ctx + (size_t)grow * 256 + g * 8;
    // granule of fragment (ks = g >> 1, token block srow >> 5): half g & 1, position (srow & 31) ^ 2g (gemm.hip's swizzle)
    const int wo = (((g >> 1) * 4 + (srow >> 5)) * 64 + (g & 1) * 32 + ((srow & 31) ^ (2 * g))) * 16;
    const uint4* w1h = reinterpret_cast<const uint4*>(p.w1h) + ((size_t)(2 * wid) * 32) * 64 + lane;  // fragment (nf, ks) at (nf * 32 + ks) * 64
    const uint4* w1l = reinterpret_cast<const uint4*>(p.w1l) + ((size_t)(2 * wid) * 32) * 64 + lane;
    f32x4 a0, a1;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        const float* q = kt < 8 ? px + kt * 32 : pc + (kt - 8) * 32;
        if (stager) {
            a0 = *reinterpret_cast<const f32x4*>(q);
            a1 = *reinterpret_cast<const f32x4*>(q + 4);
        }
    };
    auto store = [&](int stg) __attribute__((always_inline)) {
        uint4 h, l;
        if (stager) {
            split8(__builtin_bit_cast(float4, a0), __builtin_bit_cast(float4, a1), h, l);
            *reinterpret_cast<uint4*>(sm + stg * 16384 + wo) = h;
            *reinterpret_cast<uint4*>(sm + stg * 16384 + 8192 + wo) = l;
        }
    };
    uint4 wc[2][2], wn[2][2];  // [feature fragment][plane] of the current / next k-step
    auto loadw = [&](int s, uint4(&w)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            w[n][0] = w1h[(size_t)(n * 32 + s) * 64];
            w[n][1] = w1l[(size_t)(n * 32 + s) * 64];
        }
    };
    auto kstep = [&](int stg, int ks, uint4(&w)[2][2]) __attribute__((always_inline)) {
        const int apos = (hi * 32 + (lo ^ (2 * (2 * ks + hi)))) * 16;
        uint4 ah[MT], al[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int fo = stg * 16384 + (ks * 4 + m) * 1024 + apos;
            ah[m] = *reinterpret_cast<const uint4*>(sm + fo);
            al[m] = *reinterpret_cast<const uint4*>(sm + fo + 8192);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[n][m] = mfma16(w[n][0], al[m], acc[n][m]);
                acc[n][m] = mfma16(w[n][1], ah[m], acc[n][m]);
                acc[n][m] = mfma16(w[n][0], ah[m], acc[n][m]);
            }
    };
    if constexpr (VAR == 0) {
        // rolled loop, weights one k-step ahead
        issue(0);
        loadw(0, wc);
        store(0);
        issue(1);
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < 16; ++kt) {
            const int stg = kt & 1;
            loadw(2 * kt + 1, wn);
            kstep(stg, 0, wc);
            if (kt + 1 < 16) {
                store(stg ^ 1);  // its readers finished before the barrier that ended iteration kt - 1
                if (kt + 2 < 16) issue(kt + 2);
                loadw(2 * kt + 2, wc);
            }
            kstep(stg, 1, wn);
            __syncthreads();
        }
    } else {
        // Fully unrolled, weights TWO k-steps ahead in three rotating register sets.  vmcnt retires in order, so a
        // wait for a weight fragment also waits for every activation load issued before it: an activation tile gets
        // exactly as much time as the weights requested right after it.  Order per K tile (k-steps s = 2 kt, s + 1):
        //   W(s+2) | MFMAs of s | store tile kt+1 (waits X(kt+1), requested one tile ago) | X(kt+2) | W(s+3) | MFMAs of s+1
        // -> X(kt+2) must land before the MFMAs of s + 3, one whole tile later, which is also when its store needs it.
        uint4 wr[3][2][2];
        issue(0);
        loadw(0, wr[0]);
        loadw(1, wr[1]);
        store(0);
        issue(1);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            const int stg = kt & 1, s = 2 * kt;
            if (s + 2 < 32) loadw(s + 2, wr[(s + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            kstep(stg, 0, wr[s % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < 16) {
                store(stg ^ 1);
                if (kt + 2 < 16) issue(kt + 2);
            }
            if (s + 3 < 32) loadw(s + 3, wr[(s + 3) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            kstep(stg, 1, wr[(s + 1) % 3]);
            __syncthreads();
        }
    }

    FFN_STAMP(1)
    // ================================================================== bias, LayerNorm (two passes), GELU -- in registers
    // accumulator (n, m, r) of lane (lo, hi): feature 64 wid + 32 n + 8 (r >> 2) + 4 hi + (r & 3), token 32 m + lo
    float* stat = reinterpret_cast<float*>(sm + FFN_G_BYTES);  // [8][128] partial sums
    float* stat2 = stat + 1024;
    const float s1 = *p.s1;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = p.b1 ? *reinterpret_cast<const float4*>(p.b1 + 64 * wid + 32 * n + 8 * q + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[n][m][4 * q + 0] = acc[n][m][4 * q + 0] * s1 + b4.x;
                acc[n][m][4 * q + 1] = acc[n][m][4 * q + 1] * s1 + b4.y;
                acc[n][m][4 * q + 2] = acc[n][m][4 * q + 2] * s1 + b4.z;
                acc[n][m][4 * q + 3] = acc[n][m][4 * q + 3] * s1 + b4.w;
            }
        }
    if constexpr (ACT == 0) {
    float mean[MT], rstd[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[n][m][r];
        s += __shfl_xor(s, 32, 64);
        if (hi == 0) stat[wid * 128 + 32 * m + lo] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += stat[w * 128 + 32 * m + lo];
        mean[m] = s * (1.0f / 512.0f);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[n][m][r] - mean[m];
                acc[n][m][r] = d;
                s += d * d;
            }
        s += __shfl_xor(s, 32, 64);
        if (hi == 0) stat2[wid * 128 + 32 * m + lo] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += stat2[w * 128 + 32 * m + lo];
        rstd[m] = 1.0f / sqrtf(s * (1.0f / 512.0f) + 1e-5f);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 64 * wid + 32 * n + 8 * q + 4 * hi;
            const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + f0);
            const float4 e4 = *reinterpret_cast<const float4*>(p.beta + f0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[n][m][4 * q + 0] = ffn_gelu(acc[n][m][4 * q + 0] * rstd[m] * g4.x + e4.x);
                acc[n][m][4 * q + 1] = ffn_gelu(acc[n][m][4 * q + 1] * rstd[m] * g4.y + e4.y);
                acc[n][m][4 * q + 2] = ffn_gelu(acc[n][m][4 * q + 2] * rstd[m] * g4.z + e4.z);
                acc[n][m][4 * q + 3] = ffn_gelu(acc[n][m][4 * q + 3] * rstd[m] * g4.w + e4.w);
            }
            __builtin_amdgcn_sched_barrier(0);  // 16 GELU chains in flight are enough; more only costs registers
        }
    } else {
        // ACT 1: SuperGlue's MLP (BatchNorm folded into W1 / b1 at pack time), ReLU.  ACT 2 / 3: the dense matchers' MLPs,
        // LeakyReLU(0.01) (EfficientLoFTR) / ReLU (LoFTR) here and a LayerNorm AFTER the second GEMM (store phase below)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[n][m][r];
                    acc[n][m][r] = (ACT == 2) ? (v > 0.0f ? v : 0.01f * v) : fmaxf(v, 0.0f);
                }
    }

    FFN_STAMP(2)
    // ================================================================== GEMM 2: Y^T[256][128] = W2 * G^T, two K halves
    // W2 planes in the permuted K order: k-step kk = 4 w + 2 n + ks holds the features wave w hands over from
    // registers 8 ks .. 8 ks + 7 of its fragment n.
    // Wave tile 32 output features x 128 tokens.  (A 64 x 64 wave tile halves the LDS fragment reads but doubles the
    // weight fragments every wave pulls through the vector L1: measured 12.6 -> 14.6 us per half.)
    int lane2 = lane, wid2 = wid;
    // opaque copies of the thread coordinates: the address arithmetic of this section must not be hoisted into the
    // register-tight LayerNorm / GELU section above
    asm volatile("" : "+v"(lane2), "+v"(wid2));
    const int lo2 = lane2 & 31, hi2 = lane2 >> 5;
    f32x16 acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[m][r] = 0.0f;
    const uint4* w2h = reinterpret_cast<const uint4*>(p.w2h) + ((size_t)wid2 * 32) * 64 + lane2;
    const uint4* w2l = reinterpret_cast<const uint4*>(p.w2l) + ((size_t)wid2 * 32) * 64 + lane2;
    float4 res[4 * MT];  // residual rows in the order of the final row-major store, requested before the last MFMA loop
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) {
            FFN_STAMP(3)
            __syncthreads();  // every wave has read the first half
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 v0 = make_float4(acc[ph][m][8 * ks + 0], acc[ph][m][8 * ks + 1], acc[ph][m][8 * ks + 2], acc[ph][m][8 * ks + 3]);
                const float4 v1 = make_float4(acc[ph][m][8 * ks + 4], acc[ph][m][8 * ks + 5], acc[ph][m][8 * ks + 6], acc[ph][m][8 * ks + 7]);
                uint4 h, l;
                split8(v0, v1, h, l);
                // wave w's 8 fragments, hi and lo planes interleaved: 16 KB reachable from one base register
                const int off = wid2 * 16384 + lane2 * 16 + ((ks * 4 + m) * 2) * 1024;
                *reinterpret_cast<uint4*>(sm + off) = h;
                *reinterpret_cast<uint4*>(sm + off + 1024) = l;
            }
        __syncthreads();
        if (ph == 1) {
            // the hidden accumulators are dead: 64 registers take the residual rows (whole 1 KB rows per wave) while
            // the MFMAs run, so the store phase has no load latency in it
#pragma unroll
            for (int pass = 0; pass < 4 * MT; ++pass)
                res[pass] = *reinterpret_cast<const float4*>(p.x + (size_t)min(row0 + pass * 8 + wid2, p.M - 1) * 256 + lane2 * 4);
        }
        // W2 fragments: FOUR k-steps in flight in four register sets (round 6; was two).  A k-step of this GEMM is 12 MFMAs = 384 matrix cycles against
        // 24 in GEMM 1, so "two ahead" gave a fragment 0.4 us to come back from L2 where GEMM 1's get 0.8 us -- and this loop ran at 0.46 of the pipe
        // against GEMM 1's 0.63 (tools/ffn_bench.py).  The loop is unrolled over the four sets (static register names, no copies).
        uint4 wq[4][2];  // [k-step & 3][plane]
        auto wload = [&](int q, uint4(&w)[2]) __attribute__((always_inline)) {
            const int gk = (q >> 1) * 4 + ph * 2 + (q & 1);
            w[0] = w2h[(size_t)gk * 64];
            w[1] = w2l[(size_t)gk * 64];
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) wload(u, wq[u]);
        // Fragment reads placed by hand (round 6) -- the hi plane of a k-step is requested at its top and lands under its first product (which reads
        // the lo plane), the lo plane of the NEXT k-step goes into the registers that product has just released and lands under the other two;
        // the MFMAs of a k-step go product by product across the MT accumulators (order pinned with sched_barrier).  hipcc's own schedule of the
        // plain loop was ds_read_b128 -> s_waitcnt lgkmcnt(0) -> MFMA on ONE register quad, eight exposed LDS latencies per 12 MFMAs.  No extra
        // registers (a second set of hi fragments made hipcc spill six hidden activations in the LayerNorm section).  Per accumulator the
        // products and k-steps keep their order: bitwise equal.
        uint4 gh[MT], gl[MT];
        // fragment m of k-step q of this half (the layout the hand-over above wrote)
        auto goff = [&](int q, int m) __attribute__((always_inline)) { return (q >> 1) * 16384 + lane2 * 16 + (((q & 1) * 4 + m) * 2) * 1024; };
#pragma unroll
        for (int m = 0; m < MT; ++m) gl[m] = *reinterpret_cast<const uint4*>(sm + goff(0, m) + 1024);
#pragma unroll 1
        for (int kk = 0; kk < 16; kk += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = kk + u;
#pragma unroll
                for (int m = 0; m < MT; ++m) gh[m] = *reinterpret_cast<const uint4*>(sm + goff(q, m));  // lands under the first product
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc2[m] = mfma16(wq[u][0], gl[m], acc2[m]);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < 16) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) gl[m] = *reinterpret_cast<const uint4*>(sm + goff(q + 1, m) + 1024);  // lands under the other two
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc2[m] = mfma16(wq[u][1], gh[m], acc2[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc2[m] = mfma16(wq[u][0], gh[m], acc2[m]);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 4 < 16) wload(q + 4, wq[u]);  // (this k-step's MFMAs have been issued: its set is free)
            }
        }
    }

    FFN_STAMP(4)
    // ================================================================== bias, residual, row-major store through LDS
    __syncthreads();  // the second half has been read
    float* Y = reinterpret_cast<float*>(sm);
    const float s2 = *p.s2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wid2 + 8 * q + 4 * hi2;
        const float4 b4 = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float4 v;
            v.x = acc2[m][4 * q + 0] * s2 + b4.x;
            v.y = acc2[m][4 * q + 1] * s2 + b4.y;
            v.z = acc2[m][4 * q + 2] * s2 + b4.z;
            v.w = acc2[m][4 * q + 3] * s2 + b4.w;
            *reinterpret_cast<float4*>(Y + (32 * m + lo2) * FFN_YLD + f0) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4 * MT; ++pass) {
        const int row = pass * 8 + wid2;
        const int gr = row0 + row;
        if (gr < p.M) {  // wave-uniform: a wave owns the whole row
            float4 v = *reinterpret_cast<const float4*>(Y + row * FFN_YLD + lane2 * 4);
            if constexpr (ACT >= 2) {
                // LayerNorm(256) of the MLP output before the residual, a row per wave, 4 values per lane: the arithmetic of
                // lf_layernorm_kernel<4> (mean, centred sum of squares, 1 / sqrt(var + 1e-5))
                const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
                v.x -= mean, v.y -= mean, v.z -= mean, v.w -= mean;
                const float rstd = 1.0f / sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / 256.0f) + 1e-5f);
                const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + lane2 * 4), e4 = *reinterpret_cast<const float4*>(p.beta + lane2 * 4);
                v = make_float4(v.x * rstd * g4.x + e4.x, v.y * rstd * g4.y + e4.y, v.z * rstd * g4.z + e4.z, v.w * rstd * g4.w + e4.w);
            }
            *reinterpret_cast<float4*>(p.out + (size_t)gr * 256 + lane2 * 4) = make_float4(v.x + res[pass].x, v.y + res[pass].y, v.z + res[pass].z, v.w + res[pass].w);
        }
    }
    FFN_STAMP(5)
#undef FFN_STAMP
}

int ffn_launch(imcui_hip_s* h, const FfnP& p, hipStream_t stream) {
    if (h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "ffn: the fused kernel is the 3 x f16 split path (precision 1)");
    if (!p.x || !p.ctx || !p.out || !p.w1h || !p.w1l || !p.w2h || !p.w2l || !p.s1 || !p.s2 || (p.act != 1 && (!p.gamma || !p.beta)))
        return imcui_set_err(h, IMCUI_ERR_ARG, "ffn: null argument");
    if (p.act < 0 || p.act > 3) return imcui_set_err(h, IMCUI_ERR_ARG, "ffn: act=%d (0 LN+GELU, 1 ReLU, 2 LeakyReLU + post-LN, 3 ReLU + post-LN)", p.act);
    if (p.rows_per_seq > 0 && p.rows_per_seq % 128 != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "ffn: rows_per_seq=%d must be a multiple of 128", p.rows_per_seq);
    if (p.M <= 0) return IMCUI_OK;
    static const int variant = getenv("IMCUI_FFN_VARIANT") ? atoi(getenv("IMCUI_FFN_VARIANT")) : 1;
    typedef void (*kern_t)(FfnP);
    // [token tile: 128 (rolled loop), 128, 64, 32][act]
    static const kern_t kerns[4][4] = {{lg_ffn_kernel<0, 0, 4>, lg_ffn_kernel<0, 1, 4>, lg_ffn_kernel<0, 2, 4>, lg_ffn_kernel<0, 3, 4>},
                                       {lg_ffn_kernel<1, 0, 4>, lg_ffn_kernel<1, 1, 4>, lg_ffn_kernel<1, 2, 4>, lg_ffn_kernel<1, 3, 4>},
                                       {lg_ffn_kernel<1, 0, 2>, lg_ffn_kernel<1, 1, 2>, lg_ffn_kernel<1, 2, 2>, lg_ffn_kernel<1, 3, 2>},
                                       {lg_ffn_kernel<1, 0, 1>, lg_ffn_kernel<1, 1, 1>, lg_ffn_kernel<1, 2, 1>, lg_ffn_kernel<1, 3, 1>}};
    static std::atomic<unsigned long long> optin[4][4];  // > 64 KB of dynamic LDS: once per instantiation AND device (common.h)
    // token tile: the largest of 128 / 64 / 32 that still gives each of the 256 CUs a workgroup (option ffn_tile forces one)
    int tok = h->opt[OPT_FFN_TILE];
    if (tok != 128 && tok != 64 && tok != 32) tok = p.M > 256 * 64 ? 128 : p.M > 256 * 32 ? 64 : 32;
    if (p.dbg) tok = 128;  // (the lab's stamp buffer has a slot per 128 tokens)
    const int v = tok == 128 ? (variant != 0) : tok == 64 ? 2 : 3, a = p.act;
    if (!imcui_lds_optin(optin[v][a], reinterpret_cast<const void*>(kerns[v][a]), FFN_LDS_BYTES))
        return imcui_set_err(h, IMCUI_ERR_HIP, "ffn: cannot reserve %d bytes of LDS", FFN_LDS_BYTES);
    if (h->range_flag) {
        imcui_range_check(h, p.x, p.M, 256, 256, p.cnt, p.rows_per_seq, stream);
        imcui_range_check(h, p.ctx, p.M, 256, 256, p.cnt, p.rows_per_seq, stream);
    }
    imcui_prof_begin(h, PROF_GEMM, stream);
    hipLaunchKernelGGL(kerns[v][a], dim3((p.M + tok - 1) / tok), dim3(512), FFN_LDS_BYTES, stream, p);
    imcui_prof_end(h, PROF_GEMM, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// new K position 16 kk + 8 h + j  <-  original feature 16 kk + 8 (j >> 2) + 4 h + (j & 3): within every 16 features, the
// two halves of a wave hold {0-3, 8-11} and {4-7, 12-15} (accumulator rows (r & 3) + 8 (r >> 2) + 4 hi)
void ffn_permute_k(const float* src, int N, int K, float* dst) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < K; ++c) {
            const int kk = c >> 4, hh = (c >> 3) & 1, j = c & 7;
            dst[(size_t)n * K + c] = src[(size_t)n * K + 16 * kk + 8 * (j >> 2) + 4 * hh + (j & 3)];
        }
}
