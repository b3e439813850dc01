// Weight-stationary-in-registers GEMM for the projection layers (3 x f16 split arithmetic, gfx950).
//
// gemm_split_kernel (gemm.hip) moves BOTH operands through LDS and pays two barriers per 32-wide K tile; on the short-K
// projections (LightGlue / SuperGlue QKV and cross projections: K = 256, eight tiles between a prologue and a plane-writing
// epilogue) its matrix pipe was busy 0.27-0.30 of the time.  This kernel is the first GEMM of the fused FFN (ffn.hip) as a
// stand-alone launch:
//   * workgroup = 128 tokens x 256 output features, 4 waves; wave w owns features [64 w, 64 w + 64) of ALL 128 tokens
//     (2 x 4 accumulator fragments, 128 VGPRs), so no two waves share a weight fragment and the pre-split fragment-major
//     weight planes go global -> registers directly (one coalesced 16-byte load per lane = one 1 KiB MFMA fragment), one
//     k-step ahead.  The weights never touch LDS;
//   * the activations (f32) are split while they are staged into a double-buffered swizzled LDS image (gemm.hip's
//     fragment order): ONE barrier per K tile, 8 fragment reads per 24 MFMAs;
//   * for the attention layouts a 256-feature tile is exactly one of q / k / v and wave w is head w: the epilogue is
//     wave-private (no workgroup barrier) -- the wave parks 32 tokens x 64 features in its own 9 KB of LDS, reads them back
//     row-contiguous (RoPE with one float4 of cos / sin, scale, split) and stores whole 128-byte plane rows; V is parked
//     transposed and leaves as [feature][token] runs.
// Row-major epilogues (bias / ReLU / residual / activation codes of EPI_CONV) leave through the same wave-private staging
// as 256-byte row segments.
#include <stdlib.h>

#include "gemm.h"
#include "gemm_tile.h"

#define WR_BN 256
// tools/wreg_lab.hip compiles this file with WR_LAB: a device word of switches that knock out parts of the kernel (timing experiments: what the
// launch waits for).  In the product build the word is the constant 0 and every test below folds away.
#ifdef WR_LAB
__device__ int wr_lab_flags = 0;
#define WR_FLAGS (__builtin_amdgcn_readfirstlane(wr_lab_flags))
#else
#define WR_FLAGS 0
#endif
#define WR_F_NOSTORE 1   // no global stores in the epilogue
#define WR_F_ONETILE 2   // one K tile instead of K / 32
#define WR_F_NOEPI 4     // return after the main loop
#define WR_F_NOWLOAD 8   // weights loaded for the first two k-steps only (registers re-used)
#define WR_F_NOXLOAD 16  // activations loaded / staged for the first tile only
#define WR_F_NOSTORE_V 32   // no global stores in the V^T panels only (64-byte row segments)
#define WR_F_NOSTORE_QK 64  // no global stores in the q / k panels only (whole 128-byte plane rows)
// bits 8..15: the workgroups of the SECOND dispatch round (ids [ncu, 2 ncu): the second resident workgroup of every CU) start this many
// ~0.5 us later, so that the two workgroups of a CU run their matrix loop / their epilogue in opposite phases
#define WR_F_STAGGER(f) (((f) >> 8) & 255)
#define WR_STG_ROW 68    // floats per parked [token][64 features] row (272 B: conflict-free 16-byte writes)
#define WR_STG_TROW 36   // floats per parked [feature][32 tokens] row (144 B)
#define WR_WAVE_LDS 9216  // staging bytes per wave (32 x 68 x 4 = 8704; 64 x 36 x 4 = 9216)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// the pointer only depends on the workgroup: tell the compiler (buffer descriptors must live in SGPRs; cf. attention.hip)
__device__ __forceinline__ const void* wr_uniform_ptr(const void* p) {
    const size_t v = (size_t)p;
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffu));
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void*)((size_t)lo32 | ((size_t)hi32 << 32));
}
__device__ __forceinline__ void wr_wave_fence() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS traffic is done
    __builtin_amdgcn_wave_barrier();
}

// PIPE: the K loop with the weight fragments requested TWO k-steps ahead in three rotating register sets (the schedule of the fused
// FFN's first GEMM, ffn.hip VAR 1: vmcnt retires in order, so an activation tile gets exactly the lead of the weights requested after
// it); unrolled over three K tiles so that the set of a k-step is static.  PIPE = false keeps the rolled loop, one k-step ahead.
// MT: token fragments per workgroup (4 = 128 tokens, the throughput geometry; 2 / 1 = 64 / 32 tokens for launches with so few tokens
// that 128-token tiles leave most CUs idle -- one LightGlue pair is 32 row tiles.  Same thread roles and LDS addresses, the unused token
// blocks stay empty; a token's arithmetic does not depend on the tile it rides in: bitwise equal).
template <int EPI, bool SINGLE, bool PIPE = true, int MT = 4>
__global__ __launch_bounds__(256, 2) void gemm_wreg_kernel(GemmP p) {
    __shared__ uint4 smem[(4 * WR_WAVE_LDS) / 16];  // main loop: two 16 KB activation stages; epilogue: 4 x 9 KB wave staging
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    TileCtx c;
    if (!gemm_tile_setup(p, c, WR_BN, 32 * MT)) return;
    const float* A = p.A + (size_t)c.z * p.a_bs;
    const float wsc = p.wscale ? p.wscale[c.wsel] : 1.0f;
    const int labf = WR_FLAGS;
    if (WR_F_STAGGER(labf) != 0 && (int)blockIdx.x >= p.st_nct && (int)blockIdx.x < 4 * p.st_nct)  // (lab: st_nct carries the CU count)
        for (int i = 0; i < WR_F_STAGGER(labf) * ((int)blockIdx.x / p.st_nct); ++i) __builtin_amdgcn_s_sleep(16);
    const int nkt = (labf & WR_F_ONETILE) ? 1 : p.K >> 5, nks = 2 * nkt;

    f32x16 acc[2][MT];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.0f;

    // ---- activation staging: thread -> k-octet g (8 consecutive k of the 32-wide tile) of rows tid >> 2 and + 64
    const int g = tid & 3;
    const int rl0 = tid >> 2, rl1 = 64 + (tid >> 2);
    const bool st0 = MT >= 2 || rl0 < 32, st1 = MT == 4;  // which of its two rows a thread stages
    const int wo0 = (((g >> 1) * 4 + (rl0 >> 5)) * 64 + (g & 1) * 32 + ((rl0 & 31) ^ (2 * g))) * 16;
    const int wo1 = (((g >> 1) * 4 + (rl1 >> 5)) * 64 + (g & 1) * 32 + ((rl1 & 31) ^ (2 * g))) * 16;
    const float* pa0 = A + (size_t)min(c.row0 + (st0 ? rl0 : 0), c.M - 1) * p.lda + g * 8;
    const float* pa1 = A + (size_t)min(c.row0 + (st1 ? rl1 : 0), c.M - 1) * p.lda + g * 8;
    f32x4 xa0, xb0, xa1, xb1;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        if (st0) {
            xa0 = *reinterpret_cast<const f32x4*>(pa0 + kt * 32);
            xb0 = *reinterpret_cast<const f32x4*>(pa0 + kt * 32 + 4);
        }
        if (st1) {
            xa1 = *reinterpret_cast<const f32x4*>(pa1 + kt * 32);
            xb1 = *reinterpret_cast<const f32x4*>(pa1 + kt * 32 + 4);
        }
    };
    auto store = [&](int stg) __attribute__((always_inline)) {
        uint4 h, l;
        if constexpr (SINGLE) {  // one product: the nearest f16 of either operand (common.h)
            if (st0) *reinterpret_cast<uint4*>(sm + stg * 16384 + wo0) = half8_rtn(__builtin_bit_cast(float4, xa0), __builtin_bit_cast(float4, xb0));
            if (st1) *reinterpret_cast<uint4*>(sm + stg * 16384 + wo1) = half8_rtn(__builtin_bit_cast(float4, xa1), __builtin_bit_cast(float4, xb1));
            return;
        }
        if (st0) {
            split8(__builtin_bit_cast(float4, xa0), __builtin_bit_cast(float4, xb0), h, l);
            *reinterpret_cast<uint4*>(sm + stg * 16384 + wo0) = h;
            if constexpr (!SINGLE) *reinterpret_cast<uint4*>(sm + stg * 16384 + 8192 + wo0) = l;
        }
        if (st1) {
            split8(__builtin_bit_cast(float4, xa1), __builtin_bit_cast(float4, xb1), h, l);
            *reinterpret_cast<uint4*>(sm + stg * 16384 + wo1) = h;
            if constexpr (!SINGLE) *reinterpret_cast<uint4*>(sm + stg * 16384 + 8192 + wo1) = l;
        }
    };
    // ---- weights: fragments (nf, ks) of the planes [ceil(N/32)][K/16][64 lanes][8 halves] at ((nf * nks) + ks) * 64 + lane.
    // Buffer loads: two wave-uniform descriptors (hi / lo plane of the selected weight set), one 32-bit lane offset per feature
    // fragment and a scalar k-step offset -- no 64-bit address registers in the loop (the pipelined loop needs the registers)
    const int nfr = (p.N + 31) >> 5;
    const size_t plane_bytes = (size_t)nfr * 32 * p.K * 2;
    const __amdgpu_buffer_rsrc_t rWh =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wr_uniform_ptr(p.Wh + (size_t)c.wsel * p.w_stride)), 0, (unsigned)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rWl =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wr_uniform_ptr(p.Wl + (size_t)c.wsel * p.w_stride)), 0, (unsigned)plane_bytes, 0x00020000);
    unsigned wof[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int nf = min((c.col0 >> 5) + 2 * wid + n, nfr - 1);
        wof[n] = (unsigned)((nf * (p.K >> 4)) * 64 + lane) * 16u;
    }
    uint4 wc[2][2], wn[2][2];  // [feature fragment][plane] of the current / next k-step (rolled loop)
    auto loadw = [&](int s, uint4(&w)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const u32x4_t vh = __builtin_amdgcn_raw_buffer_load_b128(rWh, wof[n], (unsigned)s * 1024u, 0);
            w[n][0] = make_uint4(vh.x, vh.y, vh.z, vh.w);
            if constexpr (!SINGLE) {
                const u32x4_t vl = __builtin_amdgcn_raw_buffer_load_b128(rWl, wof[n], (unsigned)s * 1024u, 0);
                w[n][1] = make_uint4(vl.x, vl.y, vl.z, vl.w);
            }
        }
    };
    auto kstep = [&](int stg, int ks, uint4(&w)[2][2]) __attribute__((always_inline)) {
        const int apos = (hi * 32 + (lo ^ (2 * (2 * ks + hi)))) * 16;
        // two halves of two token fragments each: 16 instead of 32 fragment registers live (the pipelined loop needs them for
        // its third weight set)
#pragma unroll
        for (int mh = 0; mh < (MT + 1) / 2; ++mh) {
            constexpr int MH = MT >= 2 ? 2 : 1;  // token fragments per half
            uint4 ah[MH], al[MH];
#pragma unroll
            for (int m = 0; m < MH; ++m) {
                const int fo = stg * 16384 + (ks * 4 + 2 * mh + m) * 1024 + apos;
                ah[m] = *reinterpret_cast<const uint4*>(sm + fo);
                if constexpr (!SINGLE) al[m] = *reinterpret_cast<const uint4*>(sm + fo + 8192);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MH; ++m) {
                    // weight fragment = MFMA A operand (rows = features), activations = B (columns = tokens)
                    if constexpr (!SINGLE) {
                        acc[n][2 * mh + m] = mfma16(w[n][0], al[m], acc[n][2 * mh + m]);
                        acc[n][2 * mh + m] = mfma16(w[n][1], ah[m], acc[n][2 * mh + m]);
                    }
                    acc[n][2 * mh + m] = mfma16(w[n][0], ah[m], acc[n][2 * mh + m]);
                }
            if constexpr (PIPE) {
                if (mh == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if constexpr (PIPE) {
        uint4 w0[2][2], w1[2][2], w2[2][2];
        issue(0);
        loadw(0, w0);
        if (nks > 1) loadw(1, w1);
        store(0);
        if (nkt > 1) issue(1);
        __syncthreads();
        // tile kt (k-steps s = 2 kt, s + 1): W(s+2) | MFMAs of s | store tile kt+1, request X(kt+2) | W(s+3) | MFMAs of s+1 | barrier
#define WR_TILE(KT, WA, WB, WC)                                  \
    {                                                            \
        const int kt_ = (KT), stg_ = kt_ & 1, s_ = 2 * kt_;      \
        if (s_ + 2 < nks && !(labf & WR_F_NOWLOAD)) loadw(s_ + 2, WC); \
        __builtin_amdgcn_sched_barrier(0);                       \
        kstep(stg_, 0, WA);                                      \
        __builtin_amdgcn_sched_barrier(0);                       \
        if (kt_ + 1 < nkt && !(labf & WR_F_NOXLOAD)) {           \
            store(stg_ ^ 1);                                     \
            if (kt_ + 2 < nkt) issue(kt_ + 2);                   \
        }                                                        \
        if (s_ + 3 < nks && !(labf & WR_F_NOWLOAD)) loadw(s_ + 3, WA); \
        __builtin_amdgcn_sched_barrier(0);                       \
        kstep(stg_, 1, WB);                                      \
        __syncthreads();                                         \
    }
        int kt = 0;
#pragma unroll 1
        for (; kt + 3 <= nkt; kt += 3) {
            WR_TILE(kt, w0, w1, w2)
            WR_TILE(kt + 1, w2, w0, w1)
            WR_TILE(kt + 2, w1, w2, w0)
        }
        if (kt < nkt) WR_TILE(kt, w0, w1, w2)
        if (kt + 1 < nkt) WR_TILE(kt + 1, w2, w0, w1)
#undef WR_TILE
    } else {
    issue(0);
    loadw(0, wc);
    store(0);
    if (nkt > 1) issue(1);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        const int stg = kt & 1;
        loadw(2 * kt + 1, wn);
        kstep(stg, 0, wc);
        if (kt + 1 < nkt) {
            store(stg ^ 1);  // its readers finished before the barrier that ended iteration kt - 1
            if (kt + 2 < nkt) issue(kt + 2);
            loadw(2 * kt + 2, wc);
        }
        kstep(stg, 1, wn);
        __syncthreads();
    }
    }

    // ================================================================== epilogue (wave-private staging)
    // accumulator (n, m, r) of lane (lo, hi): feature col0 + 64 wid + 32 n + 8 (r >> 2) + 4 hi + (r & 3), token row0 + 32 m + lo
    float* st = reinterpret_cast<float*>(sm + wid * WR_WAVE_LDS);
    const int fw0 = c.col0 + 64 * wid;  // first feature of this wave
    if (fw0 >= c.N) return;            // (no workgroup barrier below)
    if (labf & WR_F_NOEPI) {
        if (acc[0][0][0] == 12345.678f) p.C[0] = acc[1][MT - 1][3];  // (keeps the accumulators alive)
        return;
    }
    bool do_store = !(labf & WR_F_NOSTORE);
    // LayerNorm folded into this layer: per-feature row sums of the (gamma-folded) weights, per-token (mean, rstd) of the raw input
    const float* lrs = nullptr;
    if constexpr (EPI == EPI_QKV_VIT || EPI == EPI_CONV) {
        if (p.ln_stats != nullptr) lrs = p.ln_rowsum + (size_t)c.wsel * p.ln_stride;
    }
    auto park_rows = [&](int m) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(st + lo * WR_STG_ROW + 32 * n + 8 * q + 4 * hi) =
                    make_float4(acc[n][m][4 * q + 0] * wsc, acc[n][m][4 * q + 1] * wsc, acc[n][m][4 * q + 2] * wsc, acc[n][m][4 * q + 3] * wsc);
    };
    if constexpr (EPI == EPI_QKV || EPI == EPI_CROSS || EPI == EPI_QKV_VIT) {
        // the 256 features of the tile are four heads of ONE of q / k / v; this wave is head `hd` of block `t`
        const int CW = p.heads * 64;
        const int t = fw0 / CW, hd = (fw0 - t * CW) >> 6;
        float* dst;
        bool vt, rope, scale;
        if (EPI == EPI_QKV || EPI == EPI_QKV_VIT) {
            const int role = t + (EPI == EPI_QKV_VIT ? p.role0 : 0);
            dst = (role == 0) ? p.Q : (role == 1) ? p.Kt : p.V;
            vt = (role == 2);
            rope = (role < 2);
            scale = (role == 0);
        } else {
            dst = (t == 0) ? p.Q : p.V;
            vt = (t == 1);
            rope = false;
            scale = (t == 0);
        }
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
        if ((labf & WR_F_NOSTORE_V) && vt) do_store = false;
        if ((labf & WR_F_NOSTORE_QK) && !vt) do_store = false;
        const int R = p.rows_per_seq;
        const int i0 = c.row0 - c.seq * R;
        if (EPI == EPI_QKV_VIT && !vt) {
            const int tab0 = i0 + (p.rope_seq_row0 ? p.rope_seq_row0[c.seq] : 0);  // this tile's first row of the RoPE2D tables
            // RoPE2D: a lane finishes 8 consecutive features of one token; their rotation partners sit 16 features away in the same
            // 32-feature half (read from the parked row as well).  `first` lanes hold the a of (a, b) -> (a cos - b sin, b cos + a sin).
            const int d0 = (lane & 7) * 8, dp = d0 ^ 16;
            const bool first = (d0 & 16) == 0;
            const int ti = (d0 >> 5) * 16 + (d0 & 8);  // table column: 16 per half, this lane's 8 frequencies
            float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba, pa = ba, pb = ba;
            if (c.bias != nullptr) {
                ba = *reinterpret_cast<const float4*>(c.bias + fw0 + d0);
                bb = *reinterpret_cast<const float4*>(c.bias + fw0 + d0 + 4);
                pa = *reinterpret_cast<const float4*>(c.bias + fw0 + dp);
                pb = *reinterpret_cast<const float4*>(c.bias + fw0 + dp + 4);
            }
            float sm_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sp_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // row sums: mine / partner
            if (lrs != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sm_[j] = lrs[fw0 + d0 + j];
                    sp_[j] = lrs[fw0 + dp + j];
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float4 tc[4][2], ts[4][2];
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {  // tables of this step's tokens, requested before the tile is parked
                    const size_t ir = (size_t)(tab0 + 32 * m + pass * 8 + (lane >> 3)) * 32 + ti;
                    tc[pass][0] = *reinterpret_cast<const float4*>(p.rope_cos + ir);
                    tc[pass][1] = *reinterpret_cast<const float4*>(p.rope_cos + ir + 4);
                    ts[pass][0] = *reinterpret_cast<const float4*>(p.rope_sin + ir);
                    ts[pass][1] = *reinterpret_cast<const float4*>(p.rope_sin + ir + 4);
                }
                park_rows(m);
                wr_wave_fence();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int tl = pass * 8 + (lane >> 3);
                    const float4 va = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + d0);
                    const float4 vb = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + d0 + 4);
                    const float4 wa = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + dp);
                    const float4 wb = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + dp + 4);
                    float mine[8] = {va.x + ba.x, va.y + ba.y, va.z + ba.z, va.w + ba.w, vb.x + bb.x, vb.y + bb.y, vb.z + bb.z, vb.w + bb.w};
                    float part[8] = {wa.x + pa.x, wa.y + pa.y, wa.z + pa.z, wa.w + pa.w, wb.x + pb.x, wb.y + pb.y, wb.z + pb.z, wb.w + pb.w};
                    if (lrs != nullptr) {
                        const float2 mr = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)(c.row0 + 32 * m + tl));
                        const float xa[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w}, xp[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                        const float ba8[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w}, pa8[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            mine[j] = mr.y * (xa[j] - mr.x * sm_[j]) + ba8[j];
                            part[j] = mr.y * (xp[j] - mr.x * sp_[j]) + pa8[j];
                        }
                    }
                    const float cw[8] = {tc[pass][0].x, tc[pass][0].y, tc[pass][0].z, tc[pass][0].w, tc[pass][1].x, tc[pass][1].y, tc[pass][1].z, tc[pass][1].w};
                    const float ss[8] = {ts[pass][0].x, ts[pass][0].y, ts[pass][0].z, ts[pass][0].w, ts[pass][1].x, ts[pass][1].y, ts[pass][1].z, ts[pass][1].w};
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float x = mine[j], y = part[j];
                        asm volatile("" : "+v"(x), "+v"(y));  // scalar fused multiply-adds, never packed (see the LightGlue branch)
                        const float r = first ? __builtin_fmaf(x, cw[j], -(y * ss[j])) : __builtin_fmaf(x, cw[j], y * ss[j]);
                        v[j] = scale ? r * p.alpha : r;
                    }
                    uint4 hv, lv;
                    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hv, lv);
                    unsigned short* o = d16 + (((size_t)c.seq * p.heads + hd) * R + i0 + 32 * m + tl) * 64 + d0;
                    if (do_store) {
                        *reinterpret_cast<uint4*>(o) = hv;
                        *reinterpret_cast<uint4*>(o + p.plane_halves) = lv;
                    }
                }
                wr_wave_fence();
            }
        } else if (!vt) {
            const int d0 = (lane & 7) * 8;
            float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
            if (c.bias != nullptr) {
                ba = *reinterpret_cast<const float4*>(c.bias + fw0 + d0);
                bb = *reinterpret_cast<const float4*>(c.bias + fw0 + d0 + 4);
            }
            // rotary tables of the wave's next 32 tokens are requested one m-step ahead (two register sets): a lane needs
            // cos / sin of 4 tokens x 4 frequencies per step, and with only two waves per SIMD nothing else hides the latency
            float4 tc[2][4], ts[2][4];
            auto load_tables = [&](int m, float4(&cs)[4], float4(&sn)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const size_t row = (size_t)(c.row0 + 32 * m + pass * 8 + (lane >> 3));
                    cs[pass] = *reinterpret_cast<const float4*>(p.rope_cos + row * 32 + (d0 >> 1));
                    sn[pass] = *reinterpret_cast<const float4*>(p.rope_sin + row * 32 + (d0 >> 1));
                }
            };
            if (rope) load_tables(0, tc[0], ts[0]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                park_rows(m);
                wr_wave_fence();
                if (rope && m + 1 < MT) load_tables(m + 1, tc[(m + 1) & 1], ts[(m + 1) & 1]);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int tl = pass * 8 + (lane >> 3);
                    const float4 va = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + d0);
                    const float4 vb = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + d0 + 4);
                    float v[8] = {va.x + ba.x, va.y + ba.y, va.z + ba.z, va.w + ba.w, vb.x + bb.x, vb.y + bb.y, vb.z + bb.z, vb.w + bb.w};
                    if (rope) {
                        const float4 cs = tc[m & 1][pass], sn = ts[m & 1][pass];
                        const float cw[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0).  Written as explicit fused
                            // multiply-adds on values made opaque to the vectoriser: hipcc's SLP pass packed these four
                            // rotations into v_pk_fma_f32 / v_pk_mul_f32 with op_sel swizzles and in-place destinations, and
                            // that sequence returned wrong even elements in lanes 48-63 of a few waves per launch on gfx950
                            // (round 3, tools/r03_diag2.py: run-to-run differences, always element 2 of a lane's 8; the
                            // scalar form is bit-stable).  The translation unit is also built with -fno-slp-vectorize.
                            float a0 = v[2 * j], a1 = v[2 * j + 1];
                            asm volatile("" : "+v"(a0), "+v"(a1));
                            v[2 * j] = __builtin_fmaf(a0, cw[j], -(a1 * ss[j]));
                            v[2 * j + 1] = __builtin_fmaf(a1, cw[j], a0 * ss[j]);
                        }
                    }
                    if (scale) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
                    }
                    uint4 hv, lv;
                    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hv, lv);
                    unsigned short* o = d16 + (((size_t)c.seq * p.heads + hd) * R + i0 + 32 * m + tl) * 64 + d0;
                    if (do_store) {
                        *reinterpret_cast<uint4*>(o) = hv;
                        *reinterpret_cast<uint4*>(o + p.plane_halves) = lv;
                    }
                }
                wr_wave_fence();
            }
        } else {
            // V^T [seq][head][64][rows]: park [feature][32 tokens], read 8 consecutive tokens of one feature per lane
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[(32 * n + 8 * (r >> 2) + 4 * hi + (r & 3)) * WR_STG_TROW + lo] = acc[n][m][r] * wsc;
                wr_wave_fence();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int d = pass * 16 + (lane >> 2), tk = (lane & 3) * 8;
                    const float bd = c.bias ? c.bias[fw0 + d] : 0.0f;
                    float4 va = *reinterpret_cast<const float4*>(st + d * WR_STG_TROW + tk);
                    float4 vb = *reinterpret_cast<const float4*>(st + d * WR_STG_TROW + tk + 4);
                    if (lrs != nullptr) {  // the lane's 8 consecutive tokens: (mean, rstd) pairs are 64 contiguous bytes
                        const float sd = lrs[fw0 + d];
                        const float4* ms4 = reinterpret_cast<const float4*>(p.ln_stats + 2 * (size_t)(c.row0 + 32 * m + tk));
                        const float4 q0 = ms4[0], q1 = ms4[1], q2 = ms4[2], q3 = ms4[3];
                        va = make_float4(q0.y * (va.x - q0.x * sd), q0.w * (va.y - q0.z * sd), q1.y * (va.z - q1.x * sd), q1.w * (va.w - q1.z * sd));
                        vb = make_float4(q2.y * (vb.x - q2.x * sd), q2.w * (vb.y - q2.z * sd), q3.y * (vb.z - q3.x * sd), q3.w * (vb.w - q3.z * sd));
                    }
                    uint4 hv, lv;
                    split8(make_float4(va.x + bd, va.y + bd, va.z + bd, va.w + bd), make_float4(vb.x + bd, vb.y + bd, vb.z + bd, vb.w + bd), hv, lv);
                    unsigned short* o = d16 + (((size_t)c.seq * p.heads + hd) * 64 + d) * R + i0 + 32 * m + tk;
                    if (do_store) {
                        *reinterpret_cast<uint4*>(o) = hv;
                        *reinterpret_cast<uint4*>(o + p.plane_halves) = lv;
                    }
                }
                wr_wave_fence();
            }
        }
    } else {
        // ---- row-major f32 output: 16 lanes cover the wave's 64 features of one token (256-byte segments)
        float* C = p.C + (size_t)c.z * p.c_bs;
        const int fl = (lane & 15) * 4;
        const int f0 = fw0 + fl;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.bias != nullptr) b4 = *reinterpret_cast<const float4*>(c.bias + f0);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lrs != nullptr) s4 = *reinterpret_cast<const float4*>(lrs + f0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            park_rows(m);
            wr_wave_fence();
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int tl = pass * 4 + (lane >> 4);
                const int row = c.row0 + 32 * m + tl;
                if (row < c.M) {
                    float4 t4 = *reinterpret_cast<const float4*>(st + tl * WR_STG_ROW + fl);
                    if (lrs != nullptr) {
                        const float2 mr = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)row);
                        t4 = make_float4(mr.y * (t4.x - mr.x * s4.x), mr.y * (t4.y - mr.x * s4.y), mr.y * (t4.z - mr.x * s4.z), mr.y * (t4.w - mr.x * s4.w));
                    }
                    float v[4] = {t4.x + b4.x, t4.y + b4.y, t4.z + b4.z, t4.w + b4.w};
                    float* dstp = C + (size_t)row * p.ldc + f0;
                    if (EPI == EPI_CONV) {
                        if (p.resid != nullptr) {
                            const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (size_t)row * p.ldr + f0);
                            v[0] += r4.x;
                            v[1] += r4.y;
                            v[2] += r4.z;
                            v[3] += r4.w;
                        }
                        if (p.act == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                        } else if (p.act == 2) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.01f * v[j];
                        } else if (p.act == 3) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = gelu_poly(v[j]);
                        }
                    } else if (EPI == EPI_BIAS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                    } else if (EPI == EPI_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                    } else if (EPI == EPI_RESID) {
                        const float4 o4 = *reinterpret_cast<const float4*>(dstp);
                        v[0] += o4.x;
                        v[1] += o4.y;
                        v[2] += o4.z;
                        v[3] += o4.w;
                    }
                    if (do_store) *reinterpret_cast<float4*>(dstp) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            wr_wave_fence();
        }
    }
}

// Which launches take this kernel: pre-split weight planes, one K slab, plain (non-batched, non-conv) A, whole 64-feature
// wave tiles and 16-byte aligned rows; the attention layouts additionally need rows_per_seq % 128 == 0 (the plane-writing epilogue
// takes seq = row0 / R for a whole 128-row tile: a tile must not straddle two sequences).  Option gemm_wreg = 0 keeps everything on
// gemm_split_kernel (A/B runs).
bool gemm_wreg_ok(const imcui_hip_s* h, const GemmP& p) {
    const int mode = h ? h->opt[OPT_GEMM_WREG] : 2;  // 0: off; 1: attention-layout projections only; 2: every eligible launch (imcui_hip_set_option)
    if (mode == 0 || (mode == 1 && p.epi != EPI_QKV && p.epi != EPI_CROSS && p.epi != EPI_QKV_VIT) || p.Wh == nullptr || p.Wl == nullptr || p.A2 != nullptr || p.conv_k > 0 || p.batch != 1 || p.mcnt || p.ncnt || p.group_rows > 1 ||
        p.rup_h > 0)
        return false;
    if (p.K % 32 != 0 || p.N % 64 != 0 || (p.lda & 3) != 0) return false;
    if (p.ln_stats != nullptr && (p.ln_rowsum == nullptr || (p.epi != EPI_QKV_VIT && p.epi != EPI_CONV))) return false;
    if (p.epi == EPI_QKV_VIT)
        return p.split_out && p.v_transposed && p.heads % 4 == 0 && p.N % (p.heads * 64) == 0 && p.rows_per_seq > 0 && p.rows_per_seq % 128 == 0 && p.M % 128 == 0 && !p.single;
    if (p.epi == EPI_QKV || p.epi == EPI_CROSS)
        return p.split_out && p.v_transposed && p.heads == 4 && p.N % 256 == 0 && p.rows_per_seq > 0 && p.rows_per_seq % 128 == 0 && p.M % 128 == 0;
    if (p.single && p.epi != EPI_CONV) return false;
    if ((p.ldc & 3) != 0 || (p.epi == EPI_CONV && p.resid != nullptr && (p.ldr & 3) != 0)) return false;
    return p.epi == EPI_BIAS || p.epi == EPI_RELU || p.epi == EPI_RESID || p.epi == EPI_CONV;
}

// small-batch row tiles (LightGlue's attention-layout projections, plain-bias and activation-epilogue launches): the largest of 128 / 64 / 32 tokens that
// still gives every CU a workgroup
template <int EPI, int MT>
static void wreg_launch_small(const GemmP& p, hipStream_t stream) {
    hipLaunchKernelGGL((gemm_wreg_kernel<EPI, false, true, MT>), dim3(cdiv(p.M, 32 * MT) * cdiv(p.N, WR_BN), 1, 1), dim3(256), 0, stream, p);
}
static int wreg_tile_tokens(const imcui_hip_s* h, const GemmP& p) {
    if (p.epi != EPI_QKV && p.epi != EPI_CROSS && p.epi != EPI_BIAS && p.epi != EPI_CONV) return 128;
    if (p.single || p.group_rows > 1 || (h && h->opt[OPT_WREG_PIPE] == 0)) return 128;
    const int forced = h ? h->opt[OPT_WREG_TILE] : 0;
    if (forced == 128 || forced == 64 || forced == 32) return forced;
    const long ncol = cdiv(p.N, WR_BN);
    if ((long)cdiv(p.M, 128) * ncol >= 256) return 128;
    if ((long)cdiv(p.M, 64) * ncol >= 256) return 64;
    return 32;
}
template <bool PIPE>
static void wreg_launch(const GemmP& p, hipStream_t stream) {
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, WR_BN), 1, 1);
    switch (p.epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_BIAS, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        case EPI_RELU: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_RELU, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_RESID, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_QKV, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        case EPI_CROSS: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_CROSS, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        case EPI_QKV_VIT: hipLaunchKernelGGL((gemm_wreg_kernel<EPI_QKV_VIT, false, PIPE>), grid, dim3(256), 0, stream, p); break;
        default:
            if (p.single)
                hipLaunchKernelGGL((gemm_wreg_kernel<EPI_CONV, true, PIPE>), grid, dim3(256), 0, stream, p);
            else
                hipLaunchKernelGGL((gemm_wreg_kernel<EPI_CONV, false, PIPE>), grid, dim3(256), 0, stream, p);
    }
}
void gemm_wreg_launch(const imcui_hip_s* h, const GemmP& p, hipStream_t stream) {
    const int tok = wreg_tile_tokens(h, p);
    if (tok != 128) {
        if (p.epi == EPI_QKV)
            tok == 64 ? wreg_launch_small<EPI_QKV, 2>(p, stream) : wreg_launch_small<EPI_QKV, 1>(p, stream);
        else if (p.epi == EPI_CROSS)
            tok == 64 ? wreg_launch_small<EPI_CROSS, 2>(p, stream) : wreg_launch_small<EPI_CROSS, 1>(p, stream);
        else if (p.epi == EPI_CONV)  // (EfficientLoFTR's projections on 300-token aggregated grids)
            tok == 64 ? wreg_launch_small<EPI_CONV, 2>(p, stream) : wreg_launch_small<EPI_CONV, 1>(p, stream);
        else
            tok == 64 ? wreg_launch_small<EPI_BIAS, 2>(p, stream) : wreg_launch_small<EPI_BIAS, 1>(p, stream);
        return;
    }
    if (h && h->opt[OPT_WREG_PIPE] == 0)  // A/B switch: 0 = the rolled K loop
        wreg_launch<false>(p, stream);
    else
        wreg_launch<true>(p, stream);
}
