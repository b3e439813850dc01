// Implicit-GEMM 3x3 convolution on the f32 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
// One workgroup (4 waves) computes an 8x16 pixel tile x 64 output channels.  The K dimension
// (9 taps x Cin) is walked as Cin/32 channel chunks x 9 taps: the (8+2)x(16+2) input patch of
// a chunk is staged once into LDS as [channel-quad][pixel][4] (image tiles are read from HBM
// in full 128-byte channel runs per pixel), the 32x64 weight slab of a tap is double-buffered
// in LDS and prefetched through registers.  Wave w owns output rows 2w, 2w+1 (32 pixels, one
// MFMA row fragment) and all 64 output channels (two column fragments).  ReLU and the 2x2
// max-pool are applied in registers: the four pixels of a pooling window live in one lane.
#include <stdlib.h>

#include "conv.h"
#include "gemm.h"

#define TH 8
#define TW 16
#define PW (TW + 2)
#define PH (TH + 2)
#define NPIX (PH * PW)  // 180
#define PSTR 181        // padded pixel stride (float4 units) -> conflict-free staging writes
#define WSTR 65         // padded cout stride (float4 units)

__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                      const float* __restrict__ bias, float* __restrict__ out, int H,
                                                      int W, int Cin, int Cout, int tiles_x, int tiles_y, int relu,
                                                      int pool) {
    __shared__ float4 Ps[8 * PSTR];
    __shared__ float4 Ws[2][8 * WSTR];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int ncout = Cout >> 6;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = t % ncout;
    int sp = t / ncout;
    const int tx = sp % tiles_x;
    sp /= tiles_x;
    const int ty = sp % tiles_y;
    const int b = sp / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, cout0 = ct * 64;

    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int nchunk = Cin >> 5;
    const float4* wp4 = reinterpret_cast<const float4*>(wp);
    // this lane's pixel inside the wave's 2x16 strip
    const int py = 2 * wid + (lo >> 4), px = lo & 15;

    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();  // everybody is done with the previous patch
        // ---- stage the input patch of this channel chunk: NPIX pixels x 8 channel quads
        for (int idx = tid; idx < NPIX * 8; idx += 256) {
            const int cq = idx & 7, pp = idx >> 3;
            const int gy = y0 - 1 + pp / PW, gx = x0 - 1 + pp % PW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(in + (((size_t)b * H + gy) * W + gx) * Cin + ch * 32 + cq * 4);
            Ps[cq * PSTR + pp] = v;
        }
        // ---- weight slab prefetch for tap 0: 8 cq x 64 cout float4 = 512 -> 2 per thread
        float4 rw0, rw1;
        auto wload = [&](int tap) __attribute__((always_inline)) {
            rw0 = wp4[((size_t)(ch * 9 + tap) * 8 + (tid >> 6)) * Cout + cout0 + (tid & 63)];
            rw1 = wp4[((size_t)(ch * 9 + tap) * 8 + 4 + (tid >> 6)) * Cout + cout0 + (tid & 63)];
        };
        wload(0);
        for (int tap = 0; tap < 9; ++tap) {
            float4* wbuf = Ws[tap & 1];
            wbuf[(tid >> 6) * WSTR + (tid & 63)] = rw0;
            wbuf[(4 + (tid >> 6)) * WSTR + (tid & 63)] = rw1;
            __syncthreads();
            if (tap + 1 < 9) wload(tap + 1);
            const int dy = tap / 3, dx = tap - dy * 3;
            const int pp = (py + dy) * PW + px + dx;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const int cq = 2 * tq + hi;
                const float4 a = Ps[cq * PSTR + pp];
                const float4 b0 = wbuf[cq * WSTR + lo];
                const float4 b1 = wbuf[cq * WSTR + 32 + lo];
                acc[0] = mfma32(a.x, b0.x, acc[0]);
                acc[1] = mfma32(a.x, b1.x, acc[1]);
                acc[0] = mfma32(a.y, b0.y, acc[0]);
                acc[1] = mfma32(a.y, b1.y, acc[1]);
                acc[0] = mfma32(a.z, b0.z, acc[0]);
                acc[1] = mfma32(a.z, b1.z, acc[1]);
                acc[0] = mfma32(a.w, b0.w, acc[0]);
                acc[1] = mfma32(a.w, b1.w, acc[1]);
            }
        }
    }

    // ---- epilogue: bias, ReLU, optional 2x2 max-pool (window = regs {r, r+1, r+8, r+9})
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = cout0 + n * 32 + lo;
        const float bv = bias[co];
        if (pool) {
            const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                float v = fmaxf(fmaxf(acc[n][r], acc[n][r + 1]), fmaxf(acc[n][r + 8], acc[n][r + 9])) + bv;
                if (relu) v = fmaxf(v, 0.0f);
                const int p = frag_row(r, hi);  // < 16: first row of the strip
                const int oy = (y0 >> 1) + wid, ox = (x0 >> 1) + (p >> 1);
                if (oy < Ho && ox < Wo) out[(((size_t)b * Ho + oy) * Wo + ox) * Cout + co] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = frag_row(r, hi);
                const int oy = y0 + 2 * wid + (p >> 4), ox = x0 + (p & 15);
                float v = acc[n][r] + bv;
                if (relu) v = fmaxf(v, 0.0f);
                if (oy < H && ox < W) out[(((size_t)b * H + oy) * W + ox) * Cout + co] = v;
            }
        }
    }
}

int conv3x3_launch(imcui_hip_s* h, const float* in, const float* wp, const float* bias, float* out, int B, int H, int W,
                   int Cin, int Cout, int relu, int pool, hipStream_t stream) {
    if (Cin % 32 != 0 || Cout % 64 != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: Cin=%d must be a multiple of 32, Cout=%d of 64", Cin, Cout);
    if (pool && ((H | W) & 1)) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: pooled layer needs even H,W (%dx%d)", H, W);
    const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, TH);
    const long nwg = (long)tiles_x * tiles_y * (Cout / 64) * B;
    if (nwg <= 0) return IMCUI_OK;
    imcui_prof_begin(h, PROF_CONV, stream);
    hipLaunchKernelGGL(conv3x3_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, in, wp, bias, out, H, W, Cin, Cout,
                       tiles_x, tiles_y, relu, pool);
    imcui_prof_end(h, PROF_CONV, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ 3 x f16 split variant
// 8x32 pixel tile x 64 output channels per workgroup; wave w owns image rows 2w, 2w+1 (two
// 32-pixel row fragments: a fragment is ONE image row, so its 16-byte LDS reads are consecutive
// and bank-conflict free) x 64 channels (two fragments) = 64 accumulators.  The f32 input patch
// of a 32-channel chunk is split into f16 (hi, lo) while it is staged into LDS as
// [channel-octet][pixel][8 halves]; weights arrive pre-split.  Each tap = two 16-channel MFMA
// steps of wh*ah + wl*ah + wh*al with the WEIGHT fragment as the A operand, so a lane ends up
// with 4 consecutive output channels of one pixel per register quad: the NHWC store is 16 bytes
// per lane, and the 2x2 max-pool is one register max (rows) + one neighbour-lane max (columns).
#define STH 8
#define STW 32
#define SPW (STW + 2)
#define SNPIX ((STH + 2) * SPW)  // 340
#define SPSTR 346                // padded pixel stride (16-byte units): conflict-free staging writes

// FUSE1A: `in` is the 1-channel image [B,H,W] and the 64-channel input of this layer is
// relu(conv1a(image)) evaluated on the fly for the patch (SuperPoint conv1a -> conv1b): the
// 79 MB/image conv1a activation never goes to HBM.  The arithmetic of conv1a is the same fmaf chain
// as conv1a_kernel, so the fused and unfused paths agree bit for bit.
#define ITW (STW + 4)  // image tile: patch + 1-pixel halo of the first conv
#define ITH (STH + 4)
__device__ __forceinline__ float conv_act(float v, int code) { return code == 2 ? (v > 0.0f ? v : 0.01f * v) : (code == 1 ? fmaxf(v, 0.0f) : v); }

// NC = 32-channel output fragments per wave: 2 -> 64 output channels per workgroup, 4 -> 128 (per tap 24 LDS fragment reads for
// 48 MFMAs instead of 16 for 24, half as many barriers per MFMA; used when Cout % 128 == 0)
// SINGLE: one f16 product per element pair (hi planes only) instead of three: see GemmP.single
template <bool FUSE1A, int NC, bool SINGLE = false>
__global__ __launch_bounds__(256, NC == 4 ? 2 : 3) void conv3x3_split_kernel(const float* __restrict__ in,
                                                            const unsigned short* __restrict__ wh,
                                                            const unsigned short* __restrict__ wl,
                                                            const float* __restrict__ wscale,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int H, int W, int Cin, int Cout, int tiles_x, int tiles_y,
                                                            int relu, int pool, const float* __restrict__ w1a,
                                                            const float* __restrict__ b1a, int cin_stride, int cout_live, ConvHead hd) {
    constexpr int WSN = NC * 32 + 1;  // padded cout stride of the weight slab (16-byte units)
    static_assert(NC == 2 || (NC == 4 && !FUSE1A), "the fused first layer has 64 output channels");
    __shared__ uint4 smem[2 * 4 * SPSTR + 2 * 2 * 4 * WSN + (FUSE1A ? (ITH * ITW + 9 * 64 + 64 + 3) / 4 + 1 : 0)];
    uint4* Ph = smem;
    uint4* Pl = smem + 4 * SPSTR;
    uint4* Wb = smem + 2 * 4 * SPSTR;  // [buf][plane][4 * WSN]
    float* img = reinterpret_cast<float*>(smem + 2 * 4 * SPSTR + 2 * 2 * 4 * WSN);  // [ITH][ITW] (FUSE1A)
    float* w1 = img + ITH * ITW;                                                       // [9][64] + bias [64]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int ncout = Cout / (32 * NC);
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = t % ncout;
    int sp = t / ncout;
    const int tx = sp % tiles_x;
    sp /= tiles_x;
    const int ty = sp % tiles_y;
    const int b = sp / tiles_y;
    const int y0 = ty * STH, x0 = tx * STW, cout0 = ct * 32 * NC;
    // 32-channel fragments of this workgroup that hold live output channels (the rest is zero padding of the layer: its
    // products are skipped, the accumulators stay 0 and the padded channels are stored as act(0 + 0) = 0)
    const int nlive = min(NC, (cout_live - cout0 + 31) >> 5);
    // `relu`: bits 0..1 = activation of the output (0 none, 1 ReLU, 2 LeakyReLU), bit 2 = ReLU applied to the INPUT while it is staged
    const bool relu_in = !FUSE1A && (relu & 4) != 0;
    relu &= 3;

    f32x16 acc[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int nchunk = Cin >> 5;
    const uint4* wh4 = reinterpret_cast<const uint4*>(wh);
    const uint4* wl4 = reinterpret_cast<const uint4*>(wl);

    if (FUSE1A) {
        // image tile (zero outside the image = conv1a's padding) and the first layer's weights
        for (int i = tid; i < ITH * ITW; i += 256) {
            const int gy = y0 - 2 + i / ITW, gx = x0 - 2 + i % ITW;
            img[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? in[((size_t)b * H + gy) * W + gx] : 0.0f;
        }
        for (int i = tid; i < 9 * 64; i += 256) w1[i] = w1a[i];
        if (tid < 64) w1[9 * 64 + tid] = b1a[tid];
    }

    // Input patch of a 32-channel chunk: 340 pixels x 4 channel octets = 1360 granules, <= 6 per thread.  For
    // the plain layers the granules of chunk ch+1 are fetched into registers while the nine taps of chunk ch
    // are multiplied (the fetch used to sit, latency exposed, between two chunks).
    constexpr int NPG = (SNPIX * 4 + 255) / 256;
    float4 pa[NPG], pc[NPG];
    unsigned pvalid = 0;
    auto patch_fetch = [&](int ch) __attribute__((always_inline)) {
        pvalid = 0;
#pragma unroll
        for (int k = 0; k < NPG; ++k) {
            const int idx = tid + 256 * k;
            const int oc = idx & 3, pp = idx >> 2;
            const int py = pp / SPW, px = pp - py * SPW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            if (idx < SNPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const float* src = in + (((size_t)b * H + gy) * W + gx) * cin_stride + ch * 32 + oc * 8;
                pa[k] = *reinterpret_cast<const float4*>(src);
                pc[k] = *reinterpret_cast<const float4*>(src + 4);
                pvalid |= 1u << k;
            }
        }
    };
    auto patch_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPG; ++k) {
            const int idx = tid + 256 * k;
            if (idx < SNPIX * 4) {
                uint4 hq = make_uint4(0u, 0u, 0u, 0u), lq = hq;
                if ((pvalid >> k) & 1u) {
                    if (relu_in) {  // the producer left the map un-activated (a residual sum that is also needed as it is)
                        pa[k] = make_float4(fmaxf(pa[k].x, 0.f), fmaxf(pa[k].y, 0.f), fmaxf(pa[k].z, 0.f), fmaxf(pa[k].w, 0.f));
                        pc[k] = make_float4(fmaxf(pc[k].x, 0.f), fmaxf(pc[k].y, 0.f), fmaxf(pc[k].z, 0.f), fmaxf(pc[k].w, 0.f));
                    }
                    if constexpr (SINGLE)
                        hq = half8_rtn(pa[k], pc[k]);  // one product: the nearest f16 of either operand
                    else
                        split8(pa[k], pc[k], hq, lq);
                }
                Ph[(idx & 3) * SPSTR + (idx >> 2)] = hq;
                if constexpr (!SINGLE) Pl[(idx & 3) * SPSTR + (idx >> 2)] = lq;
            }
        }
    };
    // NC = 4 has no registers left to hold the next chunk's patch across the nine taps (128 accumulators): it fetches the patch
    // at the chunk boundary and relies on the co-resident workgroup to cover the latency
    constexpr bool PREFETCH = (NC == 2);
    if (!FUSE1A && PREFETCH) patch_fetch(0);

    for (int ch = 0; ch < nchunk; ++ch) {
        if (!FUSE1A && !PREFETCH) patch_fetch(ch);
        __syncthreads();  // everybody is done with the previous patch (and the image tile is visible)
        if (!FUSE1A) {
            patch_store();
            if (PREFETCH && ch + 1 < nchunk) patch_fetch(ch + 1);
        }
        // fused first layer: a thread always works on the same channel octet (256 % 4 == 0), so the 9 x 8 weights
        // and the bias of that octet are fetched from LDS once per chunk, not once per granule
        float4 kw0[9], kw1[9], kb0, kb1;
        if (FUSE1A) {
            const int c0 = ch * 32 + (tid & 3) * 8;
            kb0 = *reinterpret_cast<const float4*>(w1 + 9 * 64 + c0);
            kb1 = *reinterpret_cast<const float4*>(w1 + 9 * 64 + c0 + 4);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                kw0[t9] = *reinterpret_cast<const float4*>(w1 + t9 * 64 + c0);
                kw1[t9] = *reinterpret_cast<const float4*>(w1 + t9 * 64 + c0 + 4);
            }
        }
        for (int idx = tid; FUSE1A && idx < SNPIX * 4; idx += 256) {
            const int oc = idx & 3, pp = idx >> 2;
            const int py = pp / SPW, px = pp - py * SPW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            uint4 hq = make_uint4(0u, 0u, 0u, 0u), lq = hq;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                float4 a, c;
                if (FUSE1A) {
                    // relu(conv1a) for channels ch*32 + oc*8 .. +7 of this pixel (tap-major fmaf chain)
                    a = kb0;
                    c = kb1;
#pragma unroll
                    for (int t9 = 0; t9 < 9; ++t9) {
                        const float v = img[(py + t9 / 3) * ITW + px + t9 % 3];
                        const float4 k0 = kw0[t9];
                        const float4 k1 = kw1[t9];
                        a.x = fmaf(v, k0.x, a.x);
                        a.y = fmaf(v, k0.y, a.y);
                        a.z = fmaf(v, k0.z, a.z);
                        a.w = fmaf(v, k0.w, a.w);
                        c.x = fmaf(v, k1.x, c.x);
                        c.y = fmaf(v, k1.y, c.y);
                        c.z = fmaf(v, k1.z, c.z);
                        c.w = fmaf(v, k1.w, c.w);
                    }
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                } else {
                    const float* src = in + (((size_t)b * H + gy) * W + gx) * cin_stride + ch * 32 + oc * 8;
                    a = *reinterpret_cast<const float4*>(src);
                    c = *reinterpret_cast<const float4*>(src + 4);
                }
                split8(a, c, hq, lq);
            }
            Ph[oc * SPSTR + pp] = hq;
            Pl[oc * SPSTR + pp] = lq;
        }
        // weight slab of a tap: 4 octets x 32 NC cout per plane = NC / 2 x 256 x 16 B -> NC / 2 per thread per plane (named
        // registers: an array captured by the lambda ends up in scratch memory)
        uint4 rwh, rwl, rwh1, rwl1;
        auto wload = [&](int tap) __attribute__((always_inline)) {
            const size_t o = ((size_t)(ch * 9 + tap) * 4 + (tid >> 6)) * Cout + cout0 + (tid & 63);
            rwh = wh4[o];
            if constexpr (!SINGLE) rwl = wl4[o];
            if constexpr (NC == 4) {
                rwh1 = wh4[o + 64];
                if constexpr (!SINGLE) rwl1 = wl4[o + 64];
            }
        };
        wload(0);
        for (int tap = 0; tap < 9; ++tap) {
            uint4* wbh = Wb + (tap & 1) * (2 * 4 * WSN);
            uint4* wbl = wbh + 4 * WSN;
            wbh[(tid >> 6) * WSN + (tid & 63)] = rwh;
            if constexpr (!SINGLE) wbl[(tid >> 6) * WSN + (tid & 63)] = rwl;
            if constexpr (NC == 4) {
                wbh[(tid >> 6) * WSN + (tid & 63) + 64] = rwh1;
                if constexpr (!SINGLE) wbl[(tid >> 6) * WSN + (tid & 63) + 64] = rwl1;
            }
            __syncthreads();
            if (tap + 1 < 9) wload(tap + 1);
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int oc = 2 * st + hi;
                uint4 ah[2], al[2], bh[NC], bl[NC];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int pp = (2 * wid + m + dy) * SPW + lo + dx;
                    ah[m] = Ph[oc * SPSTR + pp];
                    if constexpr (!SINGLE) al[m] = Pl[oc * SPSTR + pp];
                }
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    bh[n] = wbh[oc * WSN + n * 32 + lo];
                    if constexpr (!SINGLE) bl[n] = wbl[oc * WSN + n * 32 + lo];
                }
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    if (NC == 4 && n == NC - 1 && nlive < NC) break;  // wave-uniform: the last fragment is padding
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        // weights = MFMA A operand (rows = output channels), pixels = B (columns)
                        if constexpr (!SINGLE) {
                            acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                            acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                        }
                        acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                    }
                }
            }
        }
    }

    // ---- epilogue: lane = pixel column x0 + lo; register quad q of fragment n = channels cout0 + 32n + 8q + 4hi .. +3.
    // Written straight from the fragments a store touches 32 pixels x 32 B; the finished values are parked in LDS
    // ([pixel][64 channels], 68-float rows: conflict-free both ways) and leave as whole 256-byte channel runs.
    const float wsc = wscale[0];
    // `relu`: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  Plain layers pass a residual map (same shape as the output) in the
    // unused first-layer pointer: it is added before the activation, in the coalesced store loop.
    const float* resid = FUSE1A ? nullptr : w1a;
    const float* resid2 = FUSE1A ? nullptr : b1a;  // a second map added with the first (DPT fusion blocks: conv + skip + path)
    const bool late = resid != nullptr;
    __syncthreads();  // every wave is done with the patch / weight buffers
    float* st = reinterpret_cast<float*>(smem);
    constexpr int SROW = 68;
    // the staging tile holds 64 channels: NC / 2 passes (half = fragments 2 half, 2 half + 1)
#pragma unroll
    for (int half = 0; half < NC / 2; ++half) {
        const int coutH = cout0 + 64 * half;
        if (half > 0) __syncthreads();  // the previous half has been stored
        if (pool) {
            const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = n * 32 + 8 * q + 4 * hi;
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + coutH + cl);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float tv = fmaxf(acc[0][2 * half + n][4 * q + j], acc[1][2 * half + n][4 * q + j]);  // rows 2w, 2w+1
                        tv = fmaxf(tv, __shfl_xor(tv, 1, 64));                                            // columns x, x^1
                        v[j] = tv * wsc;
                    }
                    v[0] += b4.x;
                    v[1] += b4.y;
                    v[2] += b4.z;
                    v[3] += b4.w;
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = conv_act(v[j], relu);
                    }
                    if ((lo & 1) == 0) *reinterpret_cast<float4*>(st + (wid * 16 + (lo >> 1)) * SROW + cl) = make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {  // 64 pooled pixels x 16 channel quads
                const int px = (tid >> 4) + 16 * it, c4 = tid & 15;
                const int oy = (y0 >> 1) + (px >> 4), pxo = (x0 >> 1) + (px & 15);
                if (oy < Ho && pxo < Wo)
                    *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + pxo) * Cout + coutH + 4 * c4) =
                        *reinterpret_cast<const float4*>(st + px * SROW + 4 * c4);
            }
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {  // image rows 4 h2 .. 4 h2 + 3 of the tile (waves 2 h2, 2 h2 + 1)
                if ((wid >> 1) == h2) {
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int cl = n * 32 + 8 * q + 4 * hi;
                            const float4 b4 = *reinterpret_cast<const float4*>(bias + coutH + cl);
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                float4 v = make_float4(acc[m][2 * half + n][4 * q + 0] * wsc + b4.x, acc[m][2 * half + n][4 * q + 1] * wsc + b4.y,
                                                       acc[m][2 * half + n][4 * q + 2] * wsc + b4.z, acc[m][2 * half + n][4 * q + 3] * wsc + b4.w);
                                if (relu && !late) {
                                    v.x = conv_act(v.x, relu);
                                    v.y = conv_act(v.y, relu);
                                    v.z = conv_act(v.z, relu);
                                    v.w = conv_act(v.w, relu);
                                }
                                *reinterpret_cast<float4*>(st + (((wid & 1) * 2 + m) * 32 + lo) * SROW + cl) = v;
                            }
                        }
                }
                __syncthreads();
                if constexpr (NC == 4 && !FUSE1A) {
                    if (hd.w != nullptr) {
                        // DPT head: the 1x1 convolution to 4 channels + the point-map post-processing on the parked (activated) tile --
                        // a thread's 4 channels x 4 outputs, summed over the 16 threads of a pixel, over the two 64-channel halves in
                        // LDS; the 128-channel map itself is only written when the caller passes `out` (parity dumps)
                        const int c4 = tid & 15;
                        float4 hw[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) hw[c] = *reinterpret_cast<const float4*>(hd.w + c * 128 + 64 * half + 4 * c4);
                        float* psum = st + 128 * SROW;  // [2 h2][128 px][4]
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int px = (tid >> 4) + 16 * it;
                            const float4 v = *reinterpret_cast<const float4*>(st + px * SROW + 4 * c4);
                            float d[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                d[c] = (v.x * hw[c].x + v.y * hw[c].y) + (v.z * hw[c].z + v.w * hw[c].w);
#pragma unroll
                                for (int o = 8; o > 0; o >>= 1) d[c] += __shfl_xor(d[c], o, 64);
                            }
                            if (c4 == 0) {
                                float4* ps = reinterpret_cast<float4*>(psum + (h2 * 128 + px) * 4);
                                if (half == 0) {
                                    *ps = make_float4(d[0], d[1], d[2], d[3]);
                                } else {
                                    const float4 q = *ps;
                                    const int oy = y0 + 4 * h2 + (px >> 5), ox = x0 + (px & 31);
                                    if (oy < H && ox < W) {
                                        const float x = (q.x + d[0]) + hd.b[0], y = (q.y + d[1]) + hd.b[1], z = (q.z + d[2]) + hd.b[2], cf = (q.w + d[3]) + hd.b[3];
                                        const size_t p = ((size_t)b * H + oy) * W + ox;
                                        if (hd.raw) *reinterpret_cast<float4*>(hd.raw + p * 4) = make_float4(x, y, z, cf);
                                        const float dn = sqrtf(x * x + y * y + z * z);
                                        const float sc = expm1f(dn) / fmaxf(dn, 1e-8f);
                                        hd.pts[p * 3 + 0] = x * sc;
                                        hd.pts[p * 3 + 1] = y * sc;
                                        hd.pts[p * 3 + 2] = z * sc;
                                        hd.conf[p] = 1.0f + expf(cf);
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {  // 128 pixels x 16 channel quads
                    const int px = (tid >> 4) + 16 * it, c4 = tid & 15;
                    const int oy = y0 + 4 * h2 + (px >> 5), ox = x0 + (px & 31);
                    if (oy < H && ox < W && out != nullptr) {
                        float4 v = *reinterpret_cast<const float4*>(st + px * SROW + 4 * c4);
                        const size_t o = (((size_t)b * H + oy) * W + ox) * Cout + coutH + 4 * c4;
                        if (late) {  // residual connection: added before the activation (BasicBlock: relu(conv + identity))
                            float4 r4 = *reinterpret_cast<const float4*>(resid + o);
                            v = make_float4(v.x + r4.x, v.y + r4.y, v.z + r4.z, v.w + r4.w);
                            if (resid2) {  // (conv + first) + second, in this order
                                r4 = *reinterpret_cast<const float4*>(resid2 + o);
                                v = make_float4(v.x + r4.x, v.y + r4.y, v.z + r4.z, v.w + r4.w);
                            }
                            v = make_float4(conv_act(v.x, relu), conv_act(v.y, relu), conv_act(v.z, relu), conv_act(v.w, relu));
                        }
                        *reinterpret_cast<float4*>(out + o) = v;
                    }
                }
                if (h2 == 0) __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------ 16-row tiles for the layers with 64 output channels
// The kernel above reads 16 LDS fragments per 24 MFMAs when a workgroup has only 64 output channels (NC = 2), against 24 per 48 with
// 128 (NC = 4: matrix-pipe busy 0.63 against 0.45-0.51) -- LDS bandwidth is what the narrow layers wait for.  Here a wave owns FOUR
// image rows (a 16 x 32 pixel tile per workgroup) x 64 channels: per tap and 16-channel step 8 pixel + 4 weight fragment reads feed
// 24 MFMAs, the ratio of the wide kernel.  To keep two workgroups per CU the patch is staged 16 input channels at a time (612 pixels
// x 2 octets x (hi, lo) = 39 KB), i.e. one MFMA k-step per tap and stage; 128 accumulators per lane as NC = 4.  The halo shrinks from
// 1.33 to 1.20 patch pixels per output pixel, which is what the fused first layer (conv1a evaluated on the fly per patch pixel) pays
// its VALU time for.  Summation order: 16-channel stage -> tap (the kernel above: 32-channel chunk -> tap -> two steps).
#define TTH 16
#define TNPIX ((TTH + 2) * SPW)  // 612
#define TPSTR 618                // padded pixel stride (16-byte units)
#define TITH (TTH + 4)

template <bool FUSE1A, bool SINGLE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_tall_kernel(const float* __restrict__ in, const unsigned short* __restrict__ wh,
                                                              const unsigned short* __restrict__ wl, const float* __restrict__ wscale,
                                                              const float* __restrict__ bias, float* __restrict__ out, int H, int W, int Cin,
                                                              int Cout, int tiles_x, int tiles_y, int relu, int pool,
                                                              const float* __restrict__ w1a, const float* __restrict__ b1a, int cin_stride) {
    constexpr int WSN = 65;  // padded cout stride of the weight slab (16-byte units)
    __shared__ uint4 smem[2 * 2 * TPSTR + 2 * 2 * 2 * WSN + (FUSE1A ? (TITH * ITW + 9 * 64 + 64 + 3) / 4 + 1 : 0)];
    uint4* Ph = smem;              // [octet 2][TPSTR]
    uint4* Pl = smem + 2 * TPSTR;
    uint4* Wb = smem + 4 * TPSTR;  // [buf][plane][octet 2][WSN]
    float* img = reinterpret_cast<float*>(smem + 4 * TPSTR + 8 * WSN);  // [TITH][ITW] (FUSE1A)
    float* w1 = img + TITH * ITW;                                       // [9][64] + bias [64]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int ncout = Cout >> 6;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = t % ncout;
    int sp = t / ncout;
    const int tx = sp % tiles_x;
    sp /= tiles_x;
    const int ty = sp % tiles_y;
    const int b = sp / tiles_y;
    const int y0 = ty * TTH, x0 = tx * STW, cout0 = ct * 64;
    const bool relu_in = !FUSE1A && (relu & 4) != 0;  // as conv3x3_split_kernel
    relu &= 3;

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int nstage = Cin >> 4;
    const uint4* wh4 = reinterpret_cast<const uint4*>(wh);
    const uint4* wl4 = reinterpret_cast<const uint4*>(wl);

    if (FUSE1A) {
        for (int i = tid; i < TITH * ITW; i += 256) {
            const int gy = y0 - 2 + i / ITW, gx = x0 - 2 + i % ITW;
            img[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? in[((size_t)b * H + gy) * W + gx] : 0.0f;
        }
        for (int i = tid; i < 9 * 64; i += 256) w1[i] = w1a[i];
        if (tid < 64) w1[9 * 64 + tid] = b1a[tid];
    }

    // input patch of a 16-channel stage: 612 pixels x 2 channel octets = 1224 granules, <= 5 per thread; fetched at the stage boundary
    // (no registers to carry it across the nine taps), the co-resident workgroup covers the latency
    constexpr int NPG = (TNPIX * 2 + 255) / 256;
    for (int cs = 0; cs < nstage; ++cs) {
        float4 pa[NPG], pc[NPG];
        unsigned pvalid = 0;
        if (!FUSE1A) {
#pragma unroll
            for (int k = 0; k < NPG; ++k) {
                const int idx = tid + 256 * k;
                const int oc = idx & 1, pp = idx >> 1;
                const int py = pp / SPW, px = pp - py * SPW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                if (idx < TNPIX * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    const float* src = in + (((size_t)b * H + gy) * W + gx) * cin_stride + cs * 16 + oc * 8;
                    pa[k] = *reinterpret_cast<const float4*>(src);
                    pc[k] = *reinterpret_cast<const float4*>(src + 4);
                    pvalid |= 1u << k;
                }
            }
        }
        __syncthreads();  // everybody is done with the previous patch (and the image tile is visible)
        if (!FUSE1A) {
#pragma unroll
            for (int k = 0; k < NPG; ++k) {
                const int idx = tid + 256 * k;
                if (idx < TNPIX * 2) {
                    uint4 hq = make_uint4(0u, 0u, 0u, 0u), lq = hq;
                    if ((pvalid >> k) & 1u) {
                        if (relu_in) {
                            pa[k] = make_float4(fmaxf(pa[k].x, 0.f), fmaxf(pa[k].y, 0.f), fmaxf(pa[k].z, 0.f), fmaxf(pa[k].w, 0.f));
                            pc[k] = make_float4(fmaxf(pc[k].x, 0.f), fmaxf(pc[k].y, 0.f), fmaxf(pc[k].z, 0.f), fmaxf(pc[k].w, 0.f));
                        }
                        if constexpr (SINGLE)
                            hq = half8_rtn(pa[k], pc[k]);
                        else
                            split8(pa[k], pc[k], hq, lq);
                    }
                    Ph[(idx & 1) * TPSTR + (idx >> 1)] = hq;
                    if constexpr (!SINGLE) Pl[(idx & 1) * TPSTR + (idx >> 1)] = lq;
                }
            }
        } else {
            // fused first layer: a thread always works on the same channel octet (256 % 2 == 0): its 9 x 8 weights and bias once per stage
            float4 kw0[9], kw1[9], kb0, kb1;
            const int c0 = cs * 16 + (tid & 1) * 8;
            kb0 = *reinterpret_cast<const float4*>(w1 + 9 * 64 + c0);
            kb1 = *reinterpret_cast<const float4*>(w1 + 9 * 64 + c0 + 4);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                kw0[t9] = *reinterpret_cast<const float4*>(w1 + t9 * 64 + c0);
                kw1[t9] = *reinterpret_cast<const float4*>(w1 + t9 * 64 + c0 + 4);
            }
            for (int idx = tid; idx < TNPIX * 2; idx += 256) {
                const int oc = idx & 1, pp = idx >> 1;
                const int py = pp / SPW, px = pp - py * SPW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                uint4 hq = make_uint4(0u, 0u, 0u, 0u), lq = hq;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    // relu(conv1a) for channels c0 .. c0 + 7 of this pixel (tap-major fmaf chain, as conv1a_kernel)
                    float4 a = kb0, c = kb1;
#pragma unroll
                    for (int t9 = 0; t9 < 9; ++t9) {
                        const float v = img[(py + t9 / 3) * ITW + px + t9 % 3];
                        const float4 k0 = kw0[t9];
                        const float4 k1 = kw1[t9];
                        a.x = fmaf(v, k0.x, a.x);
                        a.y = fmaf(v, k0.y, a.y);
                        a.z = fmaf(v, k0.z, a.z);
                        a.w = fmaf(v, k0.w, a.w);
                        c.x = fmaf(v, k1.x, c.x);
                        c.y = fmaf(v, k1.y, c.y);
                        c.z = fmaf(v, k1.z, c.z);
                        c.w = fmaf(v, k1.w, c.w);
                    }
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                    split8(a, c, hq, lq);
                }
                Ph[oc * TPSTR + pp] = hq;
                Pl[oc * TPSTR + pp] = lq;
            }
        }
        // weight slab of a tap and stage: 2 octets x 64 cout per plane = 128 x 16 B: threads 0..127 carry the hi plane, 128..255 the lo plane
        uint4 rw;
        const int wplane = tid >> 7, woct = (tid >> 6) & 1, wc = tid & 63;
        auto wload = [&](int tap) __attribute__((always_inline)) {
            const size_t o = ((size_t)((cs >> 1) * 9 + tap) * 4 + (cs & 1) * 2 + woct) * Cout + cout0 + wc;
            if (wplane == 0)
                rw = wh4[o];
            else if constexpr (!SINGLE)
                rw = wl4[o];
        };
        wload(0);
        for (int tap = 0; tap < 9; ++tap) {
            uint4* wb = Wb + (tap & 1) * (4 * WSN);
            if (!SINGLE || wplane == 0) wb[wplane * (2 * WSN) + woct * WSN + wc] = rw;
            __syncthreads();
            if (tap + 1 < 9) wload(tap + 1);
            const int dy = tap / 3, dx = tap - dy * 3;
            const uint4* wbh = wb + hi * WSN;
            const uint4* wbl = wb + 2 * WSN + hi * WSN;
            uint4 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pp = (4 * wid + m + dy) * SPW + lo + dx;
                ah[m] = Ph[hi * TPSTR + pp];
                if constexpr (!SINGLE) al[m] = Pl[hi * TPSTR + pp];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = wbh[n * 32 + lo];
                if constexpr (!SINGLE) bl[n] = wbl[n * 32 + lo];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr (!SINGLE) {
                        acc[m][n] = mfma16(bh[n], al[m], acc[m][n]);
                        acc[m][n] = mfma16(bl[n], ah[m], acc[m][n]);
                    }
                    acc[m][n] = mfma16(bh[n], ah[m], acc[m][n]);
                }
        }
    }

    // ---- epilogue (as the kernel above): finished values parked in LDS [pixel][64 channels], out as whole 256-byte channel runs
    const float wsc = wscale[0];
    const float* resid = FUSE1A ? nullptr : w1a;
    const float* resid2 = FUSE1A ? nullptr : b1a;
    const bool late = resid != nullptr;
    __syncthreads();
    float* st = reinterpret_cast<float*>(smem);
    constexpr int SROW = 68;
    if (pool) {
        // rows 4w .. 4w+3 of a wave pool to 2 rows x 16 columns: 8 x 16 pooled pixels per workgroup
        const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = n * 32 + 8 * q + 4 * hi;
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + cout0 + cl);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float tv = fmaxf(acc[2 * mp][n][4 * q + j], acc[2 * mp + 1][n][4 * q + j]);
                        tv = fmaxf(tv, __shfl_xor(tv, 1, 64));
                        v[j] = tv * wsc;
                    }
                    v[0] += b4.x;
                    v[1] += b4.y;
                    v[2] += b4.z;
                    v[3] += b4.w;
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = conv_act(v[j], relu);
                    }
                    if ((lo & 1) == 0) *reinterpret_cast<float4*>(st + ((wid * 2 + mp) * 16 + (lo >> 1)) * SROW + cl) = make_float4(v[0], v[1], v[2], v[3]);
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {  // 128 pooled pixels x 16 channel quads
            const int px = (tid >> 4) + 16 * it, c4 = tid & 15;
            const int oy = (y0 >> 1) + (px >> 4), pxo = (x0 >> 1) + (px & 15);
            if (oy < Ho && pxo < Wo)
                *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + pxo) * Cout + cout0 + 4 * c4) =
                    *reinterpret_cast<const float4*>(st + px * SROW + 4 * c4);
        }
    } else {
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {  // image rows 4 h2 .. 4 h2 + 3 of the tile = wave h2
            if (wid == h2) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = n * 32 + 8 * q + 4 * hi;
                        const float4 b4 = *reinterpret_cast<const float4*>(bias + cout0 + cl);
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            float4 v = make_float4(acc[m][n][4 * q + 0] * wsc + b4.x, acc[m][n][4 * q + 1] * wsc + b4.y, acc[m][n][4 * q + 2] * wsc + b4.z,
                                                   acc[m][n][4 * q + 3] * wsc + b4.w);
                            if (relu && !late) {
                                v.x = conv_act(v.x, relu);
                                v.y = conv_act(v.y, relu);
                                v.z = conv_act(v.z, relu);
                                v.w = conv_act(v.w, relu);
                            }
                            *reinterpret_cast<float4*>(st + (m * 32 + lo) * SROW + cl) = v;
                        }
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {  // 128 pixels x 16 channel quads
                const int px = (tid >> 4) + 16 * it, c4 = tid & 15;
                const int oy = y0 + 4 * h2 + (px >> 5), ox = x0 + (px & 31);
                if (oy < H && ox < W) {
                    float4 v = *reinterpret_cast<const float4*>(st + px * SROW + 4 * c4);
                    const size_t o = (((size_t)b * H + oy) * W + ox) * Cout + cout0 + 4 * c4;
                    if (late) {
                        float4 r4 = *reinterpret_cast<const float4*>(resid + o);
                        v = make_float4(v.x + r4.x, v.y + r4.y, v.z + r4.z, v.w + r4.w);
                        if (resid2) {
                            r4 = *reinterpret_cast<const float4*>(resid2 + o);
                            v = make_float4(v.x + r4.x, v.y + r4.y, v.z + r4.z, v.w + r4.w);
                        }
                        v = make_float4(conv_act(v.x, relu), conv_act(v.y, relu), conv_act(v.z, relu), conv_act(v.w, relu));
                    }
                    *reinterpret_cast<float4*>(out + o) = v;
                }
            }
            if (h2 < 3) __syncthreads();
        }
    }
}

// IMCUI_CONV_TALL: 0 = the 8-row kernel everywhere, 1 (default) = 16-row tiles for the fused first layer, 2 = also for plain layers
// whose Cout is not a multiple of 128 (handle option "conv_tall").  Measured (profiles/r03_lab_conv_tall.txt): SuperPoint's convolutions 18.38 ms
// per 128 images with 0, 17.60 with 1, 17.94 with 2 -- the plain layers LOSE with 16-channel stages (64-byte pieces of the 256-byte
// pixel rows per stage, and no registers left to prefetch the next patch across the taps), the fused layer, whose patch is computed
// from an LDS image tile, gains 10 %.
// (both switches live in the handle since round 5 -- imcui_hip_set_option "conv_tall" / "conv_narrow", environment read once by imcui_hip_create)
static inline int conv_tall_mode(const imcui_hip_s* h) { return h->opt[OPT_CONV_TALL]; }
static inline bool conv_narrow_env(const imcui_hip_s* h) { return h->opt[OPT_CONV_NARROW] == 1; }

int conv3x3_split_launch(imcui_hip_s* h, const float* in, const unsigned short* wh, const unsigned short* wl,
                         const float* wscale, const float* bias, float* out, int B, int H, int W, int Cin, int Cout,
                         int relu, int pool, hipStream_t stream, const float* resid, int cin_stride, int cout_live, int single, const float* resid2,
                         const ConvHead* head) {
    if (cin_stride <= 0) cin_stride = Cin;
    if (cout_live <= 0 || cout_live > Cout) cout_live = Cout;
    if (cin_stride < Cin || cin_stride % 4 != 0) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: pixel stride %d for %d input channels", cin_stride, Cin);
    if (pool && resid) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: residual input and fused pooling are exclusive");
    if (resid2 && !resid) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: a second residual map needs the first");
    if (head && (Cout != 128 || pool || resid || (relu & 3) != 1 || cout_live != Cout || !head->w || !head->b || !head->pts || !head->conf ||
                 conv_narrow_env(h)))
        return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: the fused point-map head needs a 128-channel ReLU layer without pooling / residual");
    if (!head && !out) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: null output");
    if (Cin % 32 != 0 || Cout % 64 != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: Cin=%d must be a multiple of 32, Cout=%d of 64", Cin, Cout);
    if (pool && ((H | W) & 1)) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: pooled layer needs even H,W (%dx%d)", H, W);
    const int tiles_x = cdiv(W, STW);
    // 128 output channels per workgroup when the layer has them (fewer LDS fragment reads and barriers per MFMA)
    bool narrow_only = conv_narrow_env(h);  // A/B switch
    // few pixels (a single pair: 24 .. 75 tiles per image at 1/8 .. 1/4 resolution): 64-channel tiles double the workgroups while the
    // 128-channel tiling leaves CUs empty (option conv_narrow: 0 = this rule, 1 = always, 2 = never; same arithmetic per output)
    if (h->opt[OPT_CONV_NARROW] == 0 && !head && Cout % 128 == 0 && (long)tiles_x * cdiv(H, STH) * (Cout / 128) * B < 256) narrow_only = true;
    const bool wide = (Cout % 128 == 0) && !narrow_only;
    if (!wide && cout_live == Cout && conv_tall_mode(h) >= 2) {  // 64-channel layers: 16-row tiles (same LDS-read ratio as the wide kernel)
        const int tiles_t = cdiv(H, TTH);
        const long nwg_t = (long)tiles_x * tiles_t * (Cout / 64) * B;
        if (nwg_t <= 0) return IMCUI_OK;
        if (h->range_flag) imcui_range_check(h, in, (long)B * H * W, Cin, cin_stride, nullptr, 0, stream);
        imcui_prof_begin(h, PROF_CONV, stream);
        if (single)
            hipLaunchKernelGGL((conv3x3_tall_kernel<false, true>), dim3((unsigned)nwg_t), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                               tiles_x, tiles_t, relu, pool, resid, resid2, cin_stride);
        else
            hipLaunchKernelGGL((conv3x3_tall_kernel<false>), dim3((unsigned)nwg_t), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                               tiles_x, tiles_t, relu, pool, resid, resid2, cin_stride);
        imcui_prof_end(h, PROF_CONV, stream);
        IMCUI_CHECK_LAUNCH(h);
        return IMCUI_OK;
    }
    const int tiles_y = cdiv(H, STH);
    const long nwg = (long)tiles_x * tiles_y * (Cout / (wide ? 128 : 64)) * B;
    if (nwg <= 0) return IMCUI_OK;
    if (h->range_flag) imcui_range_check(h, in, (long)B * H * W, Cin, cin_stride, nullptr, 0, stream);
    imcui_prof_begin(h, PROF_CONV, stream);
    if (wide && single)
        hipLaunchKernelGGL((conv3x3_split_kernel<false, 4, true>), dim3((unsigned)nwg), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                           tiles_x, tiles_y, relu, pool, resid, resid2, cin_stride, cout_live, head ? *head : ConvHead{});
    else if (single)
        hipLaunchKernelGGL((conv3x3_split_kernel<false, 2, true>), dim3((unsigned)nwg), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                           tiles_x, tiles_y, relu, pool, resid, resid2, cin_stride, cout_live, head ? *head : ConvHead{});
    else if (wide)
        hipLaunchKernelGGL((conv3x3_split_kernel<false, 4>), dim3((unsigned)nwg), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                           tiles_x, tiles_y, relu, pool, resid, resid2, cin_stride, cout_live, head ? *head : ConvHead{});
    else
        hipLaunchKernelGGL((conv3x3_split_kernel<false, 2>), dim3((unsigned)nwg), dim3(256), 0, stream, in, wh, wl, wscale, bias, out, H, W, Cin, Cout,
                           tiles_x, tiles_y, relu, pool, resid, resid2, cin_stride, cout_live, head ? *head : ConvHead{});
    imcui_prof_end(h, PROF_CONV, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

int conv1ab_fused_split_launch(imcui_hip_s* h, const float* image, const float* w1a, const float* b1a,
                               const unsigned short* wh, const unsigned short* wl, const float* wscale, const float* bias,
                               float* out, int B, int H, int W, int pool, hipStream_t stream) {
    if (pool && ((H | W) & 1)) return imcui_set_err(h, IMCUI_ERR_ARG, "conv1ab: pooled layer needs even H,W (%dx%d)", H, W);
    const bool tall = conv_tall_mode(h) >= 1;
    const int tiles_x = cdiv(W, STW), tiles_y = cdiv(H, tall ? TTH : STH);
    const long nwg = (long)tiles_x * tiles_y * B;
    if (nwg <= 0) return IMCUI_OK;
    imcui_prof_begin(h, PROF_CONV, stream);
    if (tall)
        hipLaunchKernelGGL((conv3x3_tall_kernel<true>), dim3((unsigned)nwg), dim3(256), 0, stream, image, wh, wl, wscale, bias, out, H, W, 64, 64, tiles_x,
                           tiles_y, 1, pool, w1a, b1a, 64);
    else
        hipLaunchKernelGGL((conv3x3_split_kernel<true, 2>), dim3((unsigned)nwg), dim3(256), 0, stream, image, wh, wl, wscale, bias, out, H,
                           W, 64, 64, tiles_x, tiles_y, 1, pool, w1a, b1a, 64, 64, ConvHead{});
    imcui_prof_end(h, PROF_CONV, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ conv1a (Cin = 1)
// 16 lanes cover the 64 output channels of one pixel (4 channels each), so a wave stores
// 4 pixels x 256 B = 1 KiB of contiguous NHWC output per instruction.
__global__ __launch_bounds__(256) void conv1a_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int H,
                                                     int W, long npix) {
    const int cq = threadIdx.x & 15;
    float4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const float4*>(w + t * 64 + cq * 4);
    const float4 bv = *reinterpret_cast<const float4*>(bias + cq * 4);
    const long stride = (long)gridDim.x * 16;
    for (long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4); p < npix; p += stride) {
        const int x = (int)(p % W);
        const long q = p / W;
        const int y = (int)(q % H);
        const float* img = in + (q - y) * W;  // start of this image
        float4 a = bv;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(long)yy * W + xx] : 0.0f;
                const float4 k = wk[dy * 3 + dx];
                a.x = fmaf(v, k.x, a.x);
                a.y = fmaf(v, k.y, a.y);
                a.z = fmaf(v, k.z, a.z);
                a.w = fmaf(v, k.w, a.w);
            }
        }
        a.x = fmaxf(a.x, 0.f);
        a.y = fmaxf(a.y, 0.f);
        a.z = fmaxf(a.z, 0.f);
        a.w = fmaxf(a.w, 0.f);
        *reinterpret_cast<float4*>(out + p * 64 + cq * 4) = a;
    }
}

int conv1a_launch(imcui_hip_s* h, const float* in, const float* w, const float* bias, float* out, int B, int H, int W,
                  hipStream_t stream) {
    const long npix = (long)B * H * W;
    if (npix <= 0) return IMCUI_OK;
    long blocks = (npix + 15) / 16;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(conv1a_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, w, bias, out, H, W, npix);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ host packers
void pack_conv3x3(const float* w, int Cout, int Cin, float* dst) {
    // dst[((ch*9 + tap)*8 + cq)*Cout + co][j] = w[co][ch*32 + cq*4 + j][tap]
    const int nchunk = Cin / 32;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int tap = 0; tap < 9; ++tap)
            for (int cq = 0; cq < 8; ++cq)
                for (int co = 0; co < Cout; ++co)
                    for (int j = 0; j < 4; ++j) {
                        const int ci = ch * 32 + cq * 4 + j;
                        dst[((((size_t)ch * 9 + tap) * 8 + cq) * Cout + co) * 4 + j] = w[((size_t)co * Cin + ci) * 9 + tap];
                    }
}

float pack_conv3x3_split(const float* w, int Cout, int Cin, unsigned short* hi, unsigned short* lo) {
    // value order: [((ch*9 + tap)*4 + oc)*Cout + co][j]  <-  w[co][ch*32 + oc*8 + j][tap]
    const size_t n = (size_t)Cout * Cin * 9;
    float* tmp = (float*)malloc(n * sizeof(float));
    const int nchunk = Cin / 32;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int tap = 0; tap < 9; ++tap)
            for (int oc = 0; oc < 4; ++oc)
                for (int co = 0; co < Cout; ++co)
                    for (int j = 0; j < 8; ++j) {
                        const int ci = ch * 32 + oc * 8 + j;
                        tmp[((((size_t)ch * 9 + tap) * 4 + oc) * Cout + co) * 8 + j] = w[((size_t)co * Cin + ci) * 9 + tap];
                    }
    const float sc = split_weights_host(tmp, n, hi, lo);
    free(tmp);
    return sc;
}

void pack_conv1a(const float* w, float* dst) {
    for (int tap = 0; tap < 9; ++tap)
        for (int co = 0; co < 64; ++co) dst[tap * 64 + co] = w[co * 9 + tap];
}

float pack_conv3x3_split_from_gemm(const float* w_gemm, int Cout, int Cin, unsigned short* hi, unsigned short* lo, int cin_used) {
    // cin_used <= Cin: only the first cin_used input channels are packed (the rest are zero padding of the stored maps)
    if (cin_used <= 0) cin_used = Cin;
    float* oihw = (float*)malloc((size_t)Cout * cin_used * 9 * sizeof(float));
    if (!oihw) return 0.0f;
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < cin_used; ++ci) oihw[((size_t)co * cin_used + ci) * 9 + t] = w_gemm[((size_t)co * 9 + t) * Cin + ci];
    const float sc = pack_conv3x3_split(oihw, Cout, cin_used, hi, lo);
    free(oihw);
    return sc;
}
