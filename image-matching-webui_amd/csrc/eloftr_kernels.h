// Device kernels of the EfficientLoFTR path (imcui/hloc/matchers/eloftr.py:79 -> upstream LoFTR.forward) that are not
// GEMM-shaped: the 1 -> 64 first convolution, token aggregation (depth-wise 4x4 conv / 4x4 max-pool + LayerNorm), the
// 2-D rotary embedding, soft-max attention on the aggregated grid (8 heads x 32), bilinear up-sampling with
// align_corners=False, and the two-stage fine matching on 8x8 / 10x10 windows.  Included by eloftr.hip only.
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------ first block: 3x3 stride 2 pad 1, 1 -> 64 (+ReLU)
// The re-parameterised RepVGG block (3x3 + 1x1 + BatchNorms folded on the host).  16 lanes x 4 channels = one pixel.
// w: [9][64] tap-major, bias [64]; out NHWC [N, H/2, W/2, 64].
__global__ __launch_bounds__(256) void el_conv0_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                       int Ho, int Wo, long npix) {
    const int c4 = (threadIdx.x & 15) * 4;
    const long stride = (long)gridDim.x * 16;
    for (long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4); p < npix; p += stride) {
        const int ox = (int)(p % Wo);
        const long q = p / Wo;
        const int oy = (int)(q % Ho);
        const float* img = in + (q / Ho) * (long)H * W;
        float4 a = *reinterpret_cast<const float4*>(bias + c4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[(long)iy * W + ix] : 0.0f;
                const float4 k = *reinterpret_cast<const float4*>(w + (ky * 3 + kx) * 64 + c4);
                a.x = fmaf(v, k.x, a.x);
                a.y = fmaf(v, k.y, a.y);
                a.z = fmaf(v, k.z, a.z);
                a.w = fmaf(v, k.w, a.w);
            }
        }
        *reinterpret_cast<float4*>(out + p * 64 + c4) = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
    }
}

// ------------------------------------------------------------------ bilinear up-sampling by an integer factor, align_corners=False
// ATen upsample_bilinear2d with scale_factor given: src = (dst + 0.5) / f - 0.5, clamped at 0; NHWC, C % 4 == 0.
// `in` holds n images of h x w.  alpha multiplies the result (1 except where a constant is folded in).
__global__ __launch_bounds__(256) void el_upsample_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w,
                                                          int C, int f, long nout4) {
    const int C4 = C >> 2;
    const int Ho = f * h, Wo = f * w;
    const float rs = 1.0f / (float)f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nout4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long b = t / Ho;
        const float fy = fmaxf(rs * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(rs * ((float)ox + 0.5f) - 0.5f, 0.0f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* base = in + b * (long)h * w * C + c4 * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x0) * C);
        const float4 v01 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x1) * C);
        const float4 v10 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x0) * C);
        const float4 v11 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x1) * C);
        float4 o;
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        *reinterpret_cast<float4*>(out + i * 4) = o;
    }
}

// ------------------------------------------------------------------ token aggregation + LayerNorm (256 channels)
// mode 0: depth-wise 4x4 stride-4 convolution (queries; dw [256][16], taps row-major), mode 1: 4x4 stride-4 max-pool
// (keys / values); then LayerNorm(256) with (gamma, beta).  One block of 256 threads (= channels) per output token.
// x: NHWC [n, h, w, 256] -> y: [n, h/4 * w/4, 256].
__global__ __launch_bounds__(256) void el_aggregate_kernel(const float* __restrict__ x, const float* __restrict__ dw,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int h, int w, int mode, float* __restrict__ y) {
    __shared__ float red[4];
    const int c = threadIdx.x, lane = c & 63, wv = c >> 6;
    const int ah = h / 4, aw = w / 4;
    const int tok = blockIdx.x;  // over n * ah * aw
    const int ax = tok % aw, ay = (tok / aw) % ah, n = tok / (aw * ah);
    const float* src = x + (((size_t)n * h + ay * 4) * w + ax * 4) * 256 + c;
    float v;
    if (mode == 0) {
        v = 0.0f;  // ATen's depth-wise kernel accumulates the taps in order, starting from 0 (no bias)
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(src[((size_t)ky * w + kx) * 256], dw[c * 16 + ky * 4 + kx], v);
    } else {
        v = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaxf(v, src[((size_t)ky * w + kx) * 256]);
    }
    float s = wave_sum(v);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / 256.0f);
    __syncthreads();
    const float d = v - mean;
    s = wave_sum(d * d);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / 256.0f) + 1e-5f);
    y[(size_t)tok * 256 + c] = d * rstd * gamma[c] + beta[c];
}

// ------------------------------------------------------------------ 2-D rotary embedding on q and k (self attention only)
// Channel pair (2e, 2e+1) is rotated by angle pos * inv_freq[e >> 1], pos = 1-based row (e even) or column (e odd) of the
// aggregated grid: emb[..., 0::2] = i * inv_freq, emb[..., 1::2] = j * inv_freq, cos / sin repeat_interleave(2).
// q' = q cos + rotate_half(q) sin with rotate_half(x)[2e] = -x[2e+1], [2e+1] = x[2e].  t: [n * ah * aw, 256] in place.
__global__ __launch_bounds__(256) void el_rope_kernel(float* __restrict__ q, float* __restrict__ k, const float* __restrict__ inv_freq,
                                                      int ah, int aw, long npairs) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npairs; i += (long)gridDim.x * 256) {
        const int e = (int)(i & 127);
        const long tok = i >> 7;
        const int ax = (int)(tok % aw), ay = (int)((tok / aw) % ah);
        const float pos = (float)(((e & 1) ? ax : ay) + 1);
        const float ang = pos * inv_freq[e >> 1];
        const float cs = cosf(ang), sn = sinf(ang);
        float2 a = *reinterpret_cast<float2*>(q + i * 2);
        float2 b = *reinterpret_cast<float2*>(k + i * 2);
        // (q * cos) + (rot * sin), each product rounded, as the reference evaluates it
        *reinterpret_cast<float2*>(q + i * 2) = make_float2(a.x * cs + (-a.y) * sn, a.y * cs + a.x * sn);
        *reinterpret_cast<float2*>(k + i * 2) = make_float2(b.x * cs + (-b.y) * sn, b.y * cs + b.x * sn);
    }
}

// ------------------------------------------------------------------ soft-max attention, 8 heads x 32, <= a few thousand tokens
// Four lanes per query (8 of the 32 head dimensions each; partial dot products meet through two lane exchanges), 64
// queries per block, keys / values of the head streamed through LDS in chunks of 64.  Two passes over the keys:
// maximum first, then exp / accumulate -- the sequence is tiny (La = (H/32)(W/32) = 300 at 640x480), exact f32.
// q: [nq_seq * Lq, 256], k, v: [.. * Lk, 256]; query sequence s attends to key sequence s (pointers pre-offset).
#define EL_ATT_CHUNK 64
__global__ __launch_bounds__(256) void el_attention_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, int Lq, int Lk, float scale,
                                                           float* __restrict__ O) {
    __shared__ float Ks[EL_ATT_CHUNK * 32], Vs[EL_ATT_CHUNK * 32];
    const int head = blockIdx.y, seq = blockIdx.z;
    const int part = threadIdx.x & 3;
    const int qi = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = qi < Lq;
    float q[8], acc[8];
    const float* qp = Q + ((size_t)seq * Lq + (live ? qi : 0)) * 256 + head * 32 + part * 8;
    {
        const float4 t0 = *reinterpret_cast<const float4*>(qp), t1 = *reinterpret_cast<const float4*>(qp + 4);
        q[0] = t0.x, q[1] = t0.y, q[2] = t0.z, q[3] = t0.w, q[4] = t1.x, q[5] = t1.y, q[6] = t1.z, q[7] = t1.w;
    }
    const float* kb = K + (size_t)seq * Lk * 256 + head * 32;
    const float* vb = V + (size_t)seq * Lk * 256 + head * 32;
    auto dot = [&](int j) __attribute__((always_inline)) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s = fmaf(q[d], Ks[j * 32 + part * 8 + d], s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        return s * scale;
    };
    float m = -INFINITY;
    for (int k0 = 0; k0 < Lk; k0 += EL_ATT_CHUNK) {
        const int nk = min(EL_ATT_CHUNK, Lk - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < nk * 8; i += 256)
            *reinterpret_cast<float4*>(Ks + i * 4) = *reinterpret_cast<const float4*>(kb + (size_t)(k0 + (i >> 3)) * 256 + (i & 7) * 4);
        __syncthreads();
        for (int j = 0; j < nk; ++j) m = fmaxf(m, dot(j));
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
    float l = 0.0f;
    for (int k0 = 0; k0 < Lk; k0 += EL_ATT_CHUNK) {
        const int nk = min(EL_ATT_CHUNK, Lk - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < nk * 8; i += 256) {
            *reinterpret_cast<float4*>(Ks + i * 4) = *reinterpret_cast<const float4*>(kb + (size_t)(k0 + (i >> 3)) * 256 + (i & 7) * 4);
            *reinterpret_cast<float4*>(Vs + i * 4) = *reinterpret_cast<const float4*>(vb + (size_t)(k0 + (i >> 3)) * 256 + (i & 7) * 4);
        }
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            const float p = expf(dot(j) - m);
            l += p;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] = fmaf(p, Vs[j * 32 + part * 8 + d], acc[d]);
        }
    }
    if (live) {
        const float inv = 1.0f / l;
        float* op = O + ((size_t)seq * Lq + qi) * 256 + head * 32 + part * 8;
        *reinterpret_cast<float4*>(op) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
        *reinterpret_cast<float4*>(op + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
    }
}

// ------------------------------------------------------------------ fine level: windows + two-stage matching, one block per match
// The fine feature map is the bilinear x2 up-sampling (align_corners=False) of the 1/2-resolution map R [2B, H/2, W/2, 64]
// produced by the last fusion block; only the 8x8 (image 0, stride 8, no padding) and 10x10 (image 1, stride 8,
// padding 1 -> zeros outside the full-resolution image) windows of the matched cells are ever read, so they are
// interpolated here instead of materialising the [2B, H, W, 64] map.
//   stage 1: first 56 channels, s = a0 . a1 / 56; conf = softmax over the 64 positions x softmax over the 100
//            positions; the 8x8 interior of the 10x10 grid; argmax over 64 x 64 (first maximum in flattened order).
//   stage 2: last 8 channels: heat = softmax((b0[il] . b1[3x3 block] / sqrt 8) / 10) on the 10x10 grid at rows
//            ri + {-1,0,1}, columns rj + {-1,0,1} (negative indices wrap, as the reference's advanced indexing does);
//            key-point 1 += expectation of the normalised 3x3 grid * (3 // 2) * fine_scale.
// value of channel c at full-resolution pixel (y, x) of the x2 up-sampled map, read from a 6 x 6 patch of the
// 1/2-resolution map staged in LDS (patch origin (oy, ox) in 1/2-resolution pixels; hh x wh = size of that map)
__device__ __forceinline__ float el_fine_sample(const float* __restrict__ P, int oy, int ox, int hh, int wh, int y, int x, int c) {
    const float fy = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.0f), fx = fmaxf(0.5f * ((float)x + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hh - 1 ? 1 : 0), x1 = x0 + (x0 < wh - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float v00 = P[((y0 - oy) * 6 + (x0 - ox)) * 64 + c], v01 = P[((y0 - oy) * 6 + (x1 - ox)) * 64 + c];
    const float v10 = P[((y1 - oy) * 6 + (x0 - ox)) * 64 + c], v11 = P[((y1 - oy) * 6 + (x1 - ox)) * 64 + c];
    return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

__global__ __launch_bounds__(256) void el_fine_kernel(const float* __restrict__ R, const int* __restrict__ mb,
                                                      const int* __restrict__ mi, const int* __restrict__ mj,
                                                      const int* __restrict__ nmatch, int B, int H0, int W0, int H1, int W1,
                                                      int wc0, int wc1, float scale_c, float scale_f, float* __restrict__ kp0,
                                                      float* __restrict__ kp1, float* __restrict__ dbg_win) {
    extern __shared__ float el_sm[];
    float* F0 = el_sm;            // [64][65]  (row stride 65: conflict-free column walks)
    float* F1 = F0 + 64 * 65;     // [100][65]
    float* S = F1 + 100 * 65;     // [64][100]
    float* cmax = S + 6400;       // [100] column (over the 64 positions of window 0) max / sum
    float* csum = cmax + 100;
    float* rmax = csum + 100;     // [64] row (over the 100 positions of window 1) max / sum
    float* rsum = rmax + 64;
    __shared__ float bestv[256];
    __shared__ int besti[256];
    const int m = blockIdx.x;
    if (m >= *nmatch) return;
    const int tid = threadIdx.x;
    const int b = mb[m], ci = mi[m], cj = mj[m];
    // the B maps of image 1 follow the B maps of image 0 (sizes may differ per side)
    const int hh0 = H0 / 2, wh0 = W0 / 2, hh1 = H1 / 2, wh1 = W1 / 2;
    const float* R0 = R + (size_t)b * hh0 * wh0 * 64;
    const float* R1 = R + ((size_t)B * hh0 * wh0 + (size_t)b * hh1 * wh1) * 64;
    const int y0 = (ci / wc0) * 8, x0 = (ci % wc0) * 8;
    const int y1 = (cj / wc1) * 8 - 1, x1 = (cj % wc1) * 8 - 1;
    // Both windows read the same 6 x 6 block geometry of the 1/2-resolution map: rows 4k - 1 .. 4k + 4 for full-resolution
    // rows 8k - 1 .. 8k + 8 (src = y / 2 - 0.25, two taps).  Stage the two blocks in LDS (aliasing S, which is not live
    // yet): 2 x 2304 loads instead of 4 taps x 164 positions x 64 channels from L2.
    float* P0 = S;
    float* P1 = S + 36 * 64;
    const int oy0 = (ci / wc0) * 4 - 1, ox0 = (ci % wc0) * 4 - 1, oy1 = (cj / wc1) * 4 - 1, ox1 = (cj % wc1) * 4 - 1;
    for (int i = tid; i < 36 * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        int gy = oy0 + p / 6, gx = ox0 + p % 6;
        P0[i] = (gy >= 0 && gy < hh0 && gx >= 0 && gx < wh0) ? R0[((size_t)gy * wh0 + gx) * 64 + c] : 0.0f;
        gy = oy1 + p / 6;
        gx = ox1 + p % 6;
        P1[i] = (gy >= 0 && gy < hh1 && gx >= 0 && gx < wh1) ? R1[((size_t)gy * wh1 + gx) * 64 + c] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        F0[p * 65 + c] = el_fine_sample(P0, oy0, ox0, hh0, wh0, y0 + (p >> 3), x0 + (p & 7), c);
    }
    for (int i = tid; i < 100 * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        const int y = y1 + p / 10, x = x1 + p % 10;
        F1[p * 65 + c] = (y >= 0 && y < H1 && x >= 0 && x < W1) ? el_fine_sample(P1, oy1, ox1, hh1, wh1, y, x, c) : 0.0f;
    }
    __syncthreads();
    if (dbg_win != nullptr) {  // parity hook: the unfolded windows [cap][64 + 100][64]
        for (int i = tid; i < 164 * 64; i += 256) {
            const int p = i >> 6, c = i & 63;
            dbg_win[((size_t)m * 164 + p) * 64 + c] = p < 64 ? F0[p * 65 + c] : F1[(p - 64) * 65 + c];
        }
        __syncthreads();  // the windows are scaled in place below
    }
    // stage 1 similarities: (a0 / sqrt 56) . (a1 / sqrt 56).  The first 56 channels are scaled in place (the second stage
    // reads channels 56..63 only), then every thread owns a 4 x 5 block of the 64 x 100 matrix: 9 LDS reads per 20 FMAs.
    const float isq = 1.0f / sqrtf(56.0f);
    for (int i = tid; i < 164 * 56; i += 256) {
        const int p = i / 56, c = i - p * 56;
        float* f = p < 64 ? F0 + p * 65 + c : F1 + (p - 64) * 65 + c;
        *f = *f * isq;
    }
    __syncthreads();
    for (int t = tid; t < 320; t += 256) {
        const int l0 = (t / 20) * 4, r0 = (t % 20) * 5;
        float acc[4][5];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = 0.0f;
        for (int c = 0; c < 56; ++c) {
            float a[4], b[5];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = F0[(l0 + i) * 65 + c];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = F1[(r0 + j) * 65 + c];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) S[(l0 + i) * 100 + r0 + j] = acc[i][j];
    }
    __syncthreads();
    if (tid < 100) {
        float mx = -INFINITY;
        for (int l = 0; l < 64; ++l) mx = fmaxf(mx, S[l * 100 + tid]);
        float sm = 0.0f;
        for (int l = 0; l < 64; ++l) sm += expf(S[l * 100 + tid] - mx);
        cmax[tid] = mx;
        csum[tid] = sm;
    } else if (tid >= 128 && tid < 192) {
        const int l = tid - 128;
        float mx = -INFINITY;
        for (int r = 0; r < 100; ++r) mx = fmaxf(mx, S[l * 100 + r]);
        float sm = 0.0f;
        for (int r = 0; r < 100; ++r) sm += expf(S[l * 100 + r] - mx);
        rmax[l] = mx;
        rsum[l] = sm;
    }
    __syncthreads();
    // argmax of conf over (l, interior r) in flattened (l * 64 + r') order, first maximum
    float bv = -1.0f;
    int bi = 0x7fffffff;
    for (int i = tid; i < 4096; i += 256) {
        const int l = i >> 6, rr = i & 63;
        const int r = ((rr >> 3) + 1) * 10 + (rr & 7) + 1;
        const float s = S[l * 100 + r];
        const float cf = (expf(s - cmax[r]) / csum[r]) * (expf(s - rmax[l]) / rsum[l]);
        if (cf > bv) {  // i ascends within a thread: the first maximum is kept
            bv = cf;
            bi = i;
        }
    }
    bestv[tid] = bv;
    besti[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v = bestv[tid + o];
            const int ix = besti[tid + o];
            if (v > bestv[tid] || (v == bestv[tid] && ix < besti[tid])) {
                bestv[tid] = v;
                besti[tid] = ix;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int idx = besti[0];
        const int il = idx >> 6, ir = idx & 63;
        const float g0x = (float)(il & 7) - 4.0f + 0.5f, g0y = (float)(il >> 3) - 4.0f + 0.5f;
        const float g1x = (float)(ir & 7) - 4.0f + 0.5f, g1y = (float)(ir >> 3) - 4.0f + 0.5f;
        // stage 2
        const int ri = ir >> 3, rj = ir & 7;
        const float is8 = 1.0f / sqrtf(8.0f);
        float pv[9], mx = -INFINITY;
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const int rr = (ri + dy - 1 + 10) % 10, cc = (rj + dx - 1 + 10) % 10;
                float s = 0.0f;
                for (int c = 0; c < 8; ++c) s = fmaf(F1[(rr * 10 + cc) * 65 + 56 + c] * is8, F0[il * 65 + 56 + c], s);
                s = s / 10.0f;
                pv[dy * 3 + dx] = s;
                mx = fmaxf(mx, s);
            }
        float sm = 0.0f;
        for (int t = 0; t < 9; ++t) {
            pv[t] = expf(pv[t] - mx);
            sm += pv[t];
        }
        float ex = 0.0f, ey = 0.0f;
        for (int t = 0; t < 9; ++t) {
            const float hv = pv[t] / sm;
            ex += (float)(t % 3 - 1) * hv;
            ey += (float)(t / 3 - 1) * hv;
        }
        kp0[2 * m + 0] = (float)(ci % wc0) * scale_c + g0x * scale_f;
        kp0[2 * m + 1] = (float)(ci / wc0) * scale_c + g0y * scale_f;
        kp1[2 * m + 0] = (float)(cj % wc1) * scale_c + g1x * scale_f + ex * 1.0f * scale_f;
        kp1[2 * m + 1] = (float)(cj / wc1) * scale_c + g1y * scale_f + ey * 1.0f * scale_f;
    }
}
#define EL_FINE_SMEM ((64 * 65 + 100 * 65 + 6400 + 200 + 128) * sizeof(float))

}  // namespace
