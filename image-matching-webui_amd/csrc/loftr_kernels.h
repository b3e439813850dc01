// Device kernels of the LoFTR path that are not GEMM-shaped:
// first 7x7 conv, bilinear up-sampling, positional encoding, linear attention, LayerNorm,
// dual-softmax coarse matching, fine window gather and fine matching.  Included by loftr.hip (all of it) and eloftr.hip (LayerNorm, dual-softmax matching).
#pragma once
#include "common.h"

namespace {  // internal linkage: loftr.hip and eloftr.hip both include these kernels

// ------------------------------------------------------------------ conv1: 7x7 stride 2 pad 3, 1 -> 128 (+folded BN, ReLU)
// 16 lanes x 8 channels cover the 128 output channels of a pixel; a thread computes FOUR horizontally adjacent output
// pixels, so the two 16-byte weight loads of a tap feed 32 FMAs instead of 8 (the one-pixel version spent its time on
// 98 weight loads per 392 FMAs).  The taps of a pixel are accumulated in the same order as before (ky, kx ascending from
// the bias), so the values are unchanged.  w: [49][128], bias [128].  Wo % 4 == 0 (image widths are multiples of 8).
__global__ __launch_bounds__(256) void lf_conv7_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H,
                                                       int W, int Ho, int Wo, long npix) {
    const int c8 = (threadIdx.x & 15) * 8;
    const long nquad = npix >> 2;
    const long stride = (long)gridDim.x * 16;
    for (long qd = (long)blockIdx.x * 16 + (threadIdx.x >> 4); qd < nquad; qd += stride) {
        const long p = qd << 2;
        const int ox = (int)(p % Wo);
        const long q = p / Wo;
        const int oy = (int)(q % Ho);
        const float* img = in + (q / Ho) * (long)H * W;
        float a[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[t][j] = bias[c8 + j];
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy * 2 - 3 + ky;
            if (iy < 0 || iy >= H) continue;
            // the 13 input columns the four pixels touch in this row
            float v[13];
#pragma unroll
            for (int u = 0; u < 13; ++u) {
                const int ix = ox * 2 - 3 + u;
                v[u] = (ix >= 0 && ix < W) ? img[(long)iy * W + ix] : 0.0f;
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float4 k0 = *reinterpret_cast<const float4*>(w + (ky * 7 + kx) * 128 + c8);
                const float4 k1 = *reinterpret_cast<const float4*>(w + (ky * 7 + kx) * 128 + c8 + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float x = v[2 * t + kx];
                    a[t][0] = fmaf(x, k0.x, a[t][0]);
                    a[t][1] = fmaf(x, k0.y, a[t][1]);
                    a[t][2] = fmaf(x, k0.z, a[t][2]);
                    a[t][3] = fmaf(x, k0.w, a[t][3]);
                    a[t][4] = fmaf(x, k1.x, a[t][4]);
                    a[t][5] = fmaf(x, k1.y, a[t][5]);
                    a[t][6] = fmaf(x, k1.z, a[t][6]);
                    a[t][7] = fmaf(x, k1.w, a[t][7]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float* o = out + (p + t) * 128 + c8;
            *reinterpret_cast<float4*>(o) = make_float4(fmaxf(a[t][0], 0.f), fmaxf(a[t][1], 0.f), fmaxf(a[t][2], 0.f), fmaxf(a[t][3], 0.f));
            *reinterpret_cast<float4*>(o + 4) = make_float4(fmaxf(a[t][4], 0.f), fmaxf(a[t][5], 0.f), fmaxf(a[t][6], 0.f), fmaxf(a[t][7], 0.f));
        }
    }
}

// ------------------------------------------------------------------ bilinear x2, align_corners=True (NHWC)
// out[b, oy, ox, :] = interp(in[b], oy * (h-1)/(2h-1), ox * (w-1)/(2w-1))
__global__ __launch_bounds__(256) void lf_upsample2_kernel(const float* __restrict__ in, float* __restrict__ out, int h,
                                                           int w, int C, long nout4) {
    const int C4 = C >> 2;
    const int Ho = 2 * h, Wo = 2 * w;
    const float sy = (Ho > 1) ? (float)(h - 1) / (float)(Ho - 1) : 0.0f;
    const float sx = (Wo > 1) ? (float)(w - 1) / (float)(Wo - 1) : 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nout4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long b = t / Ho;
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* base = in + b * (long)h * w * C + c4 * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x0) * C);
        const float4 v01 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x1) * C);
        const float4 v10 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x0) * C);
        const float4 v11 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x1) * C);
        float4 o;
        // ATen upsample_bilinear2d: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        *reinterpret_cast<float4*>(out + i * 4) = o;
    }
}

// ------------------------------------------------------------------ PositionEncodingSine (added in place, NHWC tokens)
// channel 4g+0 = sin(x*div_g), 4g+1 = cos(x*div_g), 4g+2 = sin(y*div_g), 4g+3 = cos(y*div_g); x, y 1-based.
__global__ __launch_bounds__(256) void lf_posenc_kernel(float* __restrict__ feat, int hc, int wc, int C, long n,
                                                        int temp_bug_fix) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long t = i / C;
        const int x = (int)(t % wc), y = (int)((t / wc) % hc);
        const int g = c >> 2;
        // div_term = exp(arange(0, C/2, 2) * (-ln(1e4) / (C/2)))   [fixed]
        //          = exp(arange(0, C/2, 2) * (-ln(1e4) / C // 2))   [released weights: == -1.0 for C = 256]
        const float kf = (float)(2 * g);
        const float coef = temp_bug_fix ? (-9.210340371976184f / (float)(C / 2)) : floorf((-9.210340371976184f / (float)C) / 2.0f);
        const float div = expf(kf * coef);
        const float pos = (float)(((c & 2) ? y : x) + 1);
        const float arg = pos * div;
        feat[i] += (c & 1) ? cosf(arg) : sinf(arg);
    }
}

// ------------------------------------------------------------------ LayerNorm over D = 64 * VEC features, one wave per row
// mode 0: y = LN(x)   mode 1: y = res + LN(x)   (y may alias res)
// valid / group (optional): the rows come in groups of `group` rows of which only the first *valid are in use (the
// fine level: capacity B*L*25 window tokens per side, nmatch*25 of them live) -- the rest is skipped
template <int VEC>
__global__ __launch_bounds__(256) void lf_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* res,
                                                           float* y, long rows, int mode, const int* __restrict__ valid = nullptr,
                                                           long group = 0) {
    const int lane = threadIdx.x & 63;
    constexpr int D = 64 * VEC;
    // live rows: all of them, or the first *valid of every group (then the launch is a fixed-size grid-stride loop: the
    // count lives on the device and capacity-sized grids of empty workgroups cost more than the work)
    const long nval = valid ? (long)*valid : 0;
    const long nlive = valid ? (rows / group) * nval : rows;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < nlive; t += (long)gridDim.x * 4) {
        const long row = valid ? (t / nval) * group + (t % nval) : t;
        float v[VEC];
        const float* xr = x + row * D + lane * VEC;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            v[j] = xr[j];
            s += v[j];
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            v[j] -= mean;
            q += v[j] * v[j];
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float o = v[j] * rstd * gamma[lane * VEC + j] + beta[lane * VEC + j];
            if (mode == 1) o += res[row * D + lane * VEC + j];
            y[row * D + lane * VEC + j] = o;
        }
    }
}

#define LA_CHUNK 128
__device__ __forceinline__ float lf_elu1(float x) { return x > 0.0f ? x + 1.0f : (expf(x) - 1.0f) + 1.0f; }

// ------------------------------------------------------------------ linear attention, long sequences (coarse level)
// K, V: [nseq * L, heads * HD] rows of the SOURCE sequences.  Pass 1 reduces chunks of 128 tokens to
// partial KV[d][v] (K' = elu(k)+1, values pre-divided by L) and Ksum[d]; pass 2 adds the partials in a
// fixed order (deterministic, no atomics).
template <int HD>
__global__ __launch_bounds__(256) void lf_la_kv_partial_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                               int L, int heads, float* __restrict__ part, int nchunk) {
    __shared__ float Ks[LA_CHUNK * HD], Vs[LA_CHUNK * HD];
    const int chunk = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const int D = heads * HD;
    const int t0 = chunk * LA_CHUNK;
    const int nt = min(LA_CHUNK, L - t0);
    for (int i = threadIdx.x; i < LA_CHUNK * HD; i += 256) {
        const int t = i / HD, d = i - t * HD;
        float kv = 0.0f, vv = 0.0f;
        if (t < nt) {
            const size_t o = ((size_t)seq * L + t0 + t) * D + h * HD + d;
            kv = lf_elu1(K[o]);
            vv = V[o] / (float)L;
        }
        Ks[i] = kv;
        Vs[i] = vv;
    }
    __syncthreads();
    // HD*HD outputs (+ HD sums): thread -> (d, group of v)
    constexpr int NV = HD * HD / 256;  // outputs per thread: 4 (HD 32) or 1 (HD 16)
    const int d = threadIdx.x / (HD / NV), v0 = (threadIdx.x % (HD / NV)) * NV;
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0f;
    float ks = 0.0f;
    for (int t = 0; t < nt; ++t) {
        const float kk = Ks[t * HD + d];
        ks += kk;
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = fmaf(kk, Vs[t * HD + v0 + j], acc[j]);
    }
    float* o = part + (((size_t)seq * heads + h) * nchunk + chunk) * (HD * HD + HD);
#pragma unroll
    for (int j = 0; j < NV; ++j) o[d * HD + v0 + j] = acc[j];
    if (v0 == 0) o[HD * HD + d] = ks;
}

template <int HD>
__global__ void lf_la_kv_reduce_kernel(const float* __restrict__ part, int nchunk, float* __restrict__ kv) {
    const int gh = blockIdx.x;  // seq * heads + h
    constexpr int N = HD * HD + HD;
    const int i = blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s = 0.0f;
    for (int c = 0; c < nchunk; ++c) s += part[((size_t)gh * nchunk + c) * N + i];  // fixed order: deterministic
    kv[(size_t)gh * N + i] = s;
}

// message[l, h, v] = (sum_d Q'[l,h,d] KV[h][d][v]) * Z[l,h] * L,  Z = 1 / (Q'[l,h,:] . Ksum[h] + 1e-6)
// 64 tokens of one query sequence per block, ONE TOKEN PER LANE: the HD query features and the HD
// outputs of a (token, head) live in registers, KV[h] is read from LDS at wave-uniform addresses
// (broadcast), a lane reads / writes whole 128-byte lines; wave w takes heads w, w+4, ...
// (The first version ran one token per wave with a dependent LDS chain: 160 us per call at 128 x 128
// tokens against an HBM bound of ~15 us.)  The accumulation order over d is ascending as before.
template <int HD>
__global__ __launch_bounds__(256) void lf_la_apply_kernel(const float* __restrict__ Q, const float* __restrict__ kv,
                                                          int seq0, int src_seq0, int L, int Lsrc, int heads,
                                                          float* __restrict__ out) {
    extern __shared__ float la_smem[];  // [heads][HD*HD + HD] of this block's source sequence
    static_assert(HD % 4 == 0, "HD must be a multiple of 4");
    const int D = heads * HD;
    constexpr int N = HD * HD + HD;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seq = seq0 + blockIdx.y;
    const int sseq = src_seq0 + blockIdx.y;
    const long row = (long)seq * L + (long)blockIdx.x * 64 + lane;
    const bool valid = row < (long)(seq + 1) * L;
    const long rrow = valid ? row : (long)(seq + 1) * L - 1;
    for (int i = threadIdx.x; i < heads * N; i += 256) la_smem[i] = kv[(size_t)sseq * heads * N + i];
    __syncthreads();
    for (int h = wv; h < heads; h += 4) {
        float q[HD], acc[HD];
        const float4* qsrc = reinterpret_cast<const float4*>(Q + rrow * D + h * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 t = qsrc[i];
            q[4 * i + 0] = lf_elu1(t.x);
            q[4 * i + 1] = lf_elu1(t.y);
            q[4 * i + 2] = lf_elu1(t.z);
            q[4 * i + 3] = lf_elu1(t.w);
        }
#pragma unroll
        for (int v = 0; v < HD; ++v) acc[v] = 0.0f;
        const float* kh = la_smem + h * N;
        float z = 0.0f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float qd = q[d];
            z = fmaf(qd, kh[HD * HD + d], z);
#pragma unroll
            for (int v4 = 0; v4 < HD / 4; ++v4) {
                const float4 k4 = *reinterpret_cast<const float4*>(kh + d * HD + 4 * v4);
                acc[4 * v4 + 0] = fmaf(qd, k4.x, acc[4 * v4 + 0]);
                acc[4 * v4 + 1] = fmaf(qd, k4.y, acc[4 * v4 + 1]);
                acc[4 * v4 + 2] = fmaf(qd, k4.z, acc[4 * v4 + 2]);
                acc[4 * v4 + 3] = fmaf(qd, k4.w, acc[4 * v4 + 3]);
            }
        }
        const float Z = 1.0f / (z + 1e-6f);
        if (valid) {
            float4* dst = reinterpret_cast<float4*>(out + row * D + h * HD);
#pragma unroll
            for (int i = 0; i < HD / 4; ++i)
                dst[i] = make_float4(acc[4 * i + 0] * Z * (float)Lsrc, acc[4 * i + 1] * Z * (float)Lsrc, acc[4 * i + 2] * Z * (float)Lsrc,
                                     acc[4 * i + 3] * Z * (float)Lsrc);
        }
    }
}

// ------------------------------------------------------------------ dual-softmax coarse matching
// conf(i, j) = softmax over i (column statistics) * softmax over j (row statistics) of sim = (f0 / 16) . (f1 / 16)^T / 0.1; the row
// best (first column attaining it) and the column best come from simred.hip, which never stores sim (rounds 1-4 materialised it and
// read it back in four, then two passes: lf_rowstat / lf_colstat / lf_rowbest / lf_colbest, lf_stats2 / lf_best2 -- removed in round 5).

// per row decision: conf > thr, border, mutual; flag[b*L+i] = 1/0
__global__ void lf_decide_kernel(const float* __restrict__ best, const int* __restrict__ bestj,
                                 const float* __restrict__ cbest, int L, int S, int w0c, int h0c, int w1c, int h1c, int bd,
                                 float thr, int* __restrict__ flag, long n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int b = (int)(t / L), i = (int)(t - (long)b * L);
    const int j = bestj[t];
    const float v = best[t];
    if (j < 0 || j >= S) {  // a row none of whose tiles was evaluated (no confidence of it can exceed the threshold): no match
        flag[t] = 0;
        return;
    }
    const int y0 = i / w0c, x0 = i - y0 * w0c, y1 = j / w1c, x1 = j - y1 * w1c;
    bool ok = v > thr && v == cbest[(size_t)b * S + j];
    ok = ok && y0 >= bd && y0 < h0c - bd && x0 >= bd && x0 < w0c - bd && y1 >= bd && y1 < h1c - bd && x1 >= bd && x1 < w1c - bd;
    flag[t] = ok ? 1 : 0;
}
// ordered compaction of the flagged rows (torch.where order: batch-major, i ascending); single block
__global__ __launch_bounds__(1024) void lf_compact_kernel(const int* __restrict__ flag, const float* __restrict__ best,
                                                          const int* __restrict__ bestj, int L, long n, int cap,
                                                          int* __restrict__ mb, int* __restrict__ mi, int* __restrict__ mj,
                                                          float* __restrict__ mconf, int* __restrict__ nmatch) {
    __shared__ int wsum[16];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (long base = 0; base < n; base += 1024) {
        const long t = base + tid;
        const bool f = (t < n) && flag[t] != 0;
        const unsigned long long bal = __ballot(f);
        const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) woff += wsum[w];
            tot += wsum[w];
        }
        const int pos = s_run + woff + wrank;
        if (f && pos < cap) {
            mb[pos] = (int)(t / L);
            mi[pos] = (int)(t % L);
            mj[pos] = bestj[t];
            mconf[pos] = best[t];
        }
        __syncthreads();
        if (tid == 0) s_run += tot;
        __syncthreads();
    }
    if (tid == 0) *nmatch = min(s_run, cap);
}

// ------------------------------------------------------------------ fine level
// gather: row (side, m, ww) of X [2*cap*25, 256] = [ unfold 5x5 window of feat_f (128) | coarse feature (128, filled later) ]
// also gathers the coarse transformer features of the match into CG [2*cap, 256]
__global__ __launch_bounds__(256) void lf_fine_gather_kernel(const float* __restrict__ feat_f, const float* __restrict__ feat_c,
                                                             const int* __restrict__ mb, const int* __restrict__ mi,
                                                             const int* __restrict__ mj, const int* __restrict__ nmatch,
                                                             int B, int cap, int hf0, int wf0, int hc0, int wc0, int hf1, int wf1,
                                                             int hc1, int wc1, int stride, float* __restrict__ X,
                                                             float* __restrict__ CG) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x, side = blockIdx.y;
    if (m >= *nmatch) return;
    const int b = mb[m];
    const int cell = side ? mj[m] : mi[m];
    // the B images of side 1 follow the B images of side 0 in both feature buffers (sizes may differ per side)
    const int hf = side ? hf1 : hf0, wf = side ? wf1 : wf0, hc = side ? hc1 : hc0, wc = side ? wc1 : wc0;
    const float* ff = feat_f + (side ? (size_t)B * hf0 * wf0 * 128 : 0) + (size_t)b * hf * wf * 128;
    const float* fc = feat_c + (side ? (size_t)B * hc0 * wc0 * 256 : 0) + (size_t)b * hc * wc * 256;
    const int cy = cell / wc, cx = cell - cy * wc;
    const size_t wrow0 = ((size_t)side * cap + m) * 25;
    for (int ww = threadIdx.x >> 6; feat_f != nullptr && ww < 25; ww += 4) {  // (feat_f null: the windows come from the match-sparse FPN stage below)
        const int fy = cy * stride + ww / 5 - 2, fx = cx * stride + ww % 5 - 2;
        float2 v = make_float2(0.f, 0.f);
        if (fy >= 0 && fy < hf && fx >= 0 && fx < wf) v = *reinterpret_cast<const float2*>(ff + ((size_t)fy * wf + fx) * 128 + lane * 2);
        *reinterpret_cast<float2*>(X + (wrow0 + ww) * 256 + lane * 2) = v;
    }
    if (threadIdx.x < 64) {
        const float4 c = *reinterpret_cast<const float4*>(fc + (size_t)cell * 256 + lane * 4);
        *reinterpret_cast<float4*>(CG + ((size_t)side * cap + m) * 256 + lane * 4) = c;
    }
}
// ------------------------------------------------------------------ match-sparse last FPN stage (round 6)
// Only the 5x5 windows (1/2 resolution, stride 4, zero padding 2) around the matched coarse cells of the 128-channel fine map are ever read
// (lf_fine_gather_kernel).  That map is layer1_outconv2(layer1_outconv(x1) + up2(x2out)) -- a 1x1 and two 3x3 convolutions over the whole 1/2
// resolution grid, a quarter of the 1024 x 1024 step.  With few matches it is cheaper to evaluate the three layers on the 9x9 input neighbourhood
// of every window: gather -> 1x1 (GEMM, the up-sampled residual gathered beside it) -> 3x3 valid 9x9 -> 7x7 -> 3x3 valid 7x7 -> 5x5 (implicit GEMM
// over the windows as tiny images).  A position outside the image is ZERO at every level, as the dense layers' zero padding and unfold's padding make it.
// One block per window of one side: x1 [imgs, hf, wf, 128] -> G [n, 81, 128]; up2 (align_corners = True, ATen's formula and order, as the GEMM
// epilogue evaluates it) of x2q [imgs, hf / 2, wf / 2, CQ] -> T [n, 81, CQ]; V [n, 81] = 1 inside the image.
__global__ __launch_bounds__(256) void lf_win_gather_kernel(const float* __restrict__ x1, const float* __restrict__ x2q, const int* __restrict__ mb,
                                                            const int* __restrict__ cells, int m0, int n, int hf, int wf, int wc, int stride, int CQ,
                                                            float* __restrict__ G, float* __restrict__ T, unsigned char* __restrict__ V) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const int m = m0 + i;
    const int b = mb[m], cell = cells[m];
    const int cy = cell / wc, cx = cell - cy * wc;
    const int y00 = cy * stride - 4, x00 = cx * stride - 4;
    const int hq = hf >> 1, wq = wf >> 1;
    const float sy = (float)(hq - 1) / (float)(hf - 1), sx = (float)(wq - 1) / (float)(wf - 1);
    const float* xb = x1 + (size_t)b * hf * wf * 128;
    const float* qb = x2q + (size_t)b * hq * wq * CQ;
    if (threadIdx.x < 81) {
        const int y = y00 + threadIdx.x / 9, x = x00 + threadIdx.x % 9;
        V[(size_t)i * 81 + threadIdx.x] = (y >= 0 && y < hf && x >= 0 && x < wf) ? 1 : 0;
    }
    const int c1 = 128 >> 2, c2 = CQ >> 2;
    for (int e = threadIdx.x; e < 81 * c1; e += 256) {
        const int p = e / c1, c4 = e - p * c1;
        const int y = y00 + p / 9, x = x00 + p % 9;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < hf && x >= 0 && x < wf) v = *reinterpret_cast<const float4*>(xb + ((size_t)y * wf + x) * 128 + c4 * 4);
        *reinterpret_cast<float4*>(G + ((size_t)i * 81 + p) * 128 + c4 * 4) = v;
    }
    for (int e = threadIdx.x; e < 81 * c2; e += 256) {
        const int p = e / c2, c4 = e - p * c2;
        const int oy = y00 + p / 9, ox = x00 + p % 9;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy >= 0 && oy < hf && ox >= 0 && ox < wf) {
            const float fy = sy * (float)oy, fx = sx * (float)ox;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = min(y0 + 1, hq - 1), x1i = min(x0 + 1, wq - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
            const float* base = qb + c4 * 4;
            const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * wq + x0) * CQ);
            const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * wq + x1i) * CQ);
            const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * wq + x0) * CQ);
            const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * wq + x1i) * CQ);
            o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        }
        *reinterpret_cast<float4*>(T + ((size_t)i * 81 + p) * CQ + c4 * 4) = o;
    }
}
// rows of a window level (side x side positions, the interior of the 9x9 grid at offset off) that lie outside the image are set to zero:
// buf [n * side * side rows][ld floats], the first nch channels of a row
__global__ __launch_bounds__(256) void lf_win_mask_kernel(float* __restrict__ buf, const unsigned char* __restrict__ V, int n, int side, int off, long ld, int nch) {
    const int lane = threadIdx.x & 63;
    const long nrow = (long)n * side * side;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrow; r += (long)gridDim.x * 4) {
        const long i = r / (side * side);
        const int p = (int)(r - i * side * side);
        const int py = p / side + off, px = p % side + off;
        if (V[i * 81 + py * 9 + px]) continue;  // (wave-uniform)
        for (int c = lane * 4; c < nch; c += 256) *reinterpret_cast<float4*>(buf + r * ld + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// broadcast the projected coarse feature CW [2*cap, 128] into columns 128..255 of every window row
__global__ __launch_bounds__(256) void lf_fine_fill_kernel(const float* __restrict__ CW, const int* __restrict__ nmatch,
                                                           int cap, float* __restrict__ X) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x, side = blockIdx.y;
    if (m >= *nmatch) return;
    const float2 c = *reinterpret_cast<const float2*>(CW + ((size_t)side * cap + m) * 128 + lane * 2);
    for (int ww = threadIdx.x >> 6; ww < 25; ww += 4)
        *reinterpret_cast<float2*>(X + (((size_t)side * cap + m) * 25 + ww) * 256 + 128 + lane * 2) = c;
}

// linear attention on 25-token windows, D = 128, 8 heads x 16: one block per (match, side)
__global__ __launch_bounds__(256) void lf_la_window_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const int* __restrict__ nmatch,
                                                           int cap, int side0, int cross, float* __restrict__ out) {
    __shared__ float Ks[25 * 128], Vs[25 * 128], Qs[25 * 128], KV[8 * 16 * 16], KS[128];
    const int m = blockIdx.x, side = side0 + blockIdx.y;  // absolute side; all pointers are un-offset
    if (m >= *nmatch) return;
    const size_t qrow0 = ((size_t)side * cap + m) * 25;
    const size_t srow0 = ((size_t)(cross ? 1 - side : side) * cap + m) * 25;
    for (int i = threadIdx.x; i < 25 * 128; i += 256) {
        Ks[i] = lf_elu1(K[srow0 * 128 + i]);
        Vs[i] = V[srow0 * 128 + i] / 25.0f;
        Qs[i] = lf_elu1(Q[qrow0 * 128 + i]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 8 * 16 * 16; e += 256) {
        const int h = e >> 8, d = (e >> 4) & 15, v = e & 15;
        float a = 0.0f;
        for (int t = 0; t < 25; ++t) a = fmaf(Ks[t * 128 + h * 16 + d], Vs[t * 128 + h * 16 + v], a);
        KV[e] = a;
    }
    if (threadIdx.x < 128) {
        float a = 0.0f;
        for (int t = 0; t < 25; ++t) a += Ks[t * 128 + threadIdx.x];
        KS[threadIdx.x] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 25 * 128; i += 256) {
        const int t = i >> 7, c = i & 127, h = c >> 4, v = c & 15;
        float z = 0.0f, a = 0.0f;
        for (int d = 0; d < 16; ++d) {
            const float qd = Qs[t * 128 + h * 16 + d];
            z = fmaf(qd, KS[h * 16 + d], z);
            a = fmaf(qd, KV[(h * 16 + d) * 16 + v], a);
        }
        out[qrow0 * 128 + i] = a * (1.0f / (z + 1e-6f)) * 25.0f;
    }
}

// fine matching: heat = softmax(<f0[center], f1[r]> / sqrt(128)) over the 25 cells, expectation of the
// normalised grid, key-points in image pixels.  One wave per match.
__global__ __launch_bounds__(256) void lf_fine_match_kernel(const float* __restrict__ F, const int* __restrict__ mi,
                                                            const int* __restrict__ mj, const int* __restrict__ nmatch,
                                                            int cap, int w0c, int w1c, float scale_c, float scale_f,
                                                            float* __restrict__ kp0, float* __restrict__ kp1) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= *nmatch) return;
    const float* f0 = F + (((size_t)0 * cap + m) * 25 + 12) * 128;
    const float* f1 = F + ((size_t)1 * cap + m) * 25 * 128;
    const float2 c = *reinterpret_cast<const float2*>(f0 + lane * 2);
    float sim[25];
#pragma unroll
    for (int r = 0; r < 25; ++r) {
        const float2 v = *reinterpret_cast<const float2*>(f1 + r * 128 + lane * 2);
        sim[r] = wave_sum(c.x * v.x + c.y * v.y) * 0.08838834764831845f;  // 1 / sqrt(128)
    }
    float mx = sim[0];
#pragma unroll
    for (int r = 1; r < 25; ++r) mx = fmaxf(mx, sim[r]);
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 25; ++r) {
        sim[r] = expf(sim[r] - mx);
        sum += sim[r];
    }
    float ex = 0.0f, ey = 0.0f;
#pragma unroll
    for (int r = 0; r < 25; ++r) {
        const float hv = sim[r] / sum;
        ex += (-1.0f + 0.5f * (float)(r % 5)) * hv;  // linspace(-1, 1, 5)
        ey += (-1.0f + 0.5f * (float)(r / 5)) * hv;
    }
    if (lane == 0) {
        const int i = mi[m], j = mj[m];
        kp0[2 * m + 0] = (float)(i % w0c) * scale_c;
        kp0[2 * m + 1] = (float)(i / w0c) * scale_c;
        kp1[2 * m + 0] = (float)(j % w1c) * scale_c + ex * 2.0f * scale_f;  // coords * (W // 2) * scale
        kp1[2 * m + 1] = (float)(j / w1c) * scale_c + ey * 2.0f * scale_f;
    }
}

}  // namespace
