// Device kernels of the DUSt3R pair network (dust3r.hip) that are not GEMMs / convolutions / attention: patch extraction,
// LayerNorm, the 2-D rotary embedding + f16 plane split that feeds the attention kernel, the layout shuffles of the DPT
// head, bilinear x2 up-sampling and the point-map regression.  All maps are NHWC, token rows are [sequence][R][C] with R =
// tokens per image rounded up to 128 (the attention kernel's tile) and rows >= T unused.
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------ patch extraction (PatchEmbedDust3R's 16x16/16 conv as a GEMM)
// img [NI,3,H,W] in [0,1] -> A [NI*R][768], column c*256 + py*16 + px = (img[n,c,16ty+py,16tx+px] - 0.5) / 0.5 (the wrapper's
// normalisation, duster.py:60-64); rows t >= T are zero.  One thread = 4 consecutive px.
__global__ __launch_bounds__(256) void du_patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, int T,
                                                          int R, long n4) {
    const int wg = W >> 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int k4 = (int)(i % 192);
        const long row = i / 192;
        const int t = (int)(row % R);
        const long n = row / R;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const int k = k4 * 4, c = k >> 8, py = (k >> 4) & 15, px = k & 15;
            const int ty = t / wg, tx = t - ty * wg;
            const float4 s = *reinterpret_cast<const float4*>(img + ((n * 3 + c) * H + ty * 16 + py) * (long)W + tx * 16 + px);
            v = make_float4((s.x - 0.5f) / 0.5f, (s.y - 0.5f) / 0.5f, (s.z - 0.5f) / 0.5f, (s.w - 0.5f) / 0.5f);
        }
        *reinterpret_cast<float4*>(out + i * 4) = v;
    }
}

// ------------------------------------------------------------------ LayerNorm over C <= 1024 channels (C % 4 == 0), one wave per row
// two-pass moments (mean, then the centred sum of squares), eps inside the square root: torch.nn.LayerNorm.  gamma == nullptr: the
// normalisation alone, (x - mean) * rstd -- the affine part of a LayerNorm that only feeds a linear layer is folded into that layer
// at pack time (backend.py: dust3r_matrices)
__global__ __launch_bounds__(256) void du_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ out, long M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float4 v[4];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            v[k] = *reinterpret_cast<const float4*>(xr + c);
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < C) {
            const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < C) {
            if (gamma == nullptr) {
                *reinterpret_cast<float4*>(out + row * C + c) =
                    make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd, (v[k].w - mean) * rstd);
                continue;
            }
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            *reinterpret_cast<float4*>(out + row * C + c) = make_float4((v[k].x - mean) * rstd * g.x + b.x, (v[k].y - mean) * rstd * g.y + b.y,
                                                                         (v[k].z - mean) * rstd * g.z + b.z, (v[k].w - mean) * rstd * g.w + b.w);
        }
    }
}

// (mean, rstd) of every row, the same two-pass moments as du_layernorm_kernel: the statistics of a LayerNorm whose normalisation is
// applied in the epilogue of the consuming GEMM (GemmP.ln_stats)
__global__ __launch_bounds__(256) void du_rowstats_kernel(const float* __restrict__ x, float* __restrict__ stats, long M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float4 v[4];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            v[k] = *reinterpret_cast<const float4*>(xr + c);
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < C) {
            const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
}

// ------------------------------------------------------------------ q / k: RoPE2D + scale + f16 hi / lo planes [seq][head][R][64]
// src [nseq*R][ld] f32, the `heads` x 64 features of q (or k) start at column col0.  Head layout: a y half and an x half of 32
// features; inside a half feature i < 16 pairs with i + 16: (a, b) -> (a cos - b sin, b cos + a sin), angle = position *
// inv_freq[i], position = token row (y half) / column (x half) on the wg-wide grid.  One thread = 8 + 8 paired features of one
// token: (half hf, chunk c) -> features 32 hf + 8 c + [0, 8) and + 16.  Rows t >= T are written as zeros (the attention kernel
// masks keys >= cnt by probability 0, and 0 x NaN would poison the P.V product).  seq_out0 = first sequence slot written.
__global__ __launch_bounds__(256) void du_rope_split_kernel(const float* __restrict__ src, long ld, int col0, int heads, int T, int R, int wg,
                                                            const float* __restrict__ inv_freq, float alpha, int rope,
                                                            unsigned short* __restrict__ planes, size_t plane_halves, int seq_out0,
                                                            long nthreads, const int* __restrict__ seq_T, const int* __restrict__ seq_wg) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nthreads; i += (long)gridDim.x * 256) {
        const int sub = (int)(i & 3);  // (hf, c)
        const int hf = sub >> 1, c = sub & 1;
        long r = i >> 2;
        const int head = (int)(r % heads);
        r /= heads;
        const int t = (int)(r % R);
        const int seq = (int)(r / R);
        if (seq_T) {  // sequences on token grids of different shapes
            T = seq_T[seq];
            wg = seq_wg[seq];
        }
        uint4 ha = make_uint4(0u, 0u, 0u, 0u), la = ha, hb = ha, lb = ha;
        if (t < T) {
            const float* s = src + ((long)seq * R + t) * ld + col0 + head * 64 + hf * 32 + c * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(s), a1 = *reinterpret_cast<const float4*>(s + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(s + 16), b1 = *reinterpret_cast<const float4*>(s + 20);
            float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            if (rope) {
                const int ty = t / wg, tx = t - ty * wg;
                const float pos = (float)(hf ? tx : ty);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ang = pos * inv_freq[c * 8 + j];
                    const float cs = cosf(ang), sn = sinf(ang);
                    const float av = a[j], bv = b[j];
                    a[j] = av * cs + (-bv) * sn;
                    b[j] = bv * cs + av * sn;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] *= alpha;
                b[j] *= alpha;
            }
            split8(make_float4(a[0], a[1], a[2], a[3]), make_float4(a[4], a[5], a[6], a[7]), ha, la);
            split8(make_float4(b[0], b[1], b[2], b[3]), make_float4(b[4], b[5], b[6], b[7]), hb, lb);
        }
        unsigned short* o = planes + (((size_t)(seq_out0 + seq) * heads + head) * R + t) * 64 + hf * 32 + c * 8;
        *reinterpret_cast<uint4*>(o) = ha;
        *reinterpret_cast<uint4*>(o + 16) = hb;
        *reinterpret_cast<uint4*>(o + plane_halves) = la;
        *reinterpret_cast<uint4*>(o + plane_halves + 16) = lb;
    }
}

// RoPE2D tables of the token grid for the fused projection epilogue (EPI_QKV_VIT): cos / sin [R][32], entry 16 half + i =
// angle (row for half 0, column for half 1) x inv_freq[i], the same cosf / sinf of the same float angle as du_rope_split_kernel
__global__ __launch_bounds__(256) void du_rope_table_kernel(const float* __restrict__ inv_freq, int T, int R, int wg, float* __restrict__ rcos,
                                                            float* __restrict__ rsin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * 32) return;
    const int t = i >> 5, e = i & 31;
    float cs = 1.0f, sn = 0.0f;
    if (t < T) {
        const int ty = t / wg, tx = t - ty * wg;
        const float ang = (float)((e >> 4) ? tx : ty) * inv_freq[e & 15];
        cs = cosf(ang);
        sn = sinf(ang);
    }
    rcos[i] = cs;
    rsin[i] = sn;
}

// ------------------------------------------------------------------ v: f16 hi / lo planes of V^T [seq][head][64][R]
// one workgroup = 64 tokens x one head: f32 tile through LDS, each thread writes 16 consecutive tokens of one feature
__global__ __launch_bounds__(256) void du_vt_split_kernel(const float* __restrict__ src, long ld, int col0, int heads, int T, int R,
                                                          unsigned short* __restrict__ planes, size_t plane_halves, int seq_out0,
                                                          const int* __restrict__ seq_T) {
    __shared__ float tile[64][65];
    const int tiles = R >> 6;
    int b = blockIdx.x;
    const int tt = b % tiles;
    b /= tiles;
    const int head = b % heads, seq = b / heads;
    if (seq_T) T = seq_T[seq];
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int tl = (tid >> 4) + 16 * it, f4 = (tid & 15) * 4;
        const int t = tt * 64 + tl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) v = *reinterpret_cast<const float4*>(src + ((long)seq * R + t) * ld + col0 + head * 64 + f4);
        tile[tl][f4] = v.x;
        tile[tl][f4 + 1] = v.y;
        tile[tl][f4 + 2] = v.z;
        tile[tl][f4 + 3] = v.w;
    }
    __syncthreads();
    const int f = tid >> 2, g = tid & 3;  // feature, group of 16 tokens
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = tile[g * 16 + j][f];
    uint4 h0, l0, h1, l1;
    split8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), h0, l0);
    split8(make_float4(x[8], x[9], x[10], x[11]), make_float4(x[12], x[13], x[14], x[15]), h1, l1);
    unsigned short* o = planes + (((size_t)(seq_out0 + seq) * heads + head) * 64 + f) * R + tt * 64 + g * 16;
    *reinterpret_cast<uint4*>(o) = h0;
    *reinterpret_cast<uint4*>(o + 8) = h1;
    *reinterpret_cast<uint4*>(o + plane_halves) = l0;
    *reinterpret_cast<uint4*>(o + plane_halves + 8) = l1;
}

// ------------------------------------------------------------------ row gathers
// dst[(s, t)] = src[(map[s], t)] for t < rows_out: sequences of `rows_in` rows -> sequences of `rows_out` rows (C % 4 == 0).
// map == nullptr: identity.  Used to copy an image's tokens into the decoder streams that read it (rows_in = rows_out = R)
// and to drop the padding rows in front of the DPT head (rows_out = T).
__global__ __launch_bounds__(256) void du_gather_seq_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst,
                                                            int rows_in, int rows_out, int C4, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int t = (int)(r % rows_out);
        const int s = (int)(r / rows_out);
        const int ss = map ? map[s] : s;
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[((long)ss * rows_in + t) * C4 + c];
    }
}

// ------------------------------------------------------------------ transposed convolution with kernel = stride, second half
// the GEMM produced src [B*h*w][s*s*C] with column (dy*s + dx)*C + co; dst NHWC [B][s*h][s*w][C]
__global__ __launch_bounds__(256) void du_pixel_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int s,
                                                               int C4, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int ox = (int)(r % (s * w));
        r /= (s * w);
        const int oy = (int)(r % (s * h));
        const long b = r / (s * h);
        const int y = oy / s, dy = oy - y * s, x = ox / s, dx = ox - x * s;
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[(((b * h + y) * w + x) * (s * s) + dy * s + dx) * C4 + c];
    }
}

// top-left crop of NHWC maps: dst [B][h][w][C] = src [B][hs][ws][C][:, :h, :w]
__global__ __launch_bounds__(256) void du_crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int hs, int ws, int h, int w, int C4,
                                                      long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const long b = r / h;
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[((b * hs + y) * ws + x) * C4 + c];
    }
}

// ------------------------------------------------------------------ element-wise pieces of the fusion blocks
__global__ __launch_bounds__(256) void du_relu_kernel(const float* __restrict__ x, float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
}
// sum = a + b, relu_out = relu(sum)
__global__ __launch_bounds__(256) void du_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ sum,
                                                          float* __restrict__ relu_out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        const float4 sv = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
        reinterpret_cast<float4*>(sum)[i] = sv;
        reinterpret_cast<float4*>(relu_out)[i] = make_float4(fmaxf(sv.x, 0.f), fmaxf(sv.y, 0.f), fmaxf(sv.z, 0.f), fmaxf(sv.w, 0.f));
    }
}

// bilinear x2, align_corners=True (ATen's source index (h - 1) / (2h - 1) * o and evaluation order), NHWC
__global__ __launch_bounds__(256) void du_upsample2_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int C4, long n4) {
    const int Ho = 2 * h, Wo = 2 * w;
    const float sy = (float)(h - 1) / (float)(Ho - 1), sx = (float)(w - 1) / (float)(Wo - 1);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long b = t / Ho;
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float4* base = reinterpret_cast<const float4*>(in) + b * (long)h * w * C4 + c;
        const float4 v00 = base[((long)y0 * w + x0) * C4], v01 = base[((long)y0 * w + x1) * C4];
        const float4 v10 = base[((long)y1 * w + x0) * C4], v11 = base[((long)y1 * w + x1) * C4];
        float4 o;
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// ------------------------------------------------------------------ head.4 (1x1, 128 -> 4) + point-map post-processing
// feat [npix][128] (ReLU already applied), w [4][128], b [4] -> pts3d [npix][3] = xyz / max(|xyz|, 1e-8) * expm1(|xyz|), conf = 1 +
// exp(c) (depth_mode ('exp', -inf, inf), conf_mode ('exp', 1, inf)); raw [npix][4] optional (parity tests).  32 lanes per pixel.
__global__ __launch_bounds__(256) void du_regress_kernel(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ pts, float* __restrict__ conf, float* __restrict__ raw, long npix) {
    const int l = threadIdx.x & 31;
    const float4 w0 = *reinterpret_cast<const float4*>(w + 4 * l), w1 = *reinterpret_cast<const float4*>(w + 128 + 4 * l);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 256 + 4 * l), w3 = *reinterpret_cast<const float4*>(w + 384 + 4 * l);
    for (long p = (long)blockIdx.x * 8 + (threadIdx.x >> 5); p < npix; p += (long)gridDim.x * 8) {
        const float4 f = *reinterpret_cast<const float4*>(feat + p * 128 + 4 * l);
        float s0 = (f.x * w0.x + f.y * w0.y) + (f.z * w0.z + f.w * w0.w);
        float s1 = (f.x * w1.x + f.y * w1.y) + (f.z * w1.z + f.w * w1.w);
        float s2 = (f.x * w2.x + f.y * w2.y) + (f.z * w2.z + f.w * w2.w);
        float s3 = (f.x * w3.x + f.y * w3.y) + (f.z * w3.z + f.w * w3.w);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64);
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
            s3 += __shfl_xor(s3, o, 64);
        }
        if (l == 0) {
            const float x = s0 + bias[0], y = s1 + bias[1], z = s2 + bias[2], c = s3 + bias[3];
            if (raw) *reinterpret_cast<float4*>(raw + p * 4) = make_float4(x, y, z, c);
            const float d = sqrtf(x * x + y * y + z * z);
            const float sc = expm1f(d) / fmaxf(d, 1e-8f);
            pts[p * 3 + 0] = x * sc;
            pts[p * 3 + 1] = y * sc;
            pts[p * 3 + 2] = z * sc;
            conf[p] = 1.0f + expf(c);
        }
    }
}

// ------------------------------------------------------------------ MASt3R local features: pixel shuffle + L2 normalisation
// lf [P*T][(dd + 1) * 256]: column c * 256 + (y % 16) * 16 + (x % 16) of token (y / 16, x / 16) is channel c of pixel (y, x)
// (F.pixel_shuffle(., 16) of the [B, (dd + 1) * 256, H/16, W/16] view).  desc [npix][dd] = channels 0 .. dd-1 / their norm
// (desc_mode 'norm'), desc_conf [npix] = exp(channel dd) (desc_conf_mode ('exp', 0, inf)).  One thread per pixel, consecutive
// threads = consecutive positions inside a token (coalesced 64-byte runs per channel).
__global__ __launch_bounds__(256) void du_desc_kernel(const float* __restrict__ lf, float* __restrict__ desc, float* __restrict__ desc_conf, int H,
                                                      int W, int dd, long npix) {
    const int wg = W >> 4, T = (H >> 4) * wg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        // i enumerates (image, token, position inside the token)
        const int sub = (int)(i & 255);
        const long bt = i >> 8;
        const int t = (int)(bt % T);
        const long b = bt / T;
        const int y = (t / wg) * 16 + (sub >> 4), x = (t % wg) * 16 + (sub & 15);
        const float* src = lf + bt * (long)(dd + 1) * 256 + sub;
        float ss = 0.0f;
        for (int c = 0; c < dd; ++c) {
            const float u = src[(long)c * 256];
            ss += u * u;
        }
        const float nrm = sqrtf(ss);
        const long p = (b * H + y) * (long)W + x;
        for (int c = 0; c < dd; ++c) desc[p * dd + c] = src[(long)c * 256] / nrm;  // second read of the 64-byte runs: L2 hits
        desc_conf[p] = expf(src[(long)dd * 256]);
    }
}

// stream table of the decoder: smap[s] = view-1 image of pair s, smap[P + s] = its view-2 image
__global__ void du_smap_kernel(const int* __restrict__ pairs, int* __restrict__ smap, int P, int NI) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * P) {
        const int p = i < P ? i : i - P, v = i < P ? 0 : 1;
        smap[i] = min(max(pairs[2 * p + v], 0), NI - 1);
    }
}

// weight-set table of the merged decoder launches (GemmP.wsel is indexed by sequence pair): pairs of the first P streams -> set 0
// (`dec_blocks`), of the last P -> set 1 (`dec_blocks2`); `rev` the other way round
__global__ void du_wsel_kernel(int* __restrict__ fwd, int* __restrict__ rev, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
        const int side = (2 * i >= P) ? 1 : 0;
        fwd[i] = side;
        rev[i] = 1 - side;
    }
}

__global__ void du_fill_int_kernel(int* p, int v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace
