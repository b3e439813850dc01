// Device-side image preprocessing: the step BEFORE the hot path.  Restates the host code of
// imcui/hloc/extract_features.py:120-160 (`extract.preprocess`) / :80-99 (`ImageDataset.__getitem__`) / :26-40
// (`resize_image`, "cv2_area"):   uint8 image -> [cv2.cvtColor RGB2GRAY] -> astype(float32) -> cv2.resize(INTER_AREA)
// -> / 255.0, for SHRINKING resizes (the reference itself switches to INTER_LINEAR when a side grows).
// HBM-bound byte work: one thread per output pixel walks the source pixels its cell covers; float32 arithmetic in
// OpenCV's order (horizontal table pass per source row, then the vertical weights; no FMA contraction) so the result
// equals the numpy restatement oracle/preprocess.py:area_resize_f32 bit for bit (parity unpinned: cv2 is absent).
#include <math.h>

#include "common.h"

// Built with -ffp-contract=off (imcui_hip/build.py): the table passes must round like the host code they restate --
// multiply, THEN add -- and hipcc's default would fuse them into FMAs.
#include "imcui_hip.h"

// OpenCV computeResizeAreaTab (modules/imgproc/src/resize.cpp): entries (source index, weight) per destination index
extern "C" int imcui_hip_area_table(int ssize, int dsize, int* start, int* index, float* weight) {
    if (ssize <= 0 || dsize <= 0 || dsize > ssize || !start) return IMCUI_ERR_ARG;
    const double scale = (double)ssize / dsize;
    int k = 0;
    for (int dx = 0; dx < dsize; ++dx) {
        start[dx] = k;
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = fmin(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) {
            if (index) {
                index[k] = sx1 - 1;
                weight[k] = (float)((sx1 - fsx1) / cell);
            }
            ++k;
        }
        for (int sx = sx1; sx < sx2; ++sx) {
            if (index) {
                index[k] = sx;
                weight[k] = (float)(1.0 / cell);
            }
            ++k;
        }
        if (fsx2 - sx2 > 1e-3) {
            if (index) {
                index[k] = sx2;
                weight[k] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
            }
            ++k;
        }
    }
    start[dsize] = k;
    return k;
}

__device__ __forceinline__ float pp_pixel(const unsigned char* p, int C) {
    if (C == 1) return (float)p[0];
    // cv2.cvtColor(RGB2GRAY) on 8-bit pixels: (9798 R + 19235 G + 3735 B + 16384) >> 15
    return (float)((9798u * p[0] + 19235u * p[1] + 3735u * p[2] + 16384u) >> 15);
}

__global__ __launch_bounds__(256) void pp_area_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                      const int* __restrict__ xs, const int* __restrict__ xi,
                                                      const float* __restrict__ xw, const int* __restrict__ ys,
                                                      const int* __restrict__ yi, const float* __restrict__ yw,
                                                      float* __restrict__ out, int h, int w, int fast_fx, int fast_fy) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (dx >= w || dy >= h) return;
    const unsigned char* img = src + (size_t)b * H * W * C;
    float acc = 0.0f;
    if (fast_fx > 0) {
        // integer factors: OpenCV's resizeAreaFast -- plain sum of the block, row-major, times float(1 / area)
        for (int ky = 0; ky < fast_fy; ++ky)
            for (int kx = 0; kx < fast_fx; ++kx)
                acc = __fadd_rn(acc, pp_pixel(img + ((size_t)(dy * fast_fy + ky) * W + dx * fast_fx + kx) * C, C));
        acc = __fmul_rn(acc, 1.0f / (float)(fast_fx * fast_fy));
    } else {
        const int x0 = xs[dx], x1 = xs[dx + 1];
        const int y0 = ys[dy], y1 = ys[dy + 1];
        for (int j = y0; j < y1; ++j) {
            const unsigned char* row = img + (size_t)yi[j] * W * C;
            float buf = __fmul_rn(pp_pixel(row + (size_t)xi[x0] * C, C), xw[x0]);
            for (int k = x0 + 1; k < x1; ++k) buf = __fadd_rn(buf, __fmul_rn(pp_pixel(row + (size_t)xi[k] * C, C), xw[k]));
            const float t = __fmul_rn(buf, yw[j]);
            acc = (j == y0) ? t : __fadd_rn(acc, t);
        }
    }
    out[((size_t)b * h + dy) * w + dx] = __fdiv_rn(acc, 255.0f);
}

extern "C" int imcui_hip_preprocess_area_f32(imcui_hip_t* h, const unsigned char* src, int B, int H, int W, int C,
                                             const int* xstart, const int* xindex, const float* xweight, const int* ystart,
                                             const int* yindex, const float* yweight, float* out, int oh, int ow, void* stream) {
    if (!h || !src || !out || B < 0 || H <= 0 || W <= 0 || (C != 1 && C != 3) || oh <= 0 || ow <= 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "preprocess_area: bad argument");
    if (oh > H || ow > W)
        return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "preprocess_area: %dx%d -> %dx%d grows a side (the reference uses INTER_LINEAR there)", W, H, ow, oh);
    if (B == 0) return IMCUI_OK;
    const bool fast = (W % ow == 0) && (H % oh == 0);
    if (!fast && (!xstart || !xindex || !xweight || !ystart || !yindex || !yweight))
        return imcui_set_err(h, IMCUI_ERR_ARG, "preprocess_area: decimation tables missing (imcui_hip_area_table)");
    hipLaunchKernelGGL(pp_area_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), B), dim3(256), 0, (hipStream_t)stream, src, H, W, C, xstart, xindex,
                       xweight, ystart, yindex, yweight, out, oh, ow, fast ? W / ow : 0, fast ? H / oh : 0);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ growing resize: cv2.INTER_LINEAR
// `resize_image(image, size, "cv2_area")` switches to INTER_LINEAR as soon as a side grows (imcui/hloc/extract_features.py:29-31;
// `superpoint_max` force-resizes every image to 640 x 480, configs/extractors.py:29-45, so any smaller image takes this path).
// Tap table of OpenCV's resizeGeneric_ set-up for float images: half-pixel centres, fx = (float)((dx + 0.5) * scale - 0.5) with a
// double scale, sx = floor(fx), fx -= sx.  horizontal = 1: a tap outside the image is pinned to the border pixel with weight 0;
// horizontal = 0 (rows): the weight is kept and both indices are clipped.  w1 = weight of the second tap.
extern "C" int imcui_hip_linear_table(int ssize, int dsize, int horizontal, int* i0, int* i1, float* w1) {
    if (ssize <= 0 || dsize <= 0 || !i0 || !i1 || !w1) return IMCUI_ERR_ARG;
    const double inv = (double)dsize / ssize;
    const double scale = 1.0 / inv;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f = f - (float)s;
        int a = s, b = s + 1;
        if (horizontal) {
            if (s < 0) a = b = 0, f = 0.0f;
            if (s >= ssize - 1) a = b = ssize - 1, f = 0.0f;
        } else {
            a = a < 0 ? 0 : (a > ssize - 1 ? ssize - 1 : a);
            b = b < 0 ? 0 : (b > ssize - 1 ? ssize - 1 : b);
        }
        i0[d] = a;
        i1[d] = b;
        w1[d] = f;
    }
    return IMCUI_OK;
}

__global__ __launch_bounds__(256) void pp_linear_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                        const int* __restrict__ x0, const int* __restrict__ x1, const float* __restrict__ a1,
                                                        const int* __restrict__ y0, const int* __restrict__ y1, const float* __restrict__ b1,
                                                        float* __restrict__ out, int h, int w) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (dx >= w || dy >= h) return;
    const unsigned char* img = src + (size_t)b * H * W * C;
    const float wa1 = a1[dx], wa0 = __fsub_rn(1.0f, wa1), wb1 = b1[dy], wb0 = __fsub_rn(1.0f, wb1);
    const unsigned char* r0 = img + (size_t)y0[dy] * W * C;
    const unsigned char* r1 = img + (size_t)y1[dy] * W * C;
    const size_t c0 = (size_t)x0[dx] * C, c1 = (size_t)x1[dx] * C;
    // horizontal pass of the two source rows (multiply, multiply, add), then the vertical weights
    const float h0 = __fadd_rn(__fmul_rn(pp_pixel(r0 + c0, C), wa0), __fmul_rn(pp_pixel(r0 + c1, C), wa1));
    const float h1 = __fadd_rn(__fmul_rn(pp_pixel(r1 + c0, C), wa0), __fmul_rn(pp_pixel(r1 + c1, C), wa1));
    const float v = __fadd_rn(__fmul_rn(h0, wb0), __fmul_rn(h1, wb1));
    out[((size_t)b * h + dy) * w + dx] = __fdiv_rn(v, 255.0f);
}

extern "C" int imcui_hip_preprocess_linear_f32(imcui_hip_t* h, const unsigned char* src, int B, int H, int W, int C, const int* x0,
                                               const int* x1, const float* a1, const int* y0, const int* y1, const float* b1, float* out,
                                               int oh, int ow, void* stream) {
    if (!h || !src || !out || !x0 || !x1 || !a1 || !y0 || !y1 || !b1 || B < 0 || H <= 0 || W <= 0 || (C != 1 && C != 3) || oh <= 0 || ow <= 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "preprocess_linear: bad argument");
    if (B == 0) return IMCUI_OK;
    hipLaunchKernelGGL(pp_linear_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), B), dim3(256), 0, (hipStream_t)stream, src, H, W, C, x0, x1, a1, y0, y1,
                       b1, out, oh, ow);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ dfactor resize: torchvision F.resize(antialias=True)
// extract_features.py:142-148 / match_dense.py:182 round the float image down to a multiple of `dfactor` with
// `F.resize(image, size, antialias=True)` = ATen's anti-aliased bilinear kernel (_upsample_bilinear2d_aa): separable, the
// width first, every output = src[0] * w[0] followed by fused multiply-adds in tap order, float32 weights from
// _compute_indices_weights_aa (triangle filter, support = max(scale, 1), normalised).  The host table below and the kernel
// copy that arithmetic; oracle/preprocess.py: aa_resize_f32 restates it and is pinned bit for bit to torch itself.
// Returns the largest tap count; with w == nullptr only counts.
extern "C" int imcui_hip_aa_table(int in_size, int out_size, int* first, int* count, float* w, int kmax) {
    if (in_size <= 0 || out_size <= 0) return IMCUI_ERR_ARG;
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? 1.0f * scale : 1.0f;
    const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    int most = 0;
    for (int i = 0; i < out_size; ++i) {
        const float center = scale * ((float)i + 0.5f);
        int xmin = (int)((center - support) + 0.5f);
        xmin = xmin > 0 ? xmin : 0;
        int xmax = (int)((center + support) + 0.5f);
        xmax = xmax < in_size ? xmax : in_size;
        const int n = xmax - xmin;
        most = n > most ? n : most;
        if (!w) continue;
        if (n > kmax || !first || !count) return IMCUI_ERR_ARG;
        first[i] = xmin;
        count[i] = n;
        float tot = 0.0f;
        for (int j = 0; j < n; ++j) {
            float x = (((float)(j + xmin) - center) + 0.5f) * invscale;
            x = fabsf(x);
            const float t = x < 1.0f ? 1.0f - x : 0.0f;
            w[(size_t)i * kmax + j] = t;
            tot = tot + t;
        }
        for (int j = 0; j < n; ++j) w[(size_t)i * kmax + j] = tot != 0.0f ? w[(size_t)i * kmax + j] / tot : w[(size_t)i * kmax + j];
        for (int j = n; j < kmax; ++j) w[(size_t)i * kmax + j] = 0.0f;
    }
    return most;
}

__global__ __launch_bounds__(256) void pp_aa_kernel(const float* __restrict__ src, int H, int W, const int* __restrict__ xf,
                                                    const int* __restrict__ xn, const float* __restrict__ xw, int kx,
                                                    const int* __restrict__ yf, const int* __restrict__ yn, const float* __restrict__ yw,
                                                    int ky, float* __restrict__ out, int h, int w) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int pl = blockIdx.z;
    if (dx >= w || dy >= h) return;
    const float* img = src + (size_t)pl * H * W;
    const int x0 = xf[dx], nx = xn[dx], y0 = yf[dy], ny = yn[dy];
    const float* wx = xw + (size_t)dx * kx;
    const float* wy = yw + (size_t)dy * ky;
    float v = 0.0f;
    for (int j = 0; j < ny; ++j) {
        const float* row = img + (size_t)(y0 + j) * W + x0;
        float t = __fmul_rn(row[0], wx[0]);  // the width pass of this source row (rounded to float32, as ATen's intermediate tensor)
        for (int k = 1; k < nx; ++k) t = __fmaf_rn(row[k], wx[k], t);
        v = (j == 0) ? __fmul_rn(t, wy[0]) : __fmaf_rn(t, wy[j], v);
    }
    out[((size_t)pl * h + dy) * w + dx] = v;
}

extern "C" int imcui_hip_resize_aa_f32(imcui_hip_t* h, const float* src, int planes, int H, int W, const int* xfirst, const int* xcount,
                                       const float* xweight, int kx, const int* yfirst, const int* ycount, const float* yweight, int ky,
                                       float* out, int oh, int ow, void* stream) {
    if (!h || !src || !out || !xfirst || !xcount || !xweight || !yfirst || !ycount || !yweight || planes < 0 || H <= 0 || W <= 0 || oh <= 0 ||
        ow <= 0 || kx <= 0 || ky <= 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "resize_aa: bad argument");
    if (planes == 0) return IMCUI_OK;
    hipLaunchKernelGGL(pp_aa_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), planes), dim3(256), 0, (hipStream_t)stream, src, H, W, xfirst, xcount,
                       xweight, kx, yfirst, ycount, yweight, ky, out, oh, ow);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
