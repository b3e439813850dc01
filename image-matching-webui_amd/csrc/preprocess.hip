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
