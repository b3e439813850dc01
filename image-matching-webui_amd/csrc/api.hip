// Handle management, error reporting and the exported building-block entry points.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "attention.h"
#include "conv.h"
#include "ffn.h"
#include "gemm.h"
#include "imcui_hip.h"

int imcui_set_err(imcui_hip_s* h, int code, const char* fmt, ...) {
    if (h) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(h->err, sizeof h->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

// 400: round 4 -- imcui_hip_dust3r_forward[_sizes] take the size of the packed buffer and the buffer ends in a format trailer (a blob of
// an older layout is rejected instead of running with its LayerNorm affine parts dropped); imcui_hip_set_option / _get_option.
extern "C" int imcui_hip_version(void) { return 400; }

static const char* const OPT_NAMES[OPT_NCNT] = {"gemm_wreg", "wreg_pipe", "attn_variant", "attn_variant_self", "attn_variant_cross", "attn_mix_layers", "conv_tall", "conv_narrow", "simred", "ffn_tile", "wreg_tile", "attn_split", "loftr_fine_sparse"};
static int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_NCNT; ++i)
        if (strcmp(name, OPT_NAMES[i]) == 0) return i;
    return -1;
}
extern "C" int imcui_hip_set_option(imcui_hip_t* h, const char* name, int value) {
    const int i = opt_index(name);
    if (!h || i < 0) return imcui_set_err(h, IMCUI_ERR_ARG, "set_option: unknown option '%s'", name ? name : "(null)");
    h->opt[i] = value;
    return IMCUI_OK;
}
extern "C" int imcui_hip_get_option(imcui_hip_t* h, const char* name, int* value) {
    const int i = opt_index(name);
    if (!h || !value || i < 0) return imcui_set_err(h, IMCUI_ERR_ARG, "get_option: unknown option '%s'", name ? name : "(null)");
    *value = h->opt[i];
    return IMCUI_OK;
}

extern "C" int imcui_hip_create(int device, imcui_hip_t** out) {
    if (!out) return IMCUI_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return IMCUI_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return IMCUI_ERR_HIP;
    imcui_hip_s* h = (imcui_hip_s*)calloc(1, sizeof(imcui_hip_s));
    if (!h) return IMCUI_ERR_ARG;
    h->device = device;
    h->num_cu = prop.multiProcessorCount;
    h->err[0] = 0;
    h->precision = 1;
    {   // the A/B switches: environment -> handle, once (common.h)
        const char* e;
        h->opt[OPT_GEMM_WREG] = (e = getenv("IMCUI_GEMM_WREG")) ? atoi(e) : 2;
        h->opt[OPT_WREG_PIPE] = (e = getenv("IMCUI_WREG_PIPE")) ? atoi(e) : 1;
        h->opt[OPT_ATTN_VARIANT] = (e = getenv("IMCUI_ATTN_VARIANT")) ? atoi(e) : 8;  // 8 = the arithmetic of 0 with the pipelined K.Q^T schedule (attention.hip)
        h->opt[OPT_SIMRED] = (e = getenv("IMCUI_SIMRED")) ? atoi(e) : 1;
        h->opt[OPT_FFN_TILE] = (e = getenv("IMCUI_FFN_TILE")) ? atoi(e) : 0;
        h->opt[OPT_WREG_TILE] = (e = getenv("IMCUI_WREG_TILE")) ? atoi(e) : 0;
        h->opt[OPT_CONV_TALL] = (e = getenv("IMCUI_CONV_TALL")) ? atoi(e) : 1;
        h->opt[OPT_CONV_NARROW] = (e = getenv("IMCUI_CONV_NARROW")) ? atoi(e) : 0;
        h->opt[OPT_ATTN_SELF] = (e = getenv("IMCUI_ATTN_VARIANT_SELF")) ? atoi(e) : -1;
        // Round 5: LightGlue's CROSS blocks run the audited two-product P.V (variant 7) by default -- layer error <= 7.1e-6 and score error <= 4.7e-5 on the
        // three weight sets at N = M = 2048 (half the parity bar; profiles/r05_lab_attention_mix.txt), + 2.7 % end to end; the self blocks keep three
        // products (2.1e-5 / 9.6e-5 there: rejected).  -1 here restores three products everywhere.  The default is -2 = "7 while attn_variant is
        // its default (8)": an explicit attn_variant (the A/B and parity workflows that set that option alone) governs every block again.
        h->opt[OPT_ATTN_CROSS] = (e = getenv("IMCUI_ATTN_VARIANT_CROSS")) ? atoi(e) : -2;
        h->opt[OPT_ATTN_SPLIT] = (e = getenv("IMCUI_ATTN_SPLIT")) ? atoi(e) : 1;
        h->opt[OPT_LOFTR_FINE_SPARSE] = (e = getenv("IMCUI_LOFTR_FINE_SPARSE")) ? atoi(e) : 1;
        h->opt[OPT_ATTN_MIX_LAYERS] = (e = getenv("IMCUI_ATTN_MIX_LAYERS")) ? (int)strtol(e, nullptr, 0) : 0x1ff;
    }
    *out = h;
    const char* rc = getenv("IMCUI_HIP_CHECK_RANGE");
    if (rc && atoi(rc) != 0) (void)imcui_hip_set_range_check(h, 1);
    return IMCUI_OK;
}

// ---- opt-in range check of the split arithmetic ------------------------------------------------
// split2 (common.h) saturates the hi part at 65504 and loses the fp32-grade product above twice that; activations are not
// rescaled (weights are, at pack time).  Synthetic weights never come near; real checkpoints with outlier channels might.
// With the check on, every split-mode GEMM / convolution / fused-FFN launch is preceded by a scan of its f32 activation
// operand (a debugging aid: one extra read of the operand per launch); the status word is read with
// imcui_hip_get_range_status.  Off by default: no kernel changes, no cost.
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, long rows, int cols, long ld, const int* __restrict__ cnt,
                                                          int rows_per_seq, int* __restrict__ flag) {
    int bits = 0;
    const long n4 = rows * (long)(cols >> 2);
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long)gridDim.x * 256) {
        const long r = t / (cols >> 2);
        const int c = (int)(t - r * (cols >> 2)) * 4;
        if (cnt != nullptr && rows_per_seq > 0 && (int)(r % rows_per_seq) >= cnt[r / rows_per_seq]) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(fabsf(a[j]) <= 3.4028234e38f)) bits |= 2;  // NaN or Inf
            else if (fabsf(a[j]) > 65504.0f) bits |= 1;
        }
    }
    if (bits) atomicOr(flag, bits);
}
void imcui_range_check(imcui_hip_s* h, const float* x, long rows, int cols, long ld, const int* cnt, int rows_per_seq, hipStream_t s) {
    if (!h || !h->range_flag || !x || rows <= 0 || cols < 4 || (cols & 3) || (ld & 3)) return;
    const long n4 = rows * (long)(cols >> 2);
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(range_check_kernel, dim3(blocks), dim3(256), 0, s, x, rows, cols, ld, cnt, rows_per_seq, h->range_flag);
}
extern "C" int imcui_hip_set_range_check(imcui_hip_t* h, int enable) {
    if (!h) return IMCUI_ERR_ARG;
    if (enable && !h->range_flag) {
        if (hipMalloc((void**)&h->range_flag, sizeof(int)) != hipSuccess) return imcui_set_err(h, IMCUI_ERR_HIP, "range check: allocation failed");
        (void)hipMemset(h->range_flag, 0, sizeof(int));
    } else if (!enable && h->range_flag) {
        (void)hipFree(h->range_flag);
        h->range_flag = nullptr;
    }
    return IMCUI_OK;
}
extern "C" int imcui_hip_get_range_status(imcui_hip_t* h, int* status) {
    if (!h || !status) return IMCUI_ERR_ARG;
    *status = 0;
    if (!h->range_flag) return IMCUI_OK;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(status, h->range_flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return imcui_set_err(h, IMCUI_ERR_HIP, "range check: read failed");
    (void)hipMemset(h->range_flag, 0, sizeof(int));
    return IMCUI_OK;
}

extern "C" void imcui_hip_destroy(imcui_hip_t* h) {
    if (!h) return;
    for (int c = 0; c < PROF_NCLS; ++c) {
        for (int i = 0; i < 2 * h->prof_alloc[c]; ++i) (void)hipEventDestroy(h->prof_ev[c][i]);
        free(h->prof_ev[c]);
    }
    if (h->range_flag) (void)hipFree(h->range_flag);
    free(h);
}

// ---- optional HIP-event timing per kernel class ------------------------------------------
void imcui_prof_begin(imcui_hip_s* h, int cls, hipStream_t s) {
    if (!h->prof_on || h->prof_used[cls] >= PROF_MAX_EVENTS) return;
    if (!h->prof_ev[cls]) h->prof_ev[cls] = (hipEvent_t*)calloc(2 * PROF_MAX_EVENTS, sizeof(hipEvent_t));
    const int i = h->prof_used[cls];
    if (i >= h->prof_alloc[cls]) {
        (void)hipEventCreate(&h->prof_ev[cls][2 * i]);
        (void)hipEventCreate(&h->prof_ev[cls][2 * i + 1]);
        h->prof_alloc[cls] = i + 1;
    }
    (void)hipEventRecord(h->prof_ev[cls][2 * i], s);
}
void imcui_prof_end(imcui_hip_s* h, int cls, hipStream_t s) {
    if (!h->prof_on || h->prof_used[cls] >= PROF_MAX_EVENTS) return;
    (void)hipEventRecord(h->prof_ev[cls][2 * h->prof_used[cls] + 1], s);
    h->prof_used[cls] += 1;
}
extern "C" int imcui_hip_profile_enable(imcui_hip_t* h, int on) {
    if (!h) return IMCUI_ERR_ARG;
    h->prof_on = on;
    for (int c = 0; c < PROF_NCLS; ++c) h->prof_used[c] = 0;
    return IMCUI_OK;
}
extern "C" int imcui_hip_profile_read(imcui_hip_t* h, int cls, double* total_ms, int* count) {
    if (!h || cls < 0 || cls >= PROF_NCLS || !total_ms || !count) return IMCUI_ERR_ARG;
    double tot = 0.0;
    const int n = h->prof_used[cls];
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(h->prof_ev[cls][2 * i + 1]) != hipSuccess) return imcui_set_err(h, IMCUI_ERR_HIP, "profile: event sync failed");
        if (hipEventElapsedTime(&ms, h->prof_ev[cls][2 * i], h->prof_ev[cls][2 * i + 1]) != hipSuccess)
            return imcui_set_err(h, IMCUI_ERR_HIP, "profile: elapsed time failed");
        tot += ms;
    }
    *total_ms = tot;
    *count = n;
    h->prof_used[cls] = 0;
    return IMCUI_OK;
}

extern "C" int imcui_hip_set_precision(imcui_hip_t* h, int mode) {
    if (!h || (mode != 0 && mode != 1)) return IMCUI_ERR_ARG;
    h->precision = mode;
    return IMCUI_OK;
}
extern "C" int imcui_hip_get_precision(const imcui_hip_t* h) { return h ? h->precision : -1; }

extern "C" const char* imcui_hip_last_error(const imcui_hip_t* h) { return h ? h->err : "null handle"; }

extern "C" int imcui_hip_linear_f32(imcui_hip_t* h, const float* A, const float* W, const float* bias, float* C, int M, int N,
                                    int K, int relu, void* stream) {
    if (!h || !A || !W || !C) return imcui_set_err(h, IMCUI_ERR_ARG, "linear: null argument");
    GemmP g;
    g.epi = relu ? EPI_RELU : EPI_BIAS;
    g.A = A;
    g.lda = K;
    g.W = W;
    g.ldw = K;
    g.bias = bias;
    g.C = C;
    g.ldc = N;
    g.M = M;
    g.N = N;
    g.K = K;
    return gemm_launch(h, g, (hipStream_t)stream);
}

// ------------------------------------------------------------------ device-side preprocessing (step before the path)
// 4 pixels per thread: 12 bytes in (three dwords), 16 bytes out; HBM-bound byte work (7 B / pixel).
__global__ __launch_bounds__(256) void rgb_to_gray_kernel(const unsigned* __restrict__ rgb, float4* __restrict__ out, long nquad) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquad) return;
    const unsigned w0 = rgb[3 * q], w1 = rgb[3 * q + 1], w2 = rgb[3 * q + 2];
    // bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    const unsigned r0 = w0 & 255u, g0 = (w0 >> 8) & 255u, b0 = (w0 >> 16) & 255u, r1 = w0 >> 24;
    const unsigned g1 = w1 & 255u, b1 = (w1 >> 8) & 255u, r2 = (w1 >> 16) & 255u, g2 = w1 >> 24;
    const unsigned b2 = w2 & 255u, r3 = (w2 >> 8) & 255u, g3 = (w2 >> 16) & 255u, b3 = w2 >> 24;
    auto gray = [](unsigned r, unsigned g, unsigned b) { return (float)((9798u * r + 19235u * g + 3735u * b + 16384u) >> 15) / 255.0f; };
    out[q] = make_float4(gray(r0, g0, b0), gray(r1, g1, b1), gray(r2, g2, b2), gray(r3, g3, b3));
}

extern "C" int imcui_hip_rgb_to_gray_f32(imcui_hip_t* h, const unsigned char* rgb_hwc, float* out, int B, int H, int W, void* stream) {
    if (!h || !rgb_hwc || !out || B < 0 || H <= 0 || W <= 0) return imcui_set_err(h, IMCUI_ERR_ARG, "rgb_to_gray: bad argument");
    const long npix = (long)B * H * W;
    if (((long)H * W) % 4 != 0 || ((size_t)rgb_hwc & 3) != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "rgb_to_gray: H*W=%ld must be a multiple of 4 and the image 4-byte aligned", (long)H * W);
    if (npix == 0) return IMCUI_OK;
    const long nquad = npix / 4;
    hipLaunchKernelGGL(rgb_to_gray_kernel, dim3((unsigned)((nquad + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned*>(rgb_hwc), reinterpret_cast<float4*>(out), nquad);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

extern "C" float imcui_hip_linear_pack_split(const float* w, int N, int K, unsigned short* hi, unsigned short* lo) {
    if (!w || !hi || !lo || N <= 0 || K <= 0 || K % 16) return 0.0f;
    return split_weights_frag_host(w, N, K, hi, lo);
}

extern "C" int imcui_hip_linear_split_f32(imcui_hip_t* h, const float* A, const unsigned short* wh, const unsigned short* wl,
                                          const float* wscale, const float* bias, float* C, int M, int N, int K, int relu,
                                          void* stream) {
    if (!h || !A || !wh || !wl || !wscale || !C) return imcui_set_err(h, IMCUI_ERR_ARG, "linear_split: null argument");
    if (h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "linear_split: needs precision 1 (3 x f16 split)");
    GemmP g;
    g.epi = relu ? EPI_RELU : EPI_BIAS;
    g.A = A;
    g.lda = K;
    g.Wh = wh;
    g.Wl = wl;
    g.wscale = wscale;
    g.ldw = K;
    g.bias = bias;
    g.C = C;
    g.ldc = N;
    g.M = M;
    g.N = N;
    g.K = K;
    return gemm_launch(h, g, (hipStream_t)stream);
}

extern "C" int imcui_hip_qkv_split_f32(imcui_hip_t* h, const float* x, const unsigned short* wh, const unsigned short* wl,
                                       const float* wscale, const float* bias, const float* rope_cos, const float* rope_sin,
                                       const int* cnt, int nseq, int rows_per_seq, float alpha, int cross, float* q, float* k, float* v,
                                       void* stream) {
    if (!h || !x || !wh || !wl || !wscale || !cnt || !q || !v || (!cross && (!k || !rope_cos || !rope_sin)))
        return imcui_set_err(h, IMCUI_ERR_ARG, "qkv_split: null argument");
    if (h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "qkv_split: needs precision 1 (3 x f16 split)");
    if (nseq <= 0 || rows_per_seq <= 0 || rows_per_seq % 128 != 0)  // the head-major planes are laid out (and consumed by the attention kernel) in 128-row tiles
        return imcui_set_err(h, IMCUI_ERR_ARG, "qkv_split: rows_per_seq=%d must be a positive multiple of 128 (nseq=%d)", rows_per_seq, nseq);
    GemmP g;
    g.epi = cross ? EPI_CROSS : EPI_QKV;
    g.M = nseq * rows_per_seq;
    g.cnt = cnt;
    g.rows_per_seq = rows_per_seq;
    g.A = x;
    g.lda = 256;
    g.K = 256;
    g.N = cross ? 512 : 768;
    g.Wh = wh;
    g.Wl = wl;
    g.wscale = wscale;
    g.ldw = 256;
    g.bias = bias;
    g.v_transposed = 1;
    g.split_out = 1;
    g.plane_halves = (size_t)nseq * rows_per_seq * 256;
    g.Q = q;
    g.Kt = k;
    g.V = v;
    g.rope_cos = rope_cos;
    g.rope_sin = rope_sin;
    g.alpha = alpha;
    g.heads = 4;
    return gemm_launch(h, g, (hipStream_t)stream);
}

extern "C" float imcui_hip_ffn_pack_w2(const float* w2, unsigned short* hi, unsigned short* lo) {
    if (!w2 || !hi || !lo) return 0.0f;
    float* perm = (float*)malloc((size_t)256 * 512 * sizeof(float));
    if (!perm) return 0.0f;
    ffn_permute_k(w2, 256, 512, perm);
    const float sc = split_weights_frag_host(perm, 256, 512, hi, lo);
    free(perm);
    return sc;
}

extern "C" int imcui_hip_ffn_set_debug(imcui_hip_t* h, long long* stamps) {
    if (!h) return IMCUI_ERR_ARG;
    h->ffn_dbg = stamps;
    return IMCUI_OK;
}

extern "C" int imcui_hip_ffn_split_f32(imcui_hip_t* h, const float* x, const float* ctx, const unsigned short* w1h,
                                       const unsigned short* w1l, const float* s1, const float* b1, const float* gamma,
                                       const float* beta, const unsigned short* w2h, const unsigned short* w2l, const float* s2,
                                       const float* b2, float* out, int M, int act, void* stream) {
    if (!h) return IMCUI_ERR_ARG;
    FfnP p;
    p.x = x;
    p.ctx = ctx;
    p.out = out;
    p.w1h = w1h;
    p.w1l = w1l;
    p.w2h = w2h;
    p.w2l = w2l;
    p.s1 = s1;
    p.s2 = s2;
    p.b1 = b1;
    p.gamma = gamma;
    p.beta = beta;
    p.b2 = b2;
    p.M = M;
    p.act = act;
    p.dbg = h->ffn_dbg;
    return ffn_launch(h, p, (hipStream_t)stream);
}

extern "C" int imcui_hip_conv3x3_pack(const float* w_oihw, int Cout, int Cin, float* packed) {
    if (!w_oihw || !packed || Cin % 32 || Cout % 64) return IMCUI_ERR_ARG;
    pack_conv3x3(w_oihw, Cout, Cin, packed);
    return IMCUI_OK;
}

extern "C" int imcui_hip_conv3x3_f32(imcui_hip_t* h, const float* in, const float* wp, const float* bias, float* out, int B,
                                     int H, int W, int Cin, int Cout, int relu, int pool, void* stream) {
    if (!h || !in || !wp || !bias || !out) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3: null argument");
    return conv3x3_launch(h, in, wp, bias, out, B, H, W, Cin, Cout, relu, pool, (hipStream_t)stream);
}

extern "C" float imcui_hip_conv3x3_pack_split(const float* w_oihw, int Cout, int Cin, unsigned short* hi, unsigned short* lo) {
    if (!w_oihw || !hi || !lo || Cin % 32 || Cout % 64) return 0.0f;
    return pack_conv3x3_split(w_oihw, Cout, Cin, hi, lo);
}

extern "C" int imcui_hip_conv3x3_split_f32(imcui_hip_t* h, const float* in, const unsigned short* wh, const unsigned short* wl,
                                           const float* wscale, const float* bias, float* out, int B, int H, int W, int Cin,
                                           int Cout, int relu, int pool, void* stream) {
    if (!h || !in || !wh || !wl || !wscale || !bias || !out) return imcui_set_err(h, IMCUI_ERR_ARG, "conv3x3_split: null argument");
    return conv3x3_split_launch(h, in, wh, wl, wscale, bias, out, B, H, W, Cin, Cout, relu, pool, (hipStream_t)stream);
}

extern "C" int imcui_hip_conv_gemm_f32(imcui_hip_t* h, const float* in, const float* w, const float* bias, const float* resid,
                                       float* out, int B, int Hin, int Win, int Cin, int Cout, int ks, int stride, int act,
                                       void* stream) {
    if (!h || !in || !w || !out) return imcui_set_err(h, IMCUI_ERR_ARG, "conv_gemm: null argument");
    GemmP g;
    g.epi = EPI_CONV;
    const int pad = ks / 2;
    const int hout = (Hin + 2 * pad - ks) / stride + 1, wout = (Win + 2 * pad - ks) / stride + 1;
    g.A = in;
    g.conv_k = ks;
    g.conv_stride = stride;
    g.conv_pad = pad;
    g.conv_hin = Hin;
    g.conv_win = Win;
    g.conv_hout = hout;
    g.conv_wout = wout;
    g.conv_cin = Cin;
    g.W = w;
    g.K = ks * ks * Cin;
    g.ldw = g.K;
    g.bias = bias;
    g.N = Cout;
    g.M = B * hout * wout;
    g.C = out;
    g.ldc = Cout;
    g.resid = resid;
    g.ldr = Cout;
    g.act = act;
    return gemm_launch(h, g, (hipStream_t)stream);
}

extern "C" int imcui_hip_attention_f32(imcui_hip_t* h, const float* Q, const float* K, const float* V, float* O,
                                       const int* cnt, int S, int heads, int rows, int cross, int log2_domain, void* stream) {
    if (!h || !Q || !K || !V || !O || !cnt) return imcui_set_err(h, IMCUI_ERR_ARG, "attention: null argument");
    AttnP a;
    a.Q = Q;
    a.K = K;
    a.V = V;
    a.O = O;
    a.cnt = cnt;
    a.nseq = S;
    a.heads = heads;
    a.rows_per_seq = rows;
    a.cross = cross;
    a.log2_domain = log2_domain;
    return attention_launch(h, a, (hipStream_t)stream);
}
extern "C" size_t imcui_hip_attention_mx_scratch_bytes(int S, int heads, int rows) { return (S > 0 && heads > 0 && rows > 0) ? attn_v6_bytes(S, heads, rows) : 0; }
extern "C" int imcui_hip_attention_mx_f32(imcui_hip_t* h, const float* Q, const float* K, const float* V, float* O, const int* cnt, int S, int heads, int rows, int cross,
                                          void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !Q || !K || !V || !O || !cnt || !scratch) return imcui_set_err(h, IMCUI_ERR_ARG, "attention (variant 9): null argument");
    if (h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "attention variant 9 belongs to the split arithmetic (imcui_hip_set_precision(h, 1))");
    if (rows % 128 != 0 || (heads * S) % 8 != 0) return imcui_set_err(h, IMCUI_ERR_ARG, "attention (variant 9): rows %% 128 and (heads * S) %% 8 must be 0");
    if (scratch_bytes < attn_v6_bytes(S, heads, rows)) return imcui_set_err(h, IMCUI_ERR_WS, "attention (variant 9): scratch of %zu bytes needed", attn_v6_bytes(S, heads, rows));
    AttnP a;
    a.Q = Q;
    a.K = K;
    a.V = V;
    a.O = O;
    a.cnt = cnt;
    a.nseq = S;
    a.heads = heads;
    a.rows_per_seq = rows;
    a.cross = cross;
    a.log2_domain = 1;
    a.variant = 9;
    a.V6 = (unsigned char*)scratch;
    return attention_launch(h, a, (hipStream_t)stream);
}
