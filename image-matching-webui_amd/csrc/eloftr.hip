// EfficientLoFTR forward on MI355X (upstream zju3dv/EfficientLoFTR `LoFTR.forward`, called by
// imcui/hloc/matchers/eloftr.py:79 with the 'full' model type in fp32).
//
// Data path (NHWC activations, H and W multiples of 32; the two images of a pair may differ in size):
//   backbone   RepVGG 1-64-64-128-256 at 1/2, 1/2, 1/4, 1/8: every block is ONE 3x3 convolution + ReLU after the
//              host-side re-parameterisation (3x3 + 1x1 + identity branches and their BatchNorms folded, which is
//              what the reference's `reparameter()` does at load time, eloftr.py:61) -> the implicit-im2col GEMM of
//              gemm.hip; the 1 -> 64 first block is a direct kernel.
//   coarse     4 x (self, cross) aggregated attention: queries = depth-wise 4x4/4 conv, keys / values = 4x4/4
//              max-pool, LayerNorm, q/k/v projections, 2-D RoPE (self only), soft-max attention with 8 heads of
//              32 on the (H/32)(W/32) grid, output projection, bilinear x4 back to the 1/8 grid, then
//              x += LayerNorm(fc2(leaky_relu(fc1([x | up])))).  Image 1's cross attention reads the UPDATED image 0.
//   matching   dual soft-max at temperature 0.1 + threshold + border 2 + mutual nearest neighbour: the kernels of
//              the LoFTR path (loftr_kernels.h), match list in (batch, i) order.
//   fine       feature fusion 1/8 -> 1/4 -> 1/2 (1x1 and 3x3 convolutions as GEMMs, BatchNorm folded, LeakyReLU,
//              bilinear x2 with align_corners=False); the last x2 up-sampling to full resolution is evaluated only
//              on the 8x8 / 10x10 windows of the matches, inside the two-stage fine matching kernel.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv.h"
#include "eloftr_kernels.h"
#include "ffn.h"
#include "gemm.h"
#include "imcui_hip.h"
#include "loftr_kernels.h"
#include "simred.h"

// ------------------------------------------------------------------ layer table
enum {
    EL_BB0 = 0,                 // 20 backbone convolutions after the first block
    EL_TR0 = 20,                // 8 attention blocks (layer * 2 + {self, cross}) x {q, k, v, o, fc1, fc2}
    EL_OUT = EL_TR0 + 48,       // refinement_layer.out_conv (1x1, 256 -> 256, the 1/16 of the coarse features folded in)
    EL_F0_C1, EL_F0_C2, EL_F0_C3,  // out_conv_layers.0: 1x1 128 -> 256 | 3x3 256 -> 256 (+BN, LeakyReLU) | 3x3 256 -> 128
    EL_F1_C1, EL_F1_C2, EL_F1_C3,  // out_conv_layers.1: 1x1 64 -> 128  | 3x3 128 -> 128 (+BN, LeakyReLU) | 3x3 128 -> 64
    EL_NLAYERS
};
static const int EL_BB_CIN[20] = {64, 64, 64, 128, 128, 128, 128, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256};
static const int EL_BB_COUT[20] = {64, 64, 128, 128, 128, 128, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256};
static const int EL_BB_STRIDE[20] = {1, 1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

static void el_shape(int i, int* N, int* K) {
    if (i < EL_TR0) {
        *N = EL_BB_COUT[i];
        *K = 9 * EL_BB_CIN[i];
        return;
    }
    if (i < EL_OUT) {
        const int j = (i - EL_TR0) % 6;
        *N = (j == 4) ? 512 : 256;
        *K = (j >= 4) ? 512 : 256;
        return;
    }
    static const int tab[7][2] = {{256, 256}, {256, 128}, {256, 9 * 256}, {128, 9 * 256}, {128, 64}, {128, 9 * 128}, {64, 9 * 128}};
    *N = tab[i - EL_OUT][0];
    *K = tab[i - EL_OUT][1];
}
// norm vectors of block t = layer * 2 + kind: 4t + {aggregation.norm.w, .b, mlp.layer_norm.w, .b}
#define EL_NNORMS 32

struct ElLayout {
    size_t conv0_w, conv0_b;  // [9][64], [64]
    size_t w[EL_NLAYERS], b[EL_NLAYERS], wh[EL_NLAYERS], wl[EL_NLAYERS], ws[EL_NLAYERS];
    size_t c3h[EL_NLAYERS], c3l[EL_NLAYERS], c3s[EL_NLAYERS];  // 3x3 layers: planes in the layout of conv3x3_split_kernel (0 = none)
    size_t dw[8];             // depth-wise query aggregation [256][16]
    size_t wph[8], wpl[8], wps[8];  // mlp.fc2 planes with the K axis in the fused FFN kernel's order (ffn_permute_k)
    size_t norm[EL_NNORMS];
    size_t inv_freq;          // [64] rotary frequencies
    size_t total;
};
static ElLayout el_layout() {
    ElLayout l;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t r = off;
        off += align_up(n, 64);
        return r;
    };
    l.conv0_w = take(9 * 64);
    l.conv0_b = take(64);
    for (int i = 0; i < EL_NLAYERS; ++i) {
        int N, K;
        el_shape(i, &N, &K);
        const size_t npad = (size_t)((N + 31) / 32 * 32) * K;
        l.w[i] = take((size_t)N * K);
        l.b[i] = take(N);
        l.wh[i] = take(npad / 2);
        l.wl[i] = take(npad / 2);
        l.ws[i] = take(64);
        l.c3h[i] = l.c3l[i] = l.c3s[i] = 0;
        if (K % 288 == 0 && N % 64 == 0) {  // a 3x3 convolution (K = 9 Cin, Cin a multiple of 32): also packed for the patch-staging kernel
            l.c3h[i] = take((size_t)N * K / 2);
            l.c3l[i] = take((size_t)N * K / 2);
            l.c3s[i] = take(64);
        }
    }
    for (int i = 0; i < 8; ++i) l.dw[i] = take(256 * 16);
    for (int i = 0; i < 8; ++i) {
        l.wph[i] = take(256 * 512 / 2);
        l.wpl[i] = take(256 * 512 / 2);
        l.wps[i] = take(64);
    }
    for (int i = 0; i < EL_NNORMS; ++i) l.norm[i] = take(256);
    l.inv_freq = take(64);
    l.total = off;
    return l;
}

extern "C" size_t imcui_hip_eloftr_packed_floats(void) { return el_layout().total; }
extern "C" int imcui_hip_eloftr_num_layers(void) { return EL_NLAYERS; }
extern "C" int imcui_hip_eloftr_layer_shape(int i, int* N, int* K) {
    if (i < 0 || i >= EL_NLAYERS || !N || !K) return IMCUI_ERR_ARG;
    el_shape(i, N, K);
    return IMCUI_OK;
}

extern "C" int imcui_hip_eloftr_pack_weights(const float* conv0_w, const float* conv0_b, const float* const* w, const float* const* b,
                                             const float* const* dw, const float* const* norms, const float* inv_freq, float* packed) {
    if (!conv0_w || !conv0_b || !w || !b || !dw || !norms || !inv_freq || !packed) return IMCUI_ERR_ARG;
    const ElLayout l = el_layout();
    memset(packed, 0, l.total * sizeof(float));
    memcpy(packed + l.conv0_w, conv0_w, 9 * 64 * sizeof(float));
    memcpy(packed + l.conv0_b, conv0_b, 64 * sizeof(float));
    for (int i = 0; i < EL_NLAYERS; ++i) {
        int N, K;
        el_shape(i, &N, &K);
        if (!w[i]) return IMCUI_ERR_ARG;
        memcpy(packed + l.w[i], w[i], (size_t)N * K * sizeof(float));
        if (b[i]) memcpy(packed + l.b[i], b[i], (size_t)N * sizeof(float));
        packed[l.ws[i]] = split_weights_frag_host(w[i], N, K, reinterpret_cast<unsigned short*>(packed + l.wh[i]),
                                                  reinterpret_cast<unsigned short*>(packed + l.wl[i]));
        if (l.c3s[i])
            packed[l.c3s[i]] = pack_conv3x3_split_from_gemm(w[i], N, K / 9, reinterpret_cast<unsigned short*>(packed + l.c3h[i]),
                                                            reinterpret_cast<unsigned short*>(packed + l.c3l[i]));
    }
    float* perm = (float*)malloc((size_t)256 * 512 * sizeof(float));
    if (!perm) return IMCUI_ERR_ARG;
    for (int i = 0; i < 8; ++i) {
        if (!dw[i]) {
            free(perm);
            return IMCUI_ERR_ARG;
        }
        memcpy(packed + l.dw[i], dw[i], 256 * 16 * sizeof(float));
        ffn_permute_k(w[EL_TR0 + i * 6 + 5], 256, 512, perm);
        packed[l.wps[i]] = split_weights_frag_host(perm, 256, 512, reinterpret_cast<unsigned short*>(packed + l.wph[i]),
                                                   reinterpret_cast<unsigned short*>(packed + l.wpl[i]));
    }
    free(perm);
    for (int i = 0; i < EL_NNORMS; ++i) {
        if (!norms[i]) return IMCUI_ERR_ARG;
        memcpy(packed + l.norm[i], norms[i], 256 * sizeof(float));
    }
    memcpy(packed + l.inv_freq, inv_freq, 64 * sizeof(float));
    return IMCUI_OK;
}

// ------------------------------------------------------------------ workspace
struct ElWs {
    float *s0, *s1a, *x1, *s2a, *s2b, *x2, *s3a, *s3b, *fc;
    float *qa, *ka, *q, *k, *v, *att, *o, *up, *hb, *ob;
    float *rmax, *rsum, *cmax, *csum, *best, *cbest, *mconf;
    SimDsWs ds;  // the matrix-free dual-softmax (simred.hip): packed coarse features, partials, tile flags
    float *f8, *u4, *a4, *b4, *r4, *u2, *a2, *b2, *r2, *win;
    int *bestj, *flag, *mb, *mi, *mj, *nmatch;
    size_t total;
    bool ok;
};
static ElWs el_carve(void* ws, size_t bytes, int B, int H0, int W0, int H1, int W1, int dbg_windows) {
    WsAlloc a(ws, bytes);
    ElWs w;
    // per-image buffers hold the B images of side 0 followed by the B images of side 1 (sizes may differ per side)
    const size_t p2 = (size_t)B * ((size_t)(H0 / 2) * (W0 / 2) + (size_t)(H1 / 2) * (W1 / 2));
    const size_t p4 = (size_t)B * ((size_t)(H0 / 4) * (W0 / 4) + (size_t)(H1 / 4) * (W1 / 4));
    const size_t L0 = (size_t)(H0 / 8) * (W0 / 8), L1 = (size_t)(H1 / 8) * (W1 / 8);
    const size_t p8 = (size_t)B * (L0 + L1);
    const size_t pa = (size_t)B * ((size_t)(H0 / 32) * (W0 / 32) + (size_t)(H1 / 32) * (W1 / 32));
    const size_t cap = (size_t)B * L0, cap1 = (size_t)B * L1;
    w.s0 = a.get<float>(p2 * 64);
    w.s1a = a.get<float>(p2 * 64);
    w.x1 = a.get<float>(p2 * 64);
    w.s2a = a.get<float>(p4 * 128);
    w.s2b = a.get<float>(p4 * 128);
    w.x2 = a.get<float>(p4 * 128);
    w.s3a = a.get<float>(p8 * 256);
    w.s3b = a.get<float>(p8 * 256);
    w.fc = a.get<float>(p8 * 256);
    w.qa = a.get<float>(pa * 256);
    w.ka = a.get<float>(pa * 256);
    w.q = a.get<float>(pa * 256);
    w.k = a.get<float>(pa * 256);
    w.v = a.get<float>(pa * 256);
    w.att = a.get<float>(pa * 256);
    w.o = a.get<float>(pa * 256);
    w.up = a.get<float>(p8 * 256);
    w.hb = a.get<float>(p8 * 512);
    w.ob = a.get<float>(p8 * 256);
    simred_ds_carve(a, B, (int)L0, (int)L1, 256, w.ds);
    w.rmax = a.get<float>(cap);
    w.rsum = a.get<float>(cap);
    w.cmax = a.get<float>(cap1);
    w.csum = a.get<float>(cap1);
    w.best = a.get<float>(cap);
    w.cbest = a.get<float>(cap1);
    w.mconf = a.get<float>(cap);
    w.f8 = a.get<float>(p8 * 256);
    w.u4 = a.get<float>(p4 * 256);
    w.a4 = a.get<float>(p4 * 256);
    w.b4 = a.get<float>(p4 * 256);
    w.r4 = a.get<float>(p4 * 128);
    w.u2 = a.get<float>(p2 * 128);
    w.a2 = a.get<float>(p2 * 128);
    w.b2 = a.get<float>(p2 * 128);
    w.r2 = a.get<float>(p2 * 64);
    w.win = dbg_windows ? a.get<float>(cap * 164 * 64) : nullptr;
    w.bestj = a.get<int>(cap);
    w.flag = a.get<int>(cap);
    w.mb = a.get<int>(cap);
    w.mi = a.get<int>(cap);
    w.mj = a.get<int>(cap);
    w.nmatch = a.get<int>(4);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}
// debug_windows != 0 reserves (and the forward fills) the unfolded fine windows [B*L0][64 + 100][64] for the parity tests
extern "C" size_t imcui_hip_eloftr_workspace_bytes(int B, int H0, int W0, int H1, int W1, int debug_windows) {
    return el_carve(nullptr, 0, B, H0, W0, H1, W1, debug_windows).total;
}

// byte offsets of workspace buffers, for the parity tests (every per-image buffer: the B images of side 0, then side 1):
// 0 = backbone 1/2 features [.,H/2,W/2,64], 1 = 1/4 features [.,128], 2 = coarse features after the transformer [., L, 256],
// 3 = (the similarity matrix until round 4: it no longer exists), 4 = fused 1/2 map R [.,H/2,W/2,64], 5 = fine windows (debug_windows only)
extern "C" size_t imcui_hip_eloftr_debug_offset(int which, int B, int H0, int W0, int H1, int W1) {
    ElWs w = el_carve((void*)256, (size_t)-1 >> 1, B, H0, W0, H1, W1, 1);
    const char* base = (const char*)256;
    switch (which) {
        case 0: return (const char*)w.x1 - base;
        case 1: return (const char*)w.x2 - base;
        case 2: return (const char*)w.fc - base;
        case 3: return 0;
        case 4: return (const char*)w.r2 - base;
        case 5: return (const char*)w.win - base;
        default: return 0;
    }
}

// ------------------------------------------------------------------ forward
extern "C" int imcui_hip_eloftr_forward(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B, int H0,
                                        int W0, int H1, int W1, double match_threshold, float* keypoints0, float* keypoints1,
                                        float* confidence, int* batch_indexes, int* num_matches, int debug_windows, void* ws,
                                        size_t ws_bytes, void* stream_) {
    return imcui_hip_eloftr_forward_ex(h, packed, image0, image1, B, H0, W0, H1, W1, match_threshold, 0, keypoints0, keypoints1, confidence,
                                       batch_indexes, num_matches, debug_windows, ws, ws_bytes, stream_);
}

// arith 1: the reference wrapper's `precision: "fp16"` / `"mp"` (eloftr.py:32-33,43-47,63-64: `self.net.half()` / autocast) as ONE f16
// product per element pair, f32 accumulate, in the backbone and fine-fusion convolutions (the 0.36 of 0.42 TF of a pair that is
// convolution work); the transformer blocks, similarity, matching and fine stages keep the 3 x f16 split arithmetic (their
// soft-max / arg-max decisions are what the match list depends on).  arith 0 = the parity arithmetic.
extern "C" int imcui_hip_eloftr_forward_ex(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B, int H0,
                                           int W0, int H1, int W1, double match_threshold, int arith, float* keypoints0, float* keypoints1,
                                           float* confidence, int* batch_indexes, int* num_matches, int debug_windows, void* ws,
                                           size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (arith != 0 && arith != 1) return imcui_set_err(h, IMCUI_ERR_ARG, "eloftr: arith=%d (0 = 3 x f16 split products, 1 = one f16 product in the convolutions)", arith);
    if (arith == 1 && h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "eloftr: the single-product arithmetic needs precision 1");
    const int single = arith;
    if (B <= 0) return IMCUI_OK;
    if (H0 % 32 || W0 % 32 || H0 < 64 || W0 < 64 || H1 % 32 || W1 % 32 || H1 < 64 || W1 < 64)
        return imcui_set_err(h, IMCUI_ERR_ARG, "eloftr: image sizes %dx%d / %dx%d must be multiples of 32 (>= 64): the 1/8 grid is aggregated 4x4", W0, H0, W1, H1);
    if (!packed || !image0 || !image1 || !keypoints0 || !keypoints1 || !confidence || !batch_indexes || !num_matches)
        return imcui_set_err(h, IMCUI_ERR_ARG, "eloftr: null argument");
    ElWs w = el_carve(ws, ws_bytes, B, H0, W0, H1, W1, debug_windows);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "eloftr: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const ElLayout l = el_layout();
    const float* P = packed;
    const bool split = h->precision == 1;
    // per side s: image size and its 1/8 grid.  Upstream runs the backbone on the concatenated batch when both images have
    // one size and image by image otherwise; convolutions do not mix images either way.
    const int Hs[2] = {H0, H1}, Ws[2] = {W0, W1};
    const int hcs[2] = {H0 / 8, H1 / 8}, wcs[2] = {W0 / 8, W1 / 8};
    const int Ls[2] = {hcs[0] * wcs[0], hcs[1] * wcs[1]};
    const int ahs[2] = {hcs[0] / 4, hcs[1] / 4}, aws[2] = {wcs[0] / 4, wcs[1] / 4};
    const int Las[2] = {ahs[0] * aws[0], ahs[1] * aws[1]};
    const bool same = (H0 == H1 && W0 == W1);
    const int L = Ls[0], S = Ls[1];
    const int cap = B * L;
    const dim3 blk(256);
    int rc;
#define ELRUN(x)                       \
    do {                               \
        rc = (x);                      \
        if (rc != IMCUI_OK) return rc; \
    } while (0)

    auto wts = [&](GemmP& g, int li) {
        int N, K;
        el_shape(li, &N, &K);
        g.N = N;
        g.K = K;
        g.ldw = K;
        g.W = P + l.w[li];
        g.bias = P + l.b[li];
        if (split) {
            g.Wh = reinterpret_cast<const unsigned short*>(P + l.wh[li]);
            g.Wl = reinterpret_cast<const unsigned short*>(P + l.wl[li]);
            g.wscale = P + l.ws[li];
        }
    };
    auto npx = [&](int s, int div) { return (size_t)(Hs[s] / div) * (Ws[s] / div); };
    // convolution as a GEMM over the NHWC maps at input resolution 1/div: `in` / `out` / `resid` hold side 0 then side 1; one
    // launch over the 2B images when both sides have one size, one launch per side otherwise
    static const bool conv_gemm_only = getenv("IMCUI_CONV_GEMM_ONLY") != nullptr;  // A/B switch: every convolution on the implicit GEMM
    // rup: `resid` is the map at HALF the output resolution whose bilinear x2 up-sampling is the residual (evaluated in the GEMM
    // epilogue; 1x1 stride-1 layers of the split mode)
    auto conv = [&](int li, const float* in, float* out, int div, int cin, int ks, int stride, const float* resid, int act, bool rup = false) -> int {
        for (int s = 0; s < (same ? 1 : 2); ++s) {
            if (split && !conv_gemm_only && ks == 3 && stride == 1 && l.c3s[li] != 0) {
                // 3x3 stride 1: the patch-staging kernel (conv.hip) reads every input pixel once per 64 output channels; the
                // implicit GEMM re-reads it once per tap and misses L2 at these map sizes (PMC: 5-10 x the input bytes fetched)
                int N, K;
                el_shape(li, &N, &K);
                const size_t off = s ? (size_t)B * npx(0, div) : 0;
                const int r = conv3x3_split_launch(h, in + off * cin, reinterpret_cast<const unsigned short*>(P + l.c3h[li]),
                                                   reinterpret_cast<const unsigned short*>(P + l.c3l[li]), P + l.c3s[li], P + l.b[li], out + off * N,
                                                   same ? 2 * B : B, Hs[s] / div, Ws[s] / div, cin, N, act, 0, stream, resid ? resid + off * N : nullptr, 0, 0,
                                                   single);
                if (r != IMCUI_OK) return r;
                continue;
            }
            GemmP g;
            wts(g, li);
            g.epi = EPI_CONV;
            const int pad = ks / 2;
            const int hin = Hs[s] / div, win = Ws[s] / div;
            const int hout = (hin + 2 * pad - ks) / stride + 1, wout = (win + 2 * pad - ks) / stride + 1;
            const size_t ioff = s ? (size_t)B * npx(0, div) * cin : 0;
            const size_t ooff = s ? (size_t)B * npx(0, div * stride) * g.N : 0;
            g.A = in + ioff;
            g.conv_k = ks;
            g.conv_stride = stride;
            g.conv_pad = pad;
            g.conv_hin = hin;
            g.conv_win = win;
            g.conv_hout = hout;
            g.conv_wout = wout;
            g.conv_cin = cin;
            g.M = (same ? 2 * B : B) * hout * wout;
            g.C = out + ooff;
            g.ldc = g.N;
            g.resid = resid ? resid + ooff : nullptr;
            g.ldr = g.N;
            if (rup) {
                g.resid = resid + (s ? (size_t)B * npx(0, 2 * div) * g.N : 0);
                g.rup_h = Hs[s] / (2 * div);
                g.rup_w = Ws[s] / (2 * div);
                g.rup_align = 0;
            }
            g.act = act;
            g.single = (single && g.Wh != nullptr && g.rup_h == 0) ? 1 : 0;
            const int r = gemm_launch(h, g, stream);
            if (r != IMCUI_OK) return r;
        }
        return IMCUI_OK;
    };
    // ---- backbone
    {
        for (int s = 0; s < 2; ++s) {
            const long npix = (long)B * (Hs[s] / 2) * (Ws[s] / 2);
            const long blocks = min((npix + 15) / 16, (long)256 * 32);
            hipLaunchKernelGGL(el_conv0_kernel, dim3((unsigned)blocks), blk, 0, stream, s ? image1 : image0, P + l.conv0_w, P + l.conv0_b,
                               w.s0 + (s ? (size_t)B * npx(0, 2) * 64 : 0), Hs[s], Ws[s], Hs[s] / 2, Ws[s] / 2, npix);
        }
        IMCUI_CHECK_LAUNCH(h);
        // stage 1 (1/2, 64): 2 blocks; stage 2 (1/4, 128): 4; stage 3 (1/8, 256): 14
        ELRUN(conv(EL_BB0 + 0, w.s0, w.s1a, 2, 64, 3, 1, nullptr, 1));
        ELRUN(conv(EL_BB0 + 1, w.s1a, w.x1, 2, 64, 3, 1, nullptr, 1));
        ELRUN(conv(EL_BB0 + 2, w.x1, w.s2a, 2, 64, 3, 2, nullptr, 1));
        ELRUN(conv(EL_BB0 + 3, w.s2a, w.s2b, 4, 128, 3, 1, nullptr, 1));
        ELRUN(conv(EL_BB0 + 4, w.s2b, w.s2a, 4, 128, 3, 1, nullptr, 1));
        ELRUN(conv(EL_BB0 + 5, w.s2a, w.x2, 4, 128, 3, 1, nullptr, 1));
        ELRUN(conv(EL_BB0 + 6, w.x2, w.s3a, 4, 128, 3, 2, nullptr, 1));
        const float* src = w.s3a;
        for (int i = 7; i < 20; ++i) {
            float* dst = (i == 19) ? w.fc : (src == w.s3a ? w.s3b : w.s3a);
            ELRUN(conv(EL_BB0 + i, src, dst, 8, 256, 3, 1, nullptr, 1));
            src = dst;
        }
    }
    // ---- coarse transformer
    auto lin = [&](int li, const float* A, long lda, const float* A2, float* C, long rows, int act) -> int {
        GemmP g;
        wts(g, li);  // bias vectors of the transformer Linears are zero (bias=False upstream)
        g.epi = EPI_CONV;
        g.act = act;
        g.A = A;
        g.lda = lda;
        if (A2) {
            g.A2 = A2;
            g.lda2 = lda;
            g.K1 = (int)lda;
        }
        g.C = C;
        g.ldc = g.N;
        g.M = (int)rows;
        return gemm_launch(h, g, stream);
    };
    const size_t tok1 = (size_t)B * L;  // first token row of side 1
    static const bool mlp_unfused = getenv("IMCUI_LG_FFN_UNFUSED") != nullptr;  // A/B switch shared with lightglue.hip
    // block t: `ns` maps of side qs (first token row qt, grid hq x wq) attend to `ns` maps of side ss (first token row st)
    auto block = [&](int t, size_t qt, int qs, size_t st, int ss, int ns, bool rope) -> int {
        const int base = EL_TR0 + t * 6;
        float* x = w.fc + qt * 256;
        const float* src = w.fc + st * 256;
        const int Lq = Ls[qs], Laq = Las[qs], Lak = Las[ss];
        const float *gw = P + l.norm[4 * t + 0], *gb = P + l.norm[4 * t + 1];
        hipLaunchKernelGGL(el_aggregate_kernel, dim3(ns * Laq), blk, 0, stream, x, P + l.dw[t], gw, gb, hcs[qs], wcs[qs], 0, w.qa);
        hipLaunchKernelGGL(el_aggregate_kernel, dim3(ns * Lak), blk, 0, stream, src, P + l.dw[t], gw, gb, hcs[ss], wcs[ss], 1, w.ka);
        const long qrows = (long)ns * Laq, krows = (long)ns * Lak;
        int r;
        if ((r = lin(base + 0, w.qa, 256, nullptr, w.q, qrows, 0))) return r;
        if ((r = lin(base + 1, w.ka, 256, nullptr, w.k, krows, 0))) return r;
        if ((r = lin(base + 2, w.ka, 256, nullptr, w.v, krows, 0))) return r;
        if (rope) {  // self attention: queries and keys live on the same aggregated grid
            const long np = qrows * 128;
            hipLaunchKernelGGL(el_rope_kernel, dim3((unsigned)min((np + 255) / 256, (long)4096)), blk, 0, stream, w.q, w.k, P + l.inv_freq, ahs[qs],
                               aws[qs], np);
        }
        hipLaunchKernelGGL(el_attention_kernel, dim3(cdiv(Laq, 64), 8, ns), blk, 0, stream, w.q, w.k, w.v, Laq, Lak, 0.17677669529663687f, w.att);
        if ((r = lin(base + 3, w.att, 256, nullptr, w.o, qrows, 0))) return r;
        const long n4 = (long)ns * Lq * 64;
        hipLaunchKernelGGL(el_upsample_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)65536)), blk, 0, stream, w.o, w.up, ahs[qs], aws[qs], 256, 4, n4);
        const long trows = (long)ns * Lq;
        if (split && !mlp_unfused) {  // x += LayerNorm(fc2(leaky_relu(fc1([x | up])))) in one kernel (ffn.hip)
            FfnP f;
            f.act = 2;
            f.x = x;
            f.ctx = w.up;
            f.out = x;
            f.w1h = reinterpret_cast<const unsigned short*>(P + l.wh[base + 4]);
            f.w1l = reinterpret_cast<const unsigned short*>(P + l.wl[base + 4]);
            f.s1 = P + l.ws[base + 4];
            f.w2h = reinterpret_cast<const unsigned short*>(P + l.wph[t]);
            f.w2l = reinterpret_cast<const unsigned short*>(P + l.wpl[t]);
            f.s2 = P + l.wps[t];
            f.gamma = P + l.norm[4 * t + 2];
            f.beta = P + l.norm[4 * t + 3];
            f.M = (int)trows;
            return ffn_launch(h, f, stream);
        }
        if ((r = lin(base + 4, x, 256, w.up, w.hb, trows, 2))) return r;  // LeakyReLU(0.01)
        if ((r = lin(base + 5, w.hb, 512, nullptr, w.ob, trows, 0))) return r;
        hipLaunchKernelGGL(lf_layernorm_kernel<4>, dim3((unsigned)min((trows + 3) / 4, (long)65536)), blk, 0, stream, w.ob, P + l.norm[4 * t + 2],
                           P + l.norm[4 * t + 3], x, x, trows, 1, (const int*)nullptr, 0L);
        return IMCUI_OK;
    };
    for (int layer = 0; layer < 4; ++layer) {
        if (same) {
            ELRUN(block(layer * 2 + 0, 0, 0, 0, 0, 2 * B, true));  // self attention on all 2B maps
        } else {
            ELRUN(block(layer * 2 + 0, 0, 0, 0, 0, B, true));
            ELRUN(block(layer * 2 + 0, tok1, 1, tok1, 1, B, true));
        }
        ELRUN(block(layer * 2 + 1, 0, 0, tok1, 1, B, false));  // images 0 <- images 1
        ELRUN(block(layer * 2 + 1, tok1, 1, 0, 0, B, false));  // images 1 <- UPDATED images 0
    }
    IMCUI_CHECK_LAUNCH(h);

    // ---- dual soft-max coarse matching: sim = (f0 / 16) . (f1 / 16)^T / 0.1 is never stored (simred.hip: statistics pass, then the
    // confidences of the tiles that can exceed the threshold)
    ELRUN(simred_dual_softmax(h, w.ds, w.fc, 256, (long)L * 256, w.fc + tok1 * 256, 256, (long)S * 256, B, L, S, 256, 0.00390625f / 0.1f, (float)match_threshold,
                              w.rmax, w.rsum, w.cmax, w.csum, w.best, w.bestj, w.cbest, stream));
    hipLaunchKernelGGL(lf_decide_kernel, dim3(cdiv(cap, 256)), blk, 0, stream, w.best, w.bestj, w.cbest, L, S, wcs[0], hcs[0], wcs[1], hcs[1], 2,
                       (float)match_threshold, w.flag, (long)cap);
    hipLaunchKernelGGL(lf_compact_kernel, dim3(1), dim3(1024), 0, stream, w.flag, w.best, w.bestj, L, (long)cap, cap, w.mb, w.mi, w.mj,
                       w.mconf, w.nmatch);
    IMCUI_CHECK_LAUNCH(h);

    // ---- fine feature fusion
    auto upsample2 = [&](const float* in, float* out, int div, int C) {
        for (int s = 0; s < (same ? 1 : 2); ++s) {
            const int hh = Hs[s] / div, ww = Ws[s] / div;
            const long n4 = (long)(same ? 2 * B : B) * (2 * hh) * (2 * ww) * (C / 4);
            hipLaunchKernelGGL(el_upsample_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)65536)), blk, 0, stream,
                               in + (s ? (size_t)B * npx(0, div) * C : 0), out + (s ? (size_t)B * npx(0, div / 2) * C : 0), hh, ww, C, 2, n4);
        }
    };
    ELRUN(conv(EL_OUT, w.fc, w.f8, 8, 256, 1, 1, nullptr, 0));
    static const bool up_unfused = getenv("IMCUI_UPSAMPLE_UNFUSED") != nullptr;  // A/B switch: materialise the up-sampled maps
    const bool upf = split && !up_unfused;
    if (upf) {
        ELRUN(conv(EL_F0_C1, w.x2, w.a4, 4, 128, 1, 1, w.f8, 0, true));  // + bilinear x2 of out_conv's map, in the epilogue
    } else {
        upsample2(w.f8, w.u4, 8, 256);
        ELRUN(conv(EL_F0_C1, w.x2, w.a4, 4, 128, 1, 1, w.u4, 0));
    }
    ELRUN(conv(EL_F0_C2, w.a4, w.b4, 4, 256, 3, 1, nullptr, 2));
    ELRUN(conv(EL_F0_C3, w.b4, w.r4, 4, 256, 3, 1, nullptr, 0));
    if (upf) {
        ELRUN(conv(EL_F1_C1, w.x1, w.a2, 2, 64, 1, 1, w.r4, 0, true));
    } else {
        upsample2(w.r4, w.u2, 4, 128);
        ELRUN(conv(EL_F1_C1, w.x1, w.a2, 2, 64, 1, 1, w.u2, 0));
    }
    ELRUN(conv(EL_F1_C2, w.a2, w.b2, 2, 128, 3, 1, nullptr, 2));
    ELRUN(conv(EL_F1_C3, w.b2, w.r2, 2, 128, 3, 1, nullptr, 0));

    // ---- two-stage fine matching on the windows of the matches
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(el_fine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EL_FINE_SMEM);
    hipLaunchKernelGGL(el_fine_kernel, dim3(cap), blk, EL_FINE_SMEM, stream, w.r2, w.mb, w.mi, w.mj, w.nmatch, B, H0, W0, H1, W1, wcs[0], wcs[1],
                       (float)H0 / (float)hcs[0], 1.0f, keypoints0, keypoints1, w.win);
    hipMemcpyAsync(confidence, w.mconf, (size_t)cap * sizeof(float), hipMemcpyDeviceToDevice, stream);
    hipMemcpyAsync(batch_indexes, w.mb, (size_t)cap * sizeof(int), hipMemcpyDeviceToDevice, stream);
    hipMemcpyAsync(num_matches, w.nmatch, sizeof(int), hipMemcpyDeviceToDevice, stream);
    IMCUI_CHECK_LAUNCH(h);
#undef ELRUN
    return IMCUI_OK;
}
