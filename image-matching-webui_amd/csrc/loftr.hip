// LoFTR forward on MI355X (kornia.feature.LoFTR semantics, called by
// imcui/hloc/matchers/loftr.py:54).
//
// The 3x3 stride-1 convolutions of the ResNet-FPN backbone run on the patch-staging conv kernel of conv.hip (BatchNorm folded
// into weights and bias on the host, residual add + ReLU / LeakyReLU in its store loop); the strided and 1x1 convolutions and
// every Linear of the two transformers are "linear layers" W[N][K] (+bias) on the MFMA GEMM of gemm.hip (implicit-im2col
// addressing on NHWC activations); the 196-channel stage is stored padded to 256 channels.  The coarse MLPs (mlp.0, ReLU, mlp.2,
// norm2, residual) are one launch of the fused FFN kernel (ffn.hip).  Linear attention, LayerNorm, the dual-softmax coarse
// matching and the fine 5x5-window stage are the small kernels of loftr_kernels.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv.h"
#include "ffn.h"
#include "gemm.h"
#include "imcui_hip.h"
#include "loftr_kernels.h"
#include "simred.h"

// ------------------------------------------------------------------ layer table
enum {
    LF_L1_0_C1 = 0, LF_L1_0_C2, LF_L1_1_C1, LF_L1_1_C2,
    LF_L2_0_C1, LF_L2_0_C2, LF_L2_0_DS, LF_L2_1_C1, LF_L2_1_C2,
    LF_L3_0_C1, LF_L3_0_C2, LF_L3_0_DS, LF_L3_1_C1, LF_L3_1_C2,
    LF_OUT3, LF_OUT2, LF_OUT2B_0, LF_OUT2B_3, LF_OUT1, LF_OUT1B_0, LF_OUT1B_3,
    LF_COARSE0,                       // 8 layers x {q, k, v, merge, mlp0, mlp2}
    LF_DOWN = LF_COARSE0 + 48, LF_MERGEF,
    LF_FINE0,                         // 2 layers x {q, k, v, merge, mlp0, mlp2}
    LF_NLAYERS = LF_FINE0 + 12
};
#define CP 256  // the 196-channel stage, padded

static void lf_shape(int i, int* N, int* K) {
    static const int tab[21][2] = {
        {128, 9 * 128}, {128, 9 * 128}, {128, 9 * 128}, {128, 9 * 128},
        {CP, 9 * 128},  {CP, 9 * CP},   {CP, 128},      {CP, 9 * CP},   {CP, 9 * CP},
        {256, 9 * CP},  {256, 9 * 256}, {256, CP},      {256, 9 * 256}, {256, 9 * 256},
        {256, 256},     {256, CP},      {256, 9 * 256}, {CP, 9 * 256},  {CP, 128},      {CP, 9 * CP}, {128, 9 * CP},
    };
    if (i < LF_COARSE0) {
        *N = tab[i][0];
        *K = tab[i][1];
        return;
    }
    if (i == LF_DOWN || i == LF_MERGEF) {
        *N = 128;
        *K = 256;
        return;
    }
    const bool fine = i >= LF_FINE0;
    const int d = fine ? 128 : 256;
    const int j = (i - (fine ? LF_FINE0 : LF_COARSE0)) % 6;
    *N = (j == 4) ? 2 * d : d;
    *K = (j >= 4) ? 2 * d : d;
}
// 3x3 layers whose INPUT is the 196-channel stage (stored padded to CP = 256 channels): the patch-staging conv kernel reads
// only the first 224 channels (a multiple of its 32-channel chunk): 12.5 % less work in those layers
static int lf_cin_used(int li) {
    static const bool off = getenv("IMCUI_LF_NO_UNPAD") != nullptr;  // A/B switch (read at pack time and at launch time alike)
    if (off) return 0;
    switch (li) {
        case LF_L2_0_C2: case LF_L2_1_C1: case LF_L2_1_C2: case LF_OUT1B_0: case LF_OUT1B_3: return 224;
        default: return 0;
    }
}
// 3x3 layers whose OUTPUT is the 196-channel stage (N = CP = 256): the fragment of channels 224..255 is all padding and is
// not multiplied (conv3x3_split_launch's cout_live)
static int lf_cout_live(int li) {
    static const bool off = getenv("IMCUI_LF_NO_UNPAD") != nullptr;
    if (off) return 0;
    switch (li) {
        case LF_L2_0_C2: case LF_L2_1_C1: case LF_L2_1_C2: case LF_OUT2B_3: case LF_OUT1B_0: return 224;
        default: return 0;
    }
}
// norm vectors: coarse layer l: 4l + {norm1.w, norm1.b, norm2.w, norm2.b} (256), then fine (128)
#define LF_NNORMS (8 * 4 + 2 * 4)
static int lf_norm_dim(int i) { return i < 32 ? 256 : 128; }

struct LfLayout {
    size_t conv1_w, conv1_b;  // [49][128], [128]
    size_t w[LF_NLAYERS], b[LF_NLAYERS], wh[LF_NLAYERS], wl[LF_NLAYERS], ws[LF_NLAYERS];
    size_t norm[LF_NNORMS];
    size_t c3h[LF_NLAYERS], c3l[LF_NLAYERS], c3s[LF_NLAYERS];  // 3x3 convolutions: planes in the layout of conv3x3_split_kernel (0 = none)
    size_t wph[8], wpl[8], wps[8];  // coarse mlp.2 planes with the K axis in the fused FFN kernel's order (ffn_permute_k)
    size_t total;
};
static LfLayout lf_layout() {
    LfLayout l;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t r = off;
        off += align_up(n, 64);
        return r;
    };
    l.conv1_w = take(49 * 128);
    l.conv1_b = take(128);
    for (int i = 0; i < LF_NLAYERS; ++i) {
        int N, K;
        lf_shape(i, &N, &K);
        l.w[i] = take((size_t)N * K);
        l.b[i] = take(N);
        l.wh[i] = take((size_t)N * K / 2);
        l.wl[i] = take((size_t)N * K / 2);
        l.ws[i] = take(64);
    }
    for (int i = 0; i < LF_NNORMS; ++i) l.norm[i] = take(lf_norm_dim(i));
    for (int i = 0; i < 8; ++i) {
        l.wph[i] = take(256 * 512 / 2);
        l.wpl[i] = take(256 * 512 / 2);
        l.wps[i] = take(64);
    }
    for (int i = 0; i < LF_NLAYERS; ++i) {
        int N, K;
        lf_shape(i, &N, &K);
        l.c3h[i] = l.c3l[i] = l.c3s[i] = 0;
        if (i < LF_COARSE0 && K % 288 == 0 && N % 64 == 0) {  // 3x3 convolution (K = 9 Cin)
            l.c3h[i] = take((size_t)N * K / 2);
            l.c3l[i] = take((size_t)N * K / 2);
            l.c3s[i] = take(64);
        }
    }
    l.total = off;
    return l;
}

extern "C" size_t imcui_hip_loftr_packed_floats(void) { return lf_layout().total; }
extern "C" int imcui_hip_loftr_num_layers(void) { return LF_NLAYERS; }
extern "C" int imcui_hip_loftr_layer_shape(int i, int* N, int* K) {
    if (i < 0 || i >= LF_NLAYERS || !N || !K) return IMCUI_ERR_ARG;
    lf_shape(i, N, K);
    return IMCUI_OK;
}
extern "C" int imcui_hip_loftr_num_norms(void) { return LF_NNORMS; }
extern "C" int imcui_hip_loftr_norm_dim(int i) { return (i < 0 || i >= LF_NNORMS) ? -1 : lf_norm_dim(i); }

// conv1: [49][128] tap-major weights + bias (BN folded by the caller); layers: W[N][K] / bias[N]
// already in GEMM layout (convolutions as [Cout][tap][Cin_pad], BN folded, padded channels zero)
extern "C" int imcui_hip_loftr_pack_weights(const float* conv1_w, const float* conv1_b, const float* const* w,
                                            const float* const* b, const float* const* norms, float* packed) {
    if (!conv1_w || !conv1_b || !w || !b || !norms || !packed) return IMCUI_ERR_ARG;
    const LfLayout l = lf_layout();
    memset(packed, 0, l.total * sizeof(float));
    memcpy(packed + l.conv1_w, conv1_w, 49 * 128 * sizeof(float));
    memcpy(packed + l.conv1_b, conv1_b, 128 * sizeof(float));
    for (int i = 0; i < LF_NLAYERS; ++i) {
        int N, K;
        lf_shape(i, &N, &K);
        if (!w[i]) return IMCUI_ERR_ARG;
        memcpy(packed + l.w[i], w[i], (size_t)N * K * sizeof(float));
        if (b[i]) memcpy(packed + l.b[i], b[i], (size_t)N * sizeof(float));
        packed[l.ws[i]] = split_weights_frag_host(w[i], N, K, reinterpret_cast<unsigned short*>(packed + l.wh[i]),
                                                  reinterpret_cast<unsigned short*>(packed + l.wl[i]));
    }
    for (int i = 0; i < LF_NNORMS; ++i) {
        if (!norms[i]) return IMCUI_ERR_ARG;
        memcpy(packed + l.norm[i], norms[i], lf_norm_dim(i) * sizeof(float));
    }
    float* perm = (float*)malloc((size_t)256 * 512 * sizeof(float));
    if (!perm) return IMCUI_ERR_ARG;
    for (int i = 0; i < 8; ++i) {
        ffn_permute_k(w[LF_COARSE0 + i * 6 + 5], 256, 512, perm);
        packed[l.wps[i]] = split_weights_frag_host(perm, 256, 512, reinterpret_cast<unsigned short*>(packed + l.wph[i]),
                                                   reinterpret_cast<unsigned short*>(packed + l.wpl[i]));
    }
    free(perm);
    for (int i = 0; i < LF_NLAYERS; ++i)
        if (l.c3s[i]) {
            int N, K;
            lf_shape(i, &N, &K);
            packed[l.c3s[i]] = pack_conv3x3_split_from_gemm(w[i], N, K / 9, reinterpret_cast<unsigned short*>(packed + l.c3h[i]),
                                                            reinterpret_cast<unsigned short*>(packed + l.c3l[i]), lf_cin_used(i));
        }
    return IMCUI_OK;
}

// ------------------------------------------------------------------ workspace
struct LfWs {
    float *x0, *t1, *x1a, *x1, *t2, *ds2, *x2a, *x2, *t3, *ds3, *x3a, *x3, *fc, *up3, *x2o, *y2, *x2out, *up2, *x1o, *y1, *ff;
    float *q, *k, *v, *att, *m, *hb, *ob, *kvpart, *kv, *rmax, *rsum, *cmax, *csum, *best, *cbest;
    SimDsWs ds;  // the matrix-free dual-softmax (simred.hip): packed coarse features, partials, tile flags
    float *X, *CG, *CW, *F, *fq, *fk, *fv, *fatt, *fm, *fh, *fo, *mconf;
    int *bestj, *flag, *mb, *mi, *mj, *nmatch, *cnt2;
    size_t total;
    bool ok;
};
static LfWs lf_carve(void* ws, size_t bytes, int B, int H0, int W0, int H1, int W1) {
    WsAlloc a(ws, bytes);
    LfWs w;
    // per-image buffers hold the B images of side 0 followed by the B images of side 1 (sizes may differ per side)
    const size_t p2 = (size_t)B * ((size_t)(H0 / 2) * (W0 / 2) + (size_t)(H1 / 2) * (W1 / 2));
    const size_t p4 = (size_t)B * ((size_t)(H0 / 4) * (W0 / 4) + (size_t)(H1 / 4) * (W1 / 4));
    const size_t L0 = (size_t)(H0 / 8) * (W0 / 8), L1 = (size_t)(H1 / 8) * (W1 / 8);
    const size_t p8 = (size_t)B * (L0 + L1);
    const size_t cap = (size_t)B * L0, cap1 = (size_t)B * L1;
    w.x0 = a.get<float>(p2 * 128);
    w.t1 = a.get<float>(p2 * 128);
    w.x1a = a.get<float>(p2 * 128);
    w.x1 = a.get<float>(p2 * 128);
    w.t2 = a.get<float>(p4 * CP);
    w.ds2 = a.get<float>(p4 * CP);
    w.x2a = a.get<float>(p4 * CP);
    w.x2 = a.get<float>(p4 * CP);
    w.t3 = a.get<float>(p8 * 256);
    w.ds3 = a.get<float>(p8 * 256);
    w.x3a = a.get<float>(p8 * 256);
    w.x3 = a.get<float>(p8 * 256);
    w.fc = a.get<float>(p8 * 256);
    w.up3 = a.get<float>(p4 * 256);
    w.x2o = a.get<float>(p4 * 256);
    w.y2 = a.get<float>(p4 * 256);
    w.x2out = a.get<float>(p4 * CP);
    w.up2 = a.get<float>(p2 * CP);
    w.x1o = a.get<float>(p2 * CP);
    w.y1 = a.get<float>(p2 * CP);
    w.ff = a.get<float>(p2 * 128);
    w.q = a.get<float>(p8 * 256);
    w.k = a.get<float>(p8 * 256);
    w.v = a.get<float>(p8 * 256);
    w.att = a.get<float>(p8 * 256);
    w.m = a.get<float>(p8 * 256);
    w.hb = a.get<float>(p8 * 512);
    w.ob = a.get<float>(p8 * 256);
    const size_t nchunk = ((L0 > L1 ? L0 : L1) + LA_CHUNK - 1) / LA_CHUNK;
    w.kvpart = a.get<float>((size_t)2 * B * 8 * nchunk * (32 * 32 + 32));
    w.kv = a.get<float>((size_t)2 * B * 8 * (32 * 32 + 32));
    simred_ds_carve(a, B, (int)L0, (int)L1, 256, w.ds);
    w.rmax = a.get<float>(cap);
    w.rsum = a.get<float>(cap);
    w.cmax = a.get<float>(cap1);
    w.csum = a.get<float>(cap1);
    w.best = a.get<float>(cap);
    w.cbest = a.get<float>(cap1);
    w.X = a.get<float>(2 * cap * 25 * 256);
    w.CG = a.get<float>(2 * cap * 256);
    w.CW = a.get<float>(2 * cap * 128);
    w.F = a.get<float>(2 * cap * 25 * 128);
    w.fq = a.get<float>(2 * cap * 25 * 128);
    w.fk = a.get<float>(2 * cap * 25 * 128);
    w.fv = a.get<float>(2 * cap * 25 * 128);
    w.fatt = a.get<float>(2 * cap * 25 * 128);
    w.fm = a.get<float>(2 * cap * 25 * 128);
    w.fh = a.get<float>(2 * cap * 25 * 256);
    w.fo = a.get<float>(2 * cap * 25 * 128);
    w.mconf = a.get<float>(cap);
    w.bestj = a.get<int>(cap);
    w.flag = a.get<int>(cap);
    w.mb = a.get<int>(cap);
    w.mi = a.get<int>(cap);
    w.mj = a.get<int>(cap);
    w.nmatch = a.get<int>(4);
    w.cnt2 = a.get<int>(4);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}
extern "C" size_t imcui_hip_loftr_workspace_bytes(int B, int H0, int W0, int H1, int W1) { return lf_carve(nullptr, 0, B, H0, W0, H1, W1).total; }

// byte offsets of a few workspace buffers, for the parity tests: 0 = coarse features after the transformer
// [B*L0 + B*L1, 256], 1 = fine features [B*H0/2*W0/2 + B*H1/2*W1/2, 128], 2 = (the similarity matrix until round 4: gone), 3 = fine windows F
extern "C" size_t imcui_hip_loftr_debug_offset(int which, int B, int H0, int W0, int H1, int W1) {
    LfWs w = lf_carve((void*)256, (size_t)-1 >> 1, B, H0, W0, H1, W1);
    const char* base = (const char*)256;
    switch (which) {
        case 0: return (const char*)w.fc - base;
        case 1: return (const char*)w.ff - base;
        case 2: return 0;
        case 3: return (const char*)w.F - base;
        default: return 0;
    }
}

extern "C" int imcui_hip_loftr_last_fine_mode(imcui_hip_t* h, int* matches) {
    if (!h) return IMCUI_ERR_ARG;
    if (matches) *matches = h->loftr_fine_matches;
    return h->loftr_fine_mode;
}

__global__ void lf_counts_kernel(const int* nmatch, int* cnt2) {
    cnt2[0] = *nmatch;       // rows of the per-match GEMMs
    cnt2[1] = *nmatch * 25;  // rows of the per-window-token GEMMs
}
__global__ void lf_copy_int_kernel(const int* src, int* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------ forward
extern "C" int imcui_hip_loftr_forward(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B,
                                       int H0, int W0, int H1, int W1, double match_threshold, int temp_bug_fix,
                                       float* keypoints0, float* keypoints1, float* confidence, int* batch_indexes,
                                       int* num_matches, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (B <= 0) return IMCUI_OK;
    if (H0 % 8 || W0 % 8 || H0 < 32 || W0 < 32 || H1 % 8 || W1 % 8 || H1 < 32 || W1 < 32)
        return imcui_set_err(h, IMCUI_ERR_ARG, "loftr: image sizes %dx%d / %dx%d must be multiples of 8 (>= 32)", W0, H0, W1, H1);
    if (!packed || !image0 || !image1 || !keypoints0 || !keypoints1 || !confidence || !batch_indexes || !num_matches)
        return imcui_set_err(h, IMCUI_ERR_ARG, "loftr: null argument");
    // per side s: image size, 1/2 - 1/4 - 1/8 maps.  kornia runs the backbone on the concatenated batch when both
    // images have one size and on each image otherwise (LoFTR.forward); convolutions do not mix images either way.
    const int Hs[2] = {H0, H1}, Ws[2] = {W0, W1};
    const int hcs[2] = {H0 / 8, H1 / 8}, wcs[2] = {W0 / 8, W1 / 8};
    if (hcs[0] > 256 || wcs[0] > 256 || hcs[1] > 256 || wcs[1] > 256)
        return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "loftr: coarse maps %dx%d / %dx%d exceed the 256x256 positional encoding", wcs[0], hcs[0], wcs[1], hcs[1]);
    LfWs w = lf_carve(ws, ws_bytes, B, H0, W0, H1, W1);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "loftr: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const LfLayout l = lf_layout();
    const float* P = packed;
    const bool split = h->precision == 1;
    const bool same = (H0 == H1 && W0 == W1);
    const int Ls[2] = {hcs[0] * wcs[0], hcs[1] * wcs[1]};
    const int L = Ls[0], S = Ls[1];
    const int cap = B * L;
    int rc;
#define LFRUN(x)                       \
    do {                               \
        rc = (x);                      \
        if (rc != IMCUI_OK) return rc; \
    } while (0)

    auto wts = [&](GemmP& g, int li) {
        int N, K;
        lf_shape(li, &N, &K);
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
    // pixels per image of side s at resolution 1/div
    auto npx = [&](int s, int div) { return (size_t)(Hs[s] / div) * (Ws[s] / div); };
    // conv as GEMM over NHWC at input resolution 1/div: `in` / `out` / `resid` hold side 0 then side 1; one launch over
    // the 2B images when both sides have one size, one launch per side otherwise
    static const bool conv_gemm_only = getenv("IMCUI_CONV_GEMM_ONLY") != nullptr;  // A/B switch: every convolution on the implicit GEMM
    // rup: `resid` is the map at HALF the output resolution whose bilinear x2 up-sampling is the residual (evaluated in the GEMM
    // epilogue; 1x1 stride-1 layers of the split mode)
    auto conv = [&](int li, const float* in, float* out, int div, int cin, int ks, int stride, const float* resid, int act, bool rup = false) -> int {
        for (int s = 0; s < (same ? 1 : 2); ++s) {
            if (split && !conv_gemm_only && ks == 3 && stride == 1 && l.c3s[li] != 0) {
                // 3x3 stride 1: the patch-staging kernel (conv.hip) reads every input pixel once per 64 output channels; the
                // implicit GEMM re-reads it once per tap and misses L2 at these map sizes (PMC: 6-10 x the input bytes fetched)
                int N, K;
                lf_shape(li, &N, &K);
                const size_t off = s ? (size_t)B * npx(0, div) : 0;
                const int used = lf_cin_used(li) ? lf_cin_used(li) : cin;
                const int r = conv3x3_split_launch(h, in + off * cin, reinterpret_cast<const unsigned short*>(P + l.c3h[li]),
                                                   reinterpret_cast<const unsigned short*>(P + l.c3l[li]), P + l.c3s[li], P + l.b[li], out + off * N,
                                                   same ? 2 * B : B, Hs[s] / div, Ws[s] / div, used, N, act, 0, stream, resid ? resid + off * N : nullptr, cin,
                                                   lf_cout_live(li));
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
                g.rup_align = 1;
            }
            g.act = act;
            const int r = gemm_launch(h, g, stream);
            if (r != IMCUI_OK) return r;
        }
        return IMCUI_OK;
    };
    // ---- a13: ResNetFPN_8_2
    for (int s = 0; s < 2; ++s) {
        const int H2s = Hs[s] / 2, W2s = Ws[s] / 2;
        const long npix = (long)B * H2s * W2s;
        long blocks = min((npix / 4 + 15) / 16, (long)256 * 32);  // a thread group of 16 lanes does 4 pixels
        hipLaunchKernelGGL(lf_conv7_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, s ? image1 : image0, P + l.conv1_w,
                           P + l.conv1_b, w.x0 + (s ? (size_t)B * npx(0, 2) * 128 : 0), Hs[s], Ws[s], H2s, W2s, npix);
    }
    IMCUI_CHECK_LAUNCH(h);
    LFRUN(conv(LF_L1_0_C1, w.x0, w.t1, 2, 128, 3, 1, nullptr, 1));
    LFRUN(conv(LF_L1_0_C2, w.t1, w.x1a, 2, 128, 3, 1, w.x0, 1));
    LFRUN(conv(LF_L1_1_C1, w.x1a, w.t1, 2, 128, 3, 1, nullptr, 1));
    LFRUN(conv(LF_L1_1_C2, w.t1, w.x1, 2, 128, 3, 1, w.x1a, 1));
    LFRUN(conv(LF_L2_0_C1, w.x1, w.t2, 2, 128, 3, 2, nullptr, 1));
    LFRUN(conv(LF_L2_0_DS, w.x1, w.ds2, 2, 128, 1, 2, nullptr, 0));
    LFRUN(conv(LF_L2_0_C2, w.t2, w.x2a, 4, CP, 3, 1, w.ds2, 1));
    LFRUN(conv(LF_L2_1_C1, w.x2a, w.t2, 4, CP, 3, 1, nullptr, 1));
    LFRUN(conv(LF_L2_1_C2, w.t2, w.x2, 4, CP, 3, 1, w.x2a, 1));
    LFRUN(conv(LF_L3_0_C1, w.x2, w.t3, 4, CP, 3, 2, nullptr, 1));
    LFRUN(conv(LF_L3_0_DS, w.x2, w.ds3, 4, CP, 1, 2, nullptr, 0));
    LFRUN(conv(LF_L3_0_C2, w.t3, w.x3a, 8, 256, 3, 1, w.ds3, 1));
    LFRUN(conv(LF_L3_1_C1, w.x3a, w.t3, 8, 256, 3, 1, nullptr, 1));
    LFRUN(conv(LF_L3_1_C2, w.t3, w.x3, 8, 256, 3, 1, w.x3a, 1));
    LFRUN(conv(LF_OUT3, w.x3, w.fc, 8, 256, 1, 1, nullptr, 0));
    // bilinear x2 (align_corners=True) of the maps at resolution 1/div with C channels, per side
    auto upsample = [&](const float* in, float* out, int div, int C) {
        for (int s = 0; s < (same ? 1 : 2); ++s) {
            const int hh = Hs[s] / div, ww = Ws[s] / div;
            const long n4 = (long)(same ? 2 * B : B) * (2 * hh) * (2 * ww) * (C / 4);
            hipLaunchKernelGGL(lf_upsample2_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)65536)), dim3(256), 0, stream,
                               in + (s ? (size_t)B * npx(0, div) * C : 0), out + (s ? (size_t)B * npx(0, div / 2) * C : 0), hh, ww, C, n4);
        }
    };
    static const bool up_unfused = getenv("IMCUI_UPSAMPLE_UNFUSED") != nullptr;  // A/B switch: materialise the up-sampled maps
    const bool upf = split && !up_unfused;
    if (upf) {
        LFRUN(conv(LF_OUT2, w.x2, w.x2o, 4, CP, 1, 1, w.fc, 0, true));  // + bilinear x2 of layer3_outconv's map, in the epilogue
    } else {
        upsample(w.fc, w.up3, 8, 256);
        LFRUN(conv(LF_OUT2, w.x2, w.x2o, 4, CP, 1, 1, w.up3, 0));
    }
    LFRUN(conv(LF_OUT2B_0, w.x2o, w.y2, 4, 256, 3, 1, nullptr, 2));
    LFRUN(conv(LF_OUT2B_3, w.y2, w.x2out, 4, 256, 3, 1, nullptr, 0));
    // The last FPN stage (1/2 resolution: layer1_outconv + layer1_outconv2 -> the 128-channel fine map `ff`) is only ever read through the
    // 5x5 windows of the matched cells.  Option loftr_fine_sparse (default 1): it is deferred until the matches are known and evaluated on
    // the windows alone when that is the cheaper way (few matches; `sparse_fine_stage` below); 0 = always the dense maps, here, as before.
    auto dense_fine_stage = [&]() -> int {
        if (upf) {
            LFRUN(conv(LF_OUT1, w.x1, w.x1o, 2, 128, 1, 1, w.x2out, 0, true));
        } else {
            upsample(w.x2out, w.up2, 4, CP);
            LFRUN(conv(LF_OUT1, w.x1, w.x1o, 2, 128, 1, 1, w.up2, 0));
        }
        LFRUN(conv(LF_OUT1B_0, w.x1o, w.y1, 2, CP, 3, 1, nullptr, 2));
        LFRUN(conv(LF_OUT1B_3, w.y1, w.ff, 2, CP, 3, 1, nullptr, 0));
        return IMCUI_OK;
    };
    h->loftr_fine_mode = 0;
    h->loftr_fine_matches = -1;
    bool fine_deferred = h->opt[OPT_LOFTR_FINE_SPARSE] != 0;
    if (fine_deferred) {  // (a stream that is being captured into a graph cannot be waited for: dense)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) fine_deferred = false;
    }
    if (!fine_deferred) LFRUN(dense_fine_stage());

    // ---- a14: positional encoding + coarse LocalFeatureTransformer (linear attention)
    const size_t tok1 = (size_t)B * L;  // first token row of side 1
    for (int s = 0; s < (same ? 1 : 2); ++s) {
        const long n = (long)(same ? 2 * B : B) * Ls[s] * 256;
        hipLaunchKernelGGL(lf_posenc_kernel, dim3((unsigned)min((n + 255) / 256, (long)65536)), dim3(256), 0, stream,
                           w.fc + (s ? tok1 * 256 : 0), hcs[s], wcs[s], 256, n, temp_bug_fix);
    }
    auto lin = [&](int li, const float* A, long lda, const float* A2, float* C, long rows, int relu, const int* mcnt) -> int {
        GemmP g;
        wts(g, li);
        g.bias = nullptr;  // transformer Linears have no bias
        g.epi = relu ? EPI_RELU : EPI_BIAS;
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
        g.mcnt = mcnt;
        g.cnt_stride = 0;
        return gemm_launch(h, g, stream);
    };
    const int nchunk_max = cdiv(L > S ? L : S, LA_CHUNK);
    static const bool mlp_unfused = getenv("IMCUI_LG_FFN_UNFUSED") != nullptr;  // A/B switch shared with lightglue.hip
    // encoder layer: `ns` query sequences of Lq tokens starting at token row qt attend to `ns` source sequences of Lsrc
    // tokens starting at token row st; kvslot = first K'V slot (sequence index) used for the sources
    auto coarse_layer = [&](int layer, size_t qt, size_t st, int ns, int Lq, int Lsrc, int kvslot) -> int {
        const int base = LF_COARSE0 + layer * 6;
        const size_t qo = qt * 256, so = st * 256;
        const long qrows = (long)ns * Lq, srows = (long)ns * Lsrc;
        const int nchunk = cdiv(Lsrc, LA_CHUNK);
        int r;
        if ((r = lin(base + 0, w.fc + qo, 256, nullptr, w.q + qo, qrows, 0, nullptr))) return r;
        if ((r = lin(base + 1, w.fc + so, 256, nullptr, w.k + so, srows, 0, nullptr))) return r;
        if ((r = lin(base + 2, w.fc + so, 256, nullptr, w.v + so, srows, 0, nullptr))) return r;
        float* part = w.kvpart + (size_t)kvslot * 8 * nchunk_max * 1056;
        float* kv = w.kv + (size_t)kvslot * 8 * 1056;
        hipLaunchKernelGGL(lf_la_kv_partial_kernel<32>, dim3(nchunk, 8, ns), dim3(256), 0, stream, w.k + so, w.v + so, Lsrc, 8, part, nchunk);
        hipLaunchKernelGGL(lf_la_kv_reduce_kernel<32>, dim3(ns * 8, cdiv(1056, 256)), dim3(256), 0, stream, part, nchunk, kv);
        const size_t smem = (8 * 1056 + 4 * 256) * sizeof(float);
        hipLaunchKernelGGL(lf_la_apply_kernel<32>, dim3(cdiv(Lq, 64), ns), dim3(256), smem, stream, w.q + qo, kv, 0, 0, Lq, Lsrc, 8,
                           w.att + qo);
        if ((r = lin(base + 3, w.att + qo, 256, nullptr, w.m + qo, qrows, 0, nullptr))) return r;
        const float* n1w = P + l.norm[layer * 4 + 0];
        hipLaunchKernelGGL(lf_layernorm_kernel<4>, dim3((unsigned)((qrows + 3) / 4)), dim3(256), 0, stream, w.m + qo, n1w,
                           P + l.norm[layer * 4 + 1], (const float*)nullptr, w.m + qo, qrows, 0);
        if (split && !mlp_unfused) {  // x += norm2(mlp.2(relu(mlp.0([x | message])))) in one kernel (ffn.hip)
            FfnP f;
            f.act = 3;
            f.x = w.fc + qo;
            f.ctx = w.m + qo;
            f.out = w.fc + qo;
            f.w1h = reinterpret_cast<const unsigned short*>(P + l.wh[base + 4]);
            f.w1l = reinterpret_cast<const unsigned short*>(P + l.wl[base + 4]);
            f.s1 = P + l.ws[base + 4];
            f.w2h = reinterpret_cast<const unsigned short*>(P + l.wph[layer]);
            f.w2l = reinterpret_cast<const unsigned short*>(P + l.wpl[layer]);
            f.s2 = P + l.wps[layer];
            f.gamma = P + l.norm[layer * 4 + 2];
            f.beta = P + l.norm[layer * 4 + 3];
            f.M = (int)qrows;
            return ffn_launch(h, f, stream);
        }
        if ((r = lin(base + 4, w.fc + qo, 256, w.m + qo, w.hb + qt * 512, qrows, 1, nullptr))) return r;
        if ((r = lin(base + 5, w.hb + qt * 512, 512, nullptr, w.ob + qo, qrows, 0, nullptr))) return r;
        hipLaunchKernelGGL(lf_layernorm_kernel<4>, dim3((unsigned)((qrows + 3) / 4)), dim3(256), 0, stream, w.ob + qo,
                           P + l.norm[layer * 4 + 2], P + l.norm[layer * 4 + 3], w.fc + qo, w.fc + qo, qrows, 1);
        return IMCUI_OK;
    };
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lf_la_apply_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (8 * 1056 + 4 * 256) * (int)sizeof(float));
    for (int layer = 0; layer < 8; ++layer) {
        if ((layer & 1) == 0) {  // self: every image attends to itself
            if (same) {
                LFRUN(coarse_layer(layer, 0, 0, 2 * B, L, L, 0));
            } else {
                LFRUN(coarse_layer(layer, 0, 0, B, L, L, 0));
                LFRUN(coarse_layer(layer, tok1, tok1, B, S, S, B));
            }
        } else {
            LFRUN(coarse_layer(layer, 0, tok1, B, L, S, B));  // feat0 <- layer(feat0, feat1)
            LFRUN(coarse_layer(layer, tok1, 0, B, S, L, 0));  // feat1 <- layer(feat1, updated feat0)
        }
    }
    IMCUI_CHECK_LAUNCH(h);

    // ---- a15: dual-softmax coarse matching.  sim = (f0 / 16) . (f1 / 16)^T / 0.1 [B, L, S] -- 1 GiB per 1024 x 1024 pair -- is never
    // stored: simred.hip computes it tile by tile twice, once for the soft-max statistics of both directions and once
    // more, only where a confidence can exceed the threshold, for conf = softmax_i * softmax_j, its row maxima (first column) and its
    // column maxima.  (Rounds 2-4 wrote the matrix and read it back two to four times: 51.6 GB of HBM traffic per 16-pair step.)
    LFRUN(simred_dual_softmax(h, w.ds, w.fc, 256, (long)L * 256, w.fc + tok1 * 256, 256, (long)S * 256, B, L, S, 256, 0.00390625f / 0.1f, (float)match_threshold,
                              w.rmax, w.rsum, w.cmax, w.csum, w.best, w.bestj, w.cbest, stream));
    const dim3 blk(256);
    hipLaunchKernelGGL(lf_decide_kernel, dim3(cdiv(cap, 256)), blk, 0, stream, w.best, w.bestj, w.cbest, L, S, wcs[0], hcs[0], wcs[1], hcs[1],
                       2, (float)match_threshold, w.flag, (long)cap);
    hipLaunchKernelGGL(lf_compact_kernel, dim3(1), dim3(1024), 0, stream, w.flag, w.best, w.bestj, L, (long)cap, cap, w.mb, w.mi,
                       w.mj, w.mconf, w.nmatch);
    hipLaunchKernelGGL(lf_counts_kernel, dim3(1), dim3(1), 0, stream, w.nmatch, w.cnt2);
    IMCUI_CHECK_LAUNCH(h);

    // ---- a16: fine level on the 5x5 windows of the matches
    bool sparse_done = false;
    int gcap = cap;  // grid size (matches) of the fine-level launches: the capacity, or the count once it has been read back
    if (fine_deferred) {
        // the number of matches decides (one 4-byte read-back: the only host round trip of the forward pass)
        int nm = 0;
        if (hipMemcpyAsync(&nm, w.nmatch, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return imcui_set_err(h, IMCUI_ERR_HIP, "loftr: reading the match count back failed");
        h->loftr_fine_matches = nm;
        gcap = max(1, min(nm, cap));  // (the fine level below launches for the matches that exist instead of B * L0 mostly empty workgroups)
        const size_t p2 = (size_t)B * (npx(0, 2) + npx(1, 2));
        // cost model (multiply-adds; the window path runs on the implicit GEMM at about 0.75 of the dense kernels' rate: measured break-even
        // ~3700 matches per 1024 x 1024 pair, profiles/r06_lab_loftr_fine.txt -- option value 2 takes the windows whatever the count): per
        // window of one side 81 x 128 x 256 + 49 x 2304 x 256 + 25 x 2304 x 128, per dense pixel 128 x 256 + 9 x 224 x (224 + 128)
        const double sparse_cost = (double)nm * 2.0 * (81.0 * 128 * 256 + 49.0 * 2304 * 256 + 25.0 * 2304 * 128) / 0.75;
        const double dense_cost = (double)p2 * (128.0 * 256 + 9.0 * 224 * (224 + 128));
        const long wcap_chunk = (long)min((size_t)32768, p2 / 81);  // windows per chunk: the dense stage's idle buffers hold them
        if (nm == 0) {
            sparse_done = true;  // nothing reads the fine map
        } else if ((h->opt[OPT_LOFTR_FINE_SPARSE] == 2 || sparse_cost < dense_cost) && wcap_chunk >= 1) {
            float* G = w.up2;                                           // [chunk, 81, 128]
            float* T = w.x1o;                                           // [chunk, 81, CP]
            float* Y1 = w.y1;                                           // [chunk, 49, CP]
            unsigned char* V = reinterpret_cast<unsigned char*>(w.ff);  // [chunk, 81]
            for (int s = 0; s < 2; ++s) {
                const int hf = Hs[s] / 2, wf = Ws[s] / 2;
                const float* x1s = w.x1 + (s ? (size_t)B * npx(0, 2) * 128 : 0);
                const float* x2s = w.x2out + (s ? (size_t)B * npx(0, 4) * CP : 0);
                for (long m0 = 0; m0 < nm; m0 += wcap_chunk) {
                    const int n = (int)min(wcap_chunk, (long)nm - m0);
                    hipLaunchKernelGGL(lf_win_gather_kernel, dim3(n), blk, 0, stream, x1s, x2s, w.mb, s ? w.mj : w.mi, (int)m0, n, hf, wf, wcs[s], 4, CP, G, T, V);
                    IMCUI_CHECK_LAUNCH(h);
                    const unsigned mgrid = (unsigned)min((long)n * 81 / 4 + 1, (long)8192);
                    {  // layer1_outconv (1x1) + the gathered up-sampled residual, in place
                        GemmP g;
                        wts(g, LF_OUT1);
                        g.epi = EPI_CONV;
                        g.A = G;
                        g.lda = 128;
                        g.M = n * 81;
                        g.C = T;
                        g.ldc = g.N;
                        g.resid = T;
                        g.ldr = g.N;
                        g.act = 0;
                        LFRUN(gemm_launch(h, g, stream));
                        hipLaunchKernelGGL(lf_win_mask_kernel, dim3(mgrid), blk, 0, stream, T, V, n, 9, 0, (long)CP, CP);
                    }
                    auto wconv = [&](int li, const float* in, int side_in, float* out, long ldc, int act) -> int {
                        GemmP g;
                        wts(g, li);
                        g.epi = EPI_CONV;
                        g.A = in;
                        g.conv_k = 3;
                        g.conv_stride = 1;
                        g.conv_pad = 0;
                        g.conv_hin = g.conv_win = side_in;
                        g.conv_hout = g.conv_wout = side_in - 2;
                        g.conv_cin = CP;
                        g.M = n * (side_in - 2) * (side_in - 2);
                        g.C = out;
                        g.ldc = ldc;
                        g.act = act;
                        return gemm_launch(h, g, stream);
                    };
                    LFRUN(wconv(LF_OUT1B_0, T, 9, Y1, CP, 2));
                    hipLaunchKernelGGL(lf_win_mask_kernel, dim3(mgrid), blk, 0, stream, Y1, V, n, 7, 1, (long)CP, CP);
                    float* Xs = w.X + ((size_t)s * cap + m0) * 25 * 256;  // the window rows of these matches: [25][fine 128 | coarse 128]
                    LFRUN(wconv(LF_OUT1B_3, Y1, 7, Xs, 256, 0));
                    hipLaunchKernelGGL(lf_win_mask_kernel, dim3(mgrid), blk, 0, stream, Xs, V, n, 5, 2, (long)256, 128);
                    IMCUI_CHECK_LAUNCH(h);
                }
            }
            sparse_done = true;
            h->loftr_fine_mode = 1;
        } else {
            LFRUN(dense_fine_stage());
        }
    }
    hipLaunchKernelGGL(lf_fine_gather_kernel, dim3(gcap, 2), blk, 0, stream, sparse_done ? (const float*)nullptr : w.ff, w.fc, w.mb, w.mi, w.mj, w.nmatch, B, cap,
                       H0 / 2, W0 / 2, hcs[0], wcs[0], H1 / 2, W1 / 2, hcs[1], wcs[1], 4, w.X, w.CG);
    auto lin_b = [&](int li, const float* A, long lda, const float* A2, float* C, long side_rows_cap, int relu, int per_token,
                     int side0, int nsides, bool with_bias) -> int {
        // rows of side s start at s * side_rows_cap; valid rows = nmatch (* 25)
        GemmP g;
        wts(g, li);
        if (!with_bias) g.bias = nullptr;
        g.epi = relu ? EPI_RELU : EPI_BIAS;
        g.batch = nsides;
        g.A = A + (size_t)side0 * side_rows_cap * lda;
        g.lda = lda;
        g.a_bs = side_rows_cap * lda;
        if (A2) {
            g.A2 = A2 + (size_t)side0 * side_rows_cap * lda;
            g.lda2 = lda;
            g.a2_bs = side_rows_cap * lda;
            g.K1 = (int)lda;
        }
        g.C = C + (size_t)side0 * side_rows_cap * g.N;
        g.ldc = g.N;
        g.c_bs = side_rows_cap * g.N;
        g.M = (int)min(side_rows_cap, (long)gcap * (per_token ? 25 : 1));  // rows launched (the live count is read on the device: mcnt)
        g.mcnt = w.cnt2 + (per_token ? 1 : 0);
        g.cnt_stride = 0;
        return gemm_launch(h, g, stream);
    };
    LFRUN(lin_b(LF_DOWN, w.CG, 256, nullptr, w.CW, cap, 0, 0, 0, 2, true));
    hipLaunchKernelGGL(lf_fine_fill_kernel, dim3(gcap, 2), blk, 0, stream, w.CW, w.nmatch, cap, w.X);
    const long wcap = (long)cap * 25;
    LFRUN(lin_b(LF_MERGEF, w.X, 256, nullptr, w.F, wcap, 0, 1, 0, 2, true));
    auto fine_layer = [&](int layer, int side0, int nsides, int cross) -> int {
        const int base = LF_FINE0 + layer * 6;
        const int src0 = cross ? 1 - side0 : side0;
        int r;
        if ((r = lin_b(base + 0, w.F, 128, nullptr, w.fq, wcap, 0, 1, side0, nsides, false))) return r;
        if ((r = lin_b(base + 1, w.F, 128, nullptr, w.fk, wcap, 0, 1, src0, nsides, false))) return r;
        if ((r = lin_b(base + 2, w.F, 128, nullptr, w.fv, wcap, 0, 1, src0, nsides, false))) return r;
        hipLaunchKernelGGL(lf_la_window_kernel, dim3(gcap, nsides), blk, 0, stream, w.fq, w.fk, w.fv, w.nmatch, cap, side0, cross,
                           w.fatt);
        if ((r = lin_b(base + 3, w.fatt, 128, nullptr, w.fm, wcap, 0, 1, side0, nsides, false))) return r;
        const int nl = 32 + layer * 4;
        // LayerNorm of the live window tokens of the processed sides (nmatch * 25 of the wcap rows per side)
        const long rows = (long)nsides * wcap;
        float* fm0 = w.fm + (size_t)side0 * wcap * 128;
        const unsigned lngrid = (unsigned)min((rows + 3) / 4, (long)4096);
        hipLaunchKernelGGL(lf_layernorm_kernel<2>, dim3(lngrid), blk, 0, stream, fm0, P + l.norm[nl + 0],
                           P + l.norm[nl + 1], (const float*)nullptr, fm0, rows, 0, w.cnt2 + 1, wcap);
        if ((r = lin_b(base + 4, w.F, 128, w.fm, w.fh, wcap, 1, 1, side0, nsides, false))) return r;
        if ((r = lin_b(base + 5, w.fh, 256, nullptr, w.fo, wcap, 0, 1, side0, nsides, false))) return r;
        float* fo0 = w.fo + (size_t)side0 * wcap * 128;
        float* f0 = w.F + (size_t)side0 * wcap * 128;
        hipLaunchKernelGGL(lf_layernorm_kernel<2>, dim3(lngrid), blk, 0, stream, fo0, P + l.norm[nl + 2],
                           P + l.norm[nl + 3], f0, f0, rows, 1, w.cnt2 + 1, wcap);
        return IMCUI_OK;
    };
    LFRUN(fine_layer(0, 0, 2, 0));  // self on both windows
    LFRUN(fine_layer(1, 0, 1, 1));  // cross: window0 <- (window0, window1)
    LFRUN(fine_layer(1, 1, 1, 1));  //        window1 <- (window1, updated window0)
    // kornia: scale = hw0_i[0] / hw0_c[0] (= 8) for BOTH images' coarse key-points, fine offsets scaled by hw0_i[0] / hw0_f[0] (= 2)
    hipLaunchKernelGGL(lf_fine_match_kernel, dim3(cdiv(gcap, 4)), blk, 0, stream, w.F, w.mi, w.mj, w.nmatch, cap, wcs[0], wcs[1],
                       (float)H0 / (float)hcs[0], (float)H0 / (float)(H0 / 2), keypoints0, keypoints1);
    hipMemcpyAsync(confidence, w.mconf, (size_t)cap * sizeof(float), hipMemcpyDeviceToDevice, stream);
    hipLaunchKernelGGL(lf_copy_int_kernel, dim3(cdiv(cap, 256)), blk, 0, stream, w.mb, batch_indexes, cap);
    hipLaunchKernelGGL(lf_copy_int_kernel, dim3(1), dim3(64), 0, stream, w.nmatch, num_matches, 1);
    IMCUI_CHECK_LAUNCH(h);
#undef LFRUN
    return IMCUI_OK;
}
