// DUSt3R pair network on MI355X: what `dust3r.inference.inference(pairs, self.net, ...)` computes for
// imcui/hloc/matchers/duster.py:58-74 (`AsymmetricCroCo3DStereo`, ViT-L encoder / ViT-B decoder / DPT head, the
// un-vendored `third_party/dust3r` submodule; BASELINE config 5).  3 x f16 split mode only.
//
//   encoder   every IMAGE is encoded once (the reference encodes both images again for the swapped pair): 16x16 patches as a
//             GEMM, then per block LayerNorm -> qkv GEMM -> RoPE2D + f16 planes -> the flash attention kernel of attention.hip
//             (head_dim 64, the LightGlue kernel) -> projection GEMM with the residual in its epilogue -> LayerNorm -> fc1 GEMM with
//             GELU in its epilogue -> fc2 GEMM with the residual.
//   decoder   token streams [view 1 of every directed pair | view 2 of every directed pair]; the first half runs through
//             `dec_blocks`, the second through `dec_blocks2`.  Per block: keys / values of the cross attention from the OTHER
//             half's tokens as they are before the block, self attention, cross attention (attention kernel with key
//             sequence (seq + P) % 2P), MLP.
//   head      DPT over four hooked token maps (NHWC = token-major rows, so the reassembly is GEMMs plus a pixel shuffle for
//             the kernel = stride transposed convolutions), 3x3 convolutions on the patch-staging kernel of conv.hip, bilinear x2
//             (align_corners=True), then 1x1 128 -> 4 fused with the point-map post-processing.
// Token rows are padded to R = roundup(T, 128) per sequence; GEMMs skip the tiles past T, LayerNorm / element-wise kernels run
// over them harmlessly (row-wise arithmetic), the plane-split kernels write zeros there.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "attention.h"
#include "conv.h"
#include "dust3r_kernels.h"
#include "gemm.h"
#include "imcui_hip.h"

// ------------------------------------------------------------------ layer table
struct DuCfg {
    int E, enc_depth, D, dec_depth;
    int desc;  // MASt3R: width of the local descriptors (head_local_features); 0 = DUSt3R
};
static const int DU_LD[4] = {96, 192, 384, 768};
enum { DU_HEAD_BASE = 33, DU_ENC_J = 4, DU_DEC_J = 7 };
static int du_head_layers(const DuCfg& c) { return DU_HEAD_BASE + (c.desc > 0 ? 2 : 0); }
// encoder block: qkv, proj, fc1, fc2.  decoder block: qkv, proj, cross q, cross [k | v], cross proj, fc1, fc2.
// head: 0 act1 1x1 | 1 act1 transposed 4x4/4 | 2 act2 1x1 | 3 act2 transposed 2x2/2 | 4 act3 1x1 | 5 act4 1x1 | 6 act4 3x3/2 |
//       7-10 layer_rn | 11 + 5 q + {rcu1.conv1, rcu1.conv2, rcu2.conv1, rcu2.conv2, out_conv} for refinenet 4 - q | 31 head.0 | 32 head.2
//       MASt3R: 33 head_local_features.fc1 (E + D -> 4 (E + D)) | 34 .fc2 (-> (desc + 1) x 256)
static int du_l_enc(const DuCfg&, int i, int j) { return 1 + DU_ENC_J * i + j; }
static int du_l_demb(const DuCfg& c) { return 1 + DU_ENC_J * c.enc_depth; }
static int du_l_dec(const DuCfg& c, int s, int i, int j) { return du_l_demb(c) + 1 + (s * c.dec_depth + i) * DU_DEC_J + j; }
static int du_l_head(const DuCfg& c, int hd) { return du_l_demb(c) + 1 + 2 * c.dec_depth * DU_DEC_J + hd * du_head_layers(c); }
static int du_nlayers(const DuCfg& c) { return du_l_head(c, 2); }

// kind 0: GEMM weight [N][K] -> fragment-major planes; 1: 3x3 stride-1 convolution [N][9][Cin] -> planes of conv3x3_split_kernel
static void du_shape(const DuCfg& c, int li, int* N, int* K, int* kind) {
    *kind = 0;
    const int E = c.E, D = c.D;
    if (li == 0) {
        *N = E;
        *K = 768;
        return;
    }
    if (li < du_l_demb(c)) {
        const int j = (li - 1) % DU_ENC_J;
        *N = j == 0 ? 3 * E : j == 2 ? 4 * E : E;
        *K = j == 3 ? 4 * E : E;
        return;
    }
    if (li == du_l_demb(c)) {
        *N = D;
        *K = E;
        return;
    }
    if (li < du_l_head(c, 0)) {
        const int j = (li - du_l_demb(c) - 1) % DU_DEC_J;
        *N = j == 0 ? 3 * D : j == 3 ? 2 * D : j == 5 ? 4 * D : D;
        *K = j == 6 ? 4 * D : D;
        return;
    }
    const int j = (li - du_l_head(c, 0)) % du_head_layers(c);
    static const int tabN[7] = {96, 96 * 16, 192, 192 * 4, 384, 768, 768};
    if (j < 7) {
        const int tabK[7] = {E, 96, D, 192, D, D, 9 * 768};
        *N = tabN[j];
        *K = tabK[j];
        return;
    }
    if (j < 11) {
        *N = 256;
        *K = 9 * DU_LD[j - 7];
        *kind = 1;
        return;
    }
    if (j < 31) {
        const int u = (j - 11) % 5;
        *N = 256;
        *K = u == 4 ? 256 : 9 * 256;
        *kind = u == 4 ? 0 : 1;
        return;
    }
    if (j < 33) {
        *N = 128;
        *K = j == 31 ? 9 * 256 : 9 * 128;
        *kind = 1;
        return;
    }
    *N = j == 33 ? 4 * (E + D) : (c.desc + 1) * 256;
    *K = j == 33 ? E + D : 4 * (E + D);
}

// f32 vectors: encoder block i: 4 i + {norm1.w, norm1.b, norm2.w, norm2.b}; enc_norm; decoder (side s, block i): 8 (s dec_depth + i)
// + {norm1, norm2, norm_y, norm3} x {w, b}; dec_norm; head hd: head.4 weight [4][128], bias [4]; rotary inv_freq [16]
static int du_v_enc(const DuCfg&, int i, int j) { return 4 * i + j; }
static int du_v_encn(const DuCfg& c) { return 4 * c.enc_depth; }
static int du_v_dec(const DuCfg& c, int s, int i, int j) { return du_v_encn(c) + 2 + 8 * (s * c.dec_depth + i) + j; }
static int du_v_decn(const DuCfg& c) { return du_v_encn(c) + 2 + 16 * c.dec_depth; }
static int du_v_head(const DuCfg& c, int hd) { return du_v_decn(c) + 2 + 2 * hd; }
static int du_v_invf(const DuCfg& c) { return du_v_decn(c) + 6; }
static int du_nvec(const DuCfg& c) { return du_v_invf(c) + 1; }
static int du_vec_len(const DuCfg& c, int vi) {
    if (vi < du_v_encn(c) + 2) return c.E;
    if (vi < du_v_decn(c) + 2) return c.D;
    if (vi == du_v_invf(c)) return 16;
    return ((vi - du_v_decn(c) - 2) & 1) ? 4 : 512;
}

static bool du_cfg_ok(const DuCfg& c) {
    return c.E >= 64 && c.E <= 1024 && c.E % 64 == 0 && c.D >= 64 && c.D <= 1024 && c.D % 64 == 0 && c.enc_depth >= 1 && c.dec_depth >= 4 &&
           c.dec_depth % 4 == 0 && c.desc >= 0 && c.desc <= 63 && (c.desc == 0 || (c.E + c.D) % 32 == 0);
}

struct DuLayout {
    std::vector<size_t> b, wh, wl, ws, vec;
    std::vector<size_t> rs;  // GEMM layers: row sums sum_k W[n][k] of the packed matrix (folded LayerNorms, GemmP.ln_rowsum)
    size_t trailer;          // 64 words at the very end: magic, format, total floats, the five configuration integers (round 4)
    size_t total;
};
static DuLayout du_layout(const DuCfg& c) {
    DuLayout l;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t r = off;
        off += align_up(n, 64);
        return r;
    };
    const int nl = du_nlayers(c);
    l.b.resize(nl);
    l.wh.resize(nl);
    l.wl.resize(nl);
    l.ws.resize(nl);
    for (int i = 0; i < nl; ++i) {
        int N, K, kind;
        du_shape(c, i, &N, &K, &kind);
        const size_t rows = kind == 1 ? (size_t)N : (size_t)((N + 31) / 32 * 32);
        l.b[i] = take((size_t)(N + 3) / 4 * 4 + 64);  // conv3x3_split_kernel reads whole 64-channel bias groups
        l.wh[i] = take(rows * K / 2);
        l.wl[i] = take(rows * K / 2);
        l.ws[i] = take(64);
    }
    const int nv = du_nvec(c);
    l.vec.resize(nv);
    for (int i = 0; i < nv; ++i) l.vec[i] = take(du_vec_len(c, i));
    l.rs.resize(nl);
    for (int i = 0; i < nl; ++i) {  // (behind everything else: the offsets above are those of rounds 2 and 3)
        int N, K, kind;
        du_shape(c, i, &N, &K, &kind);
        l.rs[i] = kind == 0 ? take((size_t)N) : (size_t)-1;
    }
    l.trailer = take(64);
    l.total = off;
    return l;
}

// Format of the packed buffer.  The CONTENT changed in round 3 without the size telling (LayerNorm gamma / beta folded into the
// consuming matrices by the packer, row sums appended): a blob packed for an older layout, or packed from raw upstream weights by a
// caller that skipped the fold, ran and returned plausible but wrong point maps.  Since round 4 the buffer ends in a trailer
// (magic, DU_FORMAT, total size, configuration) written by imcui_hip_dust3r_pack_weights, the forward entry points take the size
// of the caller's buffer and reject anything but this library's, and imcui_hip_dust3r_check_packed verifies the trailer of a HOST copy.
// 4 = norms folded by the caller of pack_weights (backend.dust3r_matrices), row sums behind the vectors, trailer.
#define DU_MAGIC 0x494D4455u /* 'IMDU' */
#define DU_FORMAT 4u

static DuCfg du_cfg(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim) {
    return DuCfg{enc_dim, enc_depth, dec_dim, dec_depth, desc_dim};
}

extern "C" size_t imcui_hip_dust3r_packed_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    return du_cfg_ok(c) ? du_layout(c).total : 0;
}
extern "C" int imcui_hip_dust3r_num_layers(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    return du_cfg_ok(c) ? du_nlayers(c) : 0;
}
extern "C" int imcui_hip_dust3r_num_vectors(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    return du_cfg_ok(c) ? du_nvec(c) : 0;
}
extern "C" int imcui_hip_dust3r_layer_shape(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i, int* N, int* K) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || i < 0 || i >= du_nlayers(c) || !N || !K) return IMCUI_ERR_ARG;
    int kind;
    du_shape(c, i, N, K, &kind);
    return IMCUI_OK;
}
extern "C" int imcui_hip_dust3r_vector_len(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || i < 0 || i >= du_nvec(c)) return 0;
    return du_vec_len(c, i);
}

// float offsets of layer i inside the packed buffer (inspection / tests): bias [N], the two f16 planes, the 2^-e scale; kind = 0
// fragment-major GEMM planes ([ceil(N/32)][K/16][2][32][8] halves), 1 = planes of the 3x3 patch-staging convolution kernel
extern "C" int imcui_hip_dust3r_layer_offsets(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i, size_t* bias,
                                              size_t* plane_hi, size_t* plane_lo, size_t* scale, int* kind) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || i < 0 || i >= du_nlayers(c) || !bias || !plane_hi || !plane_lo || !scale || !kind) return IMCUI_ERR_ARG;
    const DuLayout l = du_layout(c);
    int N, K;
    du_shape(c, i, &N, &K, kind);
    *bias = l.b[i];
    *plane_hi = l.wh[i];
    *plane_lo = l.wl[i];
    *scale = l.ws[i];
    return IMCUI_OK;
}

// w[i]: [N][K] f32 (convolutions in the implicit-GEMM order [Cout][tap][Cin], transposed convolutions as [(dy, dx, cout)][cin]),
// b[i]: [N] or null (zero), vec[i]: the f32 vectors in the order above.  The layers are split on a few host threads.
extern "C" int imcui_hip_dust3r_pack_weights(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* const* w,
                                             const float* const* b, const float* const* vec, float* packed) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !w || !b || !vec || !packed) return IMCUI_ERR_ARG;
    const DuLayout l = du_layout(c);
    const int nl = du_nlayers(c), nv = du_nvec(c);
    for (int i = 0; i < nl; ++i)
        if (!w[i]) return IMCUI_ERR_ARG;
    for (int i = 0; i < nv; ++i)
        if (!vec[i]) return IMCUI_ERR_ARG;
    memset(packed, 0, l.total * sizeof(float));
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 1;
    if (nthr > 64) nthr = 64;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthr; ++t)
        pool.emplace_back([&, t]() {
            for (int i = (int)t; i < nl; i += (int)nthr) {
                int N, K, kind;
                du_shape(c, i, &N, &K, &kind);
                if (b[i]) memcpy(packed + l.b[i], b[i], (size_t)N * sizeof(float));
                unsigned short* hi = reinterpret_cast<unsigned short*>(packed + l.wh[i]);
                unsigned short* lo = reinterpret_cast<unsigned short*>(packed + l.wl[i]);
                packed[l.ws[i]] = kind == 1 ? pack_conv3x3_split_from_gemm(w[i], N, K / 9, hi, lo) : split_weights_frag_host(w[i], N, K, hi, lo);
                if (kind == 0)
                    for (int n = 0; n < N; ++n) {
                        double acc = 0.0;
                        for (int k = 0; k < K; ++k) acc += (double)w[i][(size_t)n * K + k];
                        packed[l.rs[i] + n] = (float)acc;
                    }
            }
        });
    for (auto& th : pool) th.join();
    // one launch can serve both decoder sides (GemmP.wsel): the scale of the dec_blocks2 layer sits right behind its dec_blocks twin
    for (int i = 0; i < c.dec_depth; ++i)
        for (int j = 0; j < DU_DEC_J; ++j) packed[l.ws[du_l_dec(c, 0, i, j)] + 1] = packed[l.ws[du_l_dec(c, 1, i, j)]];
    for (int i = 0; i < nv; ++i) memcpy(packed + l.vec[i], vec[i], (size_t)du_vec_len(c, i) * sizeof(float));
    {
        const unsigned long long tot = (unsigned long long)l.total;
        const unsigned t[9] = {DU_MAGIC, DU_FORMAT, (unsigned)(tot & 0xffffffffu), (unsigned)(tot >> 32), (unsigned)c.E, (unsigned)c.enc_depth, (unsigned)c.D, (unsigned)c.dec_depth, (unsigned)c.desc};
        memcpy(packed + l.trailer, t, sizeof t);
    }
    return IMCUI_OK;
}

extern "C" int imcui_hip_dust3r_format_version(void) { return (int)DU_FORMAT; }

// 0 when `packed` (a HOST copy of `packed_floats` floats) is a buffer this library's imcui_hip_dust3r_pack_weights wrote for this
// configuration; IMCUI_ERR_ARG otherwise (wrong size = another layout; missing / foreign trailer = not packed by this format).
extern "C" int imcui_hip_dust3r_check_packed(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed, size_t packed_floats) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !packed) return IMCUI_ERR_ARG;
    const DuLayout l = du_layout(c);
    if (packed_floats != l.total) return IMCUI_ERR_ARG;
    unsigned t[9];
    memcpy(t, packed + l.trailer, sizeof t);
    const unsigned long long tot = (unsigned long long)l.total;
    const unsigned want[9] = {DU_MAGIC, DU_FORMAT, (unsigned)(tot & 0xffffffffu), (unsigned)(tot >> 32), (unsigned)c.E, (unsigned)c.enc_depth, (unsigned)c.D, (unsigned)c.dec_depth, (unsigned)c.desc};
    return memcmp(t, want, sizeof t) == 0 ? IMCUI_OK : IMCUI_ERR_ARG;
}

// ------------------------------------------------------------------ workspace
constexpr bool DU_LN_EPILOGUE_DEFAULT = false;  // until measured
constexpr int DU_MAX_CLASSES = 4;  // distinct image sizes of one call (the wrapper's symmetrised pair has at most two)
struct DuWs {
    float *A0, *x, *xn, *qkv, *qp, *kp, *vp, *kc, *vc, *att, *hid, *fenc, *g, *y, *qc;
    float *tok0, *hook[3], *ta, *tb, *tm, *rn[4], *s0, *s1, *s2, *s3, *pa, *pb, *hd0, *hd1, *hd2, *lfh, *lfo;
    float *rcos, *rsin;  // RoPE2D tables [class][R][32] of the token grids (the fused q / k / v projection epilogue reads them)
    float* stats;        // (mean, rstd) per token row of the stage that runs (folded LayerNorms)
    int *cnt, *smap, *wsel, *wsel_rev;
    int* geo;  // images of several sizes: per-sequence token counts / grid widths / table rows, head gather maps (du_forward_impl)
    size_t geo_ints;
    size_t total;
    bool ok;
};
static int du_R_tokens(size_t T) { return (int)((T + 127) / 128 * 128); }
static int du_R(int H, int W) { return du_R_tokens((size_t)(H / 16) * (W / 16)); }
// T = tokens of the LARGEST image of the call: every sequence is padded to R rows, the head buffers hold P maps of T cells
static DuWs du_carve(void* ws, size_t bytes, const DuCfg& c, int NI, int P, size_t T) {
    WsAlloc a(ws, bytes);
    DuWs w;
    const size_t E = c.E, D = c.D;
    const size_t R = du_R_tokens(T);
    const size_t me = (size_t)NI * R, md = (size_t)2 * P * R;
    const size_t mx = me * E > md * D ? me * E : md * D;  // largest token buffer of either stage
    w.A0 = a.get<float>(me * 768);
    w.x = a.get<float>(me * E);
    w.xn = a.get<float>(mx);
    w.qkv = a.get<float>(3 * mx);
    w.qp = a.get<float>(mx);  // two f16 planes each
    w.kp = a.get<float>(mx);
    w.vp = a.get<float>(mx);
    w.kc = a.get<float>(md * D);
    w.vc = a.get<float>(md * D);
    w.att = a.get<float>(mx);
    w.hid = a.get<float>(4 * mx);
    w.fenc = a.get<float>(me * E);
    w.g = a.get<float>(me * D);
    w.y = a.get<float>(md * D);
    w.qc = a.get<float>(md * D);
    const size_t bt = (size_t)2 * P * T;  // dense token rows of both views
    w.tok0 = a.get<float>(bt * E);
    for (int k = 0; k < 3; ++k) w.hook[k] = a.get<float>(bt * D);
    const size_t pt = (size_t)P * T + 64;  // one view: P maps of T cells (+ slack: an odd grid rounds its 1/32 level up)
    w.ta = a.get<float>(pt * 768);
    w.tb = a.get<float>(pt * 1536);
    w.tm = a.get<float>(pt * 16 * 96 > pt * 768 ? pt * 16 * 96 : pt * 768);
    w.rn[0] = a.get<float>(pt * 16 * 256);
    w.rn[1] = a.get<float>(pt * 4 * 256);
    w.rn[2] = a.get<float>(pt * 256);
    w.rn[3] = a.get<float>(pt * 256);  // (h / 2)(w / 2) <= T / 4 rounded up: T is plenty
    w.s0 = a.get<float>(pt * 16 * 256);
    w.s1 = a.get<float>(pt * 16 * 256);
    w.s2 = a.get<float>(pt * 16 * 256);
    w.s3 = a.get<float>(pt * 64 * 256);
    w.pa = a.get<float>(pt * 64 * 256);
    w.pb = a.get<float>(pt * 64 * 256);
    w.hd0 = a.get<float>(pt * 64 * 128);
    w.hd1 = a.get<float>(pt * 256 * 128);
    w.hd2 = a.get<float>(pt * 256 * 128);
    w.lfh = c.desc > 0 ? a.get<float>(pt * 4 * (E + D)) : nullptr;  // MASt3R: hidden layer and output of head_local_features
    w.lfo = c.desc > 0 ? a.get<float>(pt * (size_t)(c.desc + 1) * 256) : nullptr;
    w.stats = a.get<float>(2 * (me > md ? me : md));
    w.rcos = a.get<float>((size_t)DU_MAX_CLASSES * R * 32);
    w.rsin = a.get<float>((size_t)DU_MAX_CLASSES * R * 32);
    const size_t nseq = (size_t)(NI > 2 * P ? NI : 2 * P) + 64;
    w.cnt = a.get<int>(nseq);
    w.smap = a.get<int>((size_t)2 * P + 64);
    w.wsel = a.get<int>((size_t)P + 64);
    w.wsel_rev = a.get<int>((size_t)P + 64);
    w.geo_ints = 8 * nseq;
    w.geo = a.get<int>(w.geo_ints);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}
static bool du_size_ok(int H, int W) { return H >= 32 && W >= 32 && H % 16 == 0 && W % 16 == 0 && H <= 4096 && W <= 4096; }
static bool du_dims_ok(int NI, int P, int H, int W) { return NI > 0 && P > 0 && du_size_ok(H, W); }

extern "C" size_t imcui_hip_dust3r_workspace_bytes(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, int P, int H, int W) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !du_dims_ok(NI, P, H, W)) return 0;
    return du_carve(nullptr, 0, c, NI, P, (size_t)(H / 16) * (W / 16)).total;
}

// floats the optional dump of the forward holds: (enc_depth + 2) x [NI R E] (patch embedding, every block, enc_norm), (dec_depth + 2) x
// [2P R D] (embedded streams, every block, dec_norm), then per view: layer_rn 0..3, path 4..1, the 128-channel full-resolution
// feature map, the raw [P H W 4] regression
extern "C" size_t imcui_hip_dust3r_dump_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, int P, int H, int W) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !du_dims_ok(NI, P, H, W)) return 0;
    const size_t R = du_R(H, W), h = H / 16, w = W / 16, p = P;
    size_t n = (size_t)(c.enc_depth + 2) * NI * R * c.E + (size_t)(c.dec_depth + 2) * 2 * P * R * c.D;
    const size_t per_view = p * 256 * (16 * h * w + 4 * h * w + h * w + ((h + 1) / 2) * ((w + 1) / 2))   // layer_rn
                            + p * 256 * (h * w + 4 * h * w + 16 * h * w + 64 * h * w)          // paths 4, 3, 2, 1
                            + p * (size_t)H * W * 128 + p * (size_t)H * W * 4;
    return n + 2 * per_view;
}

// ------------------------------------------------------------------ geometry of a call
// Upstream's `inference` accepts views of different sizes (it then encodes the two views of a pair separately and runs one pair
// per batch, dust3r/inference.py; the reference's drivers resize each image on its own, so two photos of different aspect ratio
// DO arrive at two sizes).  Here every image belongs to one of K <= DU_MAX_CLASSES size classes; all sequences are padded to the
// R rows of the largest grid and carry their own token count (GEMM tile skipping, attention masks), grid width and RoPE2D table;
// the DPT heads run once per (view, class) on the densely gathered maps of that class.
struct DuGeom {
    int K = 1;
    int H[DU_MAX_CLASSES], W[DU_MAX_CLASSES];
    size_t Tmax = 0;
    const int* img_class = nullptr;    // host [NI]            (K > 1)
    const int* pairs_host = nullptr;   // host [P][2]          (K > 1)
    const size_t* img_off = nullptr;   // host [NI]: float offset of image i in `images` (K > 1)
    const size_t* map_pix = nullptr;   // host [2][P]: pixel offset of the output map of (view, pair) (K > 1)
};

// images in [0,1]; pairs (device) [P][2]: directed pair p = (view 1 image, view 2 image), indices clamped to [0, NI).
static int du_forward_impl(imcui_hip_t* h, const DuCfg& c, const float* packed, const float* images, int NI, const DuGeom& geo, const int* pairs, int P,
                           int arith, float* pts3d, float* conf, float* desc, float* desc_conf, float* dump, size_t dump_floats, void* ws,
                           size_t ws_bytes, hipStream_t stream) {
    const int single = arith;  // GEMMs, 3x3 convolutions and attention with ONE f16 product per element pair
    if (((size_t)(c.E / 64) * NI) % 8 != 0 || ((size_t)(c.D / 64) * 2 * P) % 8 != 0)
        return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: heads x sequences must be a multiple of 8 (encoder %d x %d, decoder %d x %d)", c.E / 64, NI, c.D / 64, 2 * P);
    DuWs w = du_carve(ws, ws_bytes, c, NI, P, geo.Tmax);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "dust3r: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const DuLayout l = du_layout(c);
    const float* Pk = packed;
    const bool mixed = geo.K > 1;
    const int E = c.E, D = c.D, R = du_R_tokens(geo.Tmax);
    const int H = geo.H[0], W = geo.W[0], hg = H / 16, wg = W / 16, T = hg * wg;  // THE grid of a one-size call (mixed: class 0 only)
    const dim3 blk(256);
    int rc;
#define DURUN(x)                       \
    do {                               \
        rc = (x);                      \
        if (rc != IMCUI_OK) return rc; \
    } while (0)
    auto blocks = [](long n) { return dim3((unsigned)(n < 256 ? 1 : (n + 255) / 256 > 65536 * 16 ? 65536 * 16 : (n + 255) / 256)); };
    size_t dump_off = 0;
    bool dump_ok = true;
    auto dump_take = [&](size_t n) -> float* {
        if (!dump) return nullptr;
        if (dump_off + n > dump_floats) {
            dump_ok = false;
            return nullptr;
        }
        float* r = dump + dump_off;
        dump_off += n;
        return r;
    };
    auto dump_copy = [&](const float* src, size_t n) {
        float* d = dump_take(n);
        if (d) hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyDeviceToDevice, stream);
    };
    const float* vecs = Pk;
    auto V = [&](int vi) { return vecs + l.vec[vi]; };

    // per-sequence geometry of the stage that is running (encoder: one sequence per image, decoder: one per stream): token counts
    // for the GEMMs' tile skipping and the attention masks; with several image sizes also the first RoPE2D table row, the token
    // count and the grid width of every sequence (nullptr in a one-size call: the scalars T / wg / table row 0 hold for all)
    const int *cur_cnt = w.cnt, *cur_row0 = nullptr, *cur_T = nullptr, *cur_wg = nullptr;
    // folded LayerNorm: (mean, rstd) rows of the raw input of the NEXT linear layer(s), applied in their epilogue; nullptr: plain layers
    const float* cur_ln = nullptr;

    // linear layer li on `nseq` sequences of R rows (cnt live): C = act(A W^T + b + resid)
    auto lin = [&](int li, const float* A, float* C, int nseq, const float* resid, int act) -> int {
        int N, K, kind;
        du_shape(c, li, &N, &K, &kind);
        GemmP g;
        g.epi = EPI_CONV;
        g.N = N;
        g.K = K;
        g.ldw = K;
        g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[li]);
        g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[li]);
        g.wscale = Pk + l.ws[li];
        g.bias = Pk + l.b[li];
        g.A = A;
        g.lda = K;
        g.C = C;
        g.ldc = N;
        g.resid = resid;
        g.ldr = N;
        g.act = act;
        g.single = single;
        g.M = nseq * R;
        g.cnt = cur_cnt;
        g.rows_per_seq = R;
        if (cur_ln) {
            g.ln_stats = cur_ln;
            g.ln_rowsum = Pk + l.rs[li];
        }
        return gemm_launch(h, g, stream);
    };
    // decoder layer j of block i on ALL 2P streams in one launch: sequences [0, P) use the weights of `dec_blocks`, [P, 2P) those of
    // `dec_blocks2` (`swap`: the other way round -- keys / values are projected by the block of the side that will READ them).
    // The weight set is chosen per sequence PAIR (GemmP.wsel), so this needs an even P; otherwise one launch per side.
    static const bool dec_two_launches = getenv("IMCUI_DUST3R_DEC_SPLIT") != nullptr;  // A/B switch
    const bool merged = (P % 2 == 0) && !dec_two_launches;
    auto lin2 = [&](int i, int j, const float* A, long lda_rows, float* C, long ldc_rows, const float* resid, int act, bool swap) -> int {
        const int L0 = du_l_dec(c, 0, i, j), L1 = du_l_dec(c, 1, i, j);
        if (!merged) {
            const int* cnt_all = cur_cnt;
            const float* ln_all = cur_ln;
            for (int s = 0; s < 2; ++s) {
                const int li = swap ? (s ? L0 : L1) : (s ? L1 : L0);
                cur_cnt = cnt_all + (size_t)s * P;
                cur_ln = ln_all ? ln_all + 2 * (size_t)s * P * R : nullptr;
                const int r = lin(li, A + (size_t)s * P * R * lda_rows, C + (size_t)s * P * R * ldc_rows, P,
                                  resid ? resid + (size_t)s * P * R * ldc_rows : nullptr, act);
                cur_cnt = cnt_all;
                cur_ln = ln_all;
                if (r != IMCUI_OK) return r;
            }
            return IMCUI_OK;
        }
        int N, K, kind;
        du_shape(c, L0, &N, &K, &kind);
        GemmP g;
        g.epi = EPI_CONV;
        g.N = N;
        g.K = K;
        g.ldw = K;
        g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[L0]);
        g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[L0]);
        g.wscale = Pk + l.ws[L0];
        g.bias = Pk + l.b[L0];
        g.wsel = swap ? w.wsel_rev : w.wsel;
        g.w_stride = (long)(l.wh[L1] - l.wh[L0]) * 2;  // halves between the planes of the two sides (same for hi and lo)
        g.b_stride = (long)(l.b[L1] - l.b[L0]);
        g.A = A;
        g.lda = K;
        g.C = C;
        g.ldc = N;
        g.resid = resid;
        g.ldr = N;
        g.act = act;
        g.single = single;
        g.M = 2 * P * R;
        g.cnt = cur_cnt;
        g.rows_per_seq = R;
        if (cur_ln) {
            g.ln_stats = cur_ln;
            g.ln_rowsum = Pk + l.rs[L0];
            g.ln_stride = (long)(l.rs[L1] - l.rs[L0]);
        }
        return gemm_launch(h, g, stream);
    };
    // the same on dense rows (DPT head)
    auto lin_dense = [&](int li, const float* A, float* C, long rows) -> int {
        int N, K, kind;
        du_shape(c, li, &N, &K, &kind);
        GemmP g;
        g.epi = EPI_CONV;
        g.N = N;
        g.K = K;
        g.ldw = K;
        g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[li]);
        g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[li]);
        g.wscale = Pk + l.ws[li];
        g.bias = Pk + l.b[li];
        g.A = A;
        g.lda = K;
        g.C = C;
        g.ldc = N;
        g.single = single;
        g.M = (int)rows;
        return gemm_launch(h, g, stream);
    };
    auto layernorm = [&](const float* x, int vi, float* out, long rows, int C) {
        hipLaunchKernelGGL(du_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), blk, 0, stream, x, V(vi), V(vi + 1), out, rows, C, 1e-6f);
    };
    // (x - mean) * rstd: the LayerNorms in front of the q / k / v and fc1 layers, whose gamma / beta are folded into those layers at pack time
    auto normalise = [&](const float* x, float* out, long rows, int C) {
        hipLaunchKernelGGL(du_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), blk, 0, stream, x, (const float*)nullptr, (const float*)nullptr, out,
                           rows, C, 1e-6f);
    };
    // The normalisation itself can move into the epilogue of the consuming GEMM: LN(x) W^T + b = rstd (x W'^T - mean s) + b' with s = the
    // row sums of W' (packed behind the layers).  The K loop then runs on the RAW residual stream and the LayerNorm pass shrinks to the
    // statistics (half the traffic).  Parity arithmetic with the weights-in-registers kernel only; IMCUI_DUST3R_LN_EPILOGUE=0 keeps the
    // normalise-then-multiply path.
    static const bool ln_epi_env = [] {
        const char* e = getenv("IMCUI_DUST3R_LN_EPILOGUE");
        return e ? atoi(e) != 0 : DU_LN_EPILOGUE_DEFAULT;
    }();
    bool fold = false;
    if (ln_epi_env && !single) {  // would the layers take the kernel that knows the folded epilogue?
        GemmP probe;
        probe.epi = EPI_CONV;
        probe.N = 4 * c.E;
        probe.K = c.E;
        probe.lda = c.E;
        probe.ldc = 4 * c.E;
        probe.Wh = reinterpret_cast<const unsigned short*>(Pk);
        probe.Wl = probe.Wh;
        probe.ln_stats = Pk;
        probe.ln_rowsum = Pk;
        fold = gemm_wreg_ok(h, probe);
    }
    // the input of the next linear layer(s): with `fold` the raw rows + their statistics, else the normalised copy in w.xn
    auto ln_input = [&](const float* x, long rows, int C) -> const float* {
        if (fold) {
            hipLaunchKernelGGL(du_rowstats_kernel, dim3((unsigned)((rows + 3) / 4)), blk, 0, stream, x, w.stats, rows, C, 1e-6f);
            cur_ln = w.stats;
            return x;
        }
        normalise(x, w.xn, rows, C);
        return w.xn;
    };
    const float q_alpha = 0.125f * 1.44269504088896340736f;  // 1 / sqrt(64) and log2(e): the attention kernel works in base 2
    // q / k planes of `nseq` sequences read from src[:, col0 : col0 + C]
    auto rope_split = [&](const float* src, long ld, int col0, int C, int nseq, float* planes, int nseq_planes, int seq_out0, float alpha) {
        const long n = (long)nseq * R * (C / 64) * 4;
        hipLaunchKernelGGL(du_rope_split_kernel, blocks(n), blk, 0, stream, src, ld, col0, C / 64, T, R, wg, V(du_v_invf(c)), alpha, 1,
                           reinterpret_cast<unsigned short*>(planes), (size_t)nseq_planes * C * R, seq_out0, n, cur_T, cur_wg);
    };
    auto vt_split = [&](const float* src, long ld, int col0, int C, int nseq, float* planes, int nseq_planes, int seq_out0) {
        hipLaunchKernelGGL(du_vt_split_kernel, dim3((unsigned)(nseq * (C / 64) * (R / 64))), blk, 0, stream, src, ld, col0, C / 64, T, R,
                           reinterpret_cast<unsigned short*>(planes), (size_t)nseq_planes * C * R, seq_out0, cur_T);
    };
    // q / k / v projection with the attention operand planes written by the GEMM's own epilogue (EPI_QKV_VIT, gemm_wreg.hip:
    // RoPE2D from the tables, q scaled, f16 hi / lo planes, V transposed) instead of an f32 [rows][N] round trip through
    // du_rope_split_kernel / du_vt_split_kernel (5.5 % of a step in round 2).  li = layer (encoder) or (i, j) of a decoder block
    // (both sides in one launch); role0 / nblk: which of q, k, v the N = nblk x C output columns are.  IMCUI_DUST3R_QKV_UNFUSED=1
    // keeps the round-2 path (also taken when the weights-in-registers kernel is switched off or the single-product mode is on).
    static const bool qkv_unfused_env = getenv("IMCUI_DUST3R_QKV_UNFUSED") != nullptr;
    auto proj_planes = [&](int li0, int li1, bool swap, const float* A, int C, int nseq, int role0, int nblk, float* qpl, float* kpl, float* vpl,
                           bool* done) -> int {
        int N, K, kind;
        du_shape(c, li0, &N, &K, &kind);
        GemmP g;
        g.epi = EPI_QKV_VIT;
        g.N = N;
        g.K = K;
        g.ldw = K;
        g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[li0]);
        g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[li0]);
        g.wscale = Pk + l.ws[li0];
        g.bias = Pk + l.b[li0];
        if (li1 >= 0) {  // decoder: weight set per sequence pair, as lin2
            g.wsel = swap ? w.wsel_rev : w.wsel;
            g.w_stride = (long)(l.wh[li1] - l.wh[li0]) * 2;
            g.b_stride = (long)(l.b[li1] - l.b[li0]);
        }
        g.A = A;
        g.lda = K;
        g.M = nseq * R;
        g.cnt = cur_cnt;
        g.rows_per_seq = R;
        g.heads = C / 64;
        g.role0 = role0;
        g.split_out = 1;
        g.v_transposed = 1;
        g.plane_halves = (size_t)nseq * C * R;
        g.Q = qpl;
        g.Kt = kpl;
        g.V = vpl;
        g.rope_cos = w.rcos;
        g.rope_sin = w.rsin;
        g.rope_seq_row0 = cur_row0;
        g.alpha = q_alpha;
        if (cur_ln) {
            g.ln_stats = cur_ln;
            g.ln_rowsum = Pk + l.rs[li0];
            if (li1 >= 0) g.ln_stride = (long)(l.rs[li1] - l.rs[li0]);
        }
        *done = li1 != -2 && !qkv_unfused_env && !single && N == nblk * C && gemm_wreg_ok(h, g);
        if (!*done) return IMCUI_OK;
        return gemm_launch(h, g, stream);
    };
    auto attend = [&](const float* q, const float* k, const float* v, float* out, int nseq, int C, int cross) -> int {
        AttnP a;
        a.Q = q;
        a.K = k;
        a.V = v;
        a.O = out;
        a.cnt = cur_cnt;
        a.nseq = nseq;
        a.heads = C / 64;
        a.rows_per_seq = R;
        a.cross = cross;
        a.log2_domain = 1;
        a.single = single;
        return attention_launch(h, a, stream);
    };

    // ---- tables
    // several sizes: groups of the DPT heads.  Group (v, k) = the streams of view v whose image is of class k, in pair order;
    // their maps sit densely (T_k rows each) behind each other in tok0 / hook[], groups in (v, k) order.
    struct Group {
        int v, k, n, first;  // first: index of the group's first entry in hmap / imap
        size_t row0;         // first dense token row
    };
    std::vector<Group> groups;
    std::vector<int> hmap_host;  // stream of every dense map
    const int *cnt_enc = w.cnt, *cnt_dec = w.cnt, *row0_enc = nullptr, *row0_dec = nullptr, *T_enc = nullptr, *T_dec = nullptr, *wg_enc = nullptr,
              *wg_dec = nullptr, *hmap_dev = nullptr, *imap_dev = nullptr;
    {
        const int ncnt = NI > 2 * P ? NI : 2 * P;
        // stream s < P: view 1 of pair s, stream P + s: view 2
        hipLaunchKernelGGL(du_smap_kernel, dim3((unsigned)cdiv(2 * P, 256)), blk, 0, stream, pairs, w.smap, P, NI);
        hipLaunchKernelGGL(du_wsel_kernel, dim3((unsigned)cdiv(P, 256)), blk, 0, stream, w.wsel, w.wsel_rev, P);
        if (!mixed) {
            hipLaunchKernelGGL(du_fill_int_kernel, dim3((unsigned)cdiv(ncnt, 256)), blk, 0, stream, w.cnt, T, ncnt);
            hipLaunchKernelGGL(du_rope_table_kernel, dim3((unsigned)cdiv(R * 32, 256)), blk, 0, stream, V(du_v_invf(c)), T, R, wg, w.rcos, w.rsin);
        } else {
            const size_t ns = (size_t)ncnt + 64;  // slot length of the carve
            std::vector<int> g(8 * ns, 0);
            int *h_cnt_enc = g.data(), *h_cnt_dec = g.data() + ns, *h_row_enc = g.data() + 2 * ns, *h_row_dec = g.data() + 3 * ns,
                *h_wg_enc = g.data() + 4 * ns, *h_wg_dec = g.data() + 5 * ns, *h_hmap = g.data() + 6 * ns, *h_imap = g.data() + 7 * ns;
            auto Tk = [&](int k) { return (geo.H[k] / 16) * (geo.W[k] / 16); };
            for (int i = 0; i < NI; ++i) {
                const int k = geo.img_class[i];
                h_cnt_enc[i] = Tk(k);
                h_row_enc[i] = k * R;
                h_wg_enc[i] = geo.W[k] / 16;
            }
            for (int s = 0; s < 2 * P; ++s) {
                const int img = geo.pairs_host[2 * (s % P) + s / P], k = geo.img_class[img];
                h_cnt_dec[s] = Tk(k);
                h_row_dec[s] = k * R;
                h_wg_dec[s] = geo.W[k] / 16;
            }
            size_t row = 0;
            int n_maps = 0;
            for (int v = 0; v < 2; ++v)
                for (int k = 0; k < geo.K; ++k) {
                    Group gr{v, k, 0, n_maps, row};
                    for (int p = 0; p < P; ++p) {
                        const int img = geo.pairs_host[2 * p + v];
                        if (geo.img_class[img] != k) continue;
                        h_hmap[n_maps] = v * P + p;
                        h_imap[n_maps] = img;
                        hmap_host.push_back(v * P + p);
                        ++n_maps;
                        ++gr.n;
                    }
                    row += (size_t)gr.n * Tk(k);
                    if (gr.n) groups.push_back(gr);
                }
            // pageable host memory: the copy is staged before the call returns; the synchronisation makes that independent of the runtime
            if (hipMemcpyAsync(w.geo, g.data(), g.size() * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                return imcui_set_err(h, IMCUI_ERR_HIP, "dust3r: upload of the sequence tables failed");
            cnt_enc = w.geo;
            cnt_dec = w.geo + ns;
            row0_enc = w.geo + 2 * ns;
            row0_dec = w.geo + 3 * ns;
            wg_enc = w.geo + 4 * ns;
            wg_dec = w.geo + 5 * ns;
            hmap_dev = w.geo + 6 * ns;
            imap_dev = w.geo + 7 * ns;
            T_enc = cnt_enc;
            T_dec = cnt_dec;
            for (int k = 0; k < geo.K; ++k)
                hipLaunchKernelGGL(du_rope_table_kernel, dim3((unsigned)cdiv(R * 32, 256)), blk, 0, stream, V(du_v_invf(c)), Tk(k), R, geo.W[k] / 16,
                                   w.rcos + (size_t)k * R * 32, w.rsin + (size_t)k * R * 32);
        }
    }

    // ---- encoder: every image once
    cur_cnt = cnt_enc;
    cur_row0 = row0_enc;
    cur_T = T_enc;
    cur_wg = wg_enc;
    const long me = (long)NI * R;
    {
        if (!mixed) {
            const long n4 = me * 192;
            hipLaunchKernelGGL(du_patchify_kernel, blocks(n4), blk, 0, stream, images, w.A0, H, W, T, R, n4);
        } else {
            const long n4 = (long)R * 192;
            for (int i = 0; i < NI; ++i) {
                const int k = geo.img_class[i];
                hipLaunchKernelGGL(du_patchify_kernel, blocks(n4), blk, 0, stream, images + geo.img_off[i], w.A0 + (size_t)i * R * 768, geo.H[k], geo.W[k],
                                   (geo.H[k] / 16) * (geo.W[k] / 16), R, n4);
            }
        }
        IMCUI_CHECK_LAUNCH(h);
        DURUN(lin(0, w.A0, w.x, NI, nullptr, 0));
        dump_copy(w.x, (size_t)me * E);
    }
    for (int i = 0; i < c.enc_depth; ++i) {
        const float* xin = ln_input(w.x, me, E);  // norm1 (affine part inside attn.qkv)
        bool fused;
        DURUN(proj_planes(du_l_enc(c, i, 0), -1, false, xin, E, NI, 0, 3, w.qp, w.kp, w.vp, &fused));
        if (!fused) {
            DURUN(lin(du_l_enc(c, i, 0), xin, w.qkv, NI, nullptr, 0));
            rope_split(w.qkv, 3 * E, 0, E, NI, w.qp, NI, 0, q_alpha);
            rope_split(w.qkv, 3 * E, E, E, NI, w.kp, NI, 0, 1.0f);
            vt_split(w.qkv, 3 * E, 2 * E, E, NI, w.vp, NI, 0);
        }
        cur_ln = nullptr;
        IMCUI_CHECK_LAUNCH(h);
        DURUN(attend(w.qp, w.kp, w.vp, w.att, NI, E, 0));
        DURUN(lin(du_l_enc(c, i, 1), w.att, w.x, NI, w.x, 0));
        xin = ln_input(w.x, me, E);  // norm2 (inside mlp.fc1)
        DURUN(lin(du_l_enc(c, i, 2), xin, w.hid, NI, nullptr, 3));
        cur_ln = nullptr;
        DURUN(lin(du_l_enc(c, i, 3), w.hid, w.x, NI, w.x, 0));
        dump_copy(w.x, (size_t)me * E);
    }
    layernorm(w.x, du_v_encn(c), w.fenc, me, E);
    dump_copy(w.fenc, (size_t)me * E);

    // ---- decoder
    const long md = (long)2 * P * R, ms = (long)P * R;  // rows of all streams / of one side
    DURUN(lin(du_l_demb(c), w.fenc, w.g, NI, nullptr, 0));
    cur_cnt = cnt_dec;
    cur_row0 = row0_dec;
    cur_T = T_dec;
    cur_wg = wg_dec;
    {
        const long n4 = md * (D / 4);
        hipLaunchKernelGGL(du_gather_seq_kernel, blocks(n4), blk, 0, stream, w.g, w.smap, w.y, R, R, D / 4, n4);
        IMCUI_CHECK_LAUNCH(h);
        dump_copy(w.y, (size_t)md * D);
    }
    const int hooks[3] = {c.dec_depth * 2 / 4, c.dec_depth * 3 / 4, c.dec_depth};
    // the live tokens of every stream, densely: src [2P][R][C] -> dst; one size: [2P][T][C] in stream order; several: group by group
    auto gather_dense = [&](const float* src, const int* map_one_size, const int* map_groups, float* dst, int C) {
        if (!mixed) {
            const long n4 = (long)2 * P * T * (C / 4);
            hipLaunchKernelGGL(du_gather_seq_kernel, blocks(n4), blk, 0, stream, src, map_one_size, dst, R, T, C / 4, n4);
            return;
        }
        for (const Group& gr : groups) {
            const int tk = (geo.H[gr.k] / 16) * (geo.W[gr.k] / 16);
            const long n4 = (long)gr.n * tk * (C / 4);
            hipLaunchKernelGGL(du_gather_seq_kernel, blocks(n4), blk, 0, stream, src, map_groups + gr.first, dst + gr.row0 * C, R, tk, C / 4, n4);
        }
    };
    auto save_hook = [&](const float* src, int k) { gather_dense(src, nullptr, hmap_dev, w.hook[k], D); };
    for (int i = 0; i < c.dec_depth; ++i) {
        // keys / values of the cross attention: side s reads the other side's tokens as they are BEFORE this block, normalised by
        // its own norm_y and projected by its own projk / projv -> the rows of side o carry the weights of side 1 - o.  With the
        // affine parts folded into the projections, norm_y (cross keys / values) and norm1 (self attention) are ONE normalisation
        // of the block's input, for both sides at once
        const float* yin = ln_input(w.y, md, D);
        bool fused;
        DURUN(proj_planes(du_l_dec(c, 0, i, 3), merged ? du_l_dec(c, 1, i, 3) : -2, true, yin, D, 2 * P, 1, 2, nullptr, w.kc, w.vc, &fused));
        if (!fused) {
            DURUN(lin2(i, 3, yin, D, w.qkv, 2 * D, nullptr, 0, true));
            rope_split(w.qkv, 2 * D, 0, D, 2 * P, w.kc, 2 * P, 0, 1.0f);
            vt_split(w.qkv, 2 * D, D, D, 2 * P, w.vc, 2 * P, 0);
        }
        // self attention (w.xn still holds the normalised input of the block)
        DURUN(proj_planes(du_l_dec(c, 0, i, 0), merged ? du_l_dec(c, 1, i, 0) : -2, false, yin, D, 2 * P, 0, 3, w.qp, w.kp, w.vp, &fused));
        if (!fused) {
            DURUN(lin2(i, 0, yin, D, w.qkv, 3 * D, nullptr, 0, false));
            rope_split(w.qkv, 3 * D, 0, D, 2 * P, w.qp, 2 * P, 0, q_alpha);
            rope_split(w.qkv, 3 * D, D, D, 2 * P, w.kp, 2 * P, 0, 1.0f);
            vt_split(w.qkv, 3 * D, 2 * D, D, 2 * P, w.vp, 2 * P, 0);
        }
        cur_ln = nullptr;
        IMCUI_CHECK_LAUNCH(h);
        DURUN(attend(w.qp, w.kp, w.vp, w.att, 2 * P, D, 0));
        DURUN(lin2(i, 1, w.att, D, w.y, D, w.y, 0, false));
        // cross attention
        yin = ln_input(w.y, md, D);  // norm2 (inside cross_attn.projq)
        DURUN(proj_planes(du_l_dec(c, 0, i, 2), merged ? du_l_dec(c, 1, i, 2) : -2, false, yin, D, 2 * P, 0, 1, w.qp, nullptr, nullptr, &fused));
        if (!fused) {
            DURUN(lin2(i, 2, yin, D, w.qc, D, nullptr, 0, false));
            rope_split(w.qc, D, 0, D, 2 * P, w.qp, 2 * P, 0, q_alpha);
        }
        cur_ln = nullptr;
        IMCUI_CHECK_LAUNCH(h);
        DURUN(attend(w.qp, w.kc, w.vc, w.att, 2 * P, D, 2));
        DURUN(lin2(i, 4, w.att, D, w.y, D, w.y, 0, false));
        // MLP
        yin = ln_input(w.y, md, D);  // norm3 (inside mlp.fc1)
        DURUN(lin2(i, 5, yin, D, w.hid, 4 * D, nullptr, 3, false));
        cur_ln = nullptr;
        DURUN(lin2(i, 6, w.hid, 4 * D, w.y, D, w.y, 0, false));
        dump_copy(w.y, (size_t)md * D);
        for (int k = 0; k < 2; ++k)
            if (i + 1 == hooks[k]) save_hook(w.y, k);
    }
    layernorm(w.y, du_v_decn(c), w.xn, md, D);
    dump_copy(w.xn, (size_t)md * D);
    save_hook(w.xn, 2);
    gather_dense(w.fenc, w.smap, imap_dev, w.tok0, E);
    IMCUI_CHECK_LAUNCH(h);

    // ---- DPT heads: downstream_head{v + 1} on Pn maps of Hh x Wh pixels whose dense tokens start at row `row0` of tok0 / hook[].
    // One size: Pn = P, the outputs of view v are one block.  Several sizes: one call per group, every map regressed into its own
    // slot of the (view, pair)-ordered output.
    auto run_head = [&](int v, int Pn, int Hh, int Wh, size_t row0, const int* streams) -> int {
        const int hg = Hh / 16, wg = Wh / 16, T = hg * wg;
        const long pt = (long)Pn * T;
        const int L0 = du_l_head(c, v);
        // act: output activation code, + 4 = ReLU on the input while it is staged; resid (+ resid2): maps added before the activation
        auto conv3 = [&](int li, const float* in, float* out, int hh, int ww, int act, const float* resid, const float* resid2 = nullptr) -> int {
            int N, K, kind;
            du_shape(c, li, &N, &K, &kind);
            return conv3x3_split_launch(h, in, reinterpret_cast<const unsigned short*>(Pk + l.wh[li]), reinterpret_cast<const unsigned short*>(Pk + l.wl[li]),
                                        Pk + l.ws[li], Pk + l.b[li], out, Pn, hh, ww, K / 9, N, act, 0, stream, resid, 0, 0, single, resid2);
        };
        auto shuffle = [&](const float* src, float* dst, int s, int C) {
            const long n4 = pt * s * s * (C / 4);
            hipLaunchKernelGGL(du_pixel_shuffle_kernel, blocks(n4), blk, 0, stream, src, dst, hg, wg, s, C / 4, n4);
        };
        const float* tok0 = w.tok0 + row0 * E;
        const float* hk[3] = {w.hook[0] + row0 * D, w.hook[1] + row0 * D, w.hook[2] + row0 * D};
        // reassemble: 1/4, 1/8, 1/16, 1/32 (an odd token grid rounds the 1/32 level up: 3x3 stride 2 with padding 1; the x2 of the first
        // fusion block is then cropped back to the token grid, upstream's `[:, :, :layers[2].shape[2], :layers[2].shape[3]]`)
        const int h3 = (hg + 1) / 2, w3 = (wg + 1) / 2;
        DURUN(lin_dense(L0 + 0, tok0, w.ta, pt));
        DURUN(lin_dense(L0 + 1, w.ta, w.tb, pt));
        shuffle(w.tb, w.tm, 4, 96);
        DURUN(conv3(L0 + 7, w.tm, w.rn[0], 4 * hg, 4 * wg, 0, nullptr));
        DURUN(lin_dense(L0 + 2, hk[0], w.ta, pt));
        DURUN(lin_dense(L0 + 3, w.ta, w.tb, pt));
        shuffle(w.tb, w.tm, 2, 192);
        DURUN(conv3(L0 + 8, w.tm, w.rn[1], 2 * hg, 2 * wg, 0, nullptr));
        DURUN(lin_dense(L0 + 4, hk[1], w.ta, pt));
        DURUN(conv3(L0 + 9, w.ta, w.rn[2], hg, wg, 0, nullptr));
        DURUN(lin_dense(L0 + 5, hk[2], w.ta, pt));
        {
            int N, K, kind;
            du_shape(c, L0 + 6, &N, &K, &kind);
            GemmP g;  // 3x3 stride 2, padding 1: the implicit-im2col GEMM
            g.epi = EPI_CONV;
            g.N = N;
            g.K = K;
            g.ldw = K;
            g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[L0 + 6]);
            g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[L0 + 6]);
            g.wscale = Pk + l.ws[L0 + 6];
            g.bias = Pk + l.b[L0 + 6];
            g.A = w.ta;
            g.conv_k = 3;
            g.conv_stride = 2;
            g.conv_pad = 1;
            g.conv_hin = hg;
            g.conv_win = wg;
            g.conv_hout = h3;
            g.conv_wout = w3;
            g.conv_cin = 768;
            g.M = Pn * h3 * w3;
            g.C = w.tm;
            g.ldc = N;
            g.single = single;
            DURUN(gemm_launch(h, g, stream));
        }
        DURUN(conv3(L0 + 10, w.tm, w.rn[3], h3, w3, 0, nullptr));
        const int rh[4] = {4 * hg, 2 * hg, hg, h3}, rw[4] = {4 * wg, 2 * wg, wg, w3};
        if (!mixed)
            for (int k = 0; k < 4; ++k) dump_copy(w.rn[k], (size_t)Pn * rh[k] * rw[k] * 256);
        // fusion: refinenet 4, 3, 2, 1.  FeatureFusionBlock(path, skip) = out_conv(up2(RCU2(path + RCU1(skip)))), RCU(x) = x + conv2(relu(conv1(
        // relu(x)))).  The ReLU in front of a unit's first convolution is applied while that convolution stages its input, the sum
        // path + RCU1(skip) leaves the second convolution of RCU1 as a double residual (conv + skip + path), and the 1x1 out_conv runs
        // BEFORE the bilinear x2 (both are linear and the interpolation weights sum to 1, so the bias commutes; a quarter of the rows).
        // IMCUI_DUST3R_HEAD_UNFUSED=1 keeps the element-wise kernels and upstream's order (A/B and parity of the commuted form).
        static const bool head_unfused = getenv("IMCUI_DUST3R_HEAD_UNFUSED") != nullptr;
        const float* path = nullptr;
        for (int q = 0; q < 4; ++q) {
            const int lv = 3 - q, hh = rh[lv], ww = rw[lv];
            const long n4 = (long)Pn * hh * ww * 64;
            const int Lq = L0 + 11 + 5 * q;
            float* out = (q & 1) ? w.pb : w.pa;
            if (head_unfused) {
                const float* xres;
                if (q == 0) {
                    hipLaunchKernelGGL(du_relu_kernel, blocks(n4), blk, 0, stream, w.rn[3], w.s1, n4);
                    xres = w.rn[3];
                } else {
                    hipLaunchKernelGGL(du_relu_kernel, blocks(n4), blk, 0, stream, w.rn[lv], w.s0, n4);
                    DURUN(conv3(Lq + 0, w.s0, w.s1, hh, ww, 1, nullptr));
                    DURUN(conv3(Lq + 1, w.s1, w.s0, hh, ww, 0, w.rn[lv]));
                    hipLaunchKernelGGL(du_add_relu_kernel, blocks(n4), blk, 0, stream, path, w.s0, w.s2, w.s1, n4);
                    xres = w.s2;
                }
                DURUN(conv3(Lq + 2, w.s1, w.s0, hh, ww, 1, nullptr));
                DURUN(conv3(Lq + 3, w.s0, w.s1, hh, ww, 0, xres));
                hipLaunchKernelGGL(du_upsample2_kernel, blocks(4 * n4), blk, 0, stream, w.s1, w.s3, hh, ww, 64, 4 * n4);
                DURUN(lin_dense(Lq + 4, w.s3, out, (long)Pn * 4 * hh * ww));
            } else {
                const float* x2 = w.rn[3];  // input of the second unit: rn[3] itself (refinenet4 has no skip), else path + RCU1(skip)
                if (q > 0) {
                    DURUN(conv3(Lq + 0, w.rn[lv], w.s1, hh, ww, 1 + 4, nullptr));
                    DURUN(conv3(Lq + 1, w.s1, w.s2, hh, ww, 0, w.rn[lv], path));
                    x2 = w.s2;
                }
                DURUN(conv3(Lq + 2, x2, w.s0, hh, ww, 1 + 4, nullptr));
                DURUN(conv3(Lq + 3, w.s0, w.s1, hh, ww, 0, x2));
                DURUN(lin_dense(Lq + 4, w.s1, w.s3, (long)Pn * hh * ww));
                hipLaunchKernelGGL(du_upsample2_kernel, blocks(4 * n4), blk, 0, stream, w.s3, out, hh, ww, 64, 4 * n4);
            }
            path = out;
            int ph = 2 * hh, pw = 2 * ww;
            if (q == 0 && (ph != hg || pw != wg)) {
                const long c4 = (long)Pn * hg * wg * 64;
                hipLaunchKernelGGL(du_crop_kernel, blocks(c4), blk, 0, stream, out, w.hd0, ph, pw, hg, wg, 64, c4);
                path = w.hd0;
                ph = hg;
                pw = wg;
            }
            if (!mixed) dump_copy(path, (size_t)Pn * ph * pw * 256);
        }
        // head: 3x3 256 -> 128 at 1/2, x2, 3x3 128 -> 128 + ReLU, 1x1 128 -> 4 + post-processing
        DURUN(conv3(L0 + 31, path, w.hd0, Hh / 2, Wh / 2, 0, nullptr));
        {
            const long n4 = (long)Pn * Hh * Wh * 32;
            hipLaunchKernelGGL(du_upsample2_kernel, blocks(n4), blk, 0, stream, w.hd0, w.hd1, Hh / 2, Wh / 2, 32, n4);
        }
        const long npix = (long)Pn * Hh * Wh, mpix = (long)Hh * Wh;
        // one size: head.4 (1x1, 128 -> 4) and the point-map post-processing run in the epilogue of head.2's convolution; the 128-channel
        // full-resolution map is only written for the parity dump.  IMCUI_DUST3R_REGRESS_UNFUSED=1 keeps du_regress_kernel (A/B).
        static const bool regress_unfused = getenv("IMCUI_DUST3R_REGRESS_UNFUSED") != nullptr;
        if (!mixed && !regress_unfused) {
            ConvHead hd;
            hd.w = V(du_v_head(c, v));
            hd.b = V(du_v_head(c, v) + 1);
            hd.pts = pts3d + (size_t)v * npix * 3;
            hd.conf = conf + (size_t)v * npix;
            float* feat = dump_take((size_t)npix * 128);  // straight into the dump buffer
            hd.raw = dump_take((size_t)npix * 4);
            int N, K, kind;
            du_shape(c, L0 + 32, &N, &K, &kind);
            DURUN(conv3x3_split_launch(h, w.hd1, reinterpret_cast<const unsigned short*>(Pk + l.wh[L0 + 32]), reinterpret_cast<const unsigned short*>(Pk + l.wl[L0 + 32]),
                                       Pk + l.ws[L0 + 32], Pk + l.b[L0 + 32], feat, Pn, Hh, Wh, K / 9, N, 1, 0, stream, nullptr, 0, 0, single, nullptr, &hd));
        } else if (!mixed) {
            DURUN(conv3(L0 + 32, w.hd1, w.hd2, Hh, Wh, 1, nullptr));
            dump_copy(w.hd2, (size_t)npix * 128);
            float* raw = dump_take((size_t)npix * 4);
            hipLaunchKernelGGL(du_regress_kernel, blocks(npix * 32), blk, 0, stream, w.hd2, V(du_v_head(c, v)), V(du_v_head(c, v) + 1),
                               pts3d + (size_t)v * npix * 3, conf + (size_t)v * npix, raw, npix);
        } else {
            DURUN(conv3(L0 + 32, w.hd1, w.hd2, Hh, Wh, 1, nullptr));
            for (int j = 0; j < Pn; ++j) {
                const size_t o = geo.map_pix[streams[j]];
                hipLaunchKernelGGL(du_regress_kernel, blocks(mpix * 32), blk, 0, stream, w.hd2 + (size_t)j * mpix * 128, V(du_v_head(c, v)),
                                   V(du_v_head(c, v) + 1), pts3d + o * 3, conf + o, (float*)nullptr, mpix);
            }
        }
        IMCUI_CHECK_LAUNCH(h);
        if (c.desc > 0) {
            // MASt3R (mast3r.py:41-66): local features = MLP([encoder tokens | last decoder tokens]) per token, (desc + 1) x 16 x 16 values
            // = the descriptors and their confidence logit of the token's 16x16 pixels (pixel shuffle), descriptors L2-normalised
            int N, K, kind;
            du_shape(c, L0 + 33, &N, &K, &kind);
            GemmP g;
            g.epi = EPI_CONV;
            g.N = N;
            g.K = K;
            g.ldw = K;
            g.Wh = reinterpret_cast<const unsigned short*>(Pk + l.wh[L0 + 33]);
            g.Wl = reinterpret_cast<const unsigned short*>(Pk + l.wl[L0 + 33]);
            g.wscale = Pk + l.ws[L0 + 33];
            g.bias = Pk + l.b[L0 + 33];
            g.A = tok0;
            g.lda = E;
            g.A2 = hk[2];
            g.lda2 = D;
            g.K1 = E;
            g.C = w.lfh;
            g.ldc = N;
            g.act = 3;
            g.single = single;
            g.M = (int)pt;
            DURUN(gemm_launch(h, g, stream));
            DURUN(lin_dense(L0 + 34, w.lfh, w.lfo, pt));
            if (!mixed) {
                hipLaunchKernelGGL(du_desc_kernel, blocks(npix), blk, 0, stream, w.lfo, desc + (size_t)v * npix * c.desc, desc_conf + (size_t)v * npix, Hh,
                                   Wh, c.desc, npix);
            } else {
                for (int j = 0; j < Pn; ++j) {
                    const size_t o = geo.map_pix[streams[j]];
                    hipLaunchKernelGGL(du_desc_kernel, blocks(mpix), blk, 0, stream, w.lfo + (size_t)j * T * (c.desc + 1) * 256, desc + o * c.desc,
                                       desc_conf + o, Hh, Wh, c.desc, mpix);
                }
            }
            IMCUI_CHECK_LAUNCH(h);
        }
        return IMCUI_OK;
    };
    if (!mixed) {
        for (int v = 0; v < 2; ++v) DURUN(run_head(v, P, H, W, (size_t)v * P * T, nullptr));
    } else {
        for (const Group& gr : groups) DURUN(run_head(gr.v, gr.n, geo.H[gr.k], geo.W[gr.k], gr.row0, hmap_host.data() + gr.first));
    }
    if (!dump_ok) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: dump buffer too small (%zu floats)", dump_floats);
    return IMCUI_OK;
#undef DURUN
}

// ------------------------------------------------------------------ forward
// images [NI,3,H,W] in [0,1]; pairs (device) [P][2]: directed pair p = (view 1 image, view 2 image), indices clamped to [0, NI).
// arith: 0 = the 3 x f16 split products of the library's default mode (fp32-grade results), 1 = ONE f16 product per element pair in the
// GEMMs and convolutions (f32 accumulate; 11-bit operands: the class of the bf16 run BASELINE's configs[4] names).  Outputs, view-major:
// pts3d [2][P][H][W][3] (view 1 in its own frame, view 2 in view 1's frame = upstream's `pts3d_in_other_view`), conf [2][P][H][W];
// MASt3R (desc_dim > 0): desc [2][P][H][W][desc_dim] (unit norm), desc_conf [2][P][H][W].
static int du_check_common(imcui_hip_t* h, const DuCfg& c, int arith, const void* packed, size_t packed_floats, const void* images, const void* pairs,
                           const void* pts3d, const void* conf, const void* desc, const void* desc_conf) {
    if (!du_cfg_ok(c)) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: unsupported configuration (dims multiples of 64 up to 1024, dec_depth a multiple of 4)");
    if (h->precision != 1) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "dust3r: only the 3 x f16 split mode (precision 1) is implemented");
    if (arith != 0 && arith != 1) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: arith = %d (0: 3 x f16 split products, 1: one f16 product)", arith);
    if (!packed || !images || !pairs || !pts3d || !conf || (c.desc > 0 && (!desc || !desc_conf))) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: null argument");
    const size_t want = du_layout(c).total;
    if (packed_floats != want)
        return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: the packed buffer holds %zu floats, format %u of this library (imcui_hip_version %d) needs %zu: a blob of an "
                             "older layout (or of another configuration) -- pack the state dict again", packed_floats, DU_FORMAT, imcui_hip_version(), want);
    return IMCUI_OK;
}

extern "C" int imcui_hip_dust3r_forward(imcui_hip_t* h, int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed,
                                        size_t packed_floats, const float* images, int NI, int H, int W, const int* pairs, int P, int arith, float* pts3d,
                                        float* conf, float* desc, float* desc_conf, float* dump, size_t dump_floats, void* ws, size_t ws_bytes,
                                        void* stream_) {
    if (!h) return IMCUI_ERR_ARG;
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    const int rc = du_check_common(h, c, arith, packed, packed_floats, images, pairs, pts3d, conf, desc, desc_conf);
    if (rc != IMCUI_OK) return rc;
    if (!du_dims_ok(NI, P, H, W)) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: %d images of %dx%d, %d pairs: sizes must be multiples of 16", NI, W, H, P);
    DuGeom geo;
    geo.K = 1;
    geo.H[0] = H;
    geo.W[0] = W;
    geo.Tmax = (size_t)(H / 16) * (W / 16);
    return du_forward_impl(h, c, packed, images, NI, geo, pairs, P, arith, pts3d, conf, desc, desc_conf, dump, dump_floats, ws, ws_bytes,
                           (hipStream_t)stream_);
}

// ---- images of several sizes (include/imcui_hip.h) ------------------------------------------------------------------
static bool du_sizes_ok(int NI, const int* sizes, int P) {
    if (NI <= 0 || P <= 0 || !sizes) return false;
    for (int i = 0; i < NI; ++i)
        if (!du_size_ok(sizes[2 * i], sizes[2 * i + 1])) return false;
    return true;
}
static size_t du_max_tokens(int NI, const int* sizes) {
    size_t t = 0;
    for (int i = 0; i < NI; ++i) {
        const size_t ti = (size_t)(sizes[2 * i] / 16) * (sizes[2 * i + 1] / 16);
        t = ti > t ? ti : t;
    }
    return t;
}

extern "C" size_t imcui_hip_dust3r_workspace_bytes_sizes(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI,
                                                         const int* sizes, int P) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !du_sizes_ok(NI, sizes, P)) return 0;
    return du_carve(nullptr, 0, c, NI, P, du_max_tokens(NI, sizes)).total;
}

extern "C" size_t imcui_hip_dust3r_token_dump_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, const int* sizes,
                                                     int P) {
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    if (!du_cfg_ok(c) || !du_sizes_ok(NI, sizes, P)) return 0;
    const size_t R = du_R_tokens(du_max_tokens(NI, sizes));
    return (size_t)(c.enc_depth + 2) * NI * R * c.E + (size_t)(c.dec_depth + 2) * 2 * P * R * c.D;
}

extern "C" int imcui_hip_dust3r_forward_sizes(imcui_hip_t* h, int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed,
                                              size_t packed_floats, const float* images, int NI, const int* sizes, const int* pairs_host, const int* pairs, int P, int arith,
                                              float* pts3d, float* conf, float* desc, float* desc_conf, size_t* map_pixel_offsets, float* dump,
                                              size_t dump_floats, void* ws, size_t ws_bytes, void* stream_) {
    if (!h) return IMCUI_ERR_ARG;
    const DuCfg c = du_cfg(enc_dim, enc_depth, dec_dim, dec_depth, desc_dim);
    const int rc = du_check_common(h, c, arith, packed, packed_floats, images, pairs, pts3d, conf, desc, desc_conf);
    if (rc != IMCUI_OK) return rc;
    if (!pairs_host) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: null argument");
    if (!du_sizes_ok(NI, sizes, P)) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: %d images, %d pairs: every size must be a multiple of 16 (32 .. 4096)", NI, P);
    for (int i = 0; i < 2 * P; ++i)
        if (pairs_host[i] < 0 || pairs_host[i] >= NI) return imcui_set_err(h, IMCUI_ERR_ARG, "dust3r: pair table refers to image %d of %d", pairs_host[i], NI);
    DuGeom geo;
    geo.K = 0;
    std::vector<int> cls(NI);
    std::vector<size_t> off(NI), pix(2 * (size_t)P);
    size_t o = 0;
    for (int i = 0; i < NI; ++i) {
        const int Hi = sizes[2 * i], Wi = sizes[2 * i + 1];
        int k = 0;
        while (k < geo.K && (geo.H[k] != Hi || geo.W[k] != Wi)) ++k;
        if (k == geo.K) {
            if (geo.K == DU_MAX_CLASSES)
                return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "dust3r: more than %d distinct image sizes in one call", DU_MAX_CLASSES);
            geo.H[k] = Hi;
            geo.W[k] = Wi;
            ++geo.K;
        }
        cls[i] = k;
        off[i] = o;
        o += (size_t)3 * Hi * Wi;
    }
    o = 0;
    for (int v = 0; v < 2; ++v)
        for (int p = 0; p < P; ++p) {
            const int img = pairs_host[2 * p + v];
            pix[(size_t)v * P + p] = o;
            o += (size_t)sizes[2 * img] * sizes[2 * img + 1];
        }
    if (map_pixel_offsets) {
        for (size_t i = 0; i < pix.size(); ++i) map_pixel_offsets[i] = pix[i];
        map_pixel_offsets[pix.size()] = o;
    }
    geo.Tmax = du_max_tokens(NI, sizes);
    geo.img_class = cls.data();
    geo.pairs_host = pairs_host;
    geo.img_off = off.data();
    geo.map_pix = pix.data();
    // a call whose images all have one size takes the one-size path (same layouts: the offsets above are then the dense ones)
    return du_forward_impl(h, c, packed, images, NI, geo, pairs, P, arith, pts3d, conf, desc, desc_conf, dump, dump_floats, ws, ws_bytes,
                           (hipStream_t)stream_);
}
