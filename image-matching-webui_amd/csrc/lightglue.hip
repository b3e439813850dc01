// LightGlue forward on MI355X for batches of ragged key-point sets.  Replaces the
// `self.net(input)` call of imcui/hloc/matchers/lightglue.py:75 (SURVEY.md section 8a rows a8-a11,
// Appendix A.2; upstream CPU-path semantics: pruning threshold -1, shared cross similarity).
//
// Data layout.  A batch of B pairs is 2B "sequences" (image 0 / image 1 of each pair) padded to
// R = roundup(ncap, 128) rows: token row = seq * R + i.  Per-sequence valid counts live on the
// device (`cnt`), per-pair `active` flags implement early stopping, and point pruning physically
// compacts rows, so every kernel is launched for the worst case and skips dead tiles without any
// host synchronisation.  All contractions run on the matrix cores (gemm.h: exact f32 or 3 x f16 split).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "attention.h"
#include "ffn.h"
#include "gemm.h"
#include "imcui_hip.h"
#include "lightglue_assign.h"
#include "simred.h"

#define LG_LAYERS 9
#define LG_DIM 256
#define LG_HEADS 4

// ------------------------------------------------------------------ packed weights
struct LgSplit {
    size_t h, l, s;  // f16 hi / lo planes (float offsets) and 2^-e scale(s)
};
struct LgLayerOff {
    size_t wqkv, bqkv, wo, bo, w1s, b1s, gs, bs, w2s, b2s;
    size_t wx, bx, wto, bto, w1c, b1c, gc, bc, w2c, b2c;
    LgSplit sqkv, s1s, s2s, sx, s1c, s2c;  // out_proj / to_out are folded into ffn.0: no planes of their own
    LgSplit s2sp, s2cp;                    // ffn.3 planes with the K axis in the fused FFN kernel's order (ffn_permute_k)
};
struct LgLayout {
    size_t wr;
    LgLayerOff L[LG_LAYERS];
    size_t wfinal, bfinal, wmatch, bmatch, wtoken, btoken;
    LgSplit sfinal;  // [9] planes, scale array [9]
    size_t wr4, winp, binp;  // variants: posenc.Wr with scale / orientation columns [32][4], input_proj [256][<=256] + bias
    size_t total;
};

static LgLayout lg_layout() {
    LgLayout l;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t r = off;
        off += align_up(n, 64);
        return r;
    };
    l.wr = take(64);
    for (int i = 0; i < LG_LAYERS; ++i) {
        LgLayerOff& o = l.L[i];
        o.wqkv = take(768 * 256);
        o.bqkv = take(768);
        o.wo = take(256 * 256);
        o.bo = take(256);
        o.w1s = take(512 * 512);
        o.b1s = take(512);
        o.gs = take(512);
        o.bs = take(512);
        o.w2s = take(256 * 512);
        o.b2s = take(256);
        o.wx = take(512 * 256);
        o.bx = take(512);
        o.wto = take(256 * 256);
        o.bto = take(256);
        o.w1c = take(512 * 512);
        o.b1c = take(512);
        o.gc = take(512);
        o.bc = take(512);
        o.w2c = take(256 * 512);
        o.b2c = take(256);
    }
    l.wfinal = take((size_t)LG_LAYERS * 256 * 256);
    l.bfinal = take((size_t)LG_LAYERS * 256);
    l.wmatch = take((size_t)LG_LAYERS * 256);
    l.bmatch = take(64);
    l.wtoken = take((size_t)(LG_LAYERS - 1) * 256);
    l.btoken = take(64);
    auto take_split = [&](size_t n) {
        LgSplit sp;
        sp.h = take(n / 2);
        sp.l = take(n / 2);
        sp.s = take(64);
        return sp;
    };
    for (int i = 0; i < LG_LAYERS; ++i) {
        LgLayerOff& o = l.L[i];
        o.sqkv = take_split(768 * 256);
        o.s1s = take_split(512 * 512);
        o.s2s = take_split(256 * 512);
        o.sx = take_split(512 * 256);
        o.s1c = take_split(512 * 512);
        o.s2c = take_split(256 * 512);
        o.s2sp = take_split(256 * 512);
        o.s2cp = take_split(256 * 512);
    }
    l.sfinal = take_split((size_t)LG_LAYERS * 256 * 256);
    l.wr4 = take(128);
    l.winp = take(256 * 256);
    l.binp = take(256);
    l.total = off;
    return l;
}

// tensor order of the host-side packer = upstream state-dict keys
enum { T_PER_LAYER = 26 };
static const char* const LG_LAYER_KEYS[T_PER_LAYER] = {
    "transformers.%d.self_attn.Wqkv.weight",    "transformers.%d.self_attn.Wqkv.bias",
    "transformers.%d.self_attn.out_proj.weight", "transformers.%d.self_attn.out_proj.bias",
    "transformers.%d.self_attn.ffn.0.weight",    "transformers.%d.self_attn.ffn.0.bias",
    "transformers.%d.self_attn.ffn.1.weight",    "transformers.%d.self_attn.ffn.1.bias",
    "transformers.%d.self_attn.ffn.3.weight",    "transformers.%d.self_attn.ffn.3.bias",
    "transformers.%d.cross_attn.to_qk.weight",   "transformers.%d.cross_attn.to_qk.bias",
    "transformers.%d.cross_attn.to_v.weight",    "transformers.%d.cross_attn.to_v.bias",
    "transformers.%d.cross_attn.to_out.weight",  "transformers.%d.cross_attn.to_out.bias",
    "transformers.%d.cross_attn.ffn.0.weight",   "transformers.%d.cross_attn.ffn.0.bias",
    "transformers.%d.cross_attn.ffn.1.weight",   "transformers.%d.cross_attn.ffn.1.bias",
    "transformers.%d.cross_attn.ffn.3.weight",   "transformers.%d.cross_attn.ffn.3.bias",
    "log_assignment.%d.matchability.weight",     "log_assignment.%d.matchability.bias",
    "log_assignment.%d.final_proj.weight",       "log_assignment.%d.final_proj.bias",
};
// index: 0 = posenc.Wr.weight ; 1 + 26*i + j = layer tensor j ; 1 + 26*9 + 2*i (+1) = token_confidence.i.token.0.{weight,bias}
static const int LG_NUM_TENSORS = 1 + T_PER_LAYER * LG_LAYERS + 2 * (LG_LAYERS - 1);

extern "C" size_t imcui_hip_lightglue_packed_floats(void) { return lg_layout().total; }
extern "C" int imcui_hip_lightglue_num_tensors(void) { return LG_NUM_TENSORS; }
extern "C" const char* imcui_hip_lightglue_tensor_name(int i) {
    static thread_local char buf[96];
    if (i == 0) return "posenc.Wr.weight";
    if (i < 1 + T_PER_LAYER * LG_LAYERS) {
        snprintf(buf, sizeof buf, LG_LAYER_KEYS[(i - 1) % T_PER_LAYER], (i - 1) / T_PER_LAYER);
        return buf;
    }
    const int j = i - 1 - T_PER_LAYER * LG_LAYERS;
    if (j < 2 * (LG_LAYERS - 1)) {
        snprintf(buf, sizeof buf, (j & 1) ? "token_confidence.%d.token.0.bias" : "token_confidence.%d.token.0.weight", j >> 1);
        return buf;
    }
    return nullptr;
}

extern "C" int imcui_hip_lightglue_pack_weights(const float* const* t, float* packed) {
    if (!t || !packed) return IMCUI_ERR_ARG;
    for (int i = 0; i < LG_NUM_TENSORS; ++i)
        if (!t[i]) return IMCUI_ERR_ARG;
    const LgLayout l = lg_layout();
    memset(packed, 0, l.total * sizeof(float));
    memcpy(packed + l.wr, t[0], 64 * sizeof(float));
    auto cp = [&](size_t dst, const float* src, size_t n) { memcpy(packed + dst, src, n * sizeof(float)); };
    for (int i = 0; i < LG_LAYERS; ++i) {
        const float* const* s = t + 1 + T_PER_LAYER * i;
        const LgLayerOff& o = l.L[i];
        // Wqkv: upstream output feature f = h*192 + d*3 + t  ->  packed row t*256 + h*64 + d
        for (int f = 0; f < 768; ++f) {
            const int hh = f / 192, d = (f % 192) / 3, tt = f % 3;
            const int row = tt * 256 + hh * 64 + d;
            memcpy(packed + o.wqkv + (size_t)row * 256, s[0] + (size_t)f * 256, 256 * sizeof(float));
            packed[o.bqkv + row] = s[1][f];
        }
        cp(o.wo, s[2], 256 * 256);
        cp(o.bo, s[3], 256);
        cp(o.w1s, s[4], 512 * 512);
        cp(o.b1s, s[5], 512);
        cp(o.gs, s[6], 512);
        cp(o.bs, s[7], 512);
        cp(o.w2s, s[8], 256 * 512);
        cp(o.b2s, s[9], 256);
        cp(o.wx, s[10], 256 * 256);  // to_qk rows 0..255
        cp(o.bx, s[11], 256);
        cp(o.wx + 256 * 256, s[12], 256 * 256);  // to_v rows 256..511
        cp(o.bx + 256, s[13], 256);
        cp(o.wto, s[14], 256 * 256);
        cp(o.bto, s[15], 256);
        cp(o.w1c, s[16], 512 * 512);
        cp(o.b1c, s[17], 512);
        cp(o.gc, s[18], 512);
        cp(o.bc, s[19], 512);
        cp(o.w2c, s[20], 256 * 512);
        cp(o.b2c, s[21], 256);
        cp(l.wmatch + (size_t)i * 256, s[22], 256);
        packed[l.bmatch + i] = s[23][0];
        cp(l.wfinal + (size_t)i * 65536, s[24], 65536);
        cp(l.bfinal + (size_t)i * 256, s[25], 256);
    }
    const float* const* tk = t + 1 + T_PER_LAYER * LG_LAYERS;
    for (int i = 0; i < LG_LAYERS - 1; ++i) {
        cp(l.wtoken + (size_t)i * 256, tk[2 * i], 256);
        packed[l.btoken + i] = tk[2 * i + 1][0];
    }
    // out_proj / to_out are folded into ffn.0 (reference lightglue.py SelfBlock/CrossBlock.forward:
    // ffn(cat[x, out_proj(ctx)])):  cat[x, Wo ctx + bo] W1^T + b1 = cat[x, ctx] [W1a | W1b Wo]^T + (b1 + W1b bo).
    // The fold is done once here in double, so the forward reads the attention context directly and runs
    // two GEMMs fewer per layer.  The wo/bo/wto/bto slots keep the unfused tensors for inspection.
    auto fold = [&](size_t w1, size_t b1, size_t wo, size_t bo) {
        std::vector<double> row(256);
        for (int n = 0; n < 512; ++n) {
            float* w1b = packed + w1 + (size_t)n * 512 + 256;
            double bacc = packed[b1 + n];
            for (int k = 0; k < 256; ++k) row[k] = 0.0;
            for (int j = 0; j < 256; ++j) {
                const double c = w1b[j];
                const float* wor = packed + wo + (size_t)j * 256;
                for (int k = 0; k < 256; ++k) row[k] += c * (double)wor[k];
                bacc += c * (double)packed[bo + j];
            }
            for (int k = 0; k < 256; ++k) w1b[k] = (float)row[k];
            packed[b1 + n] = (float)bacc;
        }
    };
    for (int i = 0; i < LG_LAYERS; ++i) {
        const LgLayerOff& o = l.L[i];
        fold(o.w1s, o.b1s, o.wo, o.bo);
        fold(o.w1c, o.b1c, o.wto, o.bto);
    }
    // split-precision copies of every GEMM weight (taken from the packed f32 layout)
    auto sp = [&](const LgSplit& d, size_t src, int N, int K, int slot) {
        const size_t n = (size_t)N * K;
        packed[d.s + slot] = split_weights_frag_host(packed + src, N, K, reinterpret_cast<unsigned short*>(packed + d.h) + (size_t)slot * n,
                                                     reinterpret_cast<unsigned short*>(packed + d.l) + (size_t)slot * n);
    };
    float* perm = (float*)malloc((size_t)256 * 512 * sizeof(float));
    if (!perm) return IMCUI_ERR_ARG;
    for (int i = 0; i < LG_LAYERS; ++i) {
        const LgLayerOff& o = l.L[i];
        sp(o.sqkv, o.wqkv, 768, 256, 0);
        sp(o.s1s, o.w1s, 512, 512, 0);
        sp(o.s2s, o.w2s, 256, 512, 0);
        sp(o.sx, o.wx, 512, 256, 0);
        sp(o.s1c, o.w1c, 512, 512, 0);
        sp(o.s2c, o.w2c, 256, 512, 0);
        ffn_permute_k(packed + o.w2s, 256, 512, perm);
        packed[o.s2sp.s] = split_weights_frag_host(perm, 256, 512, reinterpret_cast<unsigned short*>(packed + o.s2sp.h), reinterpret_cast<unsigned short*>(packed + o.s2sp.l));
        ffn_permute_k(packed + o.w2c, 256, 512, perm);
        packed[o.s2cp.s] = split_weights_frag_host(perm, 256, 512, reinterpret_cast<unsigned short*>(packed + o.s2cp.h), reinterpret_cast<unsigned short*>(packed + o.s2cp.l));
        sp(l.sfinal, l.wfinal + (size_t)i * 65536, 256, 256, i);
    }
    free(perm);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ workspace
struct LgWs {
    float *x, *xt, *ctx, *hbuf, *q, *k, *v, *cs, *sn, *cst, *snt, *conf, *mtch, *md, *ls;
    unsigned char* v6;  // attention variant 9: fp6 planes of V^T (attention_mx.hip)
    float* apart;       // key-split attention launches of small batches: per-chunk partials (attention.hip), nullptr for large batches
    float *rmax, *rls, *cmax, *cls, *max0, *ms0;
    float *rpm, *rps, *cpm, *cps;  // assignment partials of simred.hip: rows [B][column chunks][R], columns [B][R/128][R]
    int *rpj, *cpi;
    uint4 *pk0, *pk1;  // round 5: the matching descriptors as MFMA fragments (simred_pack)
    int *cntA, *cntB, *active, *ind, *indt, *pos, *prune, *m0, *m1, *valid0, *norig, *pflag;
    size_t total;
    bool ok;
};

static LgWs lg_carve(void* ws, size_t bytes, int B, int R) {
    WsAlloc a(ws, bytes);
    LgWs w;
    const size_t rows = (size_t)2 * B * R;
    w.x = a.get<float>(rows * 256);
    w.xt = a.get<float>(rows * 256);
    w.ctx = a.get<float>(rows * 256);
    w.hbuf = a.get<float>(rows * 512);
    w.q = a.get<float>(rows * 256);
    w.k = a.get<float>(rows * 256);
    w.v = a.get<float>(rows * 256);
    w.v6 = a.get<unsigned char>(attn_v6_bytes(2 * B, LG_HEADS, R));
    // (a grid of fewer than two workgroups per CU splits its keys over workgroups: at most 1024 workgroups covers every part with <= 512 CUs)
    const bool small_grid = (long)(R / 128) * LG_HEADS * 2 * B < 1024 && R > 512;
    w.apart = small_grid ? a.get<float>(attn_part_floats(2 * B, LG_HEADS, R)) : nullptr;
    w.cs = a.get<float>(rows * 32);
    w.sn = a.get<float>(rows * 32);
    w.cst = a.get<float>(rows * 32);
    w.snt = a.get<float>(rows * 32);
    w.conf = a.get<float>(rows);
    w.mtch = a.get<float>(rows);
    w.md = a.get<float>(rows * 256);
    w.ls = a.get<float>(rows);
    w.pk0 = a.get<uint4>((size_t)B * simred_packed_uint4(R, 256));
    w.pk1 = a.get<uint4>((size_t)B * simred_packed_uint4(R, 256));
    w.rmax = a.get<float>((size_t)B * R);
    w.rls = a.get<float>((size_t)B * R);
    w.cmax = a.get<float>((size_t)B * R);
    w.cls = a.get<float>((size_t)B * R);
    w.max0 = a.get<float>((size_t)B * R);
    w.ms0 = a.get<float>((size_t)B * R);
    w.rpm = a.get<float>((size_t)B * (R / 128) * R);
    w.rps = a.get<float>((size_t)B * (R / 128) * R);
    w.cpm = a.get<float>((size_t)B * (R / 128) * R);
    w.cps = a.get<float>((size_t)B * (R / 128) * R);
    w.rpj = a.get<int>((size_t)B * (R / 128) * R);
    w.cpi = a.get<int>((size_t)B * (R / 128) * R);
    w.cntA = a.get<int>(2 * B);
    w.cntB = a.get<int>(2 * B);
    w.active = a.get<int>(B);
    w.ind = a.get<int>(rows);
    w.indt = a.get<int>(rows);
    w.pos = a.get<int>(rows);
    w.prune = a.get<int>(rows);
    w.m0 = a.get<int>((size_t)B * R);
    w.m1 = a.get<int>((size_t)B * R);
    w.valid0 = a.get<int>((size_t)B * R);
    w.norig = a.get<int>(2 * B);
    w.pflag = a.get<int>(2 * B);
    w.total = a.off;
    w.ok = a.ok;
    return w;
}

// Variant inputs (upstream LightGlue `features` table: disk / aliked 128-d descriptors -> `input_proj` Linear(input_dim, 256);
// sift / doghardnet additionally `add_scale_ori`: posenc.Wr takes (x, y, scale, orientation)).  Call after
// imcui_hip_lightglue_pack_weights on the same host buffer.
extern "C" int imcui_hip_lightglue_pack_input(const float* wr, int wr_cols, const float* w_input_proj, const float* b_input_proj,
                                              int input_dim, float* packed) {
    if (!wr || !packed || (wr_cols != 2 && wr_cols != 4) || input_dim <= 0 || input_dim > 256 || input_dim % 32 != 0) return IMCUI_ERR_ARG;
    if (input_dim != 256 && (!w_input_proj || !b_input_proj)) return IMCUI_ERR_ARG;
    const LgLayout l = lg_layout();
    for (int i = 0; i < 32; ++i)
        for (int c = 0; c < 4; ++c) packed[l.wr4 + i * 4 + c] = c < wr_cols ? wr[i * wr_cols + c] : 0.0f;
    if (wr_cols == 2) memcpy(packed + l.wr, wr, 64 * sizeof(float));
    if (input_dim != 256) {
        memcpy(packed + l.winp, w_input_proj, (size_t)256 * input_dim * sizeof(float));
        memcpy(packed + l.binp, b_input_proj, 256 * sizeof(float));
    }
    return IMCUI_OK;
}

extern "C" int imcui_hip_lightglue_set_layer_dump(imcui_hip_t* h, float* dump, size_t floats) {
    if (!h) return IMCUI_ERR_ARG;
    h->lg_dump = dump;
    h->lg_dump_floats = dump ? floats : 0;
    return IMCUI_OK;
}

extern "C" size_t imcui_hip_lightglue_workspace_bytes(int B, int ncap) {
    const int R = (int)align_up((size_t)(ncap > 0 ? ncap : 1), 128);
    return lg_carve(nullptr, 0, B, R).total;
}

// ------------------------------------------------------------------ small kernels
// init: copy descriptors, positional encoding, index maps, output defaults (one wave per row)
__global__ __launch_bounds__(256) void lg_init_kernel(const float* __restrict__ kp0, const float* __restrict__ kp1,
                                                      const float* __restrict__ d0, const float* __restrict__ d1,
                                                      const int* __restrict__ n0, const int* __restrict__ n1, int ncap,
                                                      int R, float w0, float h0, float w1, float h1,
                                                      const float* __restrict__ wr, const float* __restrict__ wr4,
                                                      const float* __restrict__ sc0, const float* __restrict__ or0,
                                                      const float* __restrict__ sc1, const float* __restrict__ or1,
                                                      int copy_desc, float* __restrict__ x,
                                                      float* __restrict__ cs, float* __restrict__ sn,
                                                      int* __restrict__ ind, int* __restrict__ prune, int prune_init,
                                                      int* __restrict__ matches0, int* __restrict__ matches1,
                                                      float* __restrict__ ms0, float* __restrict__ ms1,
                                                      int* __restrict__ prune0, int* __restrict__ prune1) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y, b = seq >> 1, img = seq & 1;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= R) return;
    const int n = img ? n1[b] : n0[b];
    const size_t row = (size_t)seq * R + i;
    if (lane == 0) {
        ind[row] = i;
        prune[row] = prune_init;
        if (i < ncap) {
            (img ? matches1 : matches0)[(size_t)b * ncap + i] = -1;
            (img ? ms1 : ms0)[(size_t)b * ncap + i] = 0.0f;
            (img ? prune1 : prune0)[(size_t)b * ncap + i] = 0;
        }
    }
    if (i >= n) return;
    if (copy_desc) {  // 256-d descriptors: input_proj is the identity
        const float* d = (img ? d1 : d0) + ((size_t)b * ncap + i) * 256;
        *reinterpret_cast<float4*>(x + row * 256 + lane * 4) = *reinterpret_cast<const float4*>(d + lane * 4);
    }
    if (lane < 32) {
        const float* kp = (img ? kp1 : kp0) + ((size_t)b * ncap + i) * 2;
        const float W = img ? w1 : w0, H = img ? h1 : h0;
        // normalize_keypoints: (k - size/2) / (max(size)/2)
        const float scale = fmaxf(W, H) / 2.0f;
        const float kx = (kp[0] - W / 2.0f) / scale, ky = (kp[1] - H / 2.0f) / scale;
        float pr;
        if (sc0 != nullptr) {  // add_scale_ori: Wr(cat[kpts_normalised, scale, orientation])
            const float s = (img ? sc1 : sc0)[(size_t)b * ncap + i], o = (img ? or1 : or0)[(size_t)b * ncap + i];
            pr = kx * wr4[lane * 4 + 0] + ky * wr4[lane * 4 + 1] + s * wr4[lane * 4 + 2] + o * wr4[lane * 4 + 3];
        } else {
            pr = kx * wr[lane * 2 + 0] + ky * wr[lane * 2 + 1];
        }
        cs[row * 32 + lane] = cosf(pr);
        sn[row * 32 + lane] = sinf(pr);
    }
}

__global__ void lg_init_pairs_kernel(const int* __restrict__ n0, const int* __restrict__ n1, int B, int* __restrict__ cnt,
                                     int* __restrict__ norig, int* __restrict__ active, int* __restrict__ stop) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int a = n0[b], c = n1[b];
    cnt[2 * b] = a;
    cnt[2 * b + 1] = c;
    norig[2 * b] = a;
    norig[2 * b + 1] = c;
    const int act = (a > 0 && c > 0) ? 1 : 0;
    active[b] = act;
    stop[b] = act ? 0 : 1;  // upstream breaks out of the layer loop at i = 0 -> "stop": 1
    if (!act) {
        cnt[2 * b] = 0;
        cnt[2 * b + 1] = 0;
    }
}

// h = GELU(LayerNorm(h)) in place, 512 features, one wave per row (8 values per lane)
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(float* __restrict__ hbuf, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const int* __restrict__ cnt,
                                                         const int* __restrict__ active, int R) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= cnt[seq] || active[seq >> 1] == 0) return;
    float* row = hbuf + ((size_t)seq * R + i) * 512;
    float4 a = *reinterpret_cast<float4*>(row + lane * 4);
    float4 c = *reinterpret_cast<float4*>(row + 256 + lane * 4);
    const float mean = wave_sum(a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w) * (1.0f / 512.0f);
    a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean;
    c.x -= mean; c.y -= mean; c.z -= mean; c.w -= mean;
    const float var = wave_sum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) *
                      (1.0f / 512.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 4), g1 = *reinterpret_cast<const float4*>(gamma + 256 + lane * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + lane * 4), b1 = *reinterpret_cast<const float4*>(beta + 256 + lane * 4);
    auto f = [&](float v, float g, float bb) {
        const float y = v * rstd * g + bb;
        return 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    };
    a.x = f(a.x, g0.x, b0.x); a.y = f(a.y, g0.y, b0.y); a.z = f(a.z, g0.z, b0.z); a.w = f(a.w, g0.w, b0.w);
    c.x = f(c.x, g1.x, b1.x); c.y = f(c.y, g1.y, b1.y); c.z = f(c.z, g1.z, b1.z); c.w = f(c.w, g1.w, b1.w);
    *reinterpret_cast<float4*>(row + lane * 4) = a;
    *reinterpret_cast<float4*>(row + 256 + lane * 4) = c;
}

// token confidence + matchability of layer `layer` (one wave per row)
__global__ __launch_bounds__(256) void lg_conf_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bt, const float* __restrict__ wm,
                                                      const float* __restrict__ bm, const int* __restrict__ cnt,
                                                      const int* __restrict__ active, int R, float* __restrict__ conf,
                                                      float* __restrict__ mtch) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= cnt[seq] || active[seq >> 1] == 0) return;
    const size_t row = (size_t)seq * R + i;
    const float4 v = *reinterpret_cast<const float4*>(x + row * 256 + lane * 4);
    const float4 a = *reinterpret_cast<const float4*>(wt + lane * 4);
    const float4 c = *reinterpret_cast<const float4*>(wm + lane * 4);
    const float zt = wave_sum(v.x * a.x + v.y * a.y + v.z * a.z + v.w * a.w) + bt[0];
    const float zm = wave_sum(v.x * c.x + v.y * c.y + v.z * c.z + v.w * c.w) + bm[0];
    if (lane == 0) {
        conf[row] = sigmoidf_(zt);
        mtch[row] = sigmoidf_(zm);
    }
}

// per pair: early-stop decision and (if still running) the pruning plan of both images.
// Reads cnt_cur, writes cnt_next for both sequences of the pair.
__global__ __launch_bounds__(256) void lg_decide_kernel(const float* __restrict__ conf, const float* __restrict__ mtch,
                                                        const int* __restrict__ cnt_cur, int* __restrict__ cnt_next,
                                                        const int* __restrict__ norig, int* __restrict__ active,
                                                        int* __restrict__ stop, int* __restrict__ pos,
                                                        const int* __restrict__ ind, int* __restrict__ prune,
                                                        int* __restrict__ pflag, int R, int layer, int do_stop,
                                                        int do_prune, int prune_min, float tau, float depth_conf,
                                                        float keep_thr) {
    __shared__ int red[4];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = cnt_cur[2 * b], c1 = cnt_cur[2 * b + 1];
    if (active[b] == 0) {
        if (tid == 0) {
            cnt_next[2 * b] = c0;
            cnt_next[2 * b + 1] = c1;
        }
        return;
    }
    if (do_stop) {
        int nun = 0;
        for (int s = 0; s < 2; ++s) {
            const int c = s ? c1 : c0;
            const float* cf = conf + ((size_t)(2 * b + s)) * R;
            for (int i = tid; i < c; i += 256) nun += (cf[i] < tau) ? 1 : 0;
        }
        nun = wave_sum_i(nun);
        if (lane == 0) red[wv] = nun;
        __syncthreads();
        if (tid == 0) {
            const int tot = red[0] + red[1] + red[2] + red[3];
            // ratio_confident = 1.0 - (#unconfident).float() / num_points   (num_points = m + n, originals)
            const float ratio = 1.0f - (float)tot / (float)(norig[2 * b] + norig[2 * b + 1]);
            s_flag = (ratio > depth_conf) ? 1 : 0;
        }
        __syncthreads();
        if (s_flag) {
            if (tid == 0) {
                active[b] = 0;
                stop[b] = layer + 1;
                cnt_next[2 * b] = c0;
                cnt_next[2 * b + 1] = c1;
            }
            return;
        }
    }
    if (!do_prune) {
        if (tid == 0) {
            cnt_next[2 * b] = c0;
            cnt_next[2 * b + 1] = c1;
        }
        return;
    }
    // upstream prunes a side only while it holds more than pruning_keypoint_thresholds[device] points
    // (cpu -1: always; cuda 1024; flash 1536): `if do_point_pruning and desc0.shape[-2] > pruning_th`
    int newc[2];
    bool pruned[2];
    for (int s = 0; s < 2; ++s) {
        const int c = s ? c1 : c0;
        const size_t base = ((size_t)(2 * b + s)) * R;
        pruned[s] = c > prune_min;
        if (tid == 0) pflag[2 * b + s] = pruned[s] ? 1 : 0;
        if (!pruned[s]) {
            newc[s] = c;
            continue;
        }
        int run = 0;
        for (int i0 = 0; i0 < c; i0 += 256) {
            const int i = i0 + tid;
            bool keep = false;
            if (i < c) {
                keep = mtch[base + i] > keep_thr;
                if (do_stop) keep = keep || (conf[base + i] <= tau);
            }
            const unsigned long long bal = __ballot(keep);
            const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) red[wv] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wv; ++w) woff += red[w];
            const int tot = red[0] + red[1] + red[2] + red[3];
            if (i < c) pos[base + i] = keep ? (run + woff + wrank) : -1;
            run += tot;
            __syncthreads();
        }
        newc[s] = run;
    }
    if (newc[0] == 0 || newc[1] == 0) {
        // a side lost all its points: upstream still bumps the prune counters of the rows each side kept
        // (`prune0[:, ind0] += 1` runs before the loop breaks at the top of the next layer); the
        // gather / copy-back kernels skip inactive pairs, so do it here (a thread re-reads its own pos[])
        for (int s = 0; s < 2; ++s) {
            if (!pruned[s]) continue;
            const int c = s ? c1 : c0;
            const size_t base = ((size_t)(2 * b + s)) * R;
            for (int i = tid; i < c; i += 256)
                if (pos[base + i] >= 0) prune[base + ind[base + i]] += 1;
        }
    }
    if (tid == 0) {
        if (newc[0] == 0 || newc[1] == 0) {
            // upstream breaks at the top of the next layer ("no keypoints" result, stop = (layer + 1) + 1)
            active[b] = 0;
            stop[b] = layer + 2;
            cnt_next[2 * b] = 0;
            cnt_next[2 * b + 1] = 0;
        } else {
            cnt_next[2 * b] = newc[0];
            cnt_next[2 * b + 1] = newc[1];
        }
    }
}

// gather surviving rows to the temp buffers (x, cos/sin, index map), bump prune counters
__global__ __launch_bounds__(256) void lg_gather_kernel(const float* __restrict__ x, float* __restrict__ xt,
                                                        const float* __restrict__ cs, const float* __restrict__ sn,
                                                        float* __restrict__ cst, float* __restrict__ snt,
                                                        const int* __restrict__ ind, int* __restrict__ indt,
                                                        int* __restrict__ prune, const int* __restrict__ pos,
                                                        const int* __restrict__ cnt_cur, const int* __restrict__ cnt_next,
                                                        const int* __restrict__ active, int R) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (active[seq >> 1] == 0 || i >= cnt_cur[seq] || cnt_next[seq] == cnt_cur[seq]) return;
    const size_t row = (size_t)seq * R + i;
    const int p = pos[row];
    if (p < 0) return;
    const size_t dst = (size_t)seq * R + p;
    *reinterpret_cast<float4*>(xt + dst * 256 + lane * 4) = *reinterpret_cast<const float4*>(x + row * 256 + lane * 4);
    if (lane < 32) {
        cst[dst * 32 + lane] = cs[row * 32 + lane];
        snt[dst * 32 + lane] = sn[row * 32 + lane];
    }
    if (lane == 0) indt[dst] = ind[row];
}
__global__ __launch_bounds__(256) void lg_copyback_kernel(float* __restrict__ x, const float* __restrict__ xt,
                                                          float* __restrict__ cs, float* __restrict__ sn,
                                                          const float* __restrict__ cst, const float* __restrict__ snt,
                                                          int* __restrict__ ind, const int* __restrict__ indt,
                                                          int* __restrict__ prune, const int* __restrict__ cnt_cur,
                                                          const int* __restrict__ cnt_next,
                                                          const int* __restrict__ active,
                                                          const int* __restrict__ pflag, int R, int do_prune) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (!do_prune || active[seq >> 1] == 0 || pflag[seq] == 0 || i >= cnt_next[seq]) return;
    const size_t row = (size_t)seq * R + i;
    const bool moved = cnt_next[seq] != cnt_cur[seq];
    if (moved) {
        *reinterpret_cast<float4*>(x + row * 256 + lane * 4) = *reinterpret_cast<const float4*>(xt + row * 256 + lane * 4);
        if (lane < 32) {
            cs[row * 32 + lane] = cst[row * 32 + lane];
            sn[row * 32 + lane] = snt[row * 32 + lane];
        }
    }
    if (lane == 0) {
        const int oi = moved ? indt[row] : ind[row];
        if (moved) ind[row] = oi;
        prune[(size_t)seq * R + oi] += 1;  // prune0[:, ind0] += 1 for the kept points
    }
}

// final per-row quantities with the head of the layer each pair stopped at:
// z = matchability logit -> ls = logsigmoid(z)
__global__ __launch_bounds__(256) void lg_match_logit_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                             const float* __restrict__ bm, const int* __restrict__ stop,
                                                             const int* __restrict__ cnt, int R, float* __restrict__ ls) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= cnt[seq]) return;
    const int layer = stop[seq >> 1] - 1;
    const size_t row = (size_t)seq * R + i;
    const float4 v = *reinterpret_cast<const float4*>(x + row * 256 + lane * 4);
    const float4 c = *reinterpret_cast<const float4*>(wm + (size_t)layer * 256 + lane * 4);
    const float z = wave_sum(v.x * c.x + v.y * c.y + v.z * c.z + v.w * c.w) + bm[layer];
    if (lane == 0) ls[row] = logsigmoidf_(z);
}

__global__ void lg_finish_stop_kernel(int* __restrict__ stop, const int* __restrict__ active, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && active[b]) stop[b] = LG_LAYERS;
}

// filter_matches + mapping through the pruning index maps, one block per pair
__global__ __launch_bounds__(256) void lg_filter_kernel(const int* __restrict__ cnt, const int* __restrict__ norig, int R,
                                                        int ncap, const int* __restrict__ m0, const int* __restrict__ m1,
                                                        const float* __restrict__ max0, const int* __restrict__ ind,
                                                        const int* __restrict__ prune, float* __restrict__ ms0tmp,
                                                        int* __restrict__ valid0, float thr, int* __restrict__ matches0,
                                                        int* __restrict__ matches1, float* __restrict__ mscores0,
                                                        float* __restrict__ mscores1, int* __restrict__ prune0,
                                                        int* __restrict__ prune1) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = cnt[2 * b], n1 = cnt[2 * b + 1];
    const int* ind0 = ind + ((size_t)2 * b) * R;
    const int* ind1 = ind + ((size_t)2 * b + 1) * R;
    const size_t pb = (size_t)b * R;
    for (int i = tid; i < norig[2 * b]; i += 256) prune0[(size_t)b * ncap + i] = prune[((size_t)2 * b) * R + i];
    for (int i = tid; i < norig[2 * b + 1]; i += 256) prune1[(size_t)b * ncap + i] = prune[((size_t)2 * b + 1) * R + i];
    for (int i = tid; i < n0; i += 256) {
        const int j = m0[pb + i];
        const bool mutual = (m1[pb + j] == i);
        const float ms = mutual ? expf(max0[pb + i]) : 0.0f;
        const bool valid = mutual && (ms > thr);
        ms0tmp[pb + i] = ms;
        valid0[pb + i] = valid ? 1 : 0;
        const size_t o = (size_t)b * ncap + ind0[i];
        matches0[o] = valid ? ind1[j] : -1;
        mscores0[o] = ms;
    }
    __syncthreads();
    for (int j = tid; j < n1; j += 256) {
        const int i = m1[pb + j];
        const bool mutual = (m0[pb + i] == j);
        const float ms = mutual ? ms0tmp[pb + i] : 0.0f;
        const bool valid = mutual && (valid0[pb + i] != 0);
        const size_t o = (size_t)b * ncap + ind1[j];
        matches1[o] = valid ? ind0[i] : -1;
        mscores1[o] = ms;
    }
}

// ------------------------------------------------------------------ forward
extern "C" int imcui_hip_lightglue_forward(imcui_hip_t* h, const float* packed, int B, int ncap, int input_dim,
                                           const float* keypoints0, const float* keypoints1, const float* descriptors0,
                                           const float* descriptors1, const float* scales0, const float* oris0,
                                           const float* scales1, const float* oris1, const int* n0, const int* n1, float w0,
                                           float h0,
                                           float w1, float h1, double depth_confidence, double width_confidence,
                                           int pruning_threshold, double filter_threshold, int* matches0, int* matches1,
                                           float* mscores0,
                                           float* mscores1, int* stop, int* prune0, int* prune1, void* ws,
                                           size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (B <= 0) return IMCUI_OK;
    if (ncap <= 0) return imcui_set_err(h, IMCUI_ERR_ARG, "lightglue: ncap=%d must be positive", ncap);
    if (input_dim <= 0 || input_dim > 256 || input_dim % 32 != 0)
        return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "lightglue: input_dim=%d must be a multiple of 32, at most 256", input_dim);
    if ((scales0 != nullptr) != (oris0 != nullptr) || (scales0 != nullptr) != (scales1 != nullptr) || (scales0 != nullptr) != (oris1 != nullptr))
        return imcui_set_err(h, IMCUI_ERR_ARG, "lightglue: scales / oris must be given for both images or not at all");
    if (!packed || !keypoints0 || !keypoints1 || !descriptors0 || !descriptors1 || !n0 || !n1 || !matches0 || !matches1 ||
        !mscores0 || !mscores1 || !stop || !prune0 || !prune1)
        return imcui_set_err(h, IMCUI_ERR_ARG, "lightglue: null argument");
    const int R = (int)align_up((size_t)ncap, 128);
    LgWs w = lg_carve(ws, ws_bytes, B, R);
    if (!ws || !w.ok) return imcui_set_err(h, IMCUI_ERR_WS, "lightglue: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const LgLayout l = lg_layout();
    const float* P = packed;
    const int S = 2 * B;
    const int do_stop = depth_confidence > 0.0;
    const int do_prune = width_confidence > 0.0;
    const float depth_f = (float)depth_confidence;
    const float keep_thr = (float)(1.0 - width_confidence);
    const float filt_f = (float)filter_threshold;
    const dim3 rowgrid(R / 4, S), blk(256);
    int rc;
#define LGRUN(x)                       \
    do {                               \
        rc = (x);                      \
        if (rc != IMCUI_OK) return rc; \
    } while (0)

    int* cnt_cur = w.cntA;
    int* cnt_next = w.cntB;
    hipLaunchKernelGGL(lg_init_pairs_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, n0, n1, B, cnt_cur, w.norig, w.active,
                       stop);
    hipLaunchKernelGGL(lg_init_kernel, rowgrid, blk, 0, stream, keypoints0, keypoints1, descriptors0, descriptors1, n0, n1,
                       ncap, R, w0, h0, w1, h1, P + l.wr, P + l.wr4, scales0, oris0, scales1, oris1, input_dim == 256 ? 1 : 0, w.x, w.cs,
                       w.sn, w.ind, w.prune, do_prune ? 1 : LG_LAYERS, matches0, matches1, mscores0, mscores1, prune0, prune1);
    IMCUI_CHECK_LAUNCH(h);
    if (input_dim != 256) {
        // upstream `input_proj` = Linear(input_dim, 256): image `img` of pair z -> rows of sequence 2 z + img
        for (int img = 0; img < 2; ++img) {
            GemmP g;
            g.epi = EPI_BIAS;
            g.batch = B;
            g.A = img ? descriptors1 : descriptors0;
            g.lda = input_dim;
            g.a_bs = (long)ncap * input_dim;
            g.W = P + l.winp;
            g.ldw = input_dim;
            g.bias = P + l.binp;
            g.C = w.x + (size_t)img * R * 256;
            g.ldc = 256;
            g.c_bs = (long)2 * R * 256;
            g.M = ncap;
            g.N = 256;
            g.K = input_dim;
            g.mcnt = img ? n1 : n0;
            g.cnt_stride = 1;
            rc = gemm_launch(h, g, stream);
            if (rc != IMCUI_OK) return rc;
        }
    }

    const bool split = h->precision == 1;
    auto base = [&](GemmP& g) {
        g.M = S * R;
        g.cnt = cnt_cur;
        g.active = w.active;
        g.rows_per_seq = R;
    };
    auto wts = [&](GemmP& g, size_t wf32, const LgSplit& sp) {
        g.W = P + wf32;
        if (split) {
            g.Wh = reinterpret_cast<const unsigned short*>(P + sp.h);
            g.Wl = reinterpret_cast<const unsigned short*>(P + sp.l);
            g.wscale = P + sp.s;
        }
    };
    // split mode: the whole FFN is one kernel (ffn.hip); IMCUI_LG_FFN_UNFUSED=1 keeps the three-launch path for A/B runs
    static const bool ffn_unfused = getenv("IMCUI_LG_FFN_UNFUSED") != nullptr;
    auto ffn = [&](const float* ctx, size_t w1, const LgSplit& s1, size_t b1, size_t gm, size_t bt, size_t w2,
                   const LgSplit& s2, const LgSplit& s2p, size_t b2) -> int {
        if (split && !ffn_unfused) {
            FfnP f;
            f.x = w.x;
            f.ctx = ctx;
            f.out = w.x;
            f.w1h = reinterpret_cast<const unsigned short*>(P + s1.h);
            f.w1l = reinterpret_cast<const unsigned short*>(P + s1.l);
            f.s1 = P + s1.s;
            f.b1 = P + b1;
            f.gamma = P + gm;
            f.beta = P + bt;
            f.w2h = reinterpret_cast<const unsigned short*>(P + s2p.h);
            f.w2l = reinterpret_cast<const unsigned short*>(P + s2p.l);
            f.s2 = P + s2p.s;
            f.b2 = P + b2;
            f.M = S * R;
            f.cnt = cnt_cur;
            f.active = w.active;
            f.rows_per_seq = R;
            return ffn_launch(h, f, stream);
        }
        GemmP g;
        base(g);
        g.epi = EPI_BIAS;
        g.A = w.x;
        g.lda = 256;
        g.A2 = ctx;  // out_proj is folded into W1 at pack time
        g.lda2 = 256;
        g.K1 = 256;
        g.K = 512;
        g.N = 512;
        wts(g, w1, s1);
        g.ldw = 512;
        g.bias = P + b1;
        g.C = w.hbuf;
        g.ldc = 512;
        int r = gemm_launch(h, g, stream);
        if (r != IMCUI_OK) return r;
        hipLaunchKernelGGL(lg_ln_gelu_kernel, rowgrid, blk, 0, stream, w.hbuf, P + gm, P + bt, cnt_cur, w.active, R);
        GemmP g2;
        base(g2);
        g2.epi = EPI_RESID;
        g2.A = w.hbuf;
        g2.lda = 512;
        g2.K = 512;
        g2.N = 256;
        wts(g2, w2, s2);
        g2.ldw = 512;
        g2.bias = P + b2;
        g2.C = w.x;
        g2.ldc = 256;
        return gemm_launch(h, g2, stream);
    };

    for (int layer = 0; layer < LG_LAYERS; ++layer) {
        const LgLayerOff& o = l.L[layer];
        // ---- SelfBlock
        {
            GemmP g;
            base(g);
            g.epi = EPI_QKV;
            g.A = w.x;
            g.lda = 256;
            g.K = 256;
            g.N = 768;
            wts(g, o.wqkv, o.sqkv);
            g.ldw = 256;
            g.bias = P + o.bqkv;
            g.v_transposed = split ? 1 : 0;
            g.split_out = split ? 1 : 0;
            g.plane_halves = (size_t)S * R * 256;
            g.Q = w.q;
            g.Kt = w.k;
            g.V = w.v;
            g.rope_cos = w.cs;
            g.rope_sin = w.sn;
            g.alpha = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) and log2(e) folded into q: the attention kernels work in base 2
            g.heads = LG_HEADS;
            LGRUN(gemm_launch(h, g, stream));
            AttnP a;
            a.Q = w.q;
            a.K = w.k;
            a.V = w.v;
            a.O = w.ctx;
            a.cnt = cnt_cur;
            a.active = w.active;
            a.nseq = S;
            a.heads = LG_HEADS;
            a.rows_per_seq = R;
            a.cross = 0;
            a.log2_domain = 1;
            a.V6 = w.v6;
            a.part = w.apart;
            if (h->opt[OPT_ATTN_SELF] >= 0 && ((h->opt[OPT_ATTN_MIX_LAYERS] >> layer) & 1)) a.variant = h->opt[OPT_ATTN_SELF];  // (per-block arithmetic mix: audit tool)
            LGRUN(attention_launch(h, a, stream));
            LGRUN(ffn(w.ctx, o.w1s, o.s1s, o.b1s, o.gs, o.bs, o.w2s, o.s2s, o.s2sp, o.b2s));
        }
        // ---- CrossBlock
        {
            GemmP g;
            base(g);
            g.epi = EPI_CROSS;
            g.A = w.x;
            g.lda = 256;
            g.K = 256;
            g.N = 512;
            wts(g, o.wx, o.sx);
            g.ldw = 256;
            g.bias = P + o.bx;
            g.v_transposed = split ? 1 : 0;
            g.split_out = split ? 1 : 0;
            g.plane_halves = (size_t)S * R * 256;
            g.Q = w.q;
            g.V = w.v;
            g.alpha = (float)(0.35355339059327373 * 1.2011224087864498);  // (64 ** -0.5) ** 0.5 and sqrt(log2 e), applied to both sides
            g.heads = LG_HEADS;
            LGRUN(gemm_launch(h, g, stream));
            AttnP a;
            a.Q = w.q;
            a.K = w.q;
            a.V = w.v;
            a.O = w.ctx;
            a.cnt = cnt_cur;
            a.active = w.active;
            a.nseq = S;
            a.heads = LG_HEADS;
            a.rows_per_seq = R;
            a.cross = 1;
            a.log2_domain = 1;
            a.V6 = w.v6;
            a.part = w.apart;
            {   // -2 (default) = the audited two-product P.V (7) while attn_variant is the default kernel (8), otherwise attn_variant governs; -1 = attn_variant
                const int cv = h->opt[OPT_ATTN_CROSS] == -2 ? (h->opt[OPT_ATTN_VARIANT] == 8 ? 7 : -1) : h->opt[OPT_ATTN_CROSS];
                if (cv >= 0 && ((h->opt[OPT_ATTN_MIX_LAYERS] >> layer) & 1)) a.variant = cv;
            }
            LGRUN(attention_launch(h, a, stream));
            LGRUN(ffn(w.ctx, o.w1c, o.s1c, o.b1c, o.gc, o.bc, o.w2c, o.s2c, o.s2cp, o.b2c));
        }
        if (h->lg_dump) {  // parity-test hook: token states after this layer (rows in their current, pruned order)
            const size_t nf = (size_t)S * R * 256;
            if ((size_t)(layer + 1) * nf > h->lg_dump_floats)
                return imcui_set_err(h, IMCUI_ERR_ARG, "lightglue: layer dump buffer too small (%zu < %zu floats)", h->lg_dump_floats,
                                     (size_t)LG_LAYERS * nf);
            if (hipMemcpyAsync(h->lg_dump + (size_t)layer * nf, w.x, nf * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return imcui_set_err(h, IMCUI_ERR_HIP, "lightglue: layer dump copy failed");
        }
        if (layer == LG_LAYERS - 1) break;  // no early stopping or adaptive width at the last layer
        if (!do_stop && !do_prune) continue;
        // ---- a10: confidence, early stop, point pruning
        const float tau = (float)fmin(fmax(0.8 + 0.1 * exp(-4.0 * layer / LG_LAYERS), 0.0), 1.0);
        hipLaunchKernelGGL(lg_conf_kernel, rowgrid, blk, 0, stream, w.x, P + l.wtoken + (size_t)layer * 256,
                           P + l.btoken + layer, P + l.wmatch + (size_t)layer * 256, P + l.bmatch + layer, cnt_cur, w.active,
                           R, w.conf, w.mtch);
        hipLaunchKernelGGL(lg_decide_kernel, dim3(B), blk, 0, stream, w.conf, w.mtch, cnt_cur, cnt_next, w.norig, w.active,
                           stop, w.pos, w.ind, w.prune, w.pflag, R, layer, do_stop, do_prune, pruning_threshold, tau, depth_f,
                           keep_thr);
        if (do_prune) {
            hipLaunchKernelGGL(lg_gather_kernel, rowgrid, blk, 0, stream, w.x, w.xt, w.cs, w.sn, w.cst, w.snt, w.ind, w.indt,
                               w.prune, w.pos, cnt_cur, cnt_next, w.active, R);
            hipLaunchKernelGGL(lg_copyback_kernel, rowgrid, blk, 0, stream, w.x, w.xt, w.cs, w.sn, w.cst, w.snt, w.ind,
                               w.indt, w.prune, cnt_cur, cnt_next, w.active, w.pflag, R, do_prune);
        }
        IMCUI_CHECK_LAUNCH(h);
        int* t = cnt_cur;
        cnt_cur = cnt_next;
        cnt_next = t;
    }

    // ---- a11: assignment with the head of the layer each pair stopped at
    hipLaunchKernelGGL(lg_finish_stop_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, stop, w.active, B);
    {
        GemmP g;  // md = final_proj(x) / 256^(1/4)
        g.M = S * R;
        g.cnt = cnt_cur;
        g.rows_per_seq = R;
        g.epi = EPI_BIAS;
        g.A = w.x;
        g.lda = 256;
        g.K = 256;
        g.N = 256;
        wts(g, l.wfinal, l.sfinal);
        g.ldw = 256;
        g.bias = P + l.bfinal;
        g.wsel = stop;
        g.wsel_off = -1;
        g.w_stride = 65536;
        g.b_stride = 256;
        g.alpha = 0.25f;
        g.C = w.md;
        g.ldc = 256;
        LGRUN(gemm_launch(h, g, stream));
    }
    hipLaunchKernelGGL(lg_match_logit_kernel, rowgrid, blk, 0, stream, w.x, P + l.wmatch, P + l.bmatch, stop, cnt_cur, R,
                       w.ls);
    {
        // Round 5: sim[b] = md0[b] . md1[b]^T is never stored (simred.hip).  Pass 1: soft-max statistics of both directions (rows carried in
        // registers across the column tiles, columns per 128-row block); pass 2: the same tiles again, the log assignment ONCE per element,
        // row best (first column) and column best (first row).  Both descriptor sets are split into MFMA fragments once, by the packer.
        const int nchunk = simred_chunks(B, R, R), nrb = R / 128;
        const int tpc = (R / 128 + nchunk - 1) / nchunk;
        simred_pack(h, w.md, 256, 1, (long)2 * R * 256, R, 256, B, cnt_cur, 2, w.pk0, stream);
        simred_pack(h, w.md + (size_t)R * 256, 256, 1, (long)2 * R * 256, R, 256, B, cnt_cur + 1, 2, w.pk1, stream);
        SimRedP sp;
        sp.mode = SR_LSE;
        sp.Ap = w.pk0, sp.Bp = w.pk1;
        sp.ap_bs = sp.bp_bs = (long)simred_packed_uint4(R, 256);
        sp.M = R, sp.N = R, sp.K = 256, sp.batch = B;
        sp.mcnt = cnt_cur, sp.ncnt = cnt_cur + 1, sp.cnt_stride = 2;
        sp.nchunk = nchunk;
        sp.r0 = w.rpm, sp.r1 = w.rps, sp.ri = w.rpj, sp.r_pitch = R;
        sp.c0 = w.cpm, sp.c1 = w.cps, sp.ci = w.cpi, sp.c_pitch = R;
        LGRUN(simred_launch(h, sp, stream));
        const dim3 mg(cdiv(R, 256), B, 2);
        hipLaunchKernelGGL(lg_stat_merge_kernel, mg, blk, 0, stream, w.rpm, w.rps, w.cpm, w.cps, cnt_cur, R, nchunk, nrb, tpc * 128, 128, 128, w.rmax, w.rls, w.cmax,
                           w.cls);
        sp.mode = SR_LGBEST;
        sp.rmax = w.rmax, sp.rsum = w.rls, sp.cmax = w.cmax, sp.csum = w.cls;
        sp.l0 = w.ls, sp.l1 = w.ls + R, sp.l0_bs = sp.l1_bs = (long)2 * R;
        LGRUN(simred_launch(h, sp, stream));
        hipLaunchKernelGGL(lg_best_merge_kernel, mg, blk, 0, stream, w.rpm, w.rpj, w.cpm, w.cpi, cnt_cur, R, nchunk, nrb, tpc * 128, 128, w.max0, w.m0, w.m1);
    }
    hipLaunchKernelGGL(lg_filter_kernel, dim3(B), blk, 0, stream, cnt_cur, w.norig, R, ncap, w.m0, w.m1, w.max0, w.ind,
                       w.prune, w.ms0, w.valid0, filt_f, matches0, matches1, mscores0, mscores1, prune0, prune1);
    IMCUI_CHECK_LAUNCH(h);
#undef LGRUN
    return IMCUI_OK;
}
