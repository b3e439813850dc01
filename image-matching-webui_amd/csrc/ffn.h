// Fused transformer FFN of the LightGlue / SuperGlue blocks (3 x f16 split mode):
//   x <- x + W2 * GELU(LayerNorm(W1 * [x | ctx] + b1)) + b2          (512 hidden features, 256-wide residual stream)
//   x <- x + W2 * ReLU(W1 * [x | ctx] + b1) + b2                     (act = 1: SuperGlue's MLP, BatchNorm folded)
//   x <- x + LayerNorm(W2 * act(W1 * [x | ctx]))                    (act = 2 / 3: the dense matchers' MLPs)
// one kernel per block instead of GEMM -> LayerNorm/GELU -> GEMM: the 512-wide hidden row never leaves the CU.
#pragma once
#include "common.h"

struct FfnP {
    const float* x = nullptr;    // [M, 256] residual stream (read as the first K slab and as the residual)
    const float* ctx = nullptr;  // [M, 256] attention context (second K slab; out_proj is folded into W1)
    float* out = nullptr;        // [M, 256], may alias x
    // fragment-major f16 hi / lo planes (split_weights_frag_host): W1 [512][512]; W2 [256][512] with its K axis in
    // the order ffn_permute_k() produces (the order GEMM 1's accumulators hand their features to GEMM 2)
    const unsigned short *w1h = nullptr, *w1l = nullptr, *w2h = nullptr, *w2l = nullptr;
    const float *s1 = nullptr, *s2 = nullptr;  // 2^-e scales of the planes (device)
    const float *b1 = nullptr, *gamma = nullptr, *beta = nullptr;  // [512]
    const float* b2 = nullptr;                                     // [256]
    // 0: LayerNorm(512) + GELU between the GEMMs (LightGlue); 1: ReLU (SuperGlue, BatchNorm folded into W1);
    // 2 / 3: LeakyReLU(0.01) / ReLU between the GEMMs and LayerNorm(256) (gamma, beta [256]) after the second, before the
    // residual (EfficientLoFTR / LoFTR coarse MLPs: x + LN(fc2(act(fc1([x | message])))), no biases: b1 = b2 = nullptr)
    int act = 0;
    int M = 0;
    // ragged sequences, as in GemmP: a 128-row tile whose first row is >= cnt[seq] or whose pair is inactive is skipped
    const int* cnt = nullptr;
    const int* active = nullptr;
    int rows_per_seq = 0;
    long long* dbg = nullptr;  // lab only: 8 wall-clock stamps per workgroup (phase boundaries, wave 0)
};

int ffn_launch(imcui_hip_s* h, const FfnP& p, hipStream_t stream);

// host: dst[n][c] = src[n][ffn_k_source(c)] -- the K order of W2 the kernel expects (K = 512)
void ffn_permute_k(const float* src, int N, int K, float* dst);
