// MFMA GEMM with fused epilogues:  C = epi(A[M,K] * W[N,K]^T + bias).
// Both operands are K-contiguous (nn.Linear weight layout / NHWC activations).
// Two arithmetic modes (imcui_hip_s::precision):
//   0  exact f32 on v_mfma_f32_32x32x2_f32
//   1  3 x f16 split (hi/lo) on v_mfma_f32_32x32x16_f16, f32 accumulate (~fp32 accuracy, ~5x rate)
#pragma once
#include "common.h"

enum GemmEpi {
    EPI_BIAS = 0,   // C = (acc + bias) * alpha          (bias may be null)
    EPI_RELU = 1,   // C = relu(acc + bias)
    EPI_RESID = 2,  // C += acc + bias                  (in-place residual)
    EPI_QKV = 3,    // LightGlue SelfBlock: bias, RoPE on q/k, q *= alpha, head-major split
    EPI_CROSS = 4,  // LightGlue CrossBlock: [qk | v] = bias, qk *= alpha, head-major split
    EPI_CONV = 5,   // C = act(acc + bias + resid): act 0 none / 1 ReLU / 2 LeakyReLU(0.01) / 3 GELU (erf)
    // (6 was EPI_SIMSTAT, rounds 3-4: the similarity stored AND reduced to soft-max partials -- replaced by simred.hip, which stores nothing)
    // ViT blocks of the DUSt3R / MASt3R networks (gemm_wreg_kernel only): the output features are `heads` x 64 wide blocks of
    // q / k / v in that order starting at block `role0` (0: q, 1: k, 2: v -- self attention [q | k | v]: role0 = 0, N = 3 C; the
    // cross-attention key / value projection [k | v]: role0 = 1, N = 2 C; its query projection: role0 = 0, N = C).  q and k get
    // CroCo's RoPE2D from the tables rope_cos / rope_sin [rows_per_seq][32] (entry 16 half + i = angle of the y (half 0) / x
    // (half 1) position of the token times inv_freq[i]; inside a 32-feature half feature i < 16 pairs with i + 16), q *= alpha;
    // everything leaves as f16 hi / lo planes, q / k [seq][head][row][64], v transposed [seq][head][64][row].
    EPI_QKV_VIT = 7,
    // NO matrix output: the similarity tile of two activation matrices (batched) is reduced while it is parked in LDS to the
    // nearest-neighbour partials of the mutual-NN matcher -- per row (best value, its column, second best value) over the tile's
    // 128 columns -> st_rpm / st_rpi / st_rps [batch][st_nct][st_rpitch], per column over each 64-row half -> st_cpm / st_cpi /
    // st_cps [batch][st_nrh][st_cpitch]; ties keep the lowest index (split mode, f32 B operand; nn.hip)
    EPI_NNSTAT = 8,
};

struct GemmP {
    int epi = EPI_BIAS;
    const float* A = nullptr;   // [M, K1] (or [M, K] when A2 == nullptr)
    long lda = 0;
    const float* A2 = nullptr;  // optional second K-slab: k >= K1 comes from A2[:, k - K1]
    long lda2 = 0;
    int K1 = 0;
    const float* W = nullptr;  // [N, K] f32
    long ldw = 0;
    // pre-split weights for precision 1 (optional; when null W is split on the fly): f16 planes of
    // W * 2^e in FRAGMENT-MAJOR order [ceil(N/32)][K/16][64 lanes][8 halves] (split_weights_frag_host),
    // wscale -> 2^-e
    const unsigned short* Wh = nullptr;
    const unsigned short* Wl = nullptr;
    const float* wscale = nullptr;
    const float* bias = nullptr;  // [N] or null
    float* C = nullptr;           // [M, N] (EPI_BIAS / RELU / RESID)
    long ldc = 0;
    int M = 0, N = 0, K = 0;
    float alpha = 1.0f;
    int group_rows = 0;  // > 1: tile order in groups of this many 128-row panels (L2 re-use when both operands are large)
    // ragged sequences: row tile r0 belongs to sequence r0 / rows_per_seq; tiles whose first
    // row is >= cnt[seq] (or whose pair active[seq >> 1] == 0) are skipped.
    const int* cnt = nullptr;
    const int* active = nullptr;
    int rows_per_seq = 0;
    // per-pair weight selection: W += (wsel[seq >> 1] + wsel_off) * w_stride, bias likewise
    const int* wsel = nullptr;
    int wsel_off = 0;
    long w_stride = 0, b_stride = 0;
    // batched mode (blockIdx.z = batch): per-batch strides and dynamic M/N
    int batch = 1;
    long a_bs = 0, a2_bs = 0, w_bs = 0, c_bs = 0;
    const int* mcnt = nullptr;  // M for batch z = mcnt[z * cnt_stride]
    const int* ncnt = nullptr;  // N for batch z = ncnt[z * cnt_stride]
    int cnt_stride = 1;
    // EPI_QKV / EPI_CROSS outputs, head-major [seq][head][row_in_seq][64]
    float* Q = nullptr;
    float* Kt = nullptr;
    float* V = nullptr;
    int v_transposed = 0;  // 1: V is written as V^T [seq][head][64][rows_per_seq] (split attention)
    int split_out = 0;     // 1: Q / K / V^T are written as f16 hi / lo planes (lo plane at +plane_halves)
    size_t plane_halves = 0;
    // implicit im2col (enabled when conv_k > 0): A is an NHWC image [B, hin, win, cin], row m of the
    // GEMM is output pixel m of a conv_k x conv_k convolution (stride, zero padding), K = conv_k^2 * cin
    // ordered (tap, channel) -- the layout of pack_conv_gemm(); cin % 32 == 0
    int conv_k = 0, conv_stride = 1, conv_pad = 0, conv_hin = 0, conv_win = 0, conv_hout = 0, conv_wout = 0, conv_cin = 0;
    const float* resid = nullptr;  // EPI_CONV: [M, N] added before the activation (may alias C)
    long ldr = 0;
    // EPI_CONV, split kernels: the residual is the bilinear x2 up-sampling of the NHWC map `resid` [images, rup_h, rup_w, ldr]
    // (rows of the GEMM = pixels of the 2 rup_h x 2 rup_w maps), evaluated in the epilogue instead of being materialised;
    // rup_align = align_corners of the interpolation (LoFTR: 1, EfficientLoFTR: 0).  0 = plain [M, N] residual.
    int rup_h = 0, rup_w = 0, rup_align = 1;
    int act = 0;
    // 1: ONE f16 product per element pair (hi planes only, f32 accumulate) instead of the three of the split arithmetic: 11-bit
    // operands, the class of a bf16 / fp16 autocast run.  EPI_CONV with pre-split weight planes only.
    int single = 0;
    float *st_rpm = nullptr, *st_rps = nullptr, *st_cpm = nullptr, *st_cps = nullptr;  // EPI_NNSTAT partials: best / second best
    int st_nct = 0, st_nrh = 0;                                                         // partial slots per row / per column
    int *st_rpi = nullptr, *st_cpi = nullptr;                                           // EPI_NNSTAT: arg-best partials
    long st_rpitch = 0, st_cpitch = 0;                                                  // EPI_NNSTAT: entries per partial slot (>= M rows / >= N columns)
    const float* rope_cos = nullptr;  // [rows, 32]
    const float* rope_sin = nullptr;
    int heads = 4;
    int role0 = 0;  // EPI_QKV_VIT: role (0 q, 1 k, 2 v) of the first heads x 64 block of output features
    // EPI_QKV_VIT with sequences on token grids of different shapes: first table row of each sequence's grid (device, per sequence);
    // nullptr: every sequence reads the table from row 0
    const int* rope_seq_row0 = nullptr;
    // LayerNorm folded into the layer (DUSt3R; gemm_wreg_kernel, EPI_CONV and EPI_QKV_VIT only): A holds the RAW rows x, the weights carry
    // gamma and the bias carries W beta (pack time), and the epilogue applies  out = rstd_row (acc - mean_row ln_rowsum[n]) + bias[n]
    // where the plain layer computes acc + bias[n].  ln_stats [M][2] = (mean, rstd) of every row of A, ln_rowsum [N] = sum_k W[n][k] of
    // the packed (gamma-folded) weights, + ln_stride floats per selected weight set (wsel)
    const float* ln_stats = nullptr;
    const float* ln_rowsum = nullptr;
    long ln_stride = 0;
};

int gemm_launch(imcui_hip_s* h, const GemmP& p, hipStream_t stream);
// gemm_wreg.hip: the weights-in-registers kernel for projection layers (split mode, pre-split weight planes); gemm_launch
// routes eligible launches to it
bool gemm_wreg_ok(const imcui_hip_s* h, const GemmP& p);
void gemm_wreg_launch(const imcui_hip_s* h, const GemmP& p, hipStream_t stream);

// host: OIHW conv weight -> GEMM weight [Cout][tap][Cin] (K order of the implicit im2col)
void pack_conv_gemm(const float* w_oihw, int Cout, int Cin, int ksize, int Cin_pad, float* dst);
// host: split an [n] f32 array into f16 hi / lo planes of w * 2^e; returns 2^-e
float split_weights_host(const float* w, size_t n, unsigned short* hi, unsigned short* lo);
// host: the same for a GEMM weight [N][K], planes written fragment-major with rows padded to 32
float split_weights_frag_host(const float* w, int N, int K, unsigned short* hi, unsigned short* lo);
