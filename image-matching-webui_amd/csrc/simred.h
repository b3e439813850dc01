// Similarity-and-reduce: the products  sim[b] = alpha * A[b] . B[b]^T  of two activation matrices that the matchers only ever REDUCE
// (nearest neighbours, soft-max statistics, dual-softmax confidences, LightGlue's log assignment) -- computed tile by tile on the matrix
// cores and reduced in registers / LDS; the matrix itself never exists in memory.  simred.hip.
//
// Replaces, per SURVEY.md section 8:  a12 `find_nn` / `mutual_check` (imcui/hloc/matchers/nearest_neighbor.py:6-24,38-66),
// a15 kornia `CoarseMatching` dual-softmax (imcui/hloc/matchers/loftr.py:54), `dual_softmax_matcher` (imcui/hloc/matchers/dual_softmax.py:8-41),
// a11 LightGlue's `log_double_softmax` + `filter_matches` (imcui/hloc/matchers/lightglue.py:54-75 -> upstream lightglue.py).
#pragma once
#include "common.h"

enum SimRedMode {
    SR_NN = 0,      // per row / per column: best value, its FIRST index, second best value
    SR_LSE = 1,     // per row / per column: max and sum exp(x - max)            (pass 1 of every soft-max based matcher)
    SR_DSBEST = 2,  // conf = softmax_col(x) * softmax_row(x): per row best (value, first column), per column best value   (pass 2, LoFTR family)
    SR_LGBEST = 3,  // LightGlue's log assignment: per row best (value, first column), per column best (value, first row)   (pass 2)
    SR_NN1 = 4,     // SR_NN without the second best (find_nn without a ratio test): r1 / c1 are not written
};

#define SR_TILE 128  // rows per workgroup block and columns per tile

struct SimRedP {
    int mode = SR_NN;
    // operands packed by simred_pack(): [batch][32-row fragment][K / 16][2 pieces][64 lanes] x 16 bytes; fragments cover
    // roundup(rows, 128) rows, rows past the (dynamic) count are zero
    const uint4* Ap = nullptr;
    const uint4* Bp = nullptr;
    long ap_bs = 0, bp_bs = 0;  // uint4 per batch
    int M = 0, N = 0, K = 0, batch = 1;  // static row / column maxima; K in {64, 128, 256}
    const int* mcnt = nullptr;           // rows of batch b = mcnt[b * cnt_stride] (device), nullptr: M
    const int* ncnt = nullptr;
    int cnt_stride = 0;
    float alpha = 1.0f;  // > 0; the nearest-neighbour modes take 1
    int nchunk = 1;  // column chunks per row block (fills the chip when batch * M / 128 is small); row outputs have one slot per chunk
    // row outputs, slot (b, chunk): value arrays [batch][nchunk][r_pitch]
    float* r0 = nullptr;  // NN: best      LSE: max   BEST: best value
    float* r1 = nullptr;  // NN: second    LSE: sum
    int* ri = nullptr;    // NN / BEST: index of the best
    long r_pitch = 0;
    // column outputs, slot (b, row block): [batch][ceil(M / 128)][c_pitch]
    float* c0 = nullptr;
    float* c1 = nullptr;
    int* ci = nullptr;  // NN / LGBEST
    long c_pitch = 0;
    // pass-2 inputs (per batch [r_pitch] / [c_pitch]): SR_DSBEST: max and sum; SR_LGBEST: max and LOG sum
    const float *rmax = nullptr, *rsum = nullptr, *cmax = nullptr, *csum = nullptr;
    const float *l0 = nullptr, *l1 = nullptr;  // SR_LGBEST: logsigmoid(z0) of row i of batch b at l0[b * l0_bs + i], logsigmoid(z1) likewise
    long l0_bs = 0, l1_bs = 0;
    // SR_DSBEST: [batch][ceil(M / 128)][ceil(N / 128)] tile flags (0 = no entry of the tile can exceed the threshold: skipped), or nullptr
    const unsigned char* flags = nullptr;
    int dbg = 0;  // lab switch (IMCUI_SR_DBG, read once): bit 0 = skip the per-column part of the epilogue, bit 1 = skip the per-row part (both: WRONG results, timing only), bit 2 = the two column halves half a tile apart (results unchanged; slower: simred.hip), bit 3 = 32-row wave tiles at every width (results unchanged; A/B of the 64-row tiles of K <= 128)
};

// pack `rows` x K floats (element (r, k) of batch b at X[b * xbs + r * ldr + k * ldk]) into the fragment order above.
// f32 = the exact-f32 arithmetic (pieces = the two k-quads of a lane), otherwise f16 hi / lo planes of the 3-product split.
void simred_pack(imcui_hip_s* h, const float* X, long ldr, long ldk, long xbs, int rows, int K, int batch, const int* cnt, int cnt_stride, uint4* out,
                 hipStream_t stream);
static inline size_t simred_packed_uint4(int rows, int K) { return (size_t)((rows + SR_TILE - 1) / SR_TILE) * SR_TILE / 32 * (K / 16) * 2 * 64; }
static inline bool simred_ok(int K) { return K == 64 || K == 128 || K == 256; }
// column chunks per row block that give the launch >= ~2 workgroups per CU (a function of the sizes only)
int simred_chunks(int batch, int M, int N);
int simred_launch(imcui_hip_s* h, const SimRedP& p, hipStream_t stream);

// ---- dual-softmax coarse matching (LoFTR, EfficientLoFTR, the DualSoftMax plugin) without the similarity matrix
//   pass 1  SR_LSE     row / column soft-max statistics
//   flags   which 128 x 128 tiles can hold a confidence above the threshold
//   pass 2  SR_DSBEST  the flagged tiles again: confidence once per element, row best (value, first column), column best
// Outputs as the two-pass kernels on the materialised matrix produced them: rmax / rsum [B][L], cmax / csum [B][S], best / bestj [B][L]
// (best = -1, bestj = 0x7fffffff for a row without any confidence above the threshold's tiles), cbest [B][S] (-1 likewise).
struct SimDsWs {
    uint4 *ap, *bp;
    float *rp0, *rp1, *cp0, *cp1;
    int* rpj;
    unsigned char* flags;
    int nchunk, nrb, nct;
};
void simred_ds_carve(WsAlloc& a, int B, int L, int S, int K, SimDsWs& w);
int simred_dual_softmax(imcui_hip_s* h, const SimDsWs& w, const float* fa, long lda, long a_bs, const float* fb, long ldb, long b_bs, int B, int L, int S, int K,
                        float alpha, float thr, float* rmax, float* rsum, float* cmax, float* csum, float* best, int* bestj, float* cbest, hipStream_t stream);
