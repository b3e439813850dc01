// Flash-style softmax(Q K^T) V on the f32 matrix cores for LightGlue (head_dim 64).
#pragma once
#include "common.h"

struct AttnP {
    // head-major operands: [seq][head][rows_per_seq][64]; Q is pre-scaled (and RoPE'd)
    const float* Q = nullptr;
    const float* K = nullptr;
    // precision 0: Q, K, V f32 [seq][head][rows][64].
    // precision 1: each buffer holds two f16 planes (hi, then lo at +nseq*heads*rows*64 halves) of
    //              Q / K [seq][head][rows][64] and of V^T [seq][head][64][rows]  (see gemm.hip split_out)
    const float* V = nullptr;
    float* O = nullptr;         // token-major context [seq * rows_per_seq + i][heads * 64]
    const int* cnt = nullptr;   // valid rows per sequence
    const int* active = nullptr;  // per pair (seq >> 1), may be null
    int nseq = 0, heads = 4, rows_per_seq = 0;
    int cross = 0;  // 0: keys/values of the same sequence; 1: of the partner image (seq ^ 1); 2: of sequence (seq + nseq / 2) % nseq
    // 1: Q (hence Q.K) was pre-multiplied by log2(e) by the producer, soft-max = exp2(s - max): the split kernel then needs no
    // multiply per probability (the layers set it; the C-ABI building block passes natural-log operands, 0)
    int log2_domain = 0;
    // 1 (split mode + log2_domain only): one f16 product per element pair, hi planes only (DUSt3R's opt-in single-product arithmetic)
    int single = 0;
    // >= 0: kernel variant for THIS launch (split mode, log2 domain) instead of the handle's "attn_variant" -- LightGlue routes its self and
    // cross blocks / a subset of its layers separately ("attn_variant_self", "attn_variant_cross", "attn_mix_layers": round 5)
    int variant = -1;
    // variant 9 (attention_mx.hip): the fp6 planes of V^T, ATTN_V6_TILE_BYTES per (sequence, head, 64-key tile) -- caller-provided scratch of
    // attn_v6_bytes(nseq, heads, rows_per_seq) bytes, filled by the launch itself from the f16 planes of V (v6_ready = 1: the producer of V
    // already wrote it).  nullptr: variant 9 is not available for this launch and falls back to variant 8.
    unsigned char* V6 = nullptr;
    int v6_ready = 0;
    // key-split launches (attention.hip, round 6): caller-provided scratch of attn_part_floats(nseq, heads, rows_per_seq) floats for the per-chunk
    // partials (O_c [chunks][nseq][heads][rows][64], then (m_c, l_c) pairs).  nullptr: one workgroup walks all the keys of its queries (same rows).
    float* part = nullptr;
};
#define ATTN_MAX_CHUNKS(R) (((R) + 511) / 512)  // chunks of 8 key tiles of 64
static inline size_t attn_part_floats(int nseq, int heads, int rows_per_seq) { return (size_t)ATTN_MAX_CHUNKS(rows_per_seq) * nseq * heads * rows_per_seq * 66; }
#define ATTN_V6_TILE_BYTES 6400  // hi6 [128 slots][16 B] + [128][8 B], lo6 likewise, 256 scale bytes
static inline size_t attn_v6_bytes(int nseq, int heads, int rows_per_seq) { return (size_t)nseq * heads * (rows_per_seq / 64) * ATTN_V6_TILE_BYTES; }
int attention_launch(imcui_hip_s* h, const AttnP& p, hipStream_t stream);
int attention_mx_launch(imcui_hip_s* h, const AttnP& p, hipStream_t stream);  // attention_mx.hip
