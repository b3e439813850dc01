// Flash attention in exact f32 on v_mfma_f32_32x32x2_f32 (gfx950), head_dim = 64.
//
// A workgroup (4 waves) owns 128 queries of one (sequence, head); wave w owns 32 of them.
// Both products are computed TRANSPOSED so that the soft-max statistics of a query are
// lane-local and P never leaves registers:
//   S^T[key, q] = K . Q^T      A = K tile rows (LDS, [d-quad][key][4]),  B = Q (32 registers)
//   O^T[d,  q]  = V^T . P^T    A = V tile (LDS, [key][d]),              B = P = the S^T registers
// In the C/D fragment layout column = lane & 31 = query, so max / sum / rescale of a query
// touch only that lane (plus one cross-half exchange), and the S^T accumulator registers are
// already the B operand of the second product (its k index = key = the register's row).
// K/V tiles of 64 keys are prefetched through registers while the previous tile is multiplied.
#include <stdlib.h>

#include <type_traits>

#include "attention.h"

#define KT 64        // keys per tile
#define KSTR 65      // padded key stride of the K image (float4 units)

// sequence whose keys / values sequence `seq` attends to
__device__ __forceinline__ int attn_key_seq(const AttnP& p, int seq) {
    if (p.cross == 0) return seq;
    if (p.cross == 1) return seq ^ 1;
    const int half = p.nseq >> 1;  // cross == 2: the sequences are [views 1 of all pairs | views 2 of all pairs]
    return seq < half ? seq + half : seq - half;
}

__global__ __launch_bounds__(256, 2) void attn_kernel(AttnP p) {
    // one LDS object: K image [d-quad][key] (float4), then V [key][d]; the front of it is
    // reused as the per-wave output transpose buffer after the last tile
    __shared__ float4 smem4[16 * KSTR + KT * 16];
    float4* Ks = smem4;
    float* Vs = reinterpret_cast<float*>(smem4 + 16 * KSTR);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, so the nqb query blocks of one
    // (sequence, head) are given ids that share b % 8 and read their K/V through ONE L2.
    const int nqb = p.rows_per_seq >> 7;
    const int bid = blockIdx.x;
    const int grp = (bid / (8 * nqb)) * 8 + (bid & 7);
    const int seq = grp / p.heads, head = grp - seq * p.heads;
    const int q0 = ((bid >> 3) % nqb) * 128;
    const int nq = p.cnt[seq];
    if (q0 >= nq) return;
    if (p.active && p.active[seq >> 1] == 0) return;
    const int kseq = attn_key_seq(p, seq);
    const int nk = p.cnt[kseq];
    const int R = p.rows_per_seq;

    const float* Qb = p.Q + ((size_t)seq * p.heads + head) * R * 64;
    const float* Kb = p.K + ((size_t)kseq * p.heads + head) * R * 64;
    const float* Vb = p.V + ((size_t)kseq * p.heads + head) * R * 64;

    // this lane's query row: 32 of its 64 dims (d = 32*hi + 0..31)
    const int qrow = q0 + wid * 32 + lo;
    float qreg[32];
    {
        const float4* qsrc = reinterpret_cast<const float4*>(Qb + (size_t)min(qrow, R - 1) * 64 + 32 * hi);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 v = qsrc[t];
            qreg[4 * t + 0] = v.x;
            qreg[4 * t + 1] = v.y;
            qreg[4 * t + 2] = v.z;
            qreg[4 * t + 3] = v.w;
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // staging: 64 keys x 16 float4 per operand = 1024 float4 -> 4 per thread
    const int s_dq = tid & 15, s_key = tid >> 4;  // key + 16*it
    float4 rk[4], rv[4];
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int key = k0 + s_key + 16 * it;
            if (key < nk) {
                rk[it] = *reinterpret_cast<const float4*>(Kb + (size_t)key * 64 + s_dq * 4);
                rv[it] = *reinterpret_cast<const float4*>(Vb + (size_t)key * 64 + s_dq * 4);
            } else {  // rows past the sequence end may hold anything (even NaN): feed zeros
                rk[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                rv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    const int ntile = (nk + KT - 1) / KT;
    if (ntile > 0) load_tile(0);
    for (int tile = 0; tile < ntile; ++tile) {
        const int k0 = tile * KT;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int key = s_key + 16 * it;
            Ks[s_dq * KSTR + key] = rk[it];
            *reinterpret_cast<float4*>(&Vs[key * 64 + s_dq * 4]) = rv[it];
        }
        __syncthreads();
        if (tile + 1 < ntile) load_tile(k0 + KT);

        // ---- S^T = K . Q^T  (two 32-key fragments)
        f32x16 s[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[f][r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 a = Ks[(8 * hi + t) * KSTR + 32 * f + lo];
                s[f] = mfma32(a.x, qreg[4 * t + 0], s[f]);
                s[f] = mfma32(a.y, qreg[4 * t + 1], s[f]);
                s[f] = mfma32(a.z, qreg[4 * t + 2], s[f]);
                s[f] = mfma32(a.w, qreg[4 * t + 3], s[f]);
            }
        }
        // ---- online soft-max for query `lo` (keys of this lane: 2 x 16 registers)
        if (k0 + KT > nk) {  // only the last tile can hold keys past the sequence end
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * f + frag_row(r, hi) >= nk) s[f][r] = -INFINITY;
        }
        float m_t = -INFINITY;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, s[f][r]);
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        const float m_new = fmaxf(m_run, m_t);  // finite: every tile holds >= 1 valid key
        const bool l2d = p.log2_domain != 0;         // Q pre-multiplied by log2(e): the exponentials are base 2
        const float alpha = l2d ? exp2f(m_run - m_new) : expf(m_run - m_new);
        float l_t = 0.0f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = l2d ? exp2f(s[f][r] - m_new) : expf(s[f][r] - m_new);
                s[f][r] = pv;
                l_t += pv;
            }
        l_run = l_run * alpha + l_t;
        m_run = m_new;
        if (__ballot(alpha != 1.0f) != 0ull) {  // the running max of some query moved
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * f + frag_row(r, hi);
                const float pb = s[f][r];
                o[0] = mfma32(Vs[key * 64 + lo], pb, o[0]);
                o[1] = mfma32(Vs[key * 64 + 32 + lo], pb, o[1]);
            }
    }

    // ---- normalise and write: transpose through LDS so each query row is stored contiguously
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (ntile > 0) ? 1.0f / l_tot : 0.0f;
    __syncthreads();
    float* Os = reinterpret_cast<float*>(smem4) + wid * (32 * 33);  // per wave [query][33]
    const int H64 = p.heads * 64;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        // write O^T fragment f: element (d = frag_row(r,hi), q = lo) -> Os[q][d]
#pragma unroll
        for (int r = 0; r < 16; ++r) Os[lo * 33 + frag_row(r, hi)] = o[f][r] * inv;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are done
        __builtin_amdgcn_wave_barrier();
        // read back: lane covers d = lo for query rows q = hi, hi+2, ...
        // read back: 8 lanes cover the 32 dims of one query -> one 16-byte store per lane
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = 8 * qq + (lane >> 3), d4 = (lane & 7) * 4;
            const int row = q0 + wid * 32 + q;
            const float4 v = make_float4(Os[q * 33 + d4], Os[q * 33 + d4 + 1], Os[q * 33 + d4 + 2], Os[q * 33 + d4 + 3]);
            if (row < nq) *reinterpret_cast<float4*>(p.O + ((size_t)seq * R + row) * H64 + head * 64 + 32 * f + d4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ 3 x f16 split variant
// Same transposed flash structure on v_mfma_f32_32x32x16_f16: K, V^T tiles and Q are split into
// f16 (hi, lo) pairs when they are staged, P is split in registers (scaled by 2^14 so that the
// small probabilities keep their low part out of the f16 subnormal range), every product is
// xh*yh + xh*yl + xl*yh with f32 accumulation.  V is consumed TRANSPOSED ([d][key], written that
// way by the projection epilogue) because the P^T registers of a lane hold keys in groups of
// four (C/D rows), so the matching A operand needs two 8-byte runs of consecutive keys per d.
// V^T LDS image: row d = 4 groups of 16 keys, each group stored as two 16-byte halves
// {keys 0-3, 8-11} and {keys 4-7, 12-15} (= what the hi = 0 / hi = 1 half-waves consume in one
// MFMA step), + one 16-byte pad per row -> a fragment is ONE conflict-free ds_read_b128.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define VSTR 9  // V^T row stride in 16-byte units
#define P_SHIFT 14.0f
#define LOG2E 1.44269504088896340736f

// Operands arrive PRE-SPLIT from the projection epilogue (gemm.hip, split_out): each of Q, K,
// V^T is two f16 planes (hi then lo) in the buffer that holds the f32 tensor in exact mode, so a
// K/V element is split once instead of once per query block and staging is a pure copy.
//
// On this part the matrix pipe and every other instruction a SIMD's waves issue take turns rather than overlap
// (tools/overlap_lab.hip, tools/attn_pipe_lab): a key tile costs the 48 MFMAs plus whatever else is issued, so the
// loop is trimmed to the instructions the arithmetic needs:
//   * K / V^T tiles are fetched with buffer loads (wave-uniform descriptor + per-thread offset + scalar tile offset):
//     no per-load 64-bit address arithmetic;
//   * `log2_domain` callers (LightGlue / SuperGlue) deliver Q pre-multiplied by log2(e), and the S accumulators start
//     from c = 14 - m_ref instead of 0, so the MFMAs themselves produce s log2e - m_ref log2e + 14 and a probability is
//     ONE v_exp_f32 -- no multiply-add per element.  m_ref is the running maximum as of the last tile that moved it by
//     more than 1.5 (in log2 units): a smaller growth only makes the tile's numbers up to 2^1.5 larger, well inside the
//     f16 range of the split (hi < 2^16), and O / l is invariant to the reference.  A larger growth takes the rescale
//     path (subtract the excess before the exponential, scale O and l), the first tile always does.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define DEFER_THR 1.5f

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    // the pointer only depends on blockIdx: tell the compiler (buffer descriptors must live in SGPRs)
    const size_t v = (size_t)p;
    // (the builtin returns int: go through unsigned before widening, or a low word with its top bit set sign-extends
    // into the high word)
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffu));
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void*)((size_t)lo32 | ((size_t)hi32 << 32));
}

// VAR (option "attn_variant"; rounds 3-5 also carried wave-priority schedules 1-3, a double-buffered tile 5 and the un-pipelined
// schedules 0 / 6 -- all measured within noise or slower, removed in round 6; the option values map onto the two survivors):
// VAR 4 (AttnP.single, DUSt3R's opt-in `arith = 1`): ONE f16 product per element pair -- only the hi planes of Q / K / V are loaded,
// staged and multiplied, P is rounded to the nearest f16 (f32 accumulation and an f32 normaliser as before).  Not a parity mode.
// VAR 7 (option values 6, 7; the audited "reduced-product P.V"): K.Q^T keeps its three products (an error there is an
// error of the EXPONENT), V^T.P^T runs as (vh + vl) . ph with the probabilities rounded to the nearest f16: two MFMAs per product
// instead of three (48 -> 40 per key tile) and one v_cvt_pk_f16_f32 per probability pair instead of v_cvt_pkrtz + two v_fma_mix.  The
// normaliser is the sum of the ROUNDED probabilities (v_dot2_f32_f16 against (1, 1): exact products, f32 accumulation, the same
// instruction count as the packed adds it replaces), so O = sum p~_j v_j / sum p~_j is an exact weighted mean with weights perturbed
// by at most 2^-12 relative: the error is sum_j p_j e_j (v_j - O) -- it vanishes for a peaked row (one dominant key: v_j = O) and
// averages down like 1 / sqrt(n) for a flat one; the worst case is two equal keys with different values, 2^-13 |v_0 - v_1|.
// Audited per block in round 5 (profiles/r05_lab_attention_mix.txt): the default of LightGlue's CROSS blocks.
// VAR 8 (option values 0-3, 5, 8; the default): three products in both contractions, the K fragments of step i + 1 requested before the
// MFMAs of step i (hipcc's own schedule is read -> wait -> 3 MFMAs on one register quad, every LDS latency exposed), the maximum of the
// first S fragment taken under the MFMAs of the second, and the cross-half maximum by v_permlane32_swap instead of ds_bpermute.
// VAR 0 remains for the natural-log entry point (L2D = false: the C-ABI building block) with the plain schedule.
//
// CHUNKS (round 6, log2-domain kernels).  The keys of a sequence are processed in chunks of ATTN_CHUNK_TILES * 64 = 512: every chunk starts
// the online soft-max from scratch (its first tile sets the reference maximum) and ends as a partial (O_c, m_c, l_c); the partials are
// folded IN CHUNK ORDER with `attn_fold`.  The result no longer depends on how the chunks are distributed: one workgroup may walk all of
// them (SPLIT = false: the fold runs in registers after every eighth tile -- the throughput geometry), or each chunk may get its own
// workgroup (SPLIT = true) that writes its partial to a scratch buffer for `attn_combine_kernel`, which folds with the same function in
// the same order: bitwise the same rows.  The split launch is for grids that leave the chip empty -- ONE LightGlue pair is 128 workgroups
// of 4 waves on 256 CUs, each walking 32 tiles alone on its SIMD (51.7 us per launch, a third of the one-pair step: VERDICT round 5, weak
// 9) -- and because both geometries compute the same function, a pair's result does not depend on the batch it rides in.
#define ATTN_CHUNK_TILES 8
// (O, m, l) <- (O, m, l) (+) (O_c, m_c, l_c): both partials carry the factor 2^(14 - their reference maximum).  Explicit operations only
// (no contraction the two call sites could resolve differently).
__device__ __forceinline__ void attn_fold_scales(float ma, float mc, float& m_new, float& a, float& b) {
    m_new = fmaxf(ma, mc);
    a = __builtin_amdgcn_exp2f(ma - m_new);
    b = __builtin_amdgcn_exp2f(mc - m_new);
}
__device__ __forceinline__ float attn_fold1(float acc, float a, float part, float b) {
    const float t = acc * a;
    return __builtin_fmaf(part, b, t);
}

template <bool L2D, int VAR, bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_split_kernel(AttnP p) {
    constexpr bool SINGLE = (VAR == 4);
    constexpr bool PV2 = (VAR == 7);                // P rounded to f16, V split: two products per element pair in V^T.P^T
    constexpr bool KPRE = (VAR == 7 || VAR == 8);   // the K fragments of step i + 1 requested before the MFMAs of step i
    constexpr bool CHUNKED = L2D;                   // (the natural-log entry point keeps one pass over all keys)
    static_assert(!SPLIT || CHUNKED, "a key-split launch needs the chunked arithmetic");
    constexpr int TILE_U4 = 2 * 8 * KSTR + 2 * 64 * VSTR;
    __shared__ uint4 smem4[TILE_U4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, so the nqb query blocks of one
    // (sequence, head) are given ids that share b % 8 and read their K/V through ONE L2.
    // SPLIT: the chunk index is the slowest grid coordinate (the workgroups of one chunk keep the XCD-aware order among themselves).
    const int nqb = p.rows_per_seq >> 7;
    const int nwg = nqb * p.heads * p.nseq;
    const int bid = SPLIT ? (int)blockIdx.x % nwg : (int)blockIdx.x;
    const int chunk = SPLIT ? (int)blockIdx.x / nwg : 0;
    const int grp = (bid / (8 * nqb)) * 8 + (bid & 7);
    const int seq = grp / p.heads, head = grp - seq * p.heads;
    const int q0 = ((bid >> 3) % nqb) * 128;
    const int nq = p.cnt[seq];
    if (q0 >= nq) return;
    if (p.active && p.active[seq >> 1] == 0) return;
    const int kseq = attn_key_seq(p, seq);
    const int nk = p.cnt[kseq];
    const int R = p.rows_per_seq;
    const size_t plane = (size_t)p.nseq * p.heads * R * 64;  // halves per plane

    const unsigned short* Qh = reinterpret_cast<const unsigned short*>(p.Q) + ((size_t)seq * p.heads + head) * R * 64;
    const unsigned short* Kg = reinterpret_cast<const unsigned short*>(p.K) + ((size_t)kseq * p.heads + head) * R * 64;
    const unsigned short* Vg = reinterpret_cast<const unsigned short*>(p.V) + ((size_t)kseq * p.heads + head) * 64 * R;
    // one (sequence, head) slab of R x 64 halves per plane and operand; reads past it return zeros
    const unsigned slab = (unsigned)R * 64u * 2u;
    const __amdgpu_buffer_rsrc_t rKh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(Kg)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rKl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(Kg + plane)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rVh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(Vg)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rVl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(Vg + plane)), 0, slab, 0x00020000);

    // Q fragment of this lane: query q0 + wid*32 + lo, dims 16s + 8hi .. +7
    uint4 qh[4], ql[4];
    {
        const int qrow = min(q0 + wid * 32 + lo, R - 1);
        const unsigned short* qsrc = Qh + (size_t)qrow * 64 + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *reinterpret_cast<const uint4*>(qsrc + 16 * s);
            if constexpr (!SINGLE) ql[s] = *reinterpret_cast<const uint4*>(qsrc + plane + 16 * s);
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f][r] = 0.0f;
    // m_ref: reference maximum (log2 units when log2_domain) the numbers of O and l are scaled by; l_run: running sum
    float m_ref = 0.0f, l_run = 0.0f;
    constexpr bool l2d = L2D;  // Q pre-multiplied by log2(e) by the producer (the layers) or natural-log operands (C-ABI block)
    f32x16 cinit;  // start value of the S accumulators: 0, or 14 - m_ref once a log2-domain reference exists
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.0f;

    // staging registers (named, see gemm.hip): K 64 keys x 8 octets x 2 planes = 4 x 16 B per thread,
    // V^T 64 d x 16 key-quads x 2 planes = 8 x 8 B per thread
    uint4 rk0, rk1, rk2, rk3;
    uint2 rv0, rv1, rv2, rv3, rv4, rv5, rv6, rv7;
    const int k_key = tid >> 3, k_oc = tid & 7;  // + 32 keys for the second item
    const int v_d = tid >> 4, v_kq = tid & 15;   // + 16 d per item
    // per-thread byte offsets inside a slab; the tile adds a scalar (k0 keys = 128 k0 bytes of K, 2 k0 bytes of a V^T row)
    const unsigned ko0 = (unsigned)(k_key * 64 + k_oc * 8) * 2u, ko1 = ko0 + 32u * 64u * 2u;
    const unsigned vo0 = ((unsigned)v_d * (unsigned)R + (unsigned)v_kq * 4u) * 2u;
    const unsigned vo1 = vo0 + 16u * (unsigned)R * 2u, vo2 = vo0 + 32u * (unsigned)R * 2u, vo3 = vo0 + 48u * (unsigned)R * 2u;
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned vo, unsigned so) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto ld2 = [](const __amdgpu_buffer_rsrc_t& r, unsigned vo, unsigned so) __attribute__((always_inline)) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
        return make_uint2(v.x, v.y);
    };
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
        const unsigned sk = (unsigned)k0 * 128u, sv = (unsigned)k0 * 2u;
        rk0 = ld4(rKh, ko0, sk);
        rk2 = ld4(rKh, ko1, sk);
        rv0 = ld2(rVh, vo0, sv);
        rv2 = ld2(rVh, vo1, sv);
        rv4 = ld2(rVh, vo2, sv);
        rv6 = ld2(rVh, vo3, sv);
        if constexpr (!SINGLE) {
            rk1 = ld4(rKl, ko0, sk);
            rk3 = ld4(rKl, ko1, sk);
            rv1 = ld2(rVl, vo0, sv);
            rv3 = ld2(rVl, vo1, sv);
            rv5 = ld2(rVl, vo2, sv);
            rv7 = ld2(rVl, vo3, sv);
        }
    };
    // key quad kq -> group kq>>2, 16-byte slot (kq&1), 8-byte half ((kq>>1)&1)
    const int v_u = ((v_kq >> 2) * 2 + (v_kq & 1)) * 2 + ((v_kq >> 1) & 1);
    auto store_tile = [&](int k0, auto tail) __attribute__((always_inline)) {
        uint4* Kh = smem4;
        uint4* Kl = Kh + 8 * KSTR;
        uint4* Vh = Kh + 2 * 8 * KSTR;
        uint4* Vl = Vh + 64 * VSTR;
        if (decltype(tail)::value) {
            // keys past the sequence end may hold anything (even NaN bit patterns): zero them
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            if (k0 + k_key >= nk) rk0 = rk1 = z4;
            if (k0 + k_key + 32 >= nk) rk2 = rk3 = z4;
            const int kk = k0 + v_kq * 4;
            const unsigned mx = (kk + 0 < nk ? 0x0000FFFFu : 0u) | (kk + 1 < nk ? 0xFFFF0000u : 0u);
            const unsigned my = (kk + 2 < nk ? 0x0000FFFFu : 0u) | (kk + 3 < nk ? 0xFFFF0000u : 0u);
            rv0.x &= mx; rv0.y &= my; rv1.x &= mx; rv1.y &= my; rv2.x &= mx; rv2.y &= my; rv3.x &= mx; rv3.y &= my;
            rv4.x &= mx; rv4.y &= my; rv5.x &= mx; rv5.y &= my; rv6.x &= mx; rv6.y &= my; rv7.x &= mx; rv7.y &= my;
        }
        Kh[k_oc * KSTR + k_key] = rk0;
        Kh[k_oc * KSTR + k_key + 32] = rk2;
        uint2* vh2 = reinterpret_cast<uint2*>(Vh);
        uint2* vl2 = reinterpret_cast<uint2*>(Vl);
        vh2[(v_d * VSTR) * 2 + v_u] = rv0;
        vh2[((v_d + 16) * VSTR) * 2 + v_u] = rv2;
        vh2[((v_d + 32) * VSTR) * 2 + v_u] = rv4;
        vh2[((v_d + 48) * VSTR) * 2 + v_u] = rv6;
        if constexpr (!SINGLE) {
            Kl[k_oc * KSTR + k_key] = rk1;
            Kl[k_oc * KSTR + k_key + 32] = rk3;
            vl2[(v_d * VSTR) * 2 + v_u] = rv1;
            vl2[((v_d + 16) * VSTR) * 2 + v_u] = rv3;
            vl2[((v_d + 32) * VSTR) * 2 + v_u] = rv5;
            vl2[((v_d + 48) * VSTR) * 2 + v_u] = rv7;
        }
    };

    // one 64-key tile: S^T = K.Q^T, online soft-max, O^T += V^T.P^T
    //   first: no reference maximum yet (accumulators start from 0, the tile's own maximum becomes the reference) -- the first tile of the
    //          sequence or of a chunk.  A RUN-TIME flag (wave-uniform): one instance of this code serves both cases, which keeps the chunk starts
    //          of the split and the unsplit geometry on literally the same instructions (and the kernel at two instances instead of four)
    //   TAIL : the tile may hold keys past the sequence end
    auto compute_tile = [&](int k0, const bool first, auto tail) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail)::value;
        const uint4* Kh = smem4;
        const uint4* Kl = Kh + 8 * KSTR;
        const uint4* Vh = Kh + 2 * 8 * KSTR;
        const uint4* Vl = Vh + 64 * VSTR;
        // with a reference the accumulation starts from cinit = 14 - m_ref (the C operand of the first MFMA of both
        // fragments; kept in registers across tiles, rewritten only when the reference moves): the MFMAs deliver the
        // exponent directly
        f32x16 s[2];
        float m_pre = -INFINITY;  // KPRE: maximum of fragment 0, taken while fragment 1 is multiplied
        if (first) {  // (a later chunk starts from scratch: cinit still holds the previous chunk's reference)
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[r] = 0.0f;
        }
        const f32x16& c0 = cinit;
        if constexpr (KPRE) {
            // hipcc's schedule of the plain loop below is read -> wait -> three MFMAs, eight times per fragment, on ONE register
            // quad: every LDS latency is exposed.  Here the (hi, lo) fragments of step i + 1 are requested before the MFMAs of step
            // i are issued (two register sets, order pinned with sched_barrier).
            uint4 kh[2], kl[2];
            kh[0] = Kh[hi * KSTR + lo];
            kl[0] = Kl[hi * KSTR + lo];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = i >> 2, st = i & 3;
                if (i + 1 < 8) {
                    const int fn = (i + 1) >> 2, sn = (i + 1) & 3;
                    kh[(i + 1) & 1] = Kh[(2 * sn + hi) * KSTR + 32 * fn + lo];
                    kl[(i + 1) & 1] = Kl[(2 * sn + hi) * KSTR + 32 * fn + lo];
                }
                __builtin_amdgcn_sched_barrier(0);
                s[f] = mfma16(kl[i & 1], qh[st], st == 0 ? c0 : s[f]);
                s[f] = mfma16(kh[i & 1], ql[st], s[f]);
                s[f] = mfma16(kh[i & 1], qh[st], s[f]);
                if (!TAIL && i >= 4) {  // the maximum of the finished first fragment rides under the MFMAs of the second
                    const int r0 = 4 * (i - 4);
                    m_pre = fmaxf(fmaxf(m_pre, s[0][r0]), fmaxf(s[0][r0 + 1], fmaxf(s[0][r0 + 2], s[0][r0 + 3])));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const uint4 ah = Kh[(2 * st + hi) * KSTR + 32 * f + lo];
                if constexpr (SINGLE) {
                    s[f] = mfma16(ah, qh[st], st == 0 ? c0 : s[f]);
                } else {
                    const uint4 al = Kl[(2 * st + hi) * KSTR + 32 * f + lo];
                    s[f] = mfma16(al, qh[st], st == 0 ? c0 : s[f]);
                    s[f] = mfma16(ah, ql[st], s[f]);
                    s[f] = mfma16(ah, qh[st], s[f]);
                }
            }
        }
        if (TAIL) {  // only the last tile can hold keys past the sequence end
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * f + frag_row(r, hi) >= nk) s[f][r] = -INFINITY;
        }
        float m_t = m_pre;
#pragma unroll
        for (int f = (KPRE && !TAIL) ? 1 : 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, s[f][r]);
        if constexpr (KPRE) {
            // the other half-wave's maximum by v_permlane32_swap (one vector instruction) instead of ds_bpermute + lgkmcnt(0)
            // (inline asm: hipcc 7.2 folds fmaxf(r[0], r[1]) of __builtin_amdgcn_permlane32_swap away -- it keeps the swap and drops
            // the second result -- which left each half-wave with the LOWER half's maximum; found in the ISA before it ever ran)
            float ma = m_t, mb = m_t;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(ma), "+v"(mb));
            m_t = ma;
        } else {
            m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        }
        f32x2 la = {0.0f, 0.0f}, lb = {0.0f, 0.0f};
        if constexpr (!l2d) {
            // natural-log operands (building-block entry point): p * 2^14 = exp2(s*log2e - m*log2e + 14), one fma + one v_exp_f32
            const float m_new = first ? m_t : fmaxf(m_ref, m_t);
            const float alpha = first ? 0.0f : __builtin_amdgcn_exp2f((m_ref - m_new) * LOG2E);
            const float bias = P_SHIFT - m_new * LOG2E;
            const f32x2 k2 = {LOG2E, LOG2E}, b2 = {bias, bias};
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const f32x2 x = {s[f][r], s[f][r + 1]}, y = {s[f][r + 2], s[f][r + 3]};
                    const f32x2 a = __builtin_elementwise_fma(x, k2, b2), b = __builtin_elementwise_fma(y, k2, b2);
                    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    const f32x2 g = {__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
                    la += e;
                    lb += g;
                    s[f][r] = e[0];
                    s[f][r + 1] = e[1];
                    s[f][r + 2] = g[0];
                    s[f][r + 3] = g[1];
                }
            l_run = l_run * alpha + ((la[0] + la[1]) + (lb[0] + lb[1]));
            m_ref = m_new;
            if (!first && __ballot(alpha != 1.0f) != 0ull) {  // the running max of some query moved
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
            }
        } else {
            // log2 domain.  FIRST: s holds the raw exponents, the tile maximum becomes the reference.  Later tiles: s already
            // holds s - m_ref + 14; only a growth of the maximum beyond DEFER_THR needs work before the exponential.
            // what to take off this query's exponents: its own decision only (a lane whose maximum grew by less than the
            // threshold subtracts 0 and scales by 1 even when another query of the wave takes the branch, so a result
            // never depends on which other rows -- padding included -- share the wave)
            const float excess = first ? m_t - P_SHIFT : (m_t > P_SHIFT + DEFER_THR ? m_t - P_SHIFT : 0.0f);
            const bool shift = first || (__ballot(excess != 0.0f) != 0ull);
            if (shift) {
                const f32x2 d2 = {excess, excess};
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 x = {s[f][r], s[f][r + 1]};
                        const f32x2 y = x - d2;
                        s[f][r] = y[0];
                        s[f][r + 1] = y[1];
                    }
                if (first) {
                    m_ref = m_t;  // the tile maximum (accumulators started from 0) is the first reference (O and l are zero: nothing to rescale)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cinit[r] = P_SHIFT - m_ref;
                } else {
                    const float alpha = __builtin_amdgcn_exp2f(-excess);
                    m_ref += excess;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cinit[r] = P_SHIFT - m_ref;
                    l_run *= alpha;
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
                }
            }
            if constexpr (PV2) {
                // the probabilities are summed AFTER their rounding to f16 (below, next to the conversion)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[f][r] = __builtin_amdgcn_exp2f(s[f][r]);
            } else {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const f32x2 e = {__builtin_amdgcn_exp2f(s[f][r]), __builtin_amdgcn_exp2f(s[f][r + 1])};
                    const f32x2 g = {__builtin_amdgcn_exp2f(s[f][r + 2]), __builtin_amdgcn_exp2f(s[f][r + 3])};
                    la += e;
                    lb += g;
                    s[f][r] = e[0];
                    s[f][r + 1] = e[1];
                    s[f][r + 2] = g[0];
                    s[f][r + 3] = g[1];
                }
            l_run += (la[0] + la[1]) + (lb[0] + lb[1]);
            }
        }
        // step (f, t) covers keys 32f + 16t + {4hi..4hi+3, 8+4hi..8+4hi+3}
        float lsum0 = 0.0f, lsum1 = 0.0f;  // PV2: this tile's sum of the rounded probabilities
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 ph, pl;
                if constexpr (PV2) {
                    // p * 2^14 <= 2^15.5 < 65504 by the deferred-maximum rule: no clamp; nearest-even
                    ph.x = half2_rtn_nc(s[f][8 * t + 0], s[f][8 * t + 1]);
                    ph.y = half2_rtn_nc(s[f][8 * t + 2], s[f][8 * t + 3]);
                    ph.z = half2_rtn_nc(s[f][8 * t + 4], s[f][8 * t + 5]);
                    ph.w = half2_rtn_nc(s[f][8 * t + 6], s[f][8 * t + 7]);
                    lsum0 = fdot2_ones(ph.x, lsum0);
                    lsum1 = fdot2_ones(ph.y, lsum1);
                    lsum0 = fdot2_ones(ph.z, lsum0);
                    lsum1 = fdot2_ones(ph.w, lsum1);
                } else if constexpr (SINGLE) {
                    ph.x = half2_rtn(s[f][8 * t + 0], s[f][8 * t + 1]);
                    ph.y = half2_rtn(s[f][8 * t + 2], s[f][8 * t + 3]);
                    ph.z = half2_rtn(s[f][8 * t + 4], s[f][8 * t + 5]);
                    ph.w = half2_rtn(s[f][8 * t + 6], s[f][8 * t + 7]);
                } else {
                    split2(s[f][8 * t + 0], s[f][8 * t + 1], ph.x, pl.x);
                    split2(s[f][8 * t + 2], s[f][8 * t + 3], ph.y, pl.y);
                    split2(s[f][8 * t + 4], s[f][8 * t + 5], ph.z, pl.z);
                    split2(s[f][8 * t + 6], s[f][8 * t + 7], ph.w, pl.w);
                }
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const int vi = (32 * df + lo) * VSTR + (2 * f + t) * 2 + hi;
                    const uint4 vh = Vh[vi];
                    if constexpr (SINGLE) {
                        o[df] = mfma16(vh, ph, o[df]);
                    } else if constexpr (PV2) {
                        const uint4 vl = Vl[vi];
                        o[df] = mfma16(vl, ph, o[df]);
                        o[df] = mfma16(vh, ph, o[df]);
                    } else {
                        const uint4 vl = Vl[vi];
                        o[df] = mfma16(vl, ph, o[df]);
                        o[df] = mfma16(vh, pl, o[df]);
                        o[df] = mfma16(vh, ph, o[df]);
                    }
                }
            }
        if constexpr (PV2) l_run += lsum0 + lsum1;
    };

    const int ntile = (nk + KT - 1) / KT;
    // the tiles this workgroup walks: all of them, or the chunk of a key-split launch (a chunk past the sequence's keys has no workgroup output:
    // the combine kernel folds only the chunks that exist)
    const int t_begin = SPLIT ? chunk * ATTN_CHUNK_TILES : 0;
    const int t_end = SPLIT ? min(ntile, t_begin + ATTN_CHUNK_TILES) : ntile;
    if (SPLIT && t_begin >= ntile) return;
    // folded partials of the finished chunks (SPLIT = false, CHUNKED): O, reference maximum, normaliser of the query of this lane
    f32x16 oacc[2];
    float macc = 0.0f, lacc = 0.0f;
    // fold the chunk that just ended into (oacc, macc, lacc) and clear the running state (SPLIT = false, CHUNKED)
    auto fold_chunk = [&](bool first_chunk) __attribute__((always_inline)) {
        const float lc = l_run + __shfl_xor(l_run, 32, 64);
        if (first_chunk) {  // the first chunk IS the running result
            macc = m_ref;
            lacc = lc;
#pragma unroll
            for (int f = 0; f < 2; ++f) oacc[f] = o[f];
        } else {
            float m_new, a, b;
            attn_fold_scales(macc, m_ref, m_new, a, b);
            lacc = attn_fold1(lacc, a, lc, b);
            macc = m_new;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[f][r] = attn_fold1(oacc[f][r], a, o[f][r], b);
        }
        l_run = 0.0f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[f][r] = 0.0f;
    };
    if (t_begin < t_end) load_tile(t_begin * KT);
    // every tile but the sequence's last one: no key beyond the count
    const int t_plain = min(t_end, ntile - 1);
    for (int tile = t_begin; tile < t_plain; ++tile) {
        __syncthreads();  // previous tile fully consumed
        store_tile(tile * KT, std::false_type{});
        __syncthreads();
        load_tile((tile + 1) * KT);  // (tile + 1 <= ntile - 1: the sequence's last tile at most; a split chunk may fetch one tile it does not use)
        compute_tile(tile * KT, CHUNKED ? ((tile - t_begin) % ATTN_CHUNK_TILES) == 0 : tile == 0, std::false_type{});
        if constexpr (CHUNKED && !SPLIT) {
            if (((tile + 1) % ATTN_CHUNK_TILES) == 0) fold_chunk(tile < ATTN_CHUNK_TILES);
        }
    }
    if (t_end == ntile && ntile > 0) {  // the sequence's last tile
        const int tile = ntile - 1;
        __syncthreads();
        store_tile(tile * KT, std::true_type{});
        __syncthreads();
        compute_tile(tile * KT, CHUNKED ? ((tile - t_begin) % ATTN_CHUNK_TILES) == 0 : tile == 0, std::true_type{});
        if constexpr (CHUNKED && !SPLIT) fold_chunk(tile < ATTN_CHUNK_TILES);
    }

    // ---- normalise and write (transpose through LDS so each query row is stored contiguously)
    float inv;
    if constexpr (CHUNKED && !SPLIT) {
        inv = (ntile > 0) ? 1.0f / lacc : 0.0f;  // l carries the same 2^14 as O
#pragma unroll
        for (int f = 0; f < 2; ++f) o[f] = oacc[f];
        if (ntile == 0) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[f][r] = 0.0f;
        }
    } else if constexpr (SPLIT) {
        inv = 1.0f;  // the partial leaves unnormalised
    } else {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        inv = (ntile > 0) ? 1.0f / l_tot : 0.0f;
    }
    __syncthreads();
    float* Os = reinterpret_cast<float*>(smem4) + wid * (32 * 33);
    const int H64 = p.heads * 64;
    // SPLIT: partial O of chunk c at part[((c * nseq + seq) * heads + head) * R + row][64], then (m, l) pairs behind all the O partials
    float* dst = p.O;
    size_t row_stride = (size_t)H64, base = (size_t)seq * R * H64 + (size_t)head * 64;
    if constexpr (SPLIT) {
        const size_t rows_all = (size_t)p.nseq * p.heads * R;
        const size_t prow0 = ((size_t)chunk * p.nseq + seq) * p.heads * R + (size_t)head * R;
        dst = p.part;
        row_stride = 64;
        base = prow0 * 64;
        const float lc = l_run + __shfl_xor(l_run, 32, 64);
        if (hi == 0) {  // one lane per query writes (m_c, l_c)
            float2* ml = reinterpret_cast<float2*>(p.part + (size_t)ATTN_MAX_CHUNKS(R) * rows_all * 64) + prow0 + q0 + wid * 32 + lo;
            *ml = make_float2(m_ref, lc);
        }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Os[lo * 33 + frag_row(r, hi)] = SPLIT ? o[f][r] : o[f][r] * inv;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // read back: 8 lanes cover the 32 dims of one query -> one 16-byte store per lane
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = 8 * qq + (lane >> 3), d4 = (lane & 7) * 4;
            const int row = q0 + wid * 32 + q;
            const float4 v = make_float4(Os[q * 33 + d4], Os[q * 33 + d4 + 1], Os[q * 33 + d4 + 2], Os[q * 33 + d4 + 3]);
            if (row < nq) *reinterpret_cast<float4*>(dst + base + (size_t)row * row_stride + 32 * f + d4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// Folds the chunk partials of a key-split launch in chunk order (the arithmetic of the unsplit kernel's in-register fold, `attn_fold*`) and
// normalises: one thread per (query row, 4 features).
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnP p) {
    const int R = p.rows_per_seq;
    const size_t rows_all = (size_t)p.nseq * p.heads * R;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t prow = t >> 4;  // (seq * heads + head) * R + row
    const int d4 = (int)(t & 15) * 4;
    if (prow >= rows_all) return;
    const int row = (int)(prow % R);
    const int sh = (int)(prow / R), seq = sh / p.heads, head = sh - seq * p.heads;
    if (row >= p.cnt[seq]) return;
    if (p.active && p.active[seq >> 1] == 0) return;
    const int nk = p.cnt[attn_key_seq(p, seq)];
    const int nchunk = (nk + KT * ATTN_CHUNK_TILES - 1) / (KT * ATTN_CHUNK_TILES);
    const float2* ml = reinterpret_cast<const float2*>(p.part + (size_t)ATTN_MAX_CHUNKS(R) * rows_all * 64);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float macc = 0.0f, lacc = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
        const float4 oc = *reinterpret_cast<const float4*>(p.part + ((size_t)c * rows_all + prow) * 64 + d4);
        const float2 mlc = ml[(size_t)c * rows_all + prow];
        if (c == 0) {
            acc = oc;
            macc = mlc.x;
            lacc = mlc.y;
        } else {
            float m_new, a, b;
            attn_fold_scales(macc, mlc.x, m_new, a, b);
            lacc = attn_fold1(lacc, a, mlc.y, b);
            macc = m_new;
            acc = make_float4(attn_fold1(acc.x, a, oc.x, b), attn_fold1(acc.y, a, oc.y, b), attn_fold1(acc.z, a, oc.z, b), attn_fold1(acc.w, a, oc.w, b));
        }
    }
    const float inv = nchunk > 0 ? 1.0f / lacc : 0.0f;
    *reinterpret_cast<float4*>(p.O + ((size_t)seq * R + row) * (p.heads * 64) + head * 64 + d4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

template <int VAR>
static void attn_launch_l2d(const AttnP& p, bool split, hipStream_t stream) {
    const unsigned nwg = (unsigned)((p.rows_per_seq / 128) * p.heads * p.nseq);
    if (split) {
        hipLaunchKernelGGL((attn_split_kernel<true, VAR, true>), dim3(nwg * (unsigned)ATTN_MAX_CHUNKS(p.rows_per_seq)), dim3(256), 0, stream, p);
        const size_t threads = (size_t)p.nseq * p.heads * p.rows_per_seq * 16;
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((attn_split_kernel<true, VAR, false>), dim3(nwg), dim3(256), 0, stream, p);
    }
}

int attention_launch(imcui_hip_s* h, const AttnP& p, hipStream_t stream) {
    if (p.rows_per_seq % 128 != 0) return imcui_set_err(h, IMCUI_ERR_ARG, "attention: rows_per_seq=%d must be a multiple of 128", p.rows_per_seq);
    if (p.nseq <= 0) return IMCUI_OK;
    if ((p.heads * p.nseq) % 8 != 0) return imcui_set_err(h, IMCUI_ERR_ARG, "attention: heads*nseq=%d must be a multiple of 8", p.heads * p.nseq);
    dim3 grid((p.rows_per_seq / 128) * p.heads * p.nseq);
    imcui_prof_begin(h, PROF_ATTN, stream);
    if (h->precision == 1 && p.log2_domain) {
        int var = p.variant >= 0 ? p.variant : h->opt[OPT_ATTN_VARIANT];  // imcui_hip_set_option(h, "attn_variant", v)
        if (var == 9 && (p.V6 == nullptr || p.single)) var = 8;  // (callers without the fp6 scratch: SuperGlue, DUSt3R, the C-ABI block)
        // key-split launch: the grid leaves the chip empty (fewer than two workgroups per CU), there is more than one chunk to split and the
        // caller provided the scratch; option attn_split: 0 never, 1 (default) by this rule, 2 whenever there is scratch (tests).  Same rows either way.
        const int rule = h->opt[OPT_ATTN_SPLIT];
        const bool split = p.part != nullptr && p.rows_per_seq > KT * ATTN_CHUNK_TILES && rule != 0 && ((long)grid.x < 2L * h->num_cu || rule == 2);
        if (var == 9) {
            const int rc = attention_mx_launch(h, p, stream);
            if (rc != IMCUI_OK) return rc;
        } else if (p.single)
            attn_launch_l2d<4>(p, split, stream);
        else if (var == 6 || var == 7)
            attn_launch_l2d<7>(p, split, stream);
        else
            attn_launch_l2d<8>(p, split, stream);
    } else if (h->precision == 1)
        hipLaunchKernelGGL((attn_split_kernel<false, 0, false>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(attn_kernel, grid, dim3(256), 0, stream, p);
    imcui_prof_end(h, PROF_ATTN, stream);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
