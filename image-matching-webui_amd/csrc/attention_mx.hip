// Flash attention, variant 9 (round 6): the split arithmetic of attention.hip with the two CORRECTION products of P.V on the block-scaled
// fp6 matrix instruction.
//
//   O^T += V^T . P^T  =  vh . ph   +   vl . ph   +   vh . pl          (variants 0 / 8: three v_mfma_f32_32x32x16_f16 per 16 keys)
//                     ~  vh . ph   +   mx6(vl) . mx6(P)   +   mx6(vh) . mx6(pl)
//
// mx6 = fp6 e2m3 elements (4 significant bits) with one e8m0 scale per 32 keys: `v_mfma_scale_f32_32x32x64_f8f6f4` multiplies 64 keys per
// instruction at about the issue cost of ONE 16-key f16 instruction (tools/mx_lab.hip: 43 ns against 39 ns), so a 64-key tile runs
// 24 (K.Q^T) + 8 (vh . ph) + 4 (corrections) = 36 matrix instructions instead of 48 (variant 8) or 40 (variant 7, which DROPS vh . pl).
// Nothing is dropped here: the correction terms are 2^-11 of the result and keep 4 bits of their own, i.e. an error of 2^-15 per key with
// random signs under a soft-max average.  CPU probe (tools/mx_corrections_probe.py, profiles/r06_lab_mx_corrections.txt): per-layer error and
// score error of LightGlue stay at the level of the three-product arithmetic (the two-product variant costs 3 - 10 x more), whereas
// the same trick on K.Q^T, on the projections / FFN and on the convolutions fails the parity bar -- there the 2^-15 lands on every
// output unaveraged (or is exponentiated).  K.Q^T keeps its three f16 products.
//
// What differs from attention.hip's kernel (same grid, same 128 queries x 64-key tiles, same deferred base-2 soft-max):
//   * KEY ORDER.  The scaled instruction wants, per lane, 32 consecutive k of one operand row; a lane's P values are the S^T accumulators
//     of fragment f, register r (C/D row (r & 3) + 8 (r >> 2) + 4 hi).  The conversion `v_cvt_scalef32_2xpk16_fp6_f32(s[0], s[1])` packs
//     field 2 r + f <- s[f][r] (measured, mx_lab), so the K tile is staged with its rows PERMUTED: accumulator (f, r) of half-wave hi holds
//     physical key 32 hi + 2 r + f.  A half-wave then owns 32 consecutive keys in field order, and the fp6 planes of V^T are plain runs
//     of 32 consecutive keys per (feature, half tile) -- nothing about the key order leaks out of this file.
//   * V^T comes as ONE f16 plane (hi) + the two fp6 planes of (hi, lo) with their scales: `attn_v6_pack_kernel` writes a 6400-byte
//     record per (sequence, head, 64-key tile) in MFMA fragment order ([128 slots][16 B] + [128 slots][8 B] per plane + 256 scale bytes),
//     so staging is a straight copy and a fragment is one conflict-free ds_read_b128 + ds_read_b64.
//   * P: ph = rtz f16 pairs (the B operand of the main product), fp6(P) and fp6(P - ph): one conversion instruction per 32 values.  Their block
//     scale is per LANE (a lane's 32 keys are one block of the instruction) and costs no maximum over the probabilities: the lane-local
//     maximum of the LOGITS is already there (it feeds the running maximum), and max P = 2^that.  A constant scale (2^13: P <= 2^15.5) was
//     built first and is NOT enough -- e2m3 spans six binades, so every probability below 1/32 of the row maximum lost its correction:
//     per-layer error 4.8e-5 on the GPU, 6.2e-5 in the CPU emulation of exactly that (`pvfix` rows of the probe).
#include <stdlib.h>

#include <type_traits>

#include "attention.h"

#define KT 64
#define KSTR 65   // padded key stride of the K image (uint4 units)
#define VHSTR 9   // V^T hi image: row d = 8 groups (hi, t, f) of 8 halves + one 16-byte pad
#define P_SHIFT 14.0f
#define DEFER_THR 1.5f
#define MX_U4 (2 * 8 * KSTR + 64 * VHSTR + ATTN_V6_TILE_BYTES / 16)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const void* mx_uniform_ptr(const void* p) {
    const size_t v = (size_t)p;
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffu));
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void*)((size_t)lo32 | ((size_t)hi32 << 32));
}
__device__ __forceinline__ int mx_key_seq(const AttnP& p, int seq) {
    if (p.cross == 0) return seq;
    if (p.cross == 1) return seq ^ 1;
    const int half = p.nseq >> 1;
    return seq < half ? seq + half : seq - half;
}
// power-of-two block scale: the smallest 2^e with amax / 2^e <= 7.5 (the largest e2m3 value); returns the e8m0 byte e + 127 in [1, 254]
__device__ __forceinline__ int mx_scale_byte(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int x = (int)((b >> 23) & 255u);         // biased exponent of amax = 1.m 2^(x - 127)
    const bool top = (b & 0x7fffffu) > 0x700000u;  // 1.m > 1.875: 1.m * 4 would exceed 7.5
    int e = x - 2 + (top ? 1 : 0);
    return e < 1 ? 1 : (e > 254 ? 254 : e);
}
__device__ __forceinline__ f32x16 mfma_fp6(const uint4& a16, const uint2& a8, const u32x6& b, const f32x16& c, int sa, int sb) {
    const i32x8 av = {(int)a16.x, (int)a16.y, (int)a16.z, (int)a16.w, (int)a8.x, (int)a8.y, 0, 0};
    const i32x8 bv = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], (int)b[4], (int)b[5], 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2, 2, 0, sa, 0, sb);  // cbsz = blgp = 2: fp6 e2m3; scales = byte 0
}

// ------------------------------------------------------------------------------------------------ the fp6 planes of V^T
// One workgroup (128 threads) per (sequence, head, 64-key tile); thread = MFMA slot s = df * 64 + lane: feature d = 32 df + (lane & 31),
// keys k0 + 32 (lane >> 5) .. + 31.  Reads the f16 hi / lo planes of V^T [seq][head][64][R] (gemm_wreg's EPI_QKV / EPI_CROSS epilogue),
// writes the tile record: hi6 [128][16 B] | hi6 [128][8 B] | lo6 [128][16 B] | lo6 [128][8 B] | scales [64 lanes][df0 hi, df0 lo, df1 hi, df1 lo].
// Keys at or beyond the sequence's count become zeros (padding rows may hold anything; they must not set a block's scale).
__global__ __launch_bounds__(128) void attn_v6_pack_kernel(const unsigned short* __restrict__ V, size_t plane, unsigned char* __restrict__ V6, const int* __restrict__ cnt,
                                                           const int* __restrict__ active, int heads, int R) {
    const int ntile = R >> 6;
    const int tile = blockIdx.x % ntile, sh = blockIdx.x / ntile;  // sh = seq * heads + head
    const int seq = sh / heads;
    const int nk = cnt[seq];
    const int k0 = tile * KT;
    if (k0 >= nk) return;
    if (active && active[seq >> 1] == 0) return;
    const int s = threadIdx.x, df = s >> 6, lane = s & 63, hi = lane >> 5, d = 32 * df + (lane & 31);
    const unsigned short* src = V + ((size_t)sh * 64 + d) * R + k0 + 32 * hi;
    unsigned char* rec = V6 + ((size_t)sh * ntile + tile) * ATTN_V6_TILE_BYTES;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        float v[32];
        float amax = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 w = *reinterpret_cast<const uint4*>(src + (size_t)pl * plane + 8 * q);
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x2 h2 = __builtin_bit_cast(f16x2, ww[j]);
                const int key = k0 + 32 * hi + 8 * q + 2 * j;
                v[8 * q + 2 * j] = key < nk ? (float)h2[0] : 0.0f;
                v[8 * q + 2 * j + 1] = key + 1 < nk ? (float)h2[1] : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
        const int sb = mx_scale_byte(amax);
        const float scale = __builtin_bit_cast(float, (unsigned)sb << 23);
        f32x16 a, b;  // the conversion packs field 2 i <- a[i], field 2 i + 1 <- b[i]: natural key order
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = v[2 * i];
            b[i] = v[2 * i + 1];
        }
        const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
        *reinterpret_cast<uint4*>(rec + pl * 3072 + s * 16) = make_uint4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<uint2*>(rec + pl * 3072 + 2048 + s * 8) = make_uint2(r[4], r[5]);
        rec[6144 + lane * 4 + 2 * df + pl] = (unsigned char)sb;
    }
}

// ------------------------------------------------------------------------------------------------ the attention kernel
__global__ __launch_bounds__(256, 2) void attn_mx_kernel(AttnP p) {
    __shared__ uint4 smem4[MX_U4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int nqb = p.rows_per_seq >> 7;
    const int bid = blockIdx.x;
    const int grp = (bid / (8 * nqb)) * 8 + (bid & 7);  // XCD-aware: the query blocks of one (sequence, head) share an L2 (attention.hip)
    const int seq = grp / p.heads, head = grp - seq * p.heads;
    const int q0 = ((bid >> 3) % nqb) * 128;
    const int nq = p.cnt[seq];
    if (q0 >= nq) return;
    if (p.active && p.active[seq >> 1] == 0) return;
    const int kseq = mx_key_seq(p, seq);
    const int nk = p.cnt[kseq];
    const int R = p.rows_per_seq;
    const size_t plane = (size_t)p.nseq * p.heads * R * 64;  // halves per plane

    const unsigned short* Qh = reinterpret_cast<const unsigned short*>(p.Q) + ((size_t)seq * p.heads + head) * R * 64;
    const unsigned short* Kg = reinterpret_cast<const unsigned short*>(p.K) + ((size_t)kseq * p.heads + head) * R * 64;
    const unsigned short* Vg = reinterpret_cast<const unsigned short*>(p.V) + ((size_t)kseq * p.heads + head) * 64 * R;
    const unsigned char* V6g = p.V6 + ((size_t)kseq * p.heads + head) * (size_t)(R >> 6) * ATTN_V6_TILE_BYTES;
    const unsigned slab = (unsigned)R * 64u * 2u;
    const __amdgpu_buffer_rsrc_t rKh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mx_uniform_ptr(Kg)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rKl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mx_uniform_ptr(Kg + plane)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rVh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mx_uniform_ptr(Vg)), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV6 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(mx_uniform_ptr(V6g))), 0, (unsigned)(R >> 6) * ATTN_V6_TILE_BYTES, 0x00020000);

    uint4 qh[4], ql[4];  // Q fragment of this lane: query q0 + wid*32 + lo, dims 16 s + 8 hi .. + 7
    {
        const int qrow = min(q0 + wid * 32 + lo, R - 1);
        const unsigned short* qsrc = Qh + (size_t)qrow * 64 + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *reinterpret_cast<const uint4*>(qsrc + 16 * s);
            ql[s] = *reinterpret_cast<const uint4*>(qsrc + plane + 16 * s);
        }
    }
    f32x16 o[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f][r] = 0.0f;
    float m_ref = 0.0f, l_run = 0.0f;
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.0f;

    // ---- staging registers: K 64 keys x 8 octets x 2 planes (4 x 16 B per thread), V^T hi 64 d x 8 key octets (2 x 16 B), fp6 record 400 x 16 B
    uint4 rk0, rk1, rk2, rk3, rv0, rv1, r60, r61;
    const int k_key = tid >> 3, k_oc = tid & 7;  // K: physical keys k_key and k_key + 32
    const int v_d = tid >> 3, v_o = tid & 7;     // V^T: rows v_d and v_d + 32, keys 8 v_o .. + 7
    const unsigned ko0 = (unsigned)(k_key * 64 + k_oc * 8) * 2u, ko1 = ko0 + 32u * 64u * 2u;
    const unsigned vo0 = ((unsigned)v_d * (unsigned)R + (unsigned)v_o * 8u) * 2u, vo1 = vo0 + 32u * (unsigned)R * 2u;
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned vo, unsigned so) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
        const unsigned sk = (unsigned)k0 * 128u, sv = (unsigned)k0 * 2u, s6 = (unsigned)(k0 >> 6) * ATTN_V6_TILE_BYTES;
        rk0 = ld4(rKh, ko0, sk);
        rk2 = ld4(rKh, ko1, sk);
        rv0 = ld4(rVh, vo0, sv);
        rv1 = ld4(rVh, vo1, sv);
        rk1 = ld4(rKl, ko0, sk);
        rk3 = ld4(rKl, ko1, sk);
        r60 = ld4(rV6, (unsigned)tid * 16u, s6);
        if (tid < ATTN_V6_TILE_BYTES / 16 - 256) r61 = ld4(rV6, (unsigned)(256 + tid) * 16u, s6);
    };
    // K image row of physical key kappa (half-wave hi_k = kappa >> 5, field jf = kappa & 31 = 2 r + f): accumulator row 32 f + frag_row(r, hi_k)
    const int kr = k_key >> 1, kf = k_key & 1;
    const int k_row0 = 32 * kf + (kr & 3) + 8 * (kr >> 2);  // physical key k_key (hi_k = 0); key k_key + 32 sits 4 rows further (hi_k = 1)
    // V^T hi image: 8-byte slot of (row d, group G = (hi_k * 2 + t) * 2 + f, half o & 1) for the even (f = 0) and odd (f = 1) keys of the octet
    const int v_g = ((v_o >> 2) * 2 + ((v_o >> 1) & 1)) * 2;
    auto store_tile = [&](int k0, auto tail) __attribute__((always_inline)) {
        uint4* Kh = smem4;
        uint4* Kl = Kh + 8 * KSTR;
        uint2* Vh2 = reinterpret_cast<uint2*>(Kh + 2 * 8 * KSTR);
        uint4* V6s = Kh + 2 * 8 * KSTR + 64 * VHSTR;
        if (decltype(tail)::value) {  // keys past the sequence end may hold anything (even NaN bit patterns): zero them
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            if (k0 + k_key >= nk) rk0 = rk1 = z4;
            if (k0 + k_key + 32 >= nk) rk2 = rk3 = z4;
            const int kk = k0 + v_o * 8;
            auto mask = [&](uint4& v) __attribute__((always_inline)) {
                v.x &= (kk + 0 < nk ? 0x0000FFFFu : 0u) | (kk + 1 < nk ? 0xFFFF0000u : 0u);
                v.y &= (kk + 2 < nk ? 0x0000FFFFu : 0u) | (kk + 3 < nk ? 0xFFFF0000u : 0u);
                v.z &= (kk + 4 < nk ? 0x0000FFFFu : 0u) | (kk + 5 < nk ? 0xFFFF0000u : 0u);
                v.w &= (kk + 6 < nk ? 0x0000FFFFu : 0u) | (kk + 7 < nk ? 0xFFFF0000u : 0u);
            };
            mask(rv0);
            mask(rv1);
        }
        Kh[k_oc * KSTR + k_row0] = rk0;
        Kh[k_oc * KSTR + k_row0 + 4] = rk2;
        Kl[k_oc * KSTR + k_row0] = rk1;
        Kl[k_oc * KSTR + k_row0 + 4] = rk3;
        // even keys of the octet -> group f = 0, odd keys -> group f = 1 (v_perm_b32: bytes 0-3 = second operand, 4-7 = first)
        auto scatter = [&](const uint4& v, int d) __attribute__((always_inline)) {
            const uint2 ev = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.w, v.z, 0x05040100u));
            const uint2 od = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x07060302u), __builtin_amdgcn_perm(v.w, v.z, 0x07060302u));
            Vh2[(d * VHSTR + v_g) * 2 + (v_o & 1)] = ev;
            Vh2[(d * VHSTR + v_g + 1) * 2 + (v_o & 1)] = od;
        };
        scatter(rv0, v_d);
        scatter(rv1, v_d + 32);
        V6s[tid] = r60;
        if (tid < ATTN_V6_TILE_BYTES / 16 - 256) V6s[256 + tid] = r61;
    };

    auto compute_tile = [&](int k0, auto first, auto tail) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value, TAIL = decltype(tail)::value;
        const uint4* Kh = smem4;
        const uint4* Kl = Kh + 8 * KSTR;
        const uint4* Vh = Kh + 2 * 8 * KSTR;
        const uint4* V6a = Vh + 64 * VHSTR;  // [plane][128 slots] 16-byte pieces at +192 uint4 per plane
        f32x16 s[2];
        float m_pre = -INFINITY;
        {   // S^T = K . Q^T in three f16 products, fragments of step i + 1 requested before the MFMAs of step i (attention.hip, variant 8)
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
                s[f] = mfma16(kl[i & 1], qh[st], st == 0 ? cinit : s[f]);
                s[f] = mfma16(kh[i & 1], ql[st], s[f]);
                s[f] = mfma16(kh[i & 1], qh[st], s[f]);
                if (!TAIL && i >= 4) {
                    const int r0 = 4 * (i - 4);
                    m_pre = fmaxf(fmaxf(m_pre, s[0][r0]), fmaxf(s[0][r0 + 1], fmaxf(s[0][r0 + 2], s[0][r0 + 3])));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (TAIL) {  // accumulator (f, r) of this half-wave is physical key k0 + 32 hi + 2 r + f
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * hi + 2 * r + f >= nk) s[f][r] = -INFINITY;
        }
        float m_t = m_pre;
#pragma unroll
        for (int f = TAIL ? 0 : 1; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, s[f][r]);
        const float m_loc = m_t;  // the maximum over THIS lane's 32 keys: its exponent becomes the lane's fp6 block scale below
        {
            float ma = m_t, mb = m_t;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(ma), "+v"(mb));
            m_t = ma;
        }
        f32x2 la = {0.0f, 0.0f}, lb = {0.0f, 0.0f};
        int pe;  // block scale of this lane's 32 probabilities: the smallest 2^pe with max P / 2^pe <= 7.5 (max P = 2^(m_loc - excess))
        {
            const float excess = FIRST ? m_t - P_SHIFT : (m_t > P_SHIFT + DEFER_THR ? m_t - P_SHIFT : 0.0f);
            pe = (int)fminf(fmaxf(ceilf((m_loc - excess) - 2.9068906f), -100.0f), 100.0f);  // log2(7.5) = 2.90689; a fully masked lane (-inf) clamps
            const bool shift = FIRST || (__ballot(excess != 0.0f) != 0ull);
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
                if (FIRST) {
                    m_ref = m_t;
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
        // ---- P: fp6 of the probabilities (block scale 2^pe), their rtz f16 halves, fp6 of the remainders (P - ph < 2^-10 of the block's
        // binade, so 2^(pe - 10) cannot saturate)
        const int sb_h = pe + 127, sb_l = pe + 117;  // e8m0 bytes of the two P operands
        const u32x6 p6h = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s[0], s[1], __builtin_bit_cast(float, (unsigned)sb_h << 23));
        unsigned ph[2][8];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const fp16x2_t h2 = __builtin_amdgcn_cvt_pkrtz(s[f][2 * j], s[f][2 * j + 1]);
                const unsigned hb = __builtin_bit_cast(unsigned, h2);
                ph[f][j] = hb;
                float l0, l1;  // p - (float)ph, exact in f32: one mixed-precision fma per element
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(s[f][2 * j]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(s[f][2 * j + 1]));
                s[f][2 * j] = l0;
                s[f][2 * j + 1] = l1;
            }
        const u32x6 p6l = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s[0], s[1], __builtin_bit_cast(float, (unsigned)sb_l << 23));
        // ---- O^T += V^T . P^T: the two fp6 corrections first, then the f16 main product
        const unsigned sc = reinterpret_cast<const unsigned*>(V6a)[1536 + lane];  // scale bytes (df0 hi, df0 lo, df1 hi, df1 lo)
        const uint2* V6b = reinterpret_cast<const uint2*>(V6a);
#pragma unroll
        for (int df = 0; df < 2; ++df)  // (the two accumulators alternate: no back-to-back dependent matrix instructions)
            o[df] = mfma_fp6(V6a[192 + df * 64 + lane], V6b[(3072 + 2048) / 8 + df * 64 + lane], p6h, o[df], (int)(sc >> (16 * df + 8)), sb_h);  // lo6(V) . fp6(P)
#pragma unroll
        for (int df = 0; df < 2; ++df)
            o[df] = mfma_fp6(V6a[df * 64 + lane], V6b[2048 / 8 + df * 64 + lane], p6l, o[df], (int)(sc >> (16 * df)), sb_l);  // hi6(V) . fp6(P - ph)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 pb = make_uint4(ph[f][4 * t], ph[f][4 * t + 1], ph[f][4 * t + 2], ph[f][4 * t + 3]);  // registers 8 t .. 8 t + 7 of s[f]
#pragma unroll
                for (int df = 0; df < 2; ++df) o[df] = mfma16(Vh[(32 * df + lo) * VHSTR + (hi * 2 + t) * 2 + f], pb, o[df]);
            }
    };

    const int ntile = (nk + KT - 1) / KT;
    if (ntile > 0) load_tile(0);
    if (ntile > 1) {
        __syncthreads();
        store_tile(0, std::false_type{});
        __syncthreads();
        load_tile(KT);
        compute_tile(0, std::true_type{}, std::false_type{});
    }
    for (int tile = 1; tile + 1 < ntile; ++tile) {
        __syncthreads();
        store_tile(tile * KT, std::false_type{});
        __syncthreads();
        load_tile((tile + 1) * KT);
        compute_tile(tile * KT, std::false_type{}, std::false_type{});
    }
    if (ntile > 0) {
        __syncthreads();
        store_tile((ntile - 1) * KT, std::true_type{});
        __syncthreads();
        if (ntile == 1)
            compute_tile(0, std::true_type{}, std::true_type{});
        else
            compute_tile((ntile - 1) * KT, std::false_type{}, std::true_type{});
    }

    // ---- normalise and write (transpose through LDS so each query row is stored contiguously)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (ntile > 0) ? 1.0f / l_tot : 0.0f;
    __syncthreads();
    float* Os = reinterpret_cast<float*>(smem4) + wid * (32 * 33);
    const int H64 = p.heads * 64;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Os[lo * 33 + frag_row(r, hi)] = o[f][r] * inv;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
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

int attention_mx_launch(imcui_hip_s* h, const AttnP& p, hipStream_t stream) {
    if (p.V6 == nullptr) return imcui_set_err(h, IMCUI_ERR_ARG, "attention variant 9 needs AttnP.V6 (the fp6 planes of V^T)");
    const size_t plane = (size_t)p.nseq * p.heads * p.rows_per_seq * 64;
    if (!p.v6_ready)
        hipLaunchKernelGGL(attn_v6_pack_kernel, dim3((unsigned)(p.nseq * p.heads * (p.rows_per_seq >> 6))), dim3(128), 0, stream, reinterpret_cast<const unsigned short*>(p.V), plane,
                           p.V6, p.cnt, p.active, p.heads, p.rows_per_seq);
    hipLaunchKernelGGL(attn_mx_kernel, dim3((unsigned)((p.rows_per_seq / 128) * p.heads * p.nseq)), dim3(256), 0, stream, p);
    return IMCUI_OK;
}
