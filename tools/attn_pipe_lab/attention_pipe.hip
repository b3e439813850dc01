// Software-pipelined 3 x f16 split flash attention for gfx950 (head_dim 64): the kernel the LightGlue / SuperGlue
// layers run in the default arithmetic mode (imcui/hloc/matchers/lightglue.py:75 -> upstream
// SelfBlock / CrossBlock attention).
//
// Why a second kernel.  On a CDNA4 SIMD the matrix pipe and the vector ALU of CO-RESIDENT waves barely overlap
// (tools/overlap_lab.hip: an MFMA wave + a VALU wave on one SIMD take the sum of their times), while ONE wave hides
// up to ~5 independent instructions behind each of its own 32-cycle MFMAs (MI355X_MICROARCH.md).  attn_split_kernel
// (attention.hip) runs QK^T -> soft-max -> PV strictly in sequence inside a wave and relies on a second workgroup
// per CU for overlap: its matrix pipe is busy 51 % of the time.  Here every wave is ALONE on its SIMD
// (__launch_bounds__(256, 1): the whole 512-register file, one 4-wave workgroup per CU) and its loop body holds two
// independent instruction streams per phase:
//
//   phase 1   MFMA: S_i = K_i . Q^T                  |  VALU: split E_{i-1} -> (hi, lo) f16 = P and its row sum
//   phase 2   MFMA: O^T += V^T_{i-1} . P^T           |  VALU: online soft-max of S_i -> E_i = 2^(s log2e - m log2e + 14)
//
// i.e. the soft-max of tile i runs under the PV product of tile i-1 and the f16 split of tile i-1 under the QK^T
// product of tile i.  The S accumulators ping-pong between two register sets (loop unrolled by two), K tiles live in
// a 2-slot and V^T tiles in a 3-slot LDS ring (tile i+1 is written while K_i / V_{i-1} are read: one barrier per
// tile), global loads run two tiles ahead.  Fragment layouts, LDS images, arithmetic (three f16 MFMAs per product, f32
// accumulate, the 2^14 guard) and therefore the RESULTS are those of attn_split_kernel; only the order of the
// key tiles' contributions to the running maximum differs by nothing (same tile order).
#include <type_traits>

#include "attention.h"

#define KT 64
#define KSTR 65  // K image: [d-octet][key] granules of 16 B, padded key stride
#define VSTR 9   // V^T image: [d][9] granules of 16 B (8 + 1 pad)
#define P_SHIFT 14.0f
#define LOG2E 1.44269504088896340736f
#define K_SLOT (2 * 8 * KSTR)   // granules per K tile: hi plane, lo plane
#define V_SLOT (2 * 64 * VSTR)  // granules per V^T tile
#define NKSLOT 2
#define NVSLOT 3

struct SFrag {
    f32x16 s[2];  // [32-key fragment]
};
struct PFrag {
    uint4 h[4], l[4];  // [k-step]: the lane's 8 keys 32 f + 16 tt + {4hi..4hi+3, 8+4hi..8+4hi+3}, ks = 2 f + tt
};

// LLVM scheduling-group masks (__builtin_amdgcn_sched_group_barrier)
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_VMEM_RD 0x020
#define SG_DS_RD 0x100
#define SG_DS_WR 0x200
#define SG_TRANS 0x400

__global__ __launch_bounds__(256, 1) void attn_split_pipe_kernel(AttnP p) {
    __shared__ uint4 smem4[NKSLOT * K_SLOT + NVSLOT * V_SLOT];
    uint4* const Kring = smem4;
    uint4* const Vring = smem4 + NKSLOT * K_SLOT;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, so the query blocks of one (sequence, head) share an L2
    const int nqb = p.rows_per_seq >> 7;
    const int bid = blockIdx.x;
    const int grp = (bid / (8 * nqb)) * 8 + (bid & 7);
    const int seq = grp / p.heads, head = grp - seq * p.heads;
    const int q0 = ((bid >> 3) % nqb) * 128;
    const int nq = p.cnt[seq];
    if (q0 >= nq) return;
    if (p.active && p.active[seq >> 1] == 0) return;
    const int kseq = p.cross ? (seq ^ 1) : seq;
    const int nk = p.cnt[kseq];
    const int R = p.rows_per_seq;
    const size_t plane = (size_t)p.nseq * p.heads * R * 64;  // halves per plane

    const unsigned short* Qh = reinterpret_cast<const unsigned short*>(p.Q) + ((size_t)seq * p.heads + head) * R * 64;
    const unsigned short* Kg = reinterpret_cast<const unsigned short*>(p.K) + ((size_t)kseq * p.heads + head) * R * 64;
    const unsigned short* Vg = reinterpret_cast<const unsigned short*>(p.V) + ((size_t)kseq * p.heads + head) * 64 * R;

    // Q fragment of this lane: query q0 + wid*32 + lo, dims 16 s + 8 hi .. +7
    uint4 qh[4], ql[4];
    {
        const int qrow = min(q0 + wid * 32 + lo, R - 1);
        const unsigned short* qsrc = Qh + (size_t)qrow * 64 + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *reinterpret_cast<const uint4*>(qsrc + 16 * s);
            ql[s] = *reinterpret_cast<const uint4*>(qsrc + plane + 16 * s);
        }
    }

    f32x16 o[2];  // [d fragment]
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[df][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f, alpha = 1.0f;

    // ---- staging (as attn_split_kernel): K 64 keys x 8 octets x 2 planes, V^T 64 d x 16 key-quads x 2 planes
    uint4 rk0, rk1, rk2, rk3;
    uint2 rv0, rv1, rv2, rv3, rv4, rv5, rv6, rv7;
    const int k_key = tid >> 3, k_oc = tid & 7;
    const int v_d = tid >> 4, v_kq = tid & 15;
    const int v_u = ((v_kq >> 2) * 2 + (v_kq & 1)) * 2 + ((v_kq >> 1) & 1);
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
        const unsigned short* ks = Kg + (size_t)(k0 + k_key) * 64 + k_oc * 8;
        rk0 = *reinterpret_cast<const uint4*>(ks);
        rk1 = *reinterpret_cast<const uint4*>(ks + plane);
        rk2 = *reinterpret_cast<const uint4*>(ks + 32 * 64);
        rk3 = *reinterpret_cast<const uint4*>(ks + 32 * 64 + plane);
        const unsigned short* vs = Vg + (size_t)v_d * R + k0 + v_kq * 4;
        rv0 = *reinterpret_cast<const uint2*>(vs);
        rv1 = *reinterpret_cast<const uint2*>(vs + plane);
        rv2 = *reinterpret_cast<const uint2*>(vs + (size_t)16 * R);
        rv3 = *reinterpret_cast<const uint2*>(vs + (size_t)16 * R + plane);
        rv4 = *reinterpret_cast<const uint2*>(vs + (size_t)32 * R);
        rv5 = *reinterpret_cast<const uint2*>(vs + (size_t)32 * R + plane);
        rv6 = *reinterpret_cast<const uint2*>(vs + (size_t)48 * R);
        rv7 = *reinterpret_cast<const uint2*>(vs + (size_t)48 * R + plane);
    };
    auto store_tile = [&](int t, auto full) __attribute__((always_inline)) {
        const int k0 = t * KT;
        if (!decltype(full)::value && k0 + KT > nk) {
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
        uint4* Kh = Kring + (t % NKSLOT) * K_SLOT;
        uint4* Kl = Kh + 8 * KSTR;
        Kh[k_oc * KSTR + k_key] = rk0;
        Kl[k_oc * KSTR + k_key] = rk1;
        Kh[k_oc * KSTR + k_key + 32] = rk2;
        Kl[k_oc * KSTR + k_key + 32] = rk3;
        uint2* vh2 = reinterpret_cast<uint2*>(Vring + (t % NVSLOT) * V_SLOT);
        uint2* vl2 = vh2 + 64 * VSTR * 2;
        vh2[(v_d * VSTR) * 2 + v_u] = rv0;
        vl2[(v_d * VSTR) * 2 + v_u] = rv1;
        vh2[((v_d + 16) * VSTR) * 2 + v_u] = rv2;
        vl2[((v_d + 16) * VSTR) * 2 + v_u] = rv3;
        vh2[((v_d + 32) * VSTR) * 2 + v_u] = rv4;
        vl2[((v_d + 32) * VSTR) * 2 + v_u] = rv5;
        vh2[((v_d + 48) * VSTR) * 2 + v_u] = rv6;
        vl2[((v_d + 48) * VSTR) * 2 + v_u] = rv7;
    };

    // ---- the pieces of an iteration
    // S = K_t . Q^T: the two 32-key fragments are independent accumulator chains and alternate
    auto qk = [&](int t, SFrag& S) __attribute__((always_inline)) {
        const uint4* Kh = Kring + (t % NKSLOT) * K_SLOT;
        const uint4* Kl = Kh + 8 * KSTR;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) S.s[f][r] = 0.0f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const uint4 ah0 = Kh[(2 * st + hi) * KSTR + lo], al0 = Kl[(2 * st + hi) * KSTR + lo];
            const uint4 ah1 = Kh[(2 * st + hi) * KSTR + 32 + lo], al1 = Kl[(2 * st + hi) * KSTR + 32 + lo];
            S.s[0] = mfma16(al0, qh[st], S.s[0]);
            S.s[1] = mfma16(al1, qh[st], S.s[1]);
            S.s[0] = mfma16(ah0, ql[st], S.s[0]);
            S.s[1] = mfma16(ah1, ql[st], S.s[1]);
            S.s[0] = mfma16(ah0, qh[st], S.s[0]);
            S.s[1] = mfma16(ah1, qh[st], S.s[1]);
        }
    };
    // P = split(E) into f16 (hi, lo), and the lane's share of the row sum of E
    auto split_p = [&](SFrag& E, PFrag& P, float& l_t) __attribute__((always_inline)) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                uint4& ph = P.h[2 * f + tt];
                uint4& pl = P.l[2 * f + tt];
                split2(E.s[f][8 * tt + 0], E.s[f][8 * tt + 1], ph.x, pl.x);
                split2(E.s[f][8 * tt + 2], E.s[f][8 * tt + 3], ph.y, pl.y);
                split2(E.s[f][8 * tt + 4], E.s[f][8 * tt + 5], ph.z, pl.z);
                split2(E.s[f][8 * tt + 6], E.s[f][8 * tt + 7], ph.w, pl.w);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    a0 += E.s[f][8 * tt + j];
                    a1 += E.s[f][8 * tt + j + 1];
                }
            }
        l_t = a0 + a1;
    };
    // O^T += V^T_t . P^T: the two d fragments are independent accumulator chains and alternate
    auto pv = [&](int t, PFrag& P) __attribute__((always_inline)) {
        const uint4* Vh = Vring + (t % NVSLOT) * V_SLOT;
        const uint4* Vl = Vh + 64 * VSTR;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int v0 = lo * VSTR + ks * 2 + hi, v1 = (32 + lo) * VSTR + ks * 2 + hi;
            const uint4 vh0 = Vh[v0], vl0 = Vl[v0], vh1 = Vh[v1], vl1 = Vl[v1];
            o[0] = mfma16(vl0, P.h[ks], o[0]);
            o[1] = mfma16(vl1, P.h[ks], o[1]);
            o[0] = mfma16(vh0, P.l[ks], o[0]);
            o[1] = mfma16(vh1, P.l[ks], o[1]);
            o[0] = mfma16(vh0, P.h[ks], o[0]);
            o[1] = mfma16(vh1, P.h[ks], o[1]);
        }
    };
    // running maximum of tile t and E = 2^(s log2e - m log2e + 14) in place; the row sum is taken by split_p one
    // iteration later (under the next QK^T product), `alpha` = rescale factor of O and l for this tile
    auto softmax = [&](int t, SFrag& S, auto full) __attribute__((always_inline)) {
        const int k0 = t * KT;
        if (!decltype(full)::value && k0 + KT > nk) {  // only the last tile can hold keys past the sequence end
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * f + frag_row(r, hi) >= nk) S.s[f][r] = -INFINITY;
        }
        float m_t = -INFINITY;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, S.s[f][r]);
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        const float m_new = fmaxf(m_run, m_t);  // finite: every tile holds >= 1 valid key
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const float bias = P_SHIFT - m_new * LOG2E;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) S.s[f][r] = __builtin_amdgcn_exp2f(fmaf(S.s[f][r], LOG2E, bias));
        m_run = m_new;
    };
    // O and l move to the scale of the tile whose E is about to be added (skipped while no running maximum moved)
    auto rescale = [&]() __attribute__((always_inline)) {
        if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
            for (int df = 0; df < 2; ++df)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[df][r] *= alpha;
            l_run *= alpha;
        }
    };

    const int ntile = (nk + KT - 1) / KT;
    // One iteration i (1 <= i <= ntile - 1):
    //   phase 1   MFMA S_new = QK(i)   | VALU P = split(E_prev), row sum of E_prev ; LDS <- tile i+1 ; registers <- tile i+2
    //   phase 2   MFMA O += PV(i-1)    | VALU soft-max(S_new)
    // After the (rare, wave-uniform) rescale branch the iteration is ONE basic block; the sched_group_barrier pipeline
    // tells the scheduler how to interleave the independent streams: per matrix step (2 fragment reads + 3 MFMAs of
    // each of the two accumulator chains = 6 MFMAs) a share of the vector work of that phase.
    auto iteration = [&](int i, SFrag& S_new, SFrag& E_prev, auto full) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full)::value;
        rescale();
        PFrag P;
        float l_t;
        if (FULL || i + 1 < ntile) store_tile(i + 1, full);
        if (FULL || i + 2 < ntile) load_tile((i + 2) * KT);
        qk(i, S_new);
        split_p(E_prev, P, l_t);
        l_run += l_t;
        pv(i - 1, P);
        softmax(i, S_new, full);
        __syncthreads();
    };


    // ---- steady-state iteration, hand-interleaved.  One wave alone on its SIMD issues in order, and a 32-cycle MFMA
    // hides about five other instructions issued behind it (MI355X_MICROARCH.md): the iteration is therefore written
    // as 48 matrix steps, each followed by its share of the vector / LDS / global work of the phase, and every step ends
    // in a scheduling fence so that hipcc keeps exactly this order.
    //   phase 1, MFMA k = 0..23 of S_new = K_i . Q^T:
    //        k < 16   one split2 of E_prev (3 VALU) + its two row-sum adds
    //        k >= 16  tile i+1 registers -> LDS and tile i+2 global -> the same registers (12 + 12 instructions)
    //        every 6th the four K fragments of the next k-step are requested from LDS
    //   phase 2, MFMA k = 0..23 of O^T += V^T_{i-1} . P^T:
    //        k < 8    four of the 32 running-maximum updates
    //        k = 8, 9 cross-half maximum (one v_permlane32_swap), new maximum, rescale factor, exponent bias
    //        k >= 10  three E = exp2(fma(s, log2e, bias)) per step
    auto iteration_full = [&](int i, SFrag& S_new, SFrag& E_prev) __attribute__((always_inline)) {
        rescale();
        const uint4* Kh = Kring + (i % NKSLOT) * K_SLOT;
        const uint4* Kl = Kh + 8 * KSTR;
        const uint4* Vh = Vring + ((i - 1) % NVSLOT) * V_SLOT;
        const uint4* Vl = Vh + 64 * VSTR;
        uint4* Kh_w = Kring + ((i + 1) % NKSLOT) * K_SLOT;
        uint4* Kl_w = Kh_w + 8 * KSTR;
        uint2* vh2 = reinterpret_cast<uint2*>(Vring + ((i + 1) % NVSLOT) * V_SLOT);
        uint2* vl2 = vh2 + 64 * VSTR * 2;
        const unsigned short* ks = Kg + (size_t)((i + 2) * KT + k_key) * 64 + k_oc * 8;
        const unsigned short* vs = Vg + (size_t)v_d * R + (i + 2) * KT + v_kq * 4;
        PFrag P;
        uint4 fr[2][4];  // fragment double buffer: [parity of the k-step][hi0, lo0, hi1, lo1]
        auto kfrag = [&](int st, uint4 (&d)[4]) __attribute__((always_inline)) {
            d[0] = Kh[(2 * st + hi) * KSTR + lo];
            d[1] = Kl[(2 * st + hi) * KSTR + lo];
            d[2] = Kh[(2 * st + hi) * KSTR + 32 + lo];
            d[3] = Kl[(2 * st + hi) * KSTR + 32 + lo];
        };
        auto vfrag = [&](int ksx, uint4 (&d)[4]) __attribute__((always_inline)) {
            const int v0 = lo * VSTR + ksx * 2 + hi, v1 = (32 + lo) * VSTR + ksx * 2 + hi;
            d[0] = Vh[v0];
            d[1] = Vl[v0];
            d[2] = Vh[v1];
            d[3] = Vl[v1];
        };
        float a0 = 0.0f, a1 = 0.0f;
        kfrag(0, fr[0]);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) S_new.s[f][r] = 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- phase 1
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int st = k / 6, u = k % 6;
            uint4(&c)[4] = fr[st & 1];
            if (u == 0 && st + 1 < 4) kfrag(st + 1, fr[(st + 1) & 1]);
            if (u == 0 && st + 1 == 4) vfrag(0, fr[0]);  // first V^T fragments for phase 2
            const int chain = u & 1;
            const uint4 a = (u < 2) ? c[2 * chain + 1] : c[2 * chain];  // lo plane first, then hi, hi
            const uint4 b = (u >= 2 && u < 4) ? ql[st] : qh[st];
            S_new.s[chain] = mfma16(a, b, S_new.s[chain]);
            if (k < 16) {
                const int f = k >> 3, tt = (k >> 2) & 1, j = k & 3;
                const float e0 = E_prev.s[f][8 * tt + 2 * j], e1 = E_prev.s[f][8 * tt + 2 * j + 1];
                unsigned h, l;
                split2(e0, e1, h, l);
                uint4& ph = P.h[2 * f + tt];
                uint4& pl = P.l[2 * f + tt];
                if (j == 0) { ph.x = h; pl.x = l; }
                if (j == 1) { ph.y = h; pl.y = l; }
                if (j == 2) { ph.z = h; pl.z = l; }
                if (j == 3) { ph.w = h; pl.w = l; }
                a0 += e0;
                a1 += e1;
            } else {
                const int m = k - 16;
                if (m == 0) { Kh_w[k_oc * KSTR + k_key] = rk0; rk0 = *reinterpret_cast<const uint4*>(ks); }
                if (m == 1) { Kl_w[k_oc * KSTR + k_key] = rk1; rk1 = *reinterpret_cast<const uint4*>(ks + plane); }
                if (m == 2) { Kh_w[k_oc * KSTR + k_key + 32] = rk2; rk2 = *reinterpret_cast<const uint4*>(ks + 32 * 64); }
                if (m == 3) { Kl_w[k_oc * KSTR + k_key + 32] = rk3; rk3 = *reinterpret_cast<const uint4*>(ks + 32 * 64 + plane); }
                if (m == 0) { vh2[(v_d * VSTR) * 2 + v_u] = rv0; rv0 = *reinterpret_cast<const uint2*>(vs); }
                if (m == 1) { vl2[(v_d * VSTR) * 2 + v_u] = rv1; rv1 = *reinterpret_cast<const uint2*>(vs + plane); }
                if (m == 2) { vh2[((v_d + 16) * VSTR) * 2 + v_u] = rv2; rv2 = *reinterpret_cast<const uint2*>(vs + (size_t)16 * R); }
                if (m == 3) { vl2[((v_d + 16) * VSTR) * 2 + v_u] = rv3; rv3 = *reinterpret_cast<const uint2*>(vs + (size_t)16 * R + plane); }
                if (m == 4) { vh2[((v_d + 32) * VSTR) * 2 + v_u] = rv4; rv4 = *reinterpret_cast<const uint2*>(vs + (size_t)32 * R); }
                if (m == 5) { vl2[((v_d + 32) * VSTR) * 2 + v_u] = rv5; rv5 = *reinterpret_cast<const uint2*>(vs + (size_t)32 * R + plane); }
                if (m == 6) { vh2[((v_d + 48) * VSTR) * 2 + v_u] = rv6; rv6 = *reinterpret_cast<const uint2*>(vs + (size_t)48 * R); }
                if (m == 7) { vl2[((v_d + 48) * VSTR) * 2 + v_u] = rv7; rv7 = *reinterpret_cast<const uint2*>(vs + (size_t)48 * R + plane); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run += a0 + a1;
        // ---------------- phase 2
        float m0 = -INFINITY, m1 = -INFINITY, bias = 0.0f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int ksx = k / 6, u = k % 6;
            uint4(&c)[4] = fr[ksx & 1];
            if (u == 0 && ksx + 1 < 4) vfrag(ksx + 1, fr[(ksx + 1) & 1]);
            const int chain = u & 1;
            const uint4 a = (u < 2) ? c[2 * chain + 1] : c[2 * chain];  // V lo . P hi, V hi . P lo, V hi . P hi
            const uint4 b = (u >= 2 && u < 4) ? P.l[ksx] : P.h[ksx];
            o[chain] = mfma16(a, b, o[chain]);
            if (k < 8) {
                const int f = k >> 2, r0 = (k & 3) * 4;
                m0 = fmaxf(m0, S_new.s[f][r0 + 0]);
                m1 = fmaxf(m1, S_new.s[f][r0 + 1]);
                m0 = fmaxf(m0, S_new.s[f][r0 + 2]);
                m1 = fmaxf(m1, S_new.s[f][r0 + 3]);
            } else if (k == 8) {
                m0 = fmaxf(m0, m1);
                // the other half-wave holds the other 32 keys of the same query
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
                m0 = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            } else if (k == 9) {
                const float m_new = fmaxf(m_run, m0);  // finite: every tile holds >= 1 valid key
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
                bias = P_SHIFT - m_new * LOG2E;
                m_run = m_new;
            } else if (k < 21) {
                const int e0 = (k - 10) * 3;
#pragma unroll
                for (int e = e0; e < e0 + 3 && e < 32; ++e) {
                    const int f = e >> 4, r = e & 15;
                    S_new.s[f][r] = __builtin_amdgcn_exp2f(fmaf(S_new.s[f][r], LOG2E, bias));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    SFrag Sa, Sb;
    if (ntile > 0) {
        load_tile(0);
        store_tile(0, std::false_type{});
        if (ntile > 1) load_tile(KT);
        __syncthreads();
        // i = 0: nothing to multiply yet
        if (ntile > 1) store_tile(1, std::false_type{});
        if (ntile > 2) load_tile(2 * KT);
        qk(0, Sa);
        softmax(0, Sa, std::false_type{});
        __syncthreads();
        int i = 1;
        for (; i + 3 < ntile; i += 2) {  // tiles i .. i+3 exist and none of i, i+1, i+2 is the last
            iteration_full(i, Sb, Sa);
            iteration_full(i + 1, Sa, Sb);
        }
        for (; i + 1 < ntile; i += 2) {
            iteration(i, Sb, Sa, std::false_type{});
            iteration(i + 1, Sa, Sb, std::false_type{});
        }
        PFrag P;
        float l_t;
        if (i < ntile) {  // odd tile left over
            iteration(i, Sb, Sa, std::false_type{});
            rescale();
            split_p(Sb, P, l_t);
            l_run += l_t;
            pv(i, P);
        } else {
            rescale();
            split_p(Sa, P, l_t);
            l_run += l_t;
            pv(ntile - 1, P);
        }
    }

    // ---- normalise and write (transpose through LDS so each query row is stored contiguously)
    __syncthreads();
    float* Os = reinterpret_cast<float*>(smem4) + wid * (32 * 33);
    const int H64 = p.heads * 64;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (ntile > 0) ? 1.0f / l_tot : 0.0f;  // l carries the same 2^14 as O
#pragma unroll
    for (int df = 0; df < 2; ++df) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Os[lo * 33 + frag_row(r, hi)] = o[df][r] * inv;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // read back: 8 lanes cover the 32 dims of one query -> one 16-byte store per lane
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = 8 * qq + (lane >> 3), d4 = (lane & 7) * 4;
            const int row = q0 + wid * 32 + q;
            const float4 v = make_float4(Os[q * 33 + d4], Os[q * 33 + d4 + 1], Os[q * 33 + d4 + 2], Os[q * 33 + d4 + 3]);
            if (row < nq) *reinterpret_cast<float4*>(p.O + ((size_t)seq * R + row) * H64 + head * 64 + 32 * df + d4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

void attention_pipe_launch(const AttnP& p, hipStream_t stream) {
    const dim3 grid((p.rows_per_seq / 128) * p.heads * p.nseq);
    hipLaunchKernelGGL(attn_split_pipe_kernel, grid, dim3(256), 0, stream, p);
}
